#!/usr/bin/env python3
"""BASELINE configs[3] / configs[4] at the sizes they name, on ONE MI355X (288 GB HBM).

  python tools/big_config.py --genomes 7 --len 3100000000 --k 25 --L 38            # configs[3]: 21.7 Gbp, 32 GiB filter
  python tools/big_config.py --genomes 3 --len 3000000000 --k 51 --L 40 --rounds 4 # configs[4]-shaped: 9 Gbp, 128 GiB filter

The reference cannot produce a golden at these sizes in any useful time, so the checks are the size-independent ones:
  * the LDS write-combining passes against the direct (scattered atomicOr / load) kernels of the same library: candidate
    mask (sha256), every counter of every round (marks / true / false / table), the sorted junction keys (sha256), the
    (position, id) list (sha256);
  * ids recomputed on the host for a random sample of emitted occurrences: canonical packed k-mer (strand by the function-0
    hash, reference candidateoccurence.h:25-50) -> binary search in the sorted keys -> +-(rank + 1) (bifurcationstorage.h:100-153);
  * |id| <= junctions for every emitted record, occurrences <= marks, true + false == table per round, keys strictly sorted;
  * the junction stream: 12 bytes per record + one separator per sequence step (junctionapi.h:118-126).
Prints ms per phase and k-mers/s through the whole path; --json appends one summary line to a file."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from twopaco_amd import capi, synth  # noqa: E402


def sha(a):
    h = hashlib.sha256()
    a = np.ascontiguousarray(a)
    mv = memoryview(a).cast("B")
    step = 1 << 28
    for i in range(0, len(mv), step):
        h.update(mv[i:i + step])
    return h.hexdigest()[:16]


def vertex_ranges(L, rounds):
    """Equal-mass ranges of min(H, H') (density 2(1 - x)): what the split pass measures in expectation."""
    size = 1 << L
    cuts = [int(size * (1.0 - (1.0 - r / rounds) ** 0.5)) for r in range(rounds + 1)]
    cuts[-1] = size
    return [(cuts[r] + (1 if r else 0), cuts[r + 1]) for r in range(rounds)]


def run(text, args, mode, tab, ranges, fetch):
    ctx = capi.Context(0)
    ctx.set_option("insert_mode", mode)
    ctx.set_option("query_mode", mode)
    if args.budget_gb:
        ctx.set_option("part_budget_bytes", int(args.budget_gb * (1 << 30)))
    for kv in args.set:
        name, value = kv.split("=")
        ctx.set_option(name, int(value))
    ctx.set_params(args.k, args.L, args.q, tab)
    t0 = time.time()
    ctx.seq_upload(text)
    up = time.time() - t0
    for rep in range(max(1, args.repeat if mode != 1 else 1)):  # --repeat 2: the second pass runs with every buffer allocated ("warm")
        ctx.run_begin()
        rounds = []
        t_all = time.time()
        for lo, hi in ranges:
            t1 = time.time()
            ctx.filter_reset()
            ctx.pass1_insert(lo, hi, count=False)
            t2 = time.time()
            marks = ctx.pass1_query(lo, hi)
            t3 = time.time()
            st = ctx.pass2_filter()
            t4 = time.time()
            rounds.append({"lo": lo, "hi": hi, "marks": marks, **st, "insert_s": t2 - t1, "query_s": t3 - t2, "filter2_s": t4 - t3,
                           "insert_kernel_ms": ctx.kernel_ms("insert"), "query_kernel_ms": ctx.kernel_ms("query"),
                           "insert_path": ctx.stat("insert_path"), "query_path": ctx.stat("query_path"),
                           "insert_batches": ctx.stat("insert_batches"), "query_batches": ctx.stat("query_batches")})
            print("  mode %d pass %d round %s" % (mode, rep, json.dumps(rounds[-1])), flush=True)
        t5 = time.time()
        J = ctx.junctions_finalize()
        n_marked, n_valid = ctx.emit()
        t6 = time.time()
        print("  mode %d pass %d whole path %.3f s" % (mode, rep, t6 - t_all), flush=True)
    out = {"mode": mode, "upload_s": up, "rounds": rounds, "junctions": J, "marked": n_marked, "occurrences": n_valid,
           "finalize_emit_s": t6 - t5, "whole_s": t6 - t_all}
    if fetch:
        out["mask_sha"] = sha(ctx.mask_download(True))
        keys = ctx.junction_keys()
        out["keys_sha"] = sha(keys)
        g, ids = ctx.emit_fetch()
        out["ids_sha"] = sha(g) + sha(ids)
        out["_keys"], out["_g"], out["_ids"] = keys, g, ids
        rs, rl = text.rec_start, text.rec_length
        nb, nr, ssha = ctypes_stream(ctx, rs, rl)
        out["stream_bytes"], out["stream_records"], out["stream_sha"] = nb, nr, ssha
    ctx.close()
    return out


def ctypes_stream(ctx, rs, rl):
    import ctypes
    rs = np.ascontiguousarray(rs, dtype=np.uint64)
    rl = np.ascontiguousarray(rl, dtype=np.uint64)
    nb, nr = ctypes.c_uint64(0), ctypes.c_uint64(0)
    ctx._ck(capi.hip().tpc_emit_stream(ctx._h, rs.ctypes.data, rl.ctypes.data, rs.size, ctypes.byref(nb), ctypes.byref(nr)))
    h = hashlib.sha256()  # the bytes of de_bruijn.bin as the device formatted them, fetched in 256 MiB pieces
    buf = np.zeros(min(nb.value, 1 << 28), dtype=np.uint8)
    for off in range(0, nb.value, 1 << 28):
        m = min(1 << 28, nb.value - off)
        if capi.hip().tpc_emit_stream_fetch(ctx._h, off, m, buf.ctypes.data) != 0:
            raise RuntimeError("tpc_emit_stream_fetch failed")
        h.update(memoryview(buf)[:m])
    return nb.value, nr.value, h.hexdigest()[:16]


def host_ids(text, args, tab, keys, g, ids, sample, seed=1):
    """ids of a random sample of emitted occurrences recomputed on the host from the packed text."""
    rng = np.random.default_rng(seed)
    valid = np.nonzero(ids != capi.INVALID_VERTEX)[0]
    pick = valid[rng.integers(0, valid.size, min(sample, valid.size))]
    bases = text.bases
    k, L = args.k, args.L
    C = keys.shape[1]
    mask = (1 << L) - 1
    h0 = [int(tab[0][c]) for c in range(4)]

    def rotl1(x):
        return ((x << 1) & mask) | (x >> (L - 1))

    bad = 0
    keys_t = keys
    for p in pick.tolist():
        pos = int(g[p])
        codes = [(int(bases[(pos + t) >> 5]) >> (2 * ((pos + t) & 31))) & 3 for t in range(k)]
        hp = hn = 0
        for t in range(k):
            hp = rotl1(hp) ^ h0[codes[t]]
            hn = rotl1(hn) ^ h0[3 - codes[k - 1 - t]]
        rc = [3 - c for c in reversed(codes)]
        if hp < hn:
            fwd = True
        elif hn < hp:
            fwd = False
        else:  # LessSelfReverseComplement (dnachar.cpp:98-114): the k-mer against its reverse complement, as text
            fwd = "".join("ACGT"[c] for c in codes) <= "".join("ACGT"[c] for c in rc)
        canon = codes if fwd else rc
        words = [0] * C
        for i, c in enumerate(canon):
            words[i >> 5] |= c << (2 * (i & 31))
        # binary search in CompressedString::Less order (word 0 first, compressedstring.h:93-104)
        lo_i, hi_i = 0, keys_t.shape[0]
        tgt = tuple(words)
        while lo_i < hi_i:
            mid = (lo_i + hi_i) // 2
            if tuple(int(x) for x in keys_t[mid]) < tgt:
                lo_i = mid + 1
            else:
                hi_i = mid
        found = lo_i < keys_t.shape[0] and tuple(int(x) for x in keys_t[lo_i]) == tgt
        want = (lo_i + 1) if fwd else -(lo_i + 1)
        if not found or int(ids[p]) != want:
            bad += 1
    return pick.size, bad


def enumerator_leg(recs_files, args, stream_sha, stream_bytes, n_rec, tmpdir):
    """The same input through CreateEnumerator (host/vertexenumerator.cpp: FASTA parse, device split pass when rounds > 1,
    junction stream written to a file; reference VE.h:122-466).  The file's bytes do not depend on where the round boundaries
    fall -- a Bloom false positive never gets an id -- so its sha256 must equal the C-ABI run's stream, whose rounds were cut
    at the analytic quantiles; its size is 12 x (records + separators) (junctionapi.h:118-126)."""
    import re
    out = os.path.join(tmpdir, "big.bin")
    t0 = time.time()
    e = capi.Enumerator(recs_files, args.k, args.L, q=args.q, rounds=args.rounds, tmpdir=tmpdir, out=out, seed=20240229, threads=min(64, os.cpu_count() or 1))
    wall = time.time() - t0
    log = e.log
    e.close()
    rounds = [(int(a), int(b)) for a, b in re.findall(r"Round \d+, (\d+):(\d+)", log)]
    tj = [int(x) for x in re.findall(r"True junctions count = (\d+)", log)]
    fj = [int(x) for x in re.findall(r"False junctions count = (\d+)", log)]
    ht = [int(x) for x in re.findall(r"Hash table size = (\d+)", log)]
    records = int(re.search(r"True marks count: (\d+)", log).group(1))
    size = os.path.getsize(out)
    assert len(rounds) == args.rounds and len(tj) == args.rounds, log
    assert rounds[0][0] == 0 and all(rounds[i + 1][0] == rounds[i][1] + 1 for i in range(args.rounds - 1)), rounds  # VE.h:234-254
    assert all(t + f == h for t, f, h in zip(tj, fj, ht)), (tj, fj, ht)   # VE.h:384-388
    assert size == 12 * (records + (n_rec - 1)), (size, records, n_rec)    # every sequence of these inputs is long enough to emit
    h = hashlib.sha256()
    with open(out, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    os.unlink(out)
    if stream_sha is not None:
        assert size == stream_bytes and h.hexdigest()[:16] == stream_sha, "CreateEnumerator's file differs from the C-ABI run's junction stream"
    if args.rounds > 1:
        assert "Splitting the input kmers set..." in log
    print("CreateEnumerator: %.1f s wall, %d rounds %s, junctions %d, records %d, file %d bytes == 12 x (records + separators), sha256 == C-ABI stream" % (
        wall, args.rounds, rounds, sum(tj), records, size), flush=True)
    return {"wall_s": wall, "rounds": rounds, "true": tj, "false": fj, "table": ht, "records": records, "file_bytes": size}


def check(args):
    """Runs the configuration and every check; returns the summary (raises AssertionError on any difference)."""
    n = int(args.len)
    t0 = time.time()
    root = synth.random_genome(n, 12345)
    text = capi.PackedText()
    kmers = 0
    files = []
    fasta_dir = getattr(args, "fasta_dir", "") or ""
    for gidx in range(args.genomes):
        rec = synth.add_n_runs(synth.substitute(root, args.div, 4000 + gidx), 2e-6, 5000 + gidx)
        capi.host().tpch_text_add_codes(text._h, rec.ctypes.data, rec.size)
        if fasta_dir:
            files.append(os.path.join(fasta_dir, "big_%d.fa" % gidx))
            synth.write_fasta(files[-1], [rec], first_id=gidx)
        bad = np.flatnonzero(rec == 4)
        kmers += rec.size - args.k + 1
        if bad.size:  # windows that hold an N: the N characters plus k - 1 positions before every run (runs are far apart)
            runs = 1 + int(np.count_nonzero(np.diff(bad) > 1))
            kmers -= int(bad.size) + runs * (args.k - 1)
        del rec
    del root
    gen_s = time.time() - t0
    print("text: %d genomes x %d bp = %.2f G positions (2^%.2f), ~%.3f G vertex k-mers, generated + packed in %.0f s" % (
        args.genomes, n, text.length / 1e9, np.log2(text.length), kmers / 1e9, gen_s), flush=True)
    if getattr(args, "enumerator_only", False):  # CreateEnumerator alone, as the first thing this process does with the device (a cold run's phases: TWOPACO_TIMING=1)
        n_rec = len(text.rec_start)
        del text
        e = enumerator_leg(files, args, None, None, n_rec, fasta_dir)
        for f in files:
            os.unlink(f)
        return {"genomes": args.genomes, "len": n, "k": args.k, "L": args.L, "rounds": args.rounds, "generate_pack_s": gen_s, "enumerator": e}
    tab = capi.seed_table(args.q, args.L, seed=20240229)
    ranges = [(0, 1 << args.L)] if args.rounds == 1 else vertex_ranges(args.L, args.rounds)
    a = run(text, args, args.force_mode, tab, ranges, fetch=True)
    summary = {"genomes": args.genomes, "len": n, "positions": int(text.length), "k": args.k, "L": args.L, "q": args.q, "rounds": args.rounds,
               "kmers_estimate": kmers, "generate_pack_s": gen_s}
    for r in a["rounds"]:
        assert r["true"] + r["false"] == r["table"], r
    assert a["occurrences"] <= a["marked"] and a["junctions"] == sum(r["true"] for r in a["rounds"])
    keys, g, ids = a.pop("_keys"), a.pop("_g"), a.pop("_ids")
    kk = keys[:, 0] if keys.shape[1] == 1 else None
    if kk is not None:
        assert (kk[1:] > kk[:-1]).all(), "junction keys not strictly sorted"
    assert (g[1:] > g[:-1]).all(), "emitted positions not strictly increasing"
    valid = ids != capi.INVALID_VERTEX
    assert int(valid.sum()) == a["occurrences"] and (np.abs(ids[valid]) <= a["junctions"]).all() and (ids[valid] != 0).all()
    n_rec = len(text.rec_start)
    # the stream: every valid record + stub records of sequence ends (2 per sequence at most) + one separator per sequence step
    assert a["stream_records"] >= a["occurrences"] and a["stream_records"] <= a["occurrences"] + 2 * n_rec
    assert a["stream_bytes"] == 12 * (a["stream_records"] + (n_rec - 1)), (a["stream_bytes"], a["stream_records"], n_rec)
    t1 = time.time()
    checked, badn = host_ids(text, args, tab, keys, g, ids, args.sample)
    print("host id recomputation: %d sampled occurrences, %d mismatches (%.0f s)" % (checked, badn, time.time() - t1), flush=True)
    assert badn == 0
    summary["partitioned"] = a
    summary["host_id_sample"] = checked
    ins = sum(r["insert_kernel_ms"] for r in a["rounds"]); qry = sum(r["query_kernel_ms"] for r in a["rounds"])
    summary["kmers_per_s"] = kmers * 1.0 / a["whole_s"]
    print("partitioned: insert %.1f ms, query %.1f ms, whole path %.3f s = %.2f G k-mers/s (x %d rounds of hashing); marks %d junctions %d occurrences %d; stream %d bytes" % (
        ins, qry, a["whole_s"], kmers * 1.0 / a["whole_s"] / 1e9, args.rounds, sum(r["marks"] for r in a["rounds"]), a["junctions"], a["occurrences"], a["stream_bytes"]), flush=True)
    if not args.no_direct:
        del keys, g, ids
        b = run(text, args, 1, tab, ranges, fetch=True)
        for kname in ("_keys", "_g", "_ids"):
            b.pop(kname)
        same = all(b[x] == a[x] for x in ("junctions", "marked", "occurrences", "mask_sha", "keys_sha", "ids_sha", "stream_bytes", "stream_records", "stream_sha"))
        same = same and all(all(ra[x] == rb[x] for x in ("marks", "true", "false", "table")) for ra, rb in zip(a["rounds"], b["rounds"]))
        print("direct kernels: insert %.1f ms, query %.1f ms, whole path %.3f s; partitioned == direct (mask sha, counters, keys sha, ids sha, stream sha): %s" % (
            sum(r["insert_kernel_ms"] for r in b["rounds"]), sum(r["query_kernel_ms"] for r in b["rounds"]), b["whole_s"], same), flush=True)
        summary["direct"] = b
        summary["partitioned_equals_direct"] = same
        assert same
    if fasta_dir:
        del text
        summary["enumerator"] = enumerator_leg(files, args, a["stream_sha"], a["stream_bytes"], n_rec, fasta_dir)
        for f in files:
            os.unlink(f)
    return summary


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=7)
    ap.add_argument("--len", type=float, default=3.1e9)
    ap.add_argument("--k", type=int, default=25)
    ap.add_argument("--L", type=int, default=38)
    ap.add_argument("--q", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=1)
    ap.add_argument("--div", type=float, default=0.001, help="substitution rate of every genome against the common root")
    ap.add_argument("--budget-gb", type=float, default=0)
    ap.add_argument("--no-direct", action="store_true", help="skip the comparison run on the direct kernels")
    ap.add_argument("--sample", type=int, default=20000)
    ap.add_argument("--repeat", type=int, default=1, help="passes of the partitioned run on the same context; the last one is reported")
    ap.add_argument("--set", action="append", default=[], help="name=value: tpc_set_option on both runs' contexts (e.g. survivor_fp_mode=0)")
    ap.add_argument("--json", default="")
    ap.add_argument("--force-mode", type=int, default=0, help="insert_mode / query_mode of the first run: 0 automatic, 2 the partitioned passes whatever the plan's own estimate says")
    ap.add_argument("--enumerator-only", action="store_true", help="with --fasta-dir: only the CreateEnumerator leg (no C-ABI run before it: a cold process)")
    ap.add_argument("--fasta-dir", default="", help="also write the genomes as FASTA files there and run them through CreateEnumerator (file size, counters, sha256 == the C-ABI stream)")
    return ap


def main():
    args = parser().parse_args()
    summary = check(args)
    if args.json:
        with open(args.json, "a") as f:
            f.write(json.dumps(summary) + "\n")
    print("OK", flush=True)


if __name__ == "__main__":
    main()
