"""Regenerates tests/golden/ from the REAL reference binary (oracle/_ref, built from
/root/reference by oracle/Makefile) with /dev/urandom pinned by oracle/urandom_shim.c and -t 1.

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py
Committed outputs: the small FASTA inputs, each case's de_bruijn.bin (or its sha256 for the big
synthetic cases), and the reference's own log counters / round ranges in cases.json.
"""
import hashlib
import json
import os
import re
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from twopaco_amd import synth  # noqa: E402

SEED = 20240229


def write_edge_fa(path):
    """Hand-made edge cases (SURVEY 8a' item 4): record of exactly k=5 bases, shorter than k, k+1,
    empty record, leading NN, lower case + IUPAC letters, duplicated records (shared first/last
    k-mers: stub vs real id), palindromic (k+1)-mers (strand tie), N runs."""
    recs = [
        ("len_k", "ACGTA"),
        ("short", "ACG"),
        ("len_k1", "ACGTAC"),
        ("empty", ""),
        ("lead_nn", "NNACGTACGGTTACNNNNTTGACCA"),
        ("lower_iupac", "acgtRYacgtkmacgtacgtswbdhvnacgtacca"),
        ("dup_a", "GATTACAGATTACATTTGGGCCCAAATTTGGGCCC"),
        ("dup_b", "GATTACAGATTACATTTGGGCCCAAATTTGGGCCC"),
        ("palin", "AACGTTAACGTTGGATCCGGATCCAATTAATT"),
        ("branch", "TTTTTACGTAGGGGGACGTACCCCCACGTATTTTT"),
        ("n_inside", "ACGTACGTNACGTACGTNNACGTACGTACGT"),
        ("one_n", "N"),
        ("x_v_u", "ACGTXACGTVACGTUACGT"),
    ]
    with open(path, "w") as f:
        for name, s in recs:
            f.write(">%s some description\n" % name)
            for i in range(0, len(s), 17):
                f.write(s[i:i + 17] + "\n")
            if name == "lead_nn":
                f.write("\n  \n")  # blank / whitespace lines are skipped


def write_rand_fa(path, n, nchr, seed, n_rate=1 / 500.0, change=0.05, indel_frac=0.9):
    """The self-test's input shape (reference test.cpp:20-67): chr0 random with N, the others =
    chr0 with 5 % edits, 10 % of them substitutions, the rest insertions/deletions."""
    base = synth.random_genome(n, seed)
    r = synth._stream(seed, n, 7)
    base = base.copy()
    base[r < np.uint64(int(n_rate * 2.0 ** 64))] = 4
    recs = [base]
    for c in range(1, nchr):
        ev = synth._stream(seed + c, n, 8)
        kind = synth._stream(seed + c, n, 9) % np.uint64(100)
        letter = (synth._stream(seed + c, n, 10) >> np.uint64(62)).astype(np.uint8)
        out = []
        edit = ev < np.uint64(int(change * 2.0 ** 64))
        for i in range(n):
            if edit[i]:
                kd = int(kind[i])
                if kd < 10:
                    out.append(letter[i])
                elif kd < 55:
                    out.append(base[i]); out.append(letter[i])
                # else deletion
            else:
                out.append(base[i])
        recs.append(np.array(out, dtype=np.uint8))
    synth.write_fasta(path, recs, width=60)


def write_longk_fa(path, seed=13):
    """Records with long shared stretches, so that keys of 10-19 words (k = 301 / 603) have real junctions: chr0 random,
    chr1 = chr0 with a substitution every ~1500 bases, chr2 = chr0 rotated (one breakpoint), chr3 = a 1300-base repeat
    unit three times with one N, chr4 shorter than k = 603."""
    base = synth.random_genome(6000, seed)
    c1 = base.copy()
    for i in range(700, 6000, 1500):
        c1[i] = (c1[i] + 1) & 3
    c2 = np.concatenate([base[2500:], base[:2500]])
    unit = synth.random_genome(1300, seed + 1)
    c3 = np.concatenate([unit, unit, unit])
    c3[2000] = 4
    c4 = base[100:500]
    synth.write_fasta(path, [base, c1, c2, c3, c4], width=70)


def write_tracts_fa(path, seed=17):
    """Low-complexity input (round 5: positions whose window repeats the one 1 .. 6 before them send nothing in the partitioned passes,
    tpc_qpartition.hip:k_periodic_build): chr0 random with homopolymer / dinucleotide / tri- ... hepta-nucleotide tracts of 30 .. 700 bases,
    some with an N inside, one across the 16384-position tile boundary of the packed text; chr1 = chr0 with 1 % substitutions (the tracts are
    junction-rich where the copies differ); chr2 a telomere: (TTAGGG)n over 6 kbp between random flanks; chr3 poly-A only; chr4 = (CA)n only."""
    n = 40000
    base = synth.random_genome(n, seed).copy()
    units = [[0], [3], [1, 0], [2, 3], [0, 0, 1], [0, 1, 2, 3], [1, 1, 0, 3, 2], [3, 3, 0, 2, 2, 2], [0, 1, 2, 3, 0, 2, 1]]
    at = synth._stream(seed, 40, 21) % np.uint64(n - 800)
    ln = synth._stream(seed, 40, 22) % np.uint64(671) + np.uint64(30)
    for t in range(40):
        a, l = int(at[t]), int(ln[t])
        if t == 7:
            a = 16384 - 1 - 300  # (record 0 starts at text position 1: the tract straddles the first tile boundary)
        base[a:a + l] = np.resize(np.array(units[t % len(units)], dtype=np.uint8), l)
        if t % 6 == 0:
            base[a + l // 2] = 4
    c1 = synth.substitute(base, 0.01, seed + 1)
    c1 = np.where(base == 4, 4, c1).astype(np.uint8)
    flank = synth.random_genome(600, seed + 2)
    c2 = np.concatenate([flank[:300], np.resize(np.array([3, 3, 0, 2, 2, 2], dtype=np.uint8), 6000), flank[300:]])
    c3 = np.zeros(900, dtype=np.uint8)
    c4 = np.resize(np.array([1, 0], dtype=np.uint8), 1100)
    synth.write_fasta(path, [base, c1, c2, c3, c4], width=80)


def parse_log(log):
    rounds = []
    for m in re.finditer(r"Round (\d+), (\d+):(\d+)", log):
        rounds.append({"low": int(m.group(2)), "high": int(m.group(3))})
    for key, pat in [("true", r"True junctions count = (\d+)"), ("false", r"False junctions count = (\d+)"),
                     ("table", r"Hash table size = (\d+)"), ("marks", r"Candidate marks count = (\d+)")]:
        for i, m in enumerate(re.finditer(pat, log)):
            rounds[i][key] = int(m.group(1))
    return {"rounds": rounds, "true_marks": int(re.search(r"True marks count: (\d+)", log).group(1)),
            "distinct": int(re.search(r"Distinct junctions = (\d+)", log).group(1))}


def main():
    # --only a,b,c : (re)generate just these cases and merge them into cases.json (the big synthetic
    # cases take minutes to an hour and tens of GiB of filter each)
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    O.build()
    tmp = tempfile.mkdtemp(dir=os.environ.get("TPC_GOLDEN_TMP"))
    if only is None:
        shutil.copy("/root/reference/example/example.fa", os.path.join(HERE, "example.fa"))
        write_edge_fa(os.path.join(HERE, "edge.fa"))
        write_rand_fa(os.path.join(HERE, "rand6.fa"), 3000, 6, 11)
        write_rand_fa(os.path.join(HERE, "c2.fa"), 2500, 3, 12, n_rate=1 / 900.0, change=0.03)
    if only is None or any(n.startswith("lk_") for n in only):
        write_longk_fa(os.path.join(HERE, "lk.fa"))
    if only is None or any(n.startswith("tr_") for n in only):
        write_tracts_fa(os.path.join(HERE, "tracts.fa"))

    cases = []

    def case(name, fasta, k, L, q=5, rounds=1, debug=False, abundance=None, keep_bin=True, synth_spec=None, files=None, threads=1):
        if only is not None and name not in only:
            return
        if files is None:
            files = [os.path.join(HERE, fasta)]
        elif callable(files):
            files = files()
        out = os.path.join(tmp, name + ".bin")
        # threads > 1 only for the big cases: the bytes and counters of a single-round run do not depend on the thread count
        log, _ = O.run_reference(files, k, L, q=q, rounds=rounds, threads=threads, seed=SEED, out=out, tmpdir=tmp, debug=debug,
                                 abundance=abundance, timeout=6 * 3600)
        if keep_bin:
            data = open(out, "rb").read()
            digest, nbytes = hashlib.sha256(data).hexdigest(), len(data)
        else:  # the big cases: gigabytes of output, hashed as a stream
            h, nbytes = hashlib.sha256(), 0
            with open(out, "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
                    nbytes += len(blk)
            digest, data = h.hexdigest(), None
            os.remove(out)
        c = {"name": name, "fasta": fasta, "k": k, "L": L, "q": q, "n_rounds": rounds, "seed": SEED, "ref_debug_build": debug, "ref_threads": threads,
             "abundance": abundance, "bin_sha256": digest, "bin_bytes": nbytes}
        if synth_spec:
            c["synth"] = synth_spec
        c.update(parse_log(log))
        if keep_bin:
            with open(os.path.join(HERE, name + ".bin"), "wb") as f:
                f.write(data)
            c["bin"] = name + ".bin"
        cases.append(c)
        print(name, nbytes, c["distinct"], [r.get("false") for r in c["rounds"]])

    case("example_k11", "example.fa", 11, 20)
    case("example_k15_r3", "example.fa", 15, 20, rounds=3)
    case("example_k15_dbg", "example.fa", 15, 20, debug=True)
    case("example_k25_dbg_r2", "example.fa", 25, 24, rounds=2, debug=True)
    case("edge_k5", "edge.fa", 5, 16)
    case("edge_k5_dbg", "edge.fa", 5, 16, debug=True)
    case("edge_k7_fp_r2", "edge.fa", 7, 10, rounds=2)
    case("edge_k3", "edge.fa", 3, 12)
    case("rand6_k9_fp", "rand6.fa", 9, 14)
    case("rand6_k9_fp_r4", "rand6.fa", 9, 14, rounds=4)
    case("rand6_k9_dbg", "rand6.fa", 9, 20, debug=True)
    case("rand6_k25_q3", "rand6.fa", 25, 24, q=3)
    case("rand6_k3", "rand6.fa", 3, 12)
    case("rand6_k9_q1", "rand6.fa", 9, 16, q=1)
    case("rand6_k9_q8", "rand6.fa", 9, 18, q=8)
    case("rand6_k9_a3", "rand6.fa", 9, 20, abundance=3)
    case("rand6_k9_L33", "rand6.fa", 9, 33)
    case("rand6_k27", "rand6.fa", 27, 22)
    case("c2_k29", "c2.fa", 29, 22)
    case("c2_k35", "c2.fa", 35, 22)
    case("c2_k51_r2", "c2.fa", 51, 22, rounds=2)
    case("c2_k61", "c2.fa", 61, 22)
    case("c2_k125", "c2.fa", 125, 22)
    # long keys (C = 10 and C = 19 words: the reference's MAX_CAPACITY, vertexenumerator.cpp:17-70) and q beyond 8
    case("lk_k301", "lk.fa", 301, 22)
    case("lk_k603", "lk.fa", 603, 22)
    case("lk_k603_r2", "lk.fa", 603, 24, rounds=2)
    case("lk_k159", "lk.fa", 159, 20)
    case("rand6_k9_q12", "rand6.fa", 9, 20, q=12)
    # more than 16 hash functions (the reference takes any -q, constructor.cpp:83-90): the closed-form kernels of tpc_pass1_anyq.hip;
    # _fp: a filter small enough for Bloom false positives, two rounds (the split pass with 20 functions)
    case("rand6_k9_q20", "rand6.fa", 9, 22, q=20)
    case("rand6_k9_q20_fp_r2", "rand6.fa", 9, 18, q=20, rounds=2)
    # collision-free multi-round runs: "first seen" in the split pass (VE.h:559-570) is then order independent,
    # so the round ranges (VE.h:206-254) below are what ANY correct implementation must print
    case("rand6_k9_L24_r4", "rand6.fa", 9, 24, rounds=4)
    case("c2_k29_L26_r3", "c2.fa", 29, 26, rounds=3)

    # low-complexity tracts (periodic windows) at filter sizes that take the partitioned passes (L >= 28); _r2: gated rounds, collision free
    case("tr_k25_L28", "tracts.fa", 25, 28)
    case("tr_k11_L28_r2", "tracts.fa", 11, 28, rounds=2)
    case("tr_k31_L30_q3", "tracts.fa", 31, 30, q=3)

    # synthetic workloads of BASELINE.json's configs (FASTA regenerated from twopaco_amd/synth.py)
    # m2_full = the bench workload (BASELINE configs[2]); m2_s05_f38 = the f = 38 geometry (512 bins per level) on a text the
    # reference finishes in minutes; m3_f38 = configs[3]'s shape (7 genomes, 1.12 Gbp of text, 32 GiB filter, several query batches)
    for name, wl, scale, L, thr in [("m1_small", "m1", 0.02, 26, 1), ("m1_full", "m1", 1.0, None, 1), ("m2_small", "m2", 0.004, 26, 1),
                                    ("m2_full", "m2", 1.0, None, 3), ("m2_s05_f38", "m2", 0.05, 38, 3), ("m3_f38", "m3", 1.0, None, 4),
                                    # m2_x15 = the bench workload with 15 x longer genomes: 4.65 G text positions, beyond 2^32 (hours of reference time)
                                    ("m2_x15", "m2", 15.0, None, 6),
                                    # m2r = m2 with repeat families, low-complexity tracts, two genomes on the other strand and 50-300 contigs per
                                    # genome (round 5: address skew, hot exact-filter keys, the stub / separator path at full size)
                                    # -t 1: the stub ids of sequence ends come from a shared counter in the order the worker threads reach them
                                    # (VE.h:945-947) -- with 10 149 records the bytes of a multi-threaded run depend on the scheduling
                                    ("m2r_small", "m2r", 0.02, 26, 1), ("m2r_full", "m2r", 1.0, None, 1),
                                    # m2r2 = m2r + minisatellite tracts (units of 7..60 bp over 200..2000 bp): periodic windows beyond the periods the
                                    # hash kernels skip (round 6)
                                    ("m2r2_small", "m2r2", 0.02, 26, 1), ("m2r2_full", "m2r2", 1.0, None, 1)]:
        if only is not None and name not in only:
            continue

        def files(name=name, wl=wl, scale=scale):
            recs, pp = synth.workload(wl, seed=12345, scale=scale)
            return synth.fasta_files(recs, pp, tmp, prefix=name + "_")
        p = synth.workload(wl, seed=12345, scale=0.0001)[1]
        case(name, None, p["k"], L or p["L"], q=p["q"], keep_bin=False, synth_spec={"workload": wl, "seed": 12345, "scale": scale}, files=files, threads=thr)

    path = os.path.join(HERE, "cases.json")
    if only is not None:
        old = json.load(open(path))
        new = {c["name"] for c in cases}
        cases = [c for c in old if c["name"] not in new] + cases
    with open(path, "w") as f:
        json.dump(cases, f, indent=1)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
