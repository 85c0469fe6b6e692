#!/usr/bin/env python3
"""bench.py -- junction-enumeration hot path on MI355X, BASELINE.json's metric.

A "step" is one full pass of the hot path over the synthetic workload, input already packed and
resident in HBM: Bloom filter reset, first-pass insert, first-pass query, candidate compaction,
second-pass exact filter, junction key sort + id index, output-pass id lookup (junction records
left in HBM).  value = vertex k-mers through the whole path per second, whole job.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload m2|m1] [--scale S]

N > 1 is launched by torch.distributed.run (one process per GPU, RCCL); see twopaco_amd/dist.py.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from twopaco_amd import capi, synth  # noqa: E402

G_BYTES = 64          # HBM access granule of a scattered 4-byte access (SURVEY 8d planning value)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s


def one_step(ctx, abundance=(1 << 64) - 1):
    ctx.run_begin()
    ctx.filter_reset()
    ctx.pass1_insert(count=False)
    marks = ctx.pass1_query()
    st = ctx.pass2_filter(abundance)
    J = ctx.junctions_finalize()
    n_marked, n_valid = ctx.emit()
    return marks, st, J, n_valid


def cpu_baseline(recs, p, n_genomes=6, timeout=240):
    """The REAL reference binary (oracle/_ref/twopaco_ref, built from /root/reference) on a bounded
    sample of the same workload, on this host's cores.  Falls back to the C oracle (kind 'port')."""
    cores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "twopaco_ref")
    sample = recs[:n_genomes]
    kmers = synth.n_kmers(sample, p["k"])
    tmp = tempfile.mkdtemp()
    if os.path.exists(ref):
        files = []
        for i, r in enumerate(sample):
            path = os.path.join(tmp, "s%d.fa" % i)
            synth.write_fasta(path, [r], first_id=i)
            files.append(path)
        for L in (p["L"], 32):
            cmd = [ref, "-k", str(p["k"]), "-f", str(L), "-q", str(p["q"]), "-t", str(cores), "--tmpdir", tmp, "-o", os.path.join(tmp, "ref.bin")] + files
            t0 = time.time()
            try:
                res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            except subprocess.TimeoutExpired:
                continue
            wall = time.time() - t0
            if res.returncode != 0:
                continue
            log = res.stdout.decode()
            m = re.search(r"\n1\t(\d+)\t(\d+)\t", log)
            occ = int(re.search(r"True marks count: (\d+)", log).group(1))
            return {"value": kmers / wall, "unit": "k-mers/s", "cores": cores, "kind": "reference",
                    "sample": "first %d genomes of the workload (%d k-mers), k=%d q=%d f=%d, reference binary -t %d, wall %.1f s "
                              "(its log: fill %s s incl. serial filter zeroing, query %s s); %d junction occurrences"
                              % (n_genomes, kmers, p["k"], p["q"], L, cores, wall, m.group(1) if m else "?", m.group(2) if m else "?", occ),
                    "junction_occurrences_per_sec": occ / wall}
    from oracle import oracle as O
    o = O.Oracle(p["k"], min(p["L"], 32), p["q"], O.seed_table(1, p["q"], min(p["L"], 32)))
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for r in sample[:2]:
        o.add_record(letters[r].tobytes())
    kmers = synth.n_kmers(sample[:2], p["k"])
    t0 = time.time()
    o.enumerate()
    wall = time.time() - t0
    return {"value": kmers / wall, "unit": "k-mers/s", "cores": 1, "kind": "port",
            "sample": "first 2 genomes (%d k-mers), scalar C oracle, f=%d, wall %.1f s" % (kmers, min(p["L"], 32), wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="m2")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--test-first", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decomposition", default="ranges", choices=["ranges", "address"],
                    help="multi-GPU: vertex-hash ranges (default) or the address-sharded filter (power-of-two N)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("TPC_FORCE_DIST"):  # TPC_FORCE_DIST: exercise the distributed path with one rank
        from twopaco_amd import dist as tdist
        return tdist.bench_main(args, rank, world, local_rank)

    torch.cuda.set_device(0)
    recs, p = synth.workload(args.workload, scale=args.scale)
    n_kmers = synth.n_kmers(recs, p["k"])
    text = capi.PackedText.from_codes(recs)
    ctx = capi.Context(0)
    ctx.set_option("insert_test_first", args.test_first)
    ctx.set_params(p["k"], p["L"], p["q"], capi.seed_table(p["q"], p["L"], seed=12345))
    t0 = time.time()
    ctx.seq_upload(text)
    torch.cuda.synchronize()
    upload_s = time.time() - t0

    for _ in range(args.warmup):
        one_step(ctx)
    names = ["filter_reset", "insert", "query", "compact", "filter2", "scan2", "sort", "emit"]
    kms = {n: 0.0 for n in names}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        marks, st, J, n_valid = one_step(ctx)
        for n in names:
            kms[n] += max(ctx.kernel_ms(n), 0.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = {n: v / args.steps for n, v in kms.items()}

    # measured HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as
    # MI355X_MICROARCH.md prescribes) -- only valid for the workload they were collected on
    traffic_ins = traffic_qry = None
    pmc = os.path.join(ROOT, "profiles", "r01t_pmc_traffic.json")  # tools/profile_round.sh + tools/pmc_traffic.py
    if os.path.exists(pmc) and args.workload == "m2" and args.scale == 1.0:
        with open(pmc) as f:
            t = json.load(f)
        traffic_ins, traffic_qry = t["groups"]["insert"], t["groups"]["query"]
    b_ins = 0.25 + p["q"] * 2 * G_BYTES     # RFO + write-back of one granule per probe
    b_chk = 0.375 + 6 * G_BYTES            # ~6 absent-edge probes per k-mer
    ach_ins = n_kmers * b_ins / (kms["insert"] * 1e-3) / 1e9
    ach_chk = n_kmers * b_chk / (kms["query"] * 1e-3) / 1e9
    out = {
        "metric": "kmers_hashed_per_sec", "value": n_kmers * args.steps / dt, "unit": "k-mers/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s: %d genomes x %d bp E. coli-like synthetic (twopaco_amd/synth.py), k=%d q=%d f=%d, 1 round"
                               % (args.workload, len(recs), recs[0].size, p["k"], p["q"], p["L"]),
                   "kmers": n_kmers, "filter_bytes": (1 << p["L"]) // 8, "insert_test_first": args.test_first},
        "junction_occurrences_per_sec": n_valid * args.steps / dt,
        "insert_kmers_per_sec": n_kmers / (kms["insert"] * 1e-3),
        "query_kmers_per_sec": n_kmers / (kms["query"] * 1e-3),
        "kernel_ms": kms,
        "result": {"candidate_marks": marks, "junctions": J, "junction_occurrences": n_valid, **st},
        "upload_s_pcie": upload_s,
        # the dominant kernel group of a step is the first-pass query (k_q_hash + k_q_split + k_q_lookup + k_q_verify)
        "roofline": {"bound": "hbm", "kernel": "first-pass query (k_q_hash, k_q_split, k_q_lookup, k_q_verify)", "achieved": ach_chk,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_chk / HBM_PEAK_GBS, "traffic": traffic_qry,
                     "algorithmic_bytes_per_kmer": b_chk, "launch_ms": kms["query"],
                     # the same launch priced with the bytes the PMC counters saw instead of the model's
                     "measured_hbm_GBs": (traffic_qry / (kms["query"] * 1e-3) / 1e9) if traffic_qry else None,
                     "measured_hbm_frac": (traffic_qry / (kms["query"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic_qry else None},
        # the north star's roofline kernel: first-pass Bloom insert
        "roofline_insert": {"bound": "hbm", "kernel": "first-pass insert (k_part_hash, k_part_split, k_part_apply)", "achieved": ach_ins,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_ins / HBM_PEAK_GBS, "traffic": traffic_ins,
                            "algorithmic_bytes_per_kmer": b_ins, "word_level_bytes_per_kmer": 0.25 + 8 * p["q"], "launch_ms": kms["insert"],
                            "measured_hbm_GBs": (traffic_ins / (kms["insert"] * 1e-3) / 1e9) if traffic_ins else None,
                            "measured_hbm_frac": (traffic_ins / (kms["insert"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic_ins else None},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(recs, p)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
