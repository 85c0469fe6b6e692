#!/usr/bin/env python3
"""What ONE rank of the vertex-hash-range decomposition costs per step at N = 1, 2, 4, 8 ranks, measured on one GPU: rank r of N
hashes the whole text behind the gate of its range and carries 1/N of the entries, there is no data-path exchange (only the
all-gather of the junction keys, ~12 MB on the 62-genome workload, which this tool leaves out: each range keeps its own keys), so
the slowest rank's time IS the N-rank step but for that all-gather.  python tools/ranges_rank_time.py [m2] [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from twopaco_amd import capi, synth
from twopaco_amd.dist import vertex_hash_ranges

wl = sys.argv[1] if len(sys.argv) > 1 else "m2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
recs, p = synth.workload(wl)
n_kmers = synth.n_kmers(recs, p["k"])
ctx = capi.Context(0)
ctx.set_params(p["k"], p["L"], p["q"], capi.seed_table(p["q"], p["L"], seed=20240229))
ctx.seq_upload(capi.PackedText.from_codes(recs))
names = ["insert", "query", "fused", "compact", "filter2", "scan2", "sort", "emit"]


def step(lo, hi):
    ctx.run_begin()
    ctx.filter_reset()
    ctx.pass1_insert(lo, hi, count=False)
    marks = ctx.pass1_query(lo, hi)
    st = ctx.pass2_filter((1 << 64) - 1)
    J = ctx.junctions_finalize()
    ctx.emit()
    return marks, J


out = {"workload": wl, "kmers": n_kmers, "per_world": {}}
for world in [int(x) for x in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
    ranks = []
    only = os.environ.get("RANKS")
    for r, (lo, hi) in enumerate(vertex_hash_ranges(p["L"], world)):
        if only and str(r) not in only.split(","):
            continue
        step(lo, hi)
        torch.cuda.synchronize()
        kms = {n: 0.0 for n in names}
        t0 = time.perf_counter()
        for _ in range(steps):
            marks, J = step(lo, hi)
            for n in names:
                kms[n] += max(ctx.kernel_ms(n), 0.0) / steps
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        ranks.append({"rank": r, "ms_per_step": round(ms, 2), "marks": marks, "junctions": J, "kernel_ms": {k: round(v, 2) for k, v in kms.items()}})
    slow = max(x["ms_per_step"] for x in ranks)
    out["per_world"][world] = {"slowest_rank_ms": slow, "kmers_per_s": n_kmers / slow * 1e3, "speedup_vs_1": None, "ranks": ranks}
one = list(out["per_world"].values())[0]["slowest_rank_ms"]
for w in out["per_world"].values():
    w["speedup_vs_1"] = round(one / w["slowest_rank_ms"], 2)
print(json.dumps(out))
