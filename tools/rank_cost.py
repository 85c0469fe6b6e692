#!/usr/bin/env python3
"""Per-rank cost of the vertex-hash-range decomposition on ONE GPU: times the round of rank r of N
(gated insert + query + exact filter) for N = 1, 2, 4, 8.  python tools/rank_cost.py [m2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from twopaco_amd import capi, synth
from twopaco_amd.dist import vertex_hash_ranges
wl = sys.argv[1] if len(sys.argv) > 1 else "m2"
recs, p = synth.workload(wl)
text = capi.PackedText.from_codes(recs)
ctx = capi.Context(0)
ctx.set_params(p["k"], p["L"], p["q"], capi.seed_table(p["q"], p["L"], seed=12345))
ctx.seq_upload(text)
names = ["insert", "query", "compact", "filter2", "scan2"]
for N in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8)):
    for r in sorted(set([0, N // 2, N - 1])):
        lo, hi = vertex_hash_ranges(p["L"], N)[r]
        for rep in range(2):
            ctx.run_begin()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.filter_reset(); ctx.pass1_insert(lo, hi, count=False); marks = ctx.pass1_query(lo, hi); st = ctx.pass2_filter()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("N=%d rank %d: round %.2f ms  marks %d  " % (N, r, dt * 1e3, marks) + " ".join("%s %.2f" % (n, ctx.kernel_ms(n)) for n in names), flush=True)
ctx.close(); text.close()
