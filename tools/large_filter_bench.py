#!/usr/bin/env python3
"""First-pass insert / query on the M2 text for filters of 2^36 .. 2^40 bits: LDS write-combining (two levels up to
f=38, three beyond) against the direct scattered kernels.  python tools/large_filter_bench.py [L ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twopaco_amd import capi, synth
Ls = [int(x) for x in sys.argv[1:]] or [36, 38, 39, 40]
recs, p = synth.workload("m2")
text = capi.PackedText.from_codes(recs)
n = synth.n_kmers(recs, p["k"])
for L in Ls:
    for mode in (0, 2, 1):
        ctx = capi.Context(0)
        ctx.set_option("insert_mode", mode)
        ctx.set_option("query_mode", mode)
        ctx.set_params(p["k"], L, p["q"], capi.seed_table(p["q"], L, seed=12345))
        ctx.seq_upload(text)
        best = None
        for rep in range(2):
            ctx.run_begin(); ctx.filter_reset()
            ctx.pass1_insert(count=False)
            marks = ctx.pass1_query()
            t = (ctx.kernel_ms("insert"), ctx.kernel_ms("query"), max(ctx.kernel_ms("filter_reset"), 0.0))
            best = t if best is None or sum(t) < sum(best) else best
        print("f=%d %-11s insert %7.2f ms (%5.2f G k-mers/s)  query %7.2f ms  reset %5.2f ms  paths %d/%d batches %d/%d marks %d" % (
            L, {0: "auto", 2: "partitioned", 1: "direct"}[mode], best[0], n / best[0] / 1e6, best[1], best[2], ctx.stat("insert_path"), ctx.stat("query_path"),
            ctx.stat("insert_batches"), ctx.stat("query_batches"), marks), flush=True)
        ctx.close()
