#!/usr/bin/env python3
"""Larger-than-benchmark smoke: m2-like text scaled up (default 4x = 1.24 G positions, more than the 2^30 positions
one query batch can address), f=38: LDS write-combining passes in several batches against the direct kernels.
python tools/scale_smoke.py [scale] [L]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from twopaco_amd import capi, synth
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
L = int(sys.argv[2]) if len(sys.argv) > 2 else 38
t0 = time.time()
recs, p = synth.workload("m2", scale=scale)
n = synth.n_kmers(recs, p["k"])
print("generated %d records, %.3f G k-mers in %.1f s" % (len(recs), n / 1e9, time.time() - t0), flush=True)
text = capi.PackedText.from_codes(recs)
del recs
res = {}
for mode in (0, 1):
    ctx = capi.Context(0)
    ctx.set_option("insert_mode", mode)
    ctx.set_option("query_mode", mode)
    ctx.set_params(p["k"], L, p["q"], capi.seed_table(p["q"], L, seed=12345))
    ctx.seq_upload(text)
    ctx.run_begin(); ctx.filter_reset()
    t1 = time.time()
    ctx.pass1_insert(count=False)
    marks = ctx.pass1_query()
    st = ctx.pass2_filter()
    J = ctx.junctions_finalize()
    nm, nv = ctx.emit()
    wall = time.time() - t1
    mask = ctx.mask_download(False)
    res[mode] = (marks, st, J, nv, mask)
    print("%-6s insert %8.2f ms  query %8.2f ms  whole step %.3f s  paths %d/%d batches %d/%d  marks %d junctions %d occurrences %d" % (
        "auto" if mode == 0 else "direct", ctx.kernel_ms("insert"), ctx.kernel_ms("query"), wall, ctx.stat("insert_path"), ctx.stat("query_path"),
        ctx.stat("insert_batches"), ctx.stat("query_batches"), marks, J, nv), flush=True)
    ctx.close()
a, b = res[0], res[1]
assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[3] == b[3] and (a[4] == b[4]).all()
print("partitioned == direct: marks, counters, junctions, occurrences, mask")
