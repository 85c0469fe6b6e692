#!/usr/bin/env python3
"""The whole two-pass path on one workload at several filter sizes (a smaller filter under the same text: more entries per slice,
a higher fill, more first-probe survivors): ms per step and per kernel group at every f.  python tools/f_sweep.py [m2] [28,30,...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from twopaco_amd import capi, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "m2"
Ls = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "28,30,32,34,36,38").split(",")]
recs, p = synth.workload(wl)
text = capi.PackedText.from_codes(recs)
n_kmers = synth.n_kmers(recs, p["k"])
names = ["insert", "query", "fused", "compact", "filter2", "scan2", "sort", "emit"]
out = []
for L in Ls:
    ctx = capi.Context(0)
    for kv in [x for x in os.environ.get("TPC_SET", "").split(",") if x]:  # TPC_SET=name=value,...: tpc_set_option
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.set_params(p["k"], L, p["q"], capi.seed_table(p["q"], L, seed=20240229))
    ctx.seq_upload(text)

    def step():
        ctx.run_begin()
        ctx.filter_reset()
        ctx.pass1_insert(count=False)
        marks = ctx.pass1_query()
        ctx.pass2_filter((1 << 64) - 1)
        J = ctx.junctions_finalize()
        ctx.emit()
        return marks, J

    step()
    torch.cuda.synchronize()
    kms = {n: 0.0 for n in names}
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        marks, J = step()
        for n in names:
            kms[n] += max(ctx.kernel_ms(n), 0.0) / steps
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out.append({"L": L, "ms_per_step": round(ms, 2), "G_kmers_per_s": round(n_kmers / ms / 1e6, 2), "marks": marks, "junctions": J,
                "insert_path": ctx.stat("insert_path"), "query_path": ctx.stat("query_path"), "query_batches": ctx.stat("query_batches"),
                "kernel_ms": {k: round(v, 2) for k, v in kms.items()}})
    ctx.close()
print(json.dumps({"workload": wl, "kmers": n_kmers, "sweep": out}))
