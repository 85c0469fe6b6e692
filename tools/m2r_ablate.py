#!/usr/bin/env python3
"""Which of m2r's departures from m2 costs what: the step's kernel groups with none / each one / all of them applied.
python tools/m2r_ablate.py  ->  one JSON line"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from twopaco_amd import capi, synth

out = []
ALL = [(), ("families",), ("tracts",), ("strands",), ("contigs",), ("families", "tracts", "strands", "contigs")]
for feats in ([tuple(x for x in sys.argv[1].split(",") if x)] if len(sys.argv) > 1 else ALL):  # argv[1]: one feature list, e.g. "tracts" or ""
    recs, p = synth.workload("m2r", m2r_features=feats)
    text = capi.PackedText.from_codes(recs)
    n_kmers = synth.n_kmers(recs, p["k"])
    ctx = capi.Context(0)
    ctx.set_params(p["k"], p["L"], p["q"], capi.seed_table(p["q"], p["L"], seed=20240229))
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
    names = ["insert", "query", "fused", "filter2", "emit"]
    kms = {n: 0.0 for n in names}
    for _ in range(3):
        marks, J = step()
        for n in names:
            kms[n] += max(ctx.kernel_ms(n), 0.0) / 3
    out.append({"features": list(feats), "kmers": n_kmers, "marks": marks, "junctions": J, "insert_overflow": ctx.stat("insert_overflow_entries"),
                "query_overflow": ctx.stat("query_overflow_entries"), "kernel_ms": {k: round(v, 2) for k, v in kms.items()}})
    ctx.close()
    del text
print(json.dumps(out))
