import os, sys
sys.path.insert(0, os.getcwd())
import torch
from twopaco_amd import capi, synth
recs, p = synth.workload("m1", scale=0.2)
text = capi.PackedText.from_codes(recs)
def free(): return torch.cuda.mem_get_info(0)[0]
f0 = free()
for rep in range(3):
    ctx = capi.Context(0)
    ctx.set_params(p["k"], 30, p["q"], capi.seed_table(p["q"], 30, seed=1))
    ctx.seq_upload(text)
    fr = []
    for step in range(30):
        ctx.run_begin(); ctx.filter_reset(); ctx.pass1_insert(count=False); ctx.pass1_query(); ctx.pass2_filter(); ctx.junctions_finalize(); ctx.emit()
        b, n = ctx.emit_stream(text.rec_start, text.rec_length)
        if step in (2, 29): fr.append(free())
    print("context %d: free after step 3 / 30: %.1f / %.1f MiB (delta %.1f)" % (rep, fr[0] / 2**20, fr[1] / 2**20, (fr[0] - fr[1]) / 2**20))
    ctx.close()
print("free before / after everything: %.1f / %.1f MiB" % (f0 / 2**20, free() / 2**20))
