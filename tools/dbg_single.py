import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
from twopaco_amd import capi
exec(open("/root/repo/tools/dbg_shard.py").read().split("if __name__")[0])
sp = specs[int(sys.argv[1]) if len(sys.argv) > 1 else 2]
letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
code_of = np.zeros(256, dtype=np.uint8); code_of[letters] = np.arange(5, dtype=np.uint8)
text = capi.PackedText.from_codes([code_of[np.frombuffer(r, dtype=np.uint8)] for r in sp["records"]])
ctx = capi.Context(0)
for opt, val in sp["options"].items(): ctx.set_option(opt, val)
ctx.set_option("insert_mode", 2); ctx.set_option("query_mode", 2)
ctx.set_params(sp["k"], sp["L"], sp["q"], capi.seed_table(sp["q"], sp["L"], seed=sp["seed"]))
ctx.seq_upload(text)
for lo, hi in sp["ranges"]:
    ctx.filter_reset(); ctx.pass1_insert(lo, hi, count=False)
    print("marks", ctx.pass1_query(lo, hi), "paths", ctx.stat("insert_path"), ctx.stat("query_path"), flush=True)
