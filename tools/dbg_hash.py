"""level-1 (hash) overflow counts of every rank of a 4-rank address-sharded insert / query, one process (no exchange)"""
import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
exec(open("/root/repo/tools/dbg_shard.py").read().split("if __name__")[0])
from twopaco_amd import capi
sp = specs[int(sys.argv[1]) if len(sys.argv) > 1 else 2]
letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
code_of = np.zeros(256, dtype=np.uint8); code_of[letters] = np.arange(5, dtype=np.uint8)
text = capi.PackedText.from_codes([code_of[np.frombuffer(r, dtype=np.uint8)] for r in sp["records"]])
for rank in range(4):
    ctx = capi.Context(0)
    for opt, val in sp["options"].items(): ctx.set_option(opt, val)
    ctx.shard_config(rank, 4)
    ctx.set_params(sp["k"], sp["L"], sp["q"], capi.seed_table(sp["q"], sp["L"], seed=sp["seed"]))
    ctx.seq_upload(text)
    lo, hi = sp["ranges"][0]
    for which in (0, 1):
        geom = ctx.shard_plan(which, lo, hi)
        send_r = torch.empty(4 * geom["region_block_bytes"], dtype=torch.uint8, device="cuda")
        send_c = torch.zeros(4 * geom["count_block_bytes"], dtype=torch.uint8, device="cuda")
        ctx.filter_reset()
        n = ctx.shard_hash(which, 0, send_r.data_ptr(), send_c.data_ptr(), lo, hi)
        cnt = send_c.view(torch.int32).cpu().numpy()
        print("rank", rank, "pass", which, "overflow", n, "entries in regions", int(cnt.sum()), "regions", cnt.size, "max", int(cnt.max()), {k: geom[k] for k in geom if k in ("b1", "b2", "cap1", "nwg1", "batches", "pos_per_round")}, flush=True)
    ctx.close()
