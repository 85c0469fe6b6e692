import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from twopaco_amd import capi, synth, dist as tdist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
backend = sys.argv[2] if len(sys.argv) > 2 else "gloo"
torch.cuda.set_device(0)
if backend == "nccl":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo", rank=0, world_size=1)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
recs, p = synth.workload("m2", scale=scale)
text = capi.PackedText.from_codes(recs)
ctx = capi.Context(0)
ctx.set_params(p["k"], p["L"], p["q"], capi.seed_table(p["q"], p["L"], seed=20240229))
ctx.seq_upload(text)
sh = tdist.AddressSharded(ctx, dist, torch.device("cuda", 0))
for rep in range(2):
    try:
        st = tdist.address_sharded_step(sh)
        print(backend, "step ok", {k: st[k] for k in ("true", "false", "marks", "junctions", "n_valid")}, sh.stats.get("survivors"))
    except Exception as e:
        print(backend, "failed:", e)
        break
dist.destroy_process_group()
