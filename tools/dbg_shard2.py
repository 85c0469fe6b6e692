"""trial `sys.argv[1]` of tests/test_gpu_addr_shard.py::test_address_sharded_randomized[4], with the overflow / survivor counts of every rank"""
import sys, os, tempfile, pathlib
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
exec(open("/root/repo/tools/dbg_shard.py").read().split("if __name__")[0])
from twopaco_amd import capi
_h, _a = capi.Context.shard_hash, capi.Context.shard_apply
def sh(self, which, batch, *a, **k):
    n = _h(self, which, batch, *a, **k); print("DBG pid", os.getpid(), "hash pass", which, "batch", batch, "overflow", n, flush=True); return n
def sa(self, which, batch, *a, **k):
    n = _a(self, which, batch, *a, **k); print("DBG pid", os.getpid(), "apply pass", which, "batch", batch, "survivors", n, flush=True); return n
capi.Context.shard_hash, capi.Context.shard_apply = sh, sa
if __name__ == "__main__":
    from test_gpu_addr_shard import run
    sp = specs[int(sys.argv[1])]
    try:
        run([sp], world, pathlib.Path(tempfile.mkdtemp()))
        print("ok")
    except Exception as e:
        msg = str(e); j = msg.find("RuntimeError: twopaco"); print("FAILED", msg[j:j + 300])
