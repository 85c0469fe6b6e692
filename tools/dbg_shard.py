import sys, os, tempfile, pathlib
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
world = 4
rng = np.random.default_rng(99 + world)
alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
specs = []
for trial in range(10):
    k = int(rng.choice([5, 9, 15, 25, 31, 33]))
    L = int(rng.integers(14, 24))
    q = int(rng.integers(1, 8))
    slice_bits = int(rng.integers(6, min(12, L - 6) + 1))
    base = alphabet[rng.integers(0, 4, int(rng.integers(3000, 50000)))].copy()
    recs = []
    for r in range(int(rng.integers(1, 5))):
        s = base.copy()
        hits = rng.random(s.size) < 0.02
        s[hits] = alphabet[rng.integers(0, 4, int(hits.sum()))]
        if rng.random() < 0.5:
            a = int(rng.integers(0, s.size)); s[a:a + int(rng.integers(1, 60))] = ord("N")
        if rng.random() < 0.2:
            s[:int(rng.integers(1, s.size // 4))] = ord("A")
        recs.append(s.tobytes())
    size = 1 << L
    cut = sorted(int(x) for x in rng.integers(0, size, 2))
    specs.append({"records": recs, "k": k, "L": L, "q": q, "seed": int(rng.integers(1, 1 << 40)), "ranges": [(0, size), (cut[0], cut[1])],
                  "abundance": (1 << 64) - 1, "compact_exchange": trial % 3 != 2,
                  "options": {"slice_bits": slice_bits, "part_min_tiles": 1, "part_budget_bytes": int(rng.choice([40 << 30, 1 << 20]))}})
if __name__ == '__main__':
    from test_gpu_addr_shard import run
    for i, sp in enumerate(specs):
        d = {k: v for k, v in sp.items() if k != "records"}
        d["n"] = [len(r) for r in sp["records"]]
        try:
            run([sp], world, pathlib.Path(tempfile.mkdtemp()))
            print("trial", i, "ok", d, flush=True)
        except Exception as e:
            msg = str(e)
            j = msg.find("twopaco_hip")
            print("trial", i, "FAILED", d, msg[j:j+300].replace("\n", " "), flush=True)
