#!/usr/bin/env python3
"""Development: one golden case through the combined exchange at a given world size; prints where a rank's filter differs from the oracle's.
   python tools/dev_combined.py rand6_k9_fp 8 2 gather"""
import os
import sys
import tempfile
import pathlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_combined as T
    name, sb, world, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None
    tmp = pathlib.Path(tempfile.mkdtemp())
    spec, o = T.golden_spec(name, sb, tmp, mode=mode, peek=len(sys.argv) > 5)
    g = T.run(spec, world, tmp)
    lo, hi = spec["ranges"][0]
    o.fill_only(lo, hi)
    marks = o.check_only(lo, hi)
    words = 1 << (sb - 5)
    for r, gr in enumerate(g):
        f = gr["rounds"][0]["filter"]
        bad = np.nonzero(f != o.filter)[0]
        print("rank", r, "combine", gr["rounds"][0]["combine"], "fused", gr["rounds"][0]["fused"])
        print("  words differing:", bad.size, "of", f.size, "; missing bits", int(np.bitwise_count(o.filter & ~f).sum()), "extra bits", int(np.bitwise_count(f & ~o.filter).sum()),
              "set in oracle", int(np.bitwise_count(o.filter).sum()))
        if bad.size:
            sl = np.unique(bad // words)
            print("  slices affected:", sl.size, "of", f.size // words, "first", sl[:16])
        m = gr["rounds"][0]["mask"]
        print("  mask bits", int(np.bitwise_count(m).sum()), "oracle marks", marks)


if __name__ == "__main__":
    main()
