#!/usr/bin/env python3
"""Timeline of the LAST bench step in a rocprofv3 --kernel-trace result (rocpd sqlite): every kernel in start order with its
offset, duration and the idle gap in front of it -- where a step's time goes besides the big kernels.
   python tools/prof_timeline.py gpurun_out/prof_x_k/k_results.db [first-kernel-of-a-step substring]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_part_hash"
rows = list(c.execute("select name,start,end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if anchor in r[0]]
if len(starts) < 2:
    sys.exit("fewer than two steps found")
a, b = starts[-2], starts[-1]   # the last complete step: from its first kernel to the next step's first kernel
step = rows[a:b]
t0 = step[0][1]
prev_end = t0
gaps = busy = 0.0
small = 0.0
print("%-44s %10s %10s %9s" % ("kernel", "at_us", "dur_us", "gap_us"))
for name, s, e in step:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short).replace("void ", "")[:44]
    gap = (s - prev_end) / 1e3
    print("%-44s %10.1f %10.1f %9.1f" % (short, (s - t0) / 1e3, (e - s) / 1e3, gap))
    if gap > 0: gaps += gap
    busy += (e - s) / 1e3
    if (e - s) / 1e3 < 500: small += (e - s) / 1e3
    prev_end = max(prev_end, e)
tail = (rows[b][1] - prev_end) / 1e3
print("step %.1f us: kernels %.1f us (of which < 0.5 ms each: %.1f us), gaps inside %.1f us, gap before the next step %.1f us"
      % ((rows[b][1] - t0) / 1e3, busy, small, gaps, tail))
