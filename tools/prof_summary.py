#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel calls / total / average duration.
   python tools/prof_summary.py gpurun_out/prof_x/x_results.db > profiles/x_kernel_stats.csv"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
print("kernel,calls,total_ms,avg_ms,percent")
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short).replace("void ", "")
    if "rocprim" in short:
        short = "rocprim::" + ("radix_sort_onesweep" if "onesweep_iteration" in name else "radix_sort_histogram")
    print('"%s",%d,%.3f,%.3f,%.3f' % (short, calls, total / 1e3, avg / 1e3, pct))  # names hold commas: quoted
