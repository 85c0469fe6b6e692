#!/usr/bin/env python3
"""End-to-end CLI timing on the config-4-shaped workload (7 x 160 Mbp, k=25, f=38): python tools/e2e_m3.py"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twopaco_amd import synth
t0 = time.time()
recs, p = synth.workload("m3")
tmp = tempfile.mkdtemp(dir="/tmp")
files = []
for i, r in enumerate(recs):
    f = os.path.join(tmp, "g%d.fa" % i)
    synth.write_fasta(f, [r], first_id=i)
    files.append(f)
print("generated + wrote %d files in %.1f s" % (len(files), time.time() - t0))
exe = os.path.join(ROOT, "twopaco_amd", "bin", "twopaco")
for rep in range(2):
    out = os.path.join(tmp, "out%d.bin" % rep)
    t0 = time.time()
    res = subprocess.run([exe, "-k", str(p["k"]), "-f", str(p["L"]), "-q", str(p["q"]), "-t", "64", "--seed", "20240229", "-o", out] + files,
                         env=dict(os.environ, TWOPACO_TIMING="1"), capture_output=True, text=True)
    wall = time.time() - t0
    occ = int(re.search(r"True marks count: (\d+)", res.stdout).group(1))
    print(res.stderr.strip())
    print("run %d: wall %.3f s, %d junction occurrences, %.2f M occ/s, %.2f G k-mers/s end to end" % (rep, wall, occ, occ / wall / 1e6, sum(r.size for r in recs) / wall / 1e9))
    time.sleep(3)
