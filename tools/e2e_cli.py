#!/usr/bin/env python3
"""End-to-end CLI timing on a synthetic workload: writes the FASTA files, runs bin/twopaco, prints
wall time and junction occurrences per second.  python tools/e2e_cli.py [m1|m2] [threads]
E2E_EXTRA="--gpus 2 --emulate-ranks": extra CLI flags (e.g. the multi-GPU host with its ranks emulated on one device).
E2E_CONTIGS=N: every genome cut into N records of its file (contig-level assemblies: thousands of records, so the packer's
record index and the junction stream's stub / separator path are in the timing)."""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from twopaco_amd import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "m2"
threads = sys.argv[2] if len(sys.argv) > 2 else str(min(64, os.cpu_count()))
recs, p = synth.workload(wl)
tmp = tempfile.mkdtemp(dir="/tmp")
files = []
if os.environ.get("E2E_SINGLE_FILE"):  # every genome a record of one big file (exercises within-file parallel parsing)
    f = os.path.join(tmp, "all.fa")
    synth.write_fasta(f, recs)
    files.append(f)
else:
    contigs = int(os.environ.get("E2E_CONTIGS", "1"))
    for i, r in enumerate(recs):
        f = os.path.join(tmp, "g%d.fa" % i)
        if contigs > 1:
            cut = [len(r) * j // contigs for j in range(contigs + 1)]
            synth.write_fasta(f, [r[a:b] for a, b in zip(cut, cut[1:])], first_id=i * contigs)
        else:
            synth.write_fasta(f, [r], first_id=i)
        files.append(f)
exe = os.path.join(ROOT, "twopaco_amd", "bin", "twopaco")
out = os.path.join(tmp, "out.bin")
pause = float(os.environ.get("E2E_PAUSE", "3"))  # the driver releases the previous process's 30+ GiB asynchronously: measure isolated runs
for rep in range(int(os.environ.get("E2E_RUNS", "3"))):
    time.sleep(pause)
    out = os.path.join(tmp, "out%d.bin" % rep)  # a fresh file each time (truncating a cached 500 MB file costs ~90 ms)
    t0 = time.time()
    res = subprocess.run([exe, "-k", str(p["k"]), "-f", str(p["L"]), "-q", str(p["q"]), "-t", threads, "--seed", "20240229", "-o", out] + os.environ.get("E2E_EXTRA", "").split() + files,
                         env=dict(os.environ, TWOPACO_TIMING="1"), capture_output=True, text=True)
    wall = time.time() - t0
    occ = int(re.search(r"True marks count: (\d+)", res.stdout).group(1))
    print(res.stderr.strip())
    print("run %d: wall %.3f s, %d junction occurrences, %.2f M occ/s end to end, out.bin %d bytes" % (rep, wall, occ, occ / wall / 1e6, os.path.getsize(out)))
