#!/bin/bash
# Per-kernel times of one bench.py build: rocprofv3 --kernel-trace --stats over a short run, summary to gpurun_out/<tag>_kernel_stats.csv
# Usage (GPU box, repo root): bash tools/quick_prof.sh <tag> [bench args]
set -e
tag=${1:-q}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_k -o k -- python3 $root/bench.py --steps 4 --warmup 1 --cpu-baseline none --e2e-runs 0 "$@" > $out/${tag}_bench_under_prof.json 2> $out/${tag}_k.err
cd $root
db=$(find $out/prof_${tag}_k -name "*.db" | head -1)
test -n "$db"
python3 tools/prof_summary.py $db > $out/${tag}_kernel_stats.csv
head -16 $out/${tag}_kernel_stats.csv
