#!/bin/bash
# rocprofv3 passes behind profiles/: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own runs.
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -e
tag=${1:-r01k}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_k -o k -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_under_prof.json 2> $out/${tag}_k.err || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/prof_${tag}_f -o f -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/${tag}_f.err || true
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/prof_${tag}_w -o w -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/${tag}_w.err || true
cd $root
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || true
find $out -name "*.db" | head
