#!/bin/bash
# rocprofv3 passes behind profiles/: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own runs (the guide's
# HBM section: separate --pmc passes), then the plain bench line of the same build.  Any failing step fails the script.
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -euo pipefail
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
light="--cpu-baseline none --e2e-runs 0"
rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_k -o k -- python3 $root/bench.py --steps 4 --warmup 1 $light > $out/${tag}_bench_under_prof.json 2> $out/${tag}_k.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/prof_${tag}_f -o f -- python3 $root/bench.py --steps 1 --warmup 0 $light > /dev/null 2> $out/${tag}_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/prof_${tag}_w -o w -- python3 $root/bench.py --steps 1 --warmup 0 $light > /dev/null 2> $out/${tag}_w.err
cd $root
kdb=$(find $out/prof_${tag}_k -name "*.db" | head -1); fdb=$(find $out/prof_${tag}_f -name "*.db" | head -1); wdb=$(find $out/prof_${tag}_w -name "*.db" | head -1)
test -n "$kdb" && test -n "$fdb" && test -n "$wdb"
python3 tools/prof_summary.py $kdb > $out/${tag}_kernel_stats.csv
python3 tools/pmc_traffic.py $fdb $wdb > $out/${tag}_pmc_traffic.json
# the bench line reads profiles/<PMC_PROFILE of bench.py>: put the counters just collected there BEFORE the line is taken, so that it
# carries `traffic` / `frac` of THIS build (round 4's kept line said "stale" because this copy came after)
pmc_name=$(python3 -c "import re;print(re.search(r'PMC_PROFILE = \"(.*?)\"', open('bench.py').read()).group(1))")
cp $out/${tag}_pmc_traffic.json profiles/$pmc_name
python3 bench.py "${@:2}" > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cp profiles/$pmc_name $out/$pmc_name
head -14 $out/${tag}_kernel_stats.csv
