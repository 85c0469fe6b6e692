#!/bin/bash
# Same-box A/B of two builds of the library: per-kernel times of `python3 <script> <args>` under rocprofv3 for each TPC_LIB_DIR.
# Usage (GPU box, repo root): bash tools/ab_libs.sh <tag> <libdirA> <libdirB> <script> [args...]
tag=$1; A=$2; B=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for side in A B; do
  dir=$A; [ $side = B ] && dir=$B
  export TPC_LIB_DIR=$root/$dir
  rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_$side -o k -- python3 $root/"$@" > $out/${tag}_$side.out 2> $out/${tag}_$side.err
  python3 $root/tools/prof_summary.py $(find $out/prof_${tag}_$side -name "*.db" | head -1) > $out/${tag}_${side}_kernel_stats.csv
  echo "== $dir"; head -8 $out/${tag}_${side}_kernel_stats.csv
done
