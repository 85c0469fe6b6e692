#!/bin/bash
# SQ counter passes over one bench step (instruction mix, issue stalls, LDS conflicts) -> gpurun_out/<tag>_sq.csv
# Usage (GPU box, repo root): bash tools/pmc_sq.sh <tag>
set -uo pipefail
tag=${1:-sq}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
light="--steps 1 --warmup 0 --cpu-baseline none --e2e-runs 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $out/prof_${tag}_sq1 -o a -- python3 $root/bench.py $light > /dev/null 2> $out/${tag}_sq1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace -d $out/prof_${tag}_sq2 -o b -- python3 $root/bench.py $light > /dev/null 2> $out/${tag}_sq2.err
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_FLAT --kernel-trace -d $out/prof_${tag}_sq3 -o c -- python3 $root/bench.py $light > /dev/null 2> $out/${tag}_sq3.err
cd $root
python3 tools/pmc_sq.py $(find $out/prof_${tag}_sq1 $out/prof_${tag}_sq2 $out/prof_${tag}_sq3 -name "*.db") > $out/${tag}_sq.csv
cat $out/${tag}_sq.csv | head -12
