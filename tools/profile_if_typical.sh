#!/bin/bash
# The HBM-bound kernels read 3-5 % slower on some boxes of the pool (same build, same input).  Takes the round's profile set
# (tools/profile_round.sh, tools/pmc_sq.sh) only on a box whose plain step time is below the given threshold, so that the
# committed set is comparable with the earlier rounds' (the box's step time is printed either way).
# Usage: bash tools/profile_if_typical.sh <tag> <max ms per step>
set -euo pipefail
tag=$1; max=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
ms=$(python3 $root/bench.py --steps 10 --warmup 2 --e2e-runs 0 --cpu-baseline none 2>/dev/null | python3 -c "import json,sys; print([json.loads(l)['ms_per_step'] for l in sys.stdin if l.startswith('{')][0])")
echo "this box: $ms ms per step (threshold $max)"
if python3 -c "import sys; sys.exit(0 if float('$ms') <= float('$max') else 1)"; then
    bash $root/tools/profile_round.sh $tag
    bash $root/tools/pmc_sq.sh $tag
    rm -rf $root/gpurun_out/prof_${tag}_*
    echo "profile set taken"
else
    echo "skipped"
fi
