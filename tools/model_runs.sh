#!/bin/bash
# The link model's inputs on one box: bench.py at one GPU, then N = 2, 4, 8 ranks taking turns on the ONE device
# (gloo rendezvous, a file lock around every library call: a call's time is that of a rank alone on its GPU).
# Usage (on the GPU box, from the repo root): bash tools/model_runs.sh <tag> [steps]
set -uo pipefail
tag=${1:-r06}
steps=${2:-4}
out=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
mkdir -p $out
python3 bench.py --cpu-baseline none --e2e-runs 0 > $out/${tag}_model_n1.json 2> $out/${tag}_model_n1.err
for n in 2 4 8; do
    rm -f /tmp/tpc_device.lock
    TPC_DIST_BACKEND=gloo TPC_DIST_SERIALIZE=/tmp/tpc_device.lock TPC_E2E_EMULATE_RANKS=1 \
        python3 bench.py --gpus $n --steps $steps --warmup 1 --no-cpu-baseline --e2e-runs 0 > $out/${tag}_model_n$n.json 2> $out/${tag}_model_n$n.err || echo "N=$n failed"
done
python3 - $out $tag <<'P'
import json, sys
out, tag = sys.argv[1:3]
one = json.load(open("%s/%s_model_n1.json" % (out, tag)))["ms_per_step"]
print("one GPU: %.2f ms" % one)
for n in (2, 4, 8):
    try:
        d = json.load(open("%s/%s_model_n%d.json" % (out, tag, n)))
    except Exception as e:
        print(n, "no line", e)
        continue
    m = d["model"]
    print("N=%d compute %.2f wire %.2f (%s) hidden %.2f -> %.2f ms = %.2fx (no overlap %.2f = %.2fx); equals golden: %s" % (
        n, m["compute_ms"], m["wire_ms_total"], ", ".join("%.2f" % v for v in m["wire_ms"].values()), m["query_hash_and_binning_ms_under_the_exchange"],
        m["predicted_ms"], one / m["predicted_ms"], m["predicted_ms_no_overlap"], one / m["predicted_ms_no_overlap"], d.get("result_equals_reference_golden")))
    print("   ", json.dumps({k: round(v, 2) for k, v in d["call_ms_rank0_per_step"].items()}))
P
