# M2 through the CLI with 1 / 2 / 4 rounds: per-round timers and identical output bytes.  bash tools/rounds_m2.sh (GPU box, repo root)
set -e
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from twopaco_amd import synth
recs, p = synth.workload("m2")
os.makedirs("/tmp/m2fa", exist_ok=True)
for i, r in enumerate(recs):
    synth.write_fasta("/tmp/m2fa/g%d.fa" % i, [r], first_id=i)
PY
for r in 1 2 4; do
  sleep 3
  TWOPACO_TIMING=1 twopaco_amd/bin/twopaco -k 25 -f 36 -r $r -t 64 --seed 20240229 -o /tmp/m2_r$r.bin /tmp/m2fa/*.fa 2> /tmp/err_r$r.txt | grep -E "True junctions|marks count" | tr "\n" " "; echo; grep "split\|round:\|rounds (\|output complete\|histogram" /tmp/err_r$r.txt | tr "\n" ";" ; echo " rounds=$r"
done
cmp /tmp/m2_r1.bin /tmp/m2_r2.bin && cmp /tmp/m2_r1.bin /tmp/m2_r4.bin && echo "r=1,2,4 outputs identical"
