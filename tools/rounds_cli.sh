set -e
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from twopaco_amd import synth
recs, p = synth.workload("m1")
os.makedirs("/tmp/m1fa", exist_ok=True)
for i, r in enumerate(recs):
    synth.write_fasta("/tmp/m1fa/g%d.fa" % i, [r], first_id=i)
PY
for r in 1 3; do
  s=$(date +%s.%N); twopaco_amd/bin/twopaco -k 25 -f 32 -r $r -t 16 --seed 12345 -o /tmp/m1_r$r.bin /tmp/m1fa/*.fa | grep -E "Round |True junctions|Distinct|marks count" | tr "\n" " "; e=$(date +%s.%N); echo " rounds=$r wall $(python3 -c "print(round($e - $s, 3))") s"
done
cmp /tmp/m1_r1.bin /tmp/m1_r3.bin && echo "r=1 and r=3 outputs identical"
sha256sum /tmp/m1_r1.bin
