# A/B of the pipelined host layer on the 10 000-pair end-to-end job: waves in flight x wave size x first-wave ramp
for cfg in "3 2500 0" "3 2500 1250" "3 2500 640" "3 2800 0" "3 3000 0" "3 3334 0" "3 3334 1700" "3 2200 0"; do
  set -- $cfg
  echo -n "slots=$1 wave=$2 ramp=$3: "
  PO_PIPELINE_SLOTS=$1 PO_WAVE_PAIRS=$2 PO_WAVE_RAMP=$3 timeout 300 python bench.py --steps 1 --warmup 1 --no_secondary --cpu_sample 0 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['strong_scaling']; print(s['pairs_per_s'], s['seconds'], {k: round(v,1) for k,v in s['pipeline_rank0'].items()})"
done
