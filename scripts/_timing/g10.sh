cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g10_small.log 2>&1
POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_regw5.so PO_DEBUG_OCC=1 PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 5120 10000 > gpurun_out/r04_g10_small_w5.log 2>&1
for cfg in "3 2500 0" "3 3334 0" "2 5000 0" "3 5000 0" "4 2000 0" "3 2500 1250" "3 3334 1250" "4 2500 0"; do
  set -- $cfg
  echo -n "slots=$1 wave=$2 ramp=$3: "
  PO_PIPELINE_SLOTS=$1 PO_WAVE_PAIRS=$2 PO_WAVE_RAMP=$3 timeout 300 python bench.py --steps 1 --warmup 1 --no_secondary --cpu_sample 0 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['strong_scaling']; print(s['pairs_per_s'], s['seconds'], s['seconds_min'], {k: round(v,1) for k,v in s['pipeline_rank0'].items()})"
done > gpurun_out/r04_g10_e2e.log 2>&1
exit 0
