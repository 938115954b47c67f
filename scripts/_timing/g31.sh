cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "0 0 1250 0" "0 0 1250 800" "0 0 1250 1250" "0 0 1250 500" "0 4000 1250 800" "0 3000 1250 800" "3 3334 1250 800" "0 0 800 800"; do
  set -- $cfg
  echo -n "slots=$1 wave=$2 ramp=$3 tail=$4: "
  env $( [ "$1" != "0" ] && echo PO_PIPELINE_SLOTS=$1 ) $( [ "$2" != "0" ] && echo PO_WAVE_PAIRS=$2 ) PO_WAVE_RAMP=$3 PO_WAVE_TAIL=$4 timeout 300 python bench.py --steps 1 --warmup 1 --no_secondary --cpu_sample 0 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['strong_scaling']; print(s['pairs_per_s'], s['seconds'], s.get('pairs_per_s_min'), s.get('pairs_per_s_max'), {k: round(v,1) for k,v in s['pipeline_rank0'].items()})"
done > gpurun_out/r04_g31_e2e.log 2>&1
exit 0
