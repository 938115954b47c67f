# A/B on the bench's own 10 000 distinct pairs: scripts/_timing/ab_bench.sh LABEL [variant ...]  ("main" = the in-tree library)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
label=$1; shift
for v in "$@"; do
  if [ "$v" = main ]; then unset POREOVER_HIP_LIB; else export POREOVER_HIP_LIB=$GRAFT_REPO_ROOT/scripts/_timing/libporeover_hip_$v.so; fi
  echo -n "$v: "
  timeout 400 python bench.py --steps 4 --warmup 1 --no_secondary --no_strong --cpu_sample 0 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'kernel', d['roofline']['avg_launch_ms'], 'stage', d['roofline']['stage_ms'], 'mismatch', d['parity_check']['digest_mismatches'])"
done > gpurun_out/r04_ab_$label.log 2>&1
