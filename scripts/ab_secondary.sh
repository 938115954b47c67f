#!/bin/bash
# A/B of library builds on bench.py's SECONDARY legs (Bonito / row / W = 10 / flip-flop pairs, 1-D): scripts/ab_secondary.sh LABEL [variant ...]
# ("main" = the in-tree library; others = scripts/variants/libporeover_hip_NAME.so).  Writes gpurun_out/ab2_LABEL.log.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
label=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset POREOVER_HIP_LIB; else export POREOVER_HIP_LIB=$PWD/scripts/variants/libporeover_hip_$v.so; fi
  echo "$v:"
  timeout 600 python bench.py --steps 2 --warmup 1 --no_strong --cpu_sample 0 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('   headline', d['value'], 'kernel', d['roofline']['avg_launch_ms'])
for k,v in d['secondary'].items():
    if k.startswith('pair'): print('   ', k, v['pairs_per_s'], 'kernel', v['pair_beam_kernel_ms'], 'mismatches', v['parity_check']['mismatches'])
    elif isinstance(v,dict): print('   ', k, v.get('kernel_ms'))
    else: print('   ', k, v)"
done; done > gpurun_out/ab2_$label.log 2>&1
cat gpurun_out/ab2_$label.log
