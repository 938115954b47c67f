#!/bin/bash
# A/B of an environment switch on the bench's own 10 000 distinct pairs, ONE gpurun call:
#   scripts/ab_env.sh LABEL "VAR=a" "VAR=b" ...   -> gpurun_out/ab_env_LABEL.log   ("-" = nothing set)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
label=$1; shift
for rep in 1 2; do
for v in "$@"; do
  echo -n "$v: "
  ( [ "$v" != "-" ] && export $v; timeout 400 python bench.py --steps 4 --warmup 1 --no_secondary --no_strong --cpu_sample 0 2>gpurun_out/ab_env_err.log | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'stages', d['stage_ms_per_step'], 'kernel', d['roofline']['avg_launch_ms'], 'stage', d['roofline']['stage_ms'], 'mismatches', d['parity_check']['digest_mismatches'])" )
done; done > gpurun_out/ab_env_$label.log 2>&1
cat gpurun_out/ab_env_$label.log
