#!/bin/bash
# A/B of an environment switch on the headline AND the end-to-end leg: scripts/ab_env_e2e.sh LABEL "VAR=a" ...   ("-" = nothing set)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
label=$1; shift
for rep in 1 2; do
for v in "$@"; do
  echo -n "$v: "
  ( [ "$v" != "-" ] && export $v; timeout 400 python bench.py --steps 3 --warmup 1 --no_secondary --cpu_sample 0 2>gpurun_out/ab_env_err.log | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['strong_scaling']; print(d['value'], d['ms_per_step'], 'align', d['stage_ms_per_step']['align_envelope'], 'e2e', s['pairs_per_s'], s['seconds'], 'first call', s['first_call_s'], 'shard', s['shard_1250_e2e']['pairs_per_s'], 'mismatches', d['parity_check']['digest_mismatches'])" )
done; done > gpurun_out/ab_env_e2e_$label.log 2>&1
cat gpurun_out/ab_env_e2e_$label.log
