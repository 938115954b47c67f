#!/bin/bash
# A/B of the chain mode (po_set_chain_mode / PO_CHAIN_CLOSED) on the bench's own 10 000 distinct pairs, ONE gpurun call:
#   scripts/ab_chain.sh LABEL [reps]     -> gpurun_out/ab_chain_LABEL.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
label=$1; reps=${2:-2}
for rep in $(seq 1 $reps); do
for v in closed_form serial; do
  if [ "$v" = closed_form ]; then export PO_CHAIN_CLOSED=1; else unset PO_CHAIN_CLOSED; fi
  echo -n "$v: "
  timeout 400 python bench.py --steps 4 --warmup 1 --no_secondary --no_strong --cpu_sample 0 2>gpurun_out/ab_chain_err.log | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'kernel', d['roofline']['avg_launch_ms'], 'stage', d['roofline']['stage_ms'], 'parity', d['parity_check'])"
done; done > gpurun_out/ab_chain_$label.log 2>&1
cat gpurun_out/ab_chain_$label.log
