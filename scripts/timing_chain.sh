#!/bin/bash
# per-phase timers of pair slot 0 (-DPO_REG_TIMING build: scripts/build_file_variant.sh po_beam2d_reg timing -DPO_REG_TIMING) in both chain modes
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export POREOVER_HIP_LIB=$PWD/scripts/variants/libporeover_hip_timing.so
for v in closed_form serial; do
  if [ "$v" = closed_form ]; then export PO_CHAIN_CLOSED=1; else unset PO_CHAIN_CLOSED; fi
  echo "== $v"
  timeout 400 python bench.py --steps 1 --warmup 0 --no_secondary --no_strong --cpu_sample 0 2>&1 | grep -A12 "po_reg_timing" | head -14
done > gpurun_out/timing_chain_$1.log 2>&1
cat gpurun_out/timing_chain_$1.log
