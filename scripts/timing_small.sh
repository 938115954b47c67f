#!/bin/bash
# per-phase timers of pair slot 0 (-DPO_REG_TIMING variant) with the device nearly empty (16 pairs) and full (10 000 pairs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export POREOVER_HIP_LIB=$PWD/scripts/variants/libporeover_hip_timing.so
for n in 16 10000; do
  echo "== $n pairs"
  timeout 400 python bench.py --pairs $n --steps 1 --warmup 0 --no_secondary --no_strong --cpu_sample 0 2>&1 | grep -A12 "po_reg_timing" | head -13
done > gpurun_out/timing_small_$1.log 2>&1
cat gpurun_out/timing_small_$1.log
