#!/bin/bash
# per-phase timers of beam2d_reg_kernel (-DPO_REG_TIMING builds) on bench.py's secondary legs: scripts/timing_secondary.sh LABEL variant...
cd ${GRAFT_REPO_ROOT:-/root/repo}
label=$1; shift
for v in "$@"; do
  export POREOVER_HIP_LIB=$PWD/scripts/variants/libporeover_hip_$v.so
  echo "==== $v"
  timeout 600 python bench.py --steps 1 --warmup 1 --no_strong --cpu_sample 0 2>&1 | grep -v "^{" | grep -A5 "po_reg_timing"
done > gpurun_out/timing2_$label.log 2>&1
wc -l gpurun_out/timing2_$label.log
