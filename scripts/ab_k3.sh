#!/bin/bash
# A/B of library builds on the three-value models' W > 6 classes of beam2d_kernel (no bench leg): scripts/ab_k3.sh LABEL variant ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
label=$1; shift
for v in "$@"; do
  if [ "$v" = main ]; then unset POREOVER_HIP_LIB; else export POREOVER_HIP_LIB=$PWD/scripts/variants/libporeover_hip_$v.so; fi
  echo "== $v"
  for cfg in "2048 25 bonito row_col" "2048 25 bonito row" "2048 25 flipflop row_col" "2048 10 bonito row" "2048 10 flipflop row"; do
    PO_ROUTE=legacy timeout 300 python scripts/quick_time_2d.py $cfg 2>&1 | tail -1
  done
done > gpurun_out/ab_k3_$label.log 2>&1
cat gpurun_out/ab_k3_$label.log
