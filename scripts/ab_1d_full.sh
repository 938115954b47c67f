#!/bin/bash
# A/B of library builds on the 1-D beam search, 1 000 reads (one wave per SIMD) and 10 000 reads (the device full): scripts/ab_1d_full.sh LABEL variant ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
label=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset POREOVER_HIP_LIB; else export POREOVER_HIP_LIB=$PWD/scripts/variants/libporeover_hip_$v.so; fi
  echo "== $v"
  for n in 1000 10000; do
    timeout 300 python scripts/quick_time_1d.py $n 2>&1 | grep "^beam1d" | tail -2
    timeout 300 python scripts/quick_time_1d.py $n flipflop 2>&1 | grep "W=10" | tail -1
  done
done; done > gpurun_out/ab_1d_$label.log 2>&1
cat gpurun_out/ab_1d_$label.log
