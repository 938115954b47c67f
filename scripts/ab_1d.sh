#!/bin/bash
# A/B of library builds on the 1-D beam search (BASELINE configs 2 and 5: 1 000 reads, T = 4000, W = 10), ONE gpurun call:
#   scripts/ab_1d.sh LABEL variant ...    ("main" = the in-tree library)   -> gpurun_out/ab_1d_LABEL.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
label=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset POREOVER_HIP_LIB; else export POREOVER_HIP_LIB=$PWD/scripts/variants/libporeover_hip_$v.so; fi
  echo "== $v"
  timeout 300 python scripts/quick_time_1d.py 1000 2>&1 | grep -v "^viterbi" | tail -4
  timeout 300 python scripts/quick_time_1d.py 1000 flipflop 2>&1 | grep "W=10" | tail -2
done; done > gpurun_out/ab_1d_$label.log 2>&1
cat gpurun_out/ab_1d_$label.log
