#!/bin/bash
# A/B of library builds on the flip-flop Viterbi (1 000 / 10 000 reads, T = 4000) and the flip-flop pair leg's stages: scripts/ab_vit_ff.sh LABEL variant ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
label=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset POREOVER_HIP_LIB; else export POREOVER_HIP_LIB=$PWD/scripts/variants/libporeover_hip_$v.so; fi
  echo "== $v"
  for n in 1000 10000; do timeout 300 python scripts/quick_time_1d.py $n flipflop 2>&1 | grep "^viterbi"; done
done; done > gpurun_out/ab_vit_$label.log 2>&1
cat gpurun_out/ab_vit_$label.log
