#!/bin/bash
# A/B timing of pair-beam kernel variants in ONE gpurun call: scripts/ab.sh n W variant1 variant2 ...  ("base" = the regular build)
n=$1; W=$2; shift 2
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then lib=poreover_amd/libporeover_hip.so; else lib=scripts/variants/libporeover_hip_$v.so; fi
  echo -n "$v: "; POREOVER_HIP_LIB=$lib timeout 200 python scripts/quick_time_2d.py $n $W 2>&1 | tail -1
done; done
