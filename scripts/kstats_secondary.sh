#!/bin/bash
# kernel durations of bench.py's secondary legs per library variant: scripts/kstats_secondary.sh LABEL variant...
root=${GRAFT_REPO_ROOT:-/root/repo}
label=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = main ]; then unset POREOVER_HIP_LIB; else export POREOVER_HIP_LIB=$root/scripts/variants/libporeover_hip_$v.so; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/ks2_${label}_$v -- python3 $root/bench.py --steps 1 --warmup 1 --no_strong --cpu_sample 0 > $root/gpurun_out/ks2_${label}_$v.log 2>&1
  f=$(find $root/gpurun_out/ks2_${label}_$v -name "*kernel_stats.csv" | head -1)
  echo "==== $v"; [ -n "$f" ] && cp $f $root/gpurun_out/ks2_${label}_$v.csv && grep "beam2d" $f | cut -c1-150
  rm -rf $root/gpurun_out/ks2_${label}_$v
done
