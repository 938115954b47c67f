#!/bin/bash
# A/B on the bench's own 10 000 distinct pairs, in ONE gpurun call: scripts/ab_bench.sh LABEL [variant ...]
#   "main" = the in-tree library; any other name = scripts/variants/libporeover_hip_NAME.so (scripts/build_file_variant.sh builds them;
#   the directory is git-ignored, the .so files travel with the gpurun snapshot).  Writes gpurun_out/ab_LABEL.log.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
label=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset POREOVER_HIP_LIB; else export POREOVER_HIP_LIB=$PWD/scripts/variants/libporeover_hip_$v.so; fi
  echo -n "$v: "
  timeout 400 python bench.py --steps 4 --warmup 1 --no_secondary --no_strong --cpu_sample 0 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'kernel', d['roofline']['avg_launch_ms'], 'stage', d['roofline']['stage_ms'], 'mismatch', d['parity_check']['digest_mismatches'])"
done; done > gpurun_out/ab_$label.log 2>&1
cat gpurun_out/ab_$label.log
