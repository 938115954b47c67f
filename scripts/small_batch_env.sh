#!/bin/bash
# pair-beam kernel time at small launch sizes under environment switches: scripts/small_batch_env.sh LABEL "VAR=a" "-" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
label=$1; shift
for v in "$@"; do
  echo "== $v"
  ( [ "$v" != "-" ] && export $v; PO_ROUTES=reg timeout 600 python scripts/small_batch.py 1 16 256 1250 4096 2>&1 | tail -5 )
done > gpurun_out/small_batch_env_$label.log 2>&1
cat gpurun_out/small_batch_env_$label.log
