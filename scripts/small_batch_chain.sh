#!/bin/bash
# pair-beam kernel time at small launch sizes, serial chain vs closed form (PO_CHAIN_CLOSED): scripts/small_batch_chain.sh LABEL
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in serial closed_form; do
  if [ "$v" = closed_form ]; then export PO_CHAIN_CLOSED=1; else unset PO_CHAIN_CLOSED; fi
  echo "== $v"
  PO_ROUTES=reg timeout 600 python scripts/small_batch.py 1 16 256 1250 2500 4096 2>&1 | tail -6
done > gpurun_out/small_batch_chain_$1.log 2>&1
cat gpurun_out/small_batch_chain_$1.log
