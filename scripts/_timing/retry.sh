#!/bin/bash
# scripts/_timing/retry.sh TIMEOUT SCRIPT : gpurun with retries while the pod's GPU slots are busy
for i in $(seq 1 20); do
  out=$(gpurun --timeout $1 -- bash $2 2>&1 | tail -3)
  echo "$out"
  echo "$out" | grep -q "status=transient\|slot(s) on this pod are busy" || exit 0
  sleep 60
done
