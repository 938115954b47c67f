#!/bin/bash
# A/B of the pipelined host layer on the 10 000-pair end-to-end job, in ONE gpurun call: waves in flight x wave size x first-wave
# ramp x last-wave size (scripts/cold_start.py: host float32 in -> strings out, three calls; the later two are printed).
#   scripts/e2e_ab.sh "SLOTS WAVE RAMP TAIL" ...      (0 = the engine's own choice)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  set -- $cfg
  echo -n "slots=$1 wave=$2 ramp=$3 tail=$4: "
  env $( [ "$1" != 0 ] && echo PO_PIPELINE_SLOTS=$1 ) $( [ "$2" != 0 ] && echo PO_WAVE_PAIRS=$2 ) $( [ "$3" != x ] && echo PO_WAVE_RAMP=$3 ) $( [ "$4" != x ] && echo PO_WAVE_TAIL=$4 ) \
    timeout 300 python scripts/cold_start.py --reps 4 2>&1 | tail -1
done
