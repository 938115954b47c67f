#!/bin/bash
# per-kernel durations of one quick_time_2d run: scripts/kstats.sh [n] [W] [kind]
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/ks
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/ks -- python3 $root/scripts/quick_time_2d.py ${1:-10000} ${2:-5} ${3:-poreover} > $root/gpurun_out/ks.log 2>&1
f=$(find $root/gpurun_out/ks -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if any(k in r["Name"] for k in ("beam2d", "pair_prep", "fillBuffer")): print("%-60s calls %s avg %.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
