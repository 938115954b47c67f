#!/bin/bash
# SQ instruction counters of the pair beam kernel on one route: scripts/pmc_ring.sh TAG ROUTE [n] [W]
tag=$1; route=$2; n=${3:-4096}; W=${4:-5}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
PO_ROUTE=$route timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES \
  --output-format csv -d $root/gpurun_out/pmc_$tag -- python3 $root/scripts/quick_time_2d.py $n $W > $root/gpurun_out/pmc_$tag.log 2>&1
tail -2 $root/gpurun_out/pmc_$tag.log
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$root/gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:50]][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": cnt[r["Kernel_Name"][:50]] += 1
for k, v in acc.items():
    if "beam2d" in k: print(k, "launches", cnt[k], {a: "%.3e" % b for a, b in v.items()})
PY
