#!/bin/bash
# SQ wave-cycle breakdown of the pair beam kernel: scripts/pmc_sq.sh TAG [n] [W]   (set PO_B2_LEGACY=1 for the one-pair-per-wave kernel)
tag=$1; n=${2:-4096}; W=${3:-5}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES \
  --output-format csv -d $root/gpurun_out/pmc_$tag -- python3 $root/scripts/quick_time_2d.py $n $W > $root/gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$root/gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "beam2d" in k: print(k, dict(v))
PY
