#!/bin/bash
# LDS-side counters of the pair beam kernel: scripts/pmc_lds.sh VARIANT [n] [W]
v=$1; n=${2:-5120}; W=${3:-5}
root=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$v" = base ]; then export POREOVER_HIP_LIB=$root/poreover_amd/libporeover_hip.so; else export POREOVER_HIP_LIB=$root/scripts/variants/libporeover_hip_$v.so; fi
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $root/gpurun_out/pl_$v -- python3 $root/scripts/quick_time_2d.py $n $W > $root/gpurun_out/pl_$v.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(int)
for f in glob.glob("$root/gpurun_out/pl_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "beam2d_x2" in k or "beam2d_kernel" in k: print("$v", k, {a: "%.4g" % b for a, b in d.items()})
PY
tail -2 $root/gpurun_out/pl_$v.log | cut -c1-150
