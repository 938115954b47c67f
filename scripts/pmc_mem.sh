#!/bin/bash
# L2 / fabric write-side counters of the pair beam kernel for a library variant: scripts/pmc_mem.sh VARIANT [n] [W]
v=$1; n=${2:-4096}; W=${3:-5}
root=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$v" = base ]; then export POREOVER_HIP_LIB=$root/poreover_amd/libporeover_hip.so; else export POREOVER_HIP_LIB=$root/scripts/variants/libporeover_hip_$v.so; fi
cd /tmp && export TMPDIR=/tmp
for pass in "TCC_REQ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU"; do
  tag=$(echo $pass | cut -c1-7)
  timeout 400 rocprofv3 --pmc $pass --output-format csv -d $root/gpurun_out/pm_${v}_$tag -- python3 $root/scripts/quick_time_2d.py $n $W > $root/gpurun_out/pm_${v}_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob("$root/gpurun_out/pm_${v}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "beam2d_x2" in k or "beam2d_kernel" in k: print("$v", k, {a: "%.3g" % b for a, b in d.items()})
PY
