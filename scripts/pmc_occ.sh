#!/bin/bash
# SQ + LDS counters of the x2 kernel at a given residency: scripts/pmc_occ.sh VARIANT PER_CU N
v=$1; k=$2; n=$3
root=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$v" = base ]; then export POREOVER_HIP_LIB=$root/poreover_amd/libporeover_hip.so; else export POREOVER_HIP_LIB=$root/scripts/variants/libporeover_hip_$v.so; fi
export PO_X2_PER_CU=$k
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $pass | cut -c4-11)
  timeout 400 rocprofv3 --pmc $pass --output-format csv -d $root/gpurun_out/po_${v}_${k}_$tag -- python3 $root/scripts/quick_time_2d.py $n 5 > $root/gpurun_out/po_${v}_${k}_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob("$root/gpurun_out/po_${v}_${k}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "beam2d_x2" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
wc = acc["SQ_WAVE_CYCLES"]; cu = acc["SQ_BUSY_CU_CYCLES"] / 2   # counted in both passes
print("$v per_cu=$k n=$n:", "active %.1f%% wait %.1f%% issue-stall %.1f%% of wave cycles;" % (100*acc["SQ_ACTIVE_INST_ANY"]/wc, 100*acc["SQ_WAIT_ANY"]/wc, 100*acc["SQ_WAIT_INST_ANY"]/wc),
      "VALU insts/CU-cycle %.3f; LDS idx-active/CU-cycle %.3f; bank-conflict share %.2f; LDS-inst active %.3f; VMEM %.3f" % (
      acc["SQ_INSTS_VALU"]/cu, acc["SQ_LDS_IDX_ACTIVE"]/cu, acc["SQ_LDS_BANK_CONFLICT"]/max(acc["SQ_LDS_IDX_ACTIVE"],1), acc["SQ_ACTIVE_INST_LDS"]/cu, acc["SQ_ACTIVE_INST_VMEM"]/cu))
PY
tail -1 $root/gpurun_out/po_${v}_${k}_WAVE_CYC.log | cut -c1-160
