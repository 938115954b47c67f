#!/bin/bash
# kernel TRACE (start / end of every launch) of bench.py's secondary legs -> gpurun_out/ktrace2_LABEL.csv (name, start, end; ns)
root=${GRAFT_REPO_ROOT:-/root/repo}
label=$1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/kt2_$label -- python3 $root/bench.py --steps 1 --warmup 1 --no_strong --cpu_sample 0 > $root/gpurun_out/kt2_$label.log 2>&1
f=$(find $root/gpurun_out/kt2_$label -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
with open("$root/gpurun_out/ktrace2_$label.csv","w") as o:
    for r in rows:
        o.write("%s,%d,%d,%s\n"%(r["Kernel_Name"].split("(")[0].replace("void ",""),int(r["Start_Timestamp"])-t0,int(r["End_Timestamp"])-t0,r.get("Queue_Id","")))
PY
rm -rf $root/gpurun_out/kt2_$label
wc -l $root/gpurun_out/ktrace2_$label.csv
