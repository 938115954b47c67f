#!/bin/bash
# LDS counters of the kernels that keep their working set in LDS / registers: scripts/pmc_lds_round.sh TAG
#   -> gpurun_out/pmc_lds_TAG.json  (beam2d_reg_kernel at 10 000 pairs, beam2d_ring_kernel at 2 048, beam1d kernels at 1 000 reads)
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
C="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
timeout 400 rocprofv3 --pmc $C --output-format csv -d $root/gpurun_out/lds_${tag}_reg -- python3 $root/scripts/quick_time_2d.py 10000 5 > $root/gpurun_out/lds_${tag}_reg.log 2>&1
PO_ROUTE=ring timeout 400 rocprofv3 --pmc $C --output-format csv -d $root/gpurun_out/lds_${tag}_ring -- python3 $root/scripts/quick_time_2d.py 2048 5 > $root/gpurun_out/lds_${tag}_ring.log 2>&1
timeout 400 rocprofv3 --pmc $C --output-format csv -d $root/gpurun_out/lds_${tag}_b1 -- python3 $root/scripts/quick_time_1d.py 1000 > $root/gpurun_out/lds_${tag}_b1.log 2>&1
python3 - <<PY
import csv, glob, collections, json
res = {}
for part, keys in (("reg", ("beam2d_reg",)), ("b1", ("beam1d",))):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$root/gpurun_out/lds_${tag}_%s/**/*counter_collection.csv" % part, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not any(x in k for x in keys): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_INSTS_LDS": n[k] += 1
    for k, d in acc.items():
        d = dict(d); d["launches_summed"] = n[k]
        if d.get("SQ_LDS_IDX_ACTIVE"): d["bank_conflict_cycles_per_active_cycle"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
        if d.get("SQ_BUSY_CU_CYCLES"): d["lds_active_fraction_of_busy_cu_cycles"] = round(d.get("SQ_LDS_IDX_ACTIVE", 0.0) / d["SQ_BUSY_CU_CYCLES"], 4)
        res[k] = d
json.dump({"command": "rocprofv3 --pmc $C -- python3 scripts/quick_time_2d.py 10000 5 | (PO_ROUTE=ring) quick_time_2d.py 2048 5 | quick_time_1d.py 1000",
           "note": "counters summed over the device and over the launches of each script (quick_time_2d: 2 launches; quick_time_1d: W = 10 and W = 25, warm-up + timed)",
           "kernels": res}, open("$root/gpurun_out/pmc_lds_$tag.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
