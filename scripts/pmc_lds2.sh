#!/bin/bash
# LDS pipe counters of the pair beam kernel: scripts/pmc_lds2.sh TAG [n] [W]  ->  gpurun_out/pmc_lds_TAG.json
tag=$1; n=${2:-10000}; W=${3:-5}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $root/gpurun_out/pmclds_$tag -- python3 $root/scripts/quick_time_2d.py $n $W > $root/gpurun_out/pmclds_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$root/gpurun_out/pmclds_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "beam2d" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {k: {c: v[c] / max(1, cnt[k][c]) for c in v} for k, v in acc.items()}
json.dump({"command": "rocprofv3 --pmc SQ_LDS_* ... -- python3 scripts/quick_time_2d.py $n $W (per-launch averages)", "kernels": out}, open("$root/gpurun_out/pmc_lds_$tag.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
