#!/bin/bash
# HBM-side traffic of every kernel of one bench step (FETCH_SIZE and WRITE_SIZE in separate passes, as the
# MI355X guide prescribes): scripts/pmc_hbm.sh TAG [pairs]   ->  gpurun_out/pmc_hbm_TAG.json
tag=${1:-rXX}; pairs=${2:-1250}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $root/gpurun_out/hbm_${tag}_$c -- python3 $root/bench.py --pairs $pairs --gen_procs 1 --steps 1 --warmup 1 --cpu_sample 0 --no_secondary --no_strong > $root/gpurun_out/hbm_${tag}_$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    # the LAST launch of every kernel (the first launch on a fresh workspace also clears its value-store slices)
    last = {}; n = collections.defaultdict(int)
    for f in glob.glob("$root/gpurun_out/hbm_${tag}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            d = int(r["Dispatch_Id"])
            if k not in last or d > last[k][0]: last[k] = (d, 0.0)
    for f in glob.glob("$root/gpurun_out/hbm_${tag}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            n[k] += 1
            if int(r["Dispatch_Id"]) == last[k][0]: last[k] = (last[k][0], last[k][1] + float(r["Counter_Value"]))
    for k in last:
        if any(x in k for x in ("beam2d", "viterbi", "pair_prep")):
            out[k][c + "_KB_per_launch"] = round(last[k][1], 1); out[k]["launches_seen"] = n[k]
res = {"command": "rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python3 bench.py --pairs $pairs --gen_procs 1 --steps 1 --warmup 1 --cpu_sample 0 --no_secondary --no_strong",
       "note": "one pass per counter; values of the LAST launch of each kernel (steady state: workspace already tagged); KB as rocprofv3 reports; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X guide), WRITE_SIZE uncalibrated",
       "workload": "$pairs synthetic pairs, T~4000, W=5", "pairs_per_launch": $pairs, "kernels": out}
json.dump(res, open("$root/gpurun_out/pmc_hbm_$tag.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
