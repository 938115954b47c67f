#!/bin/bash
# SQ instruction mix + wave-cycle breakdown of the pair beam kernels (two --pmc passes, 8 SQ counters each):
#   scripts/pmc_sq2.sh TAG [n] [W]  ->  gpurun_out/pmc_sq_TAG.json  (copy to profiles/ to have it judged)
tag=$1; n=${2:-10000}; W=${3:-5}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
p1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAVES"
p2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"
i=0
for p in "$p1" "$p2"; do
  i=$((i+1))
  timeout 500 rocprofv3 --pmc $p --output-format csv -d $root/gpurun_out/pmcsq_${tag}_$i -- python3 $root/scripts/quick_time_2d.py $n $W > $root/gpurun_out/pmcsq_${tag}_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for i in (1, 2):
    for f in glob.glob("$root/gpurun_out/pmcsq_${tag}_%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "beam2d" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {}
for k, v in acc.items():
    d = {c: v[c] / max(1, cnt[k][c]) for c in v}   # per launch
    d["launches"] = max(cnt[k].values())
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if c in d: d[c + "_frac_of_wave_cycles"] = round(d[c] / wc, 4)
    out[k] = d
res = {"command": "rocprofv3 --pmc <8 SQ counters> --output-format csv -- python3 scripts/quick_time_2d.py $n $W (two passes; per-launch averages; SQ_*CYCLES in quad-cycles)",
       "pairs_per_launch": $n, "beam_width": $W, "kernels": out}
json.dump(res, open("$root/gpurun_out/pmc_sq_$tag.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
