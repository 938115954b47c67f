#!/bin/bash
# The round's measurement evidence in one gpurun call: scripts/profile_round.sh TAG   (writes gpurun_out/*_TAG*)
#   1. bench.py as the driver runs it                                   -> bench_TAG.json
#   2. rocprofv3 --kernel-trace --stats of the timed region only        -> kernel_stats_TAG.csv  (the dominant kernel's
#      (--no_secondary --no_strong: every launch of it is a 10 000-pair launch, so AverageNs IS roofline.avg_launch_ms)
#   3. HBM counters, separate --pmc passes, 10 000 and 1250 pairs       -> pmc_hbm_TAG_10000.json, pmc_hbm_TAG_1250.json
#   4. SQ instruction / wait counters, 10 000 pairs                     -> pmc_sq_TAG.json
#   5. the same for round 3's routing (PO_REG_NEVER=1: beam2d_kernel)   -> *_TAG_legacy*
#   6. the 1-D beam search (config 2): kernel stats + SQ counters       -> beam1d_TAG.txt
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
tail -1 gpurun_out/bench_$tag.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
prof_one() {   # $1 = suffix, env already set
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag$1 -- python3 $root/bench.py --steps 3 --warmup 1 --gen_procs 1 --cpu_sample 0 --no_secondary --no_strong > $root/gpurun_out/prof_$tag$1.log 2>&1
  f=$(find $root/gpurun_out/prof_$tag$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $root/gpurun_out/kernel_stats_$tag$1.csv && head -8 $f | cut -c1-160
  tail -1 $root/gpurun_out/prof_$tag$1.log > $root/gpurun_out/bench_under_rocprof_$tag$1.json
}
sq_one() {     # $1 = suffix
  timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $root/gpurun_out/sq_$tag$1 -- python3 $root/bench.py --steps 1 --warmup 1 --gen_procs 1 --cpu_sample 0 --no_secondary --no_strong > $root/gpurun_out/sq_$tag$1.log 2>&1
  python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); last = {}
rows = []
for f in glob.glob("$root/gpurun_out/sq_$tag$1/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d = int(r["Dispatch_Id"])
    if k not in last or d > last[k]: last[k] = d
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if int(r["Dispatch_Id"]) == last[k]: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = {k: dict(v) for k, v in acc.items() if any(x in k for x in ("beam2d", "pair_prep", "viterbi"))}
json.dump({"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- python3 bench.py --steps 1 --warmup 1 --gen_procs 1 --cpu_sample 0 --no_secondary --no_strong",
           "note": "the LAST launch of every kernel (10 000 pairs per launch), counters summed over the device", "kernels": out}, open("$root/gpurun_out/pmc_sq_$tag$1.json", "w"), indent=1)
for k, v in out.items():
    if "beam2d" in k: print(k, {a: "%.3e" % b for a, b in v.items()})
PY
}
prof_one ""
cd $root && scripts/pmc_hbm.sh ${tag}_10000 10000 > gpurun_out/pmc_hbm_${tag}_10000.txt 2>&1; tail -16 gpurun_out/pmc_hbm_${tag}_10000.txt
scripts/pmc_hbm.sh ${tag}_1250 1250 > gpurun_out/pmc_hbm_${tag}_1250.txt 2>&1
cd /tmp; sq_one ""
export PO_REG_NEVER=1
prof_one "_legacy"
cd /tmp; sq_one "_legacy"
unset PO_REG_NEVER
# 1-D beam search
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/b1_$tag -- python3 $root/scripts/quick_time_1d.py 1000 > $root/gpurun_out/beam1d_$tag.txt 2>&1
f=$(find $root/gpurun_out/b1_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/gpurun_out/kernel_stats_beam1d_$tag.csv && grep beam1d $f | cut -c1-200
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $root/gpurun_out/b1sq_$tag -- python3 $root/scripts/quick_time_1d.py 1000 >> $root/gpurun_out/beam1d_$tag.txt 2>&1
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$root/gpurun_out/b1sq_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
out = {k: dict(v, launches=n[k]) for k, v in acc.items() if "beam1d" in k}
json.dump({"command": "rocprofv3 --pmc ... -- python3 scripts/quick_time_1d.py 1000", "note": "summed over the launches of the script (W = 10 twice, W = 25 twice)", "kernels": out}, open("$root/gpurun_out/pmc_sq_beam1d_$tag.json", "w"), indent=1)
print(out)
PY
