#!/bin/bash
# bench.py plain, then under rocprofv3 --kernel-trace --stats (round-end evidence): scripts/bench_profile.sh TAG
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
tail -1 gpurun_out/bench_$tag.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/bench.py --steps 2 --warmup 1 --gen_procs 1 --cpu_sample 0 > $root/gpurun_out/prof_$tag.log 2>&1
f=$(find $root/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $root/gpurun_out/kernel_stats_$tag.csv && head -12 $f
tail -1 $root/gpurun_out/prof_$tag.log | cut -c1-300
