#!/bin/bash
# a short closing check of a HOST-side change on the library in the tree, ONE gpurun call: scripts/final_short.sh TAG HEAD [fuzz seconds per session]
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
tag=$1; head=$2; secs=${3:-100}
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/gpu_suite_$tag.txt; cat gpurun_out/gpu_suite_$tag.txt
( time timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err ) 2> gpurun_out/bench_time_$tag.txt; tail -3 gpurun_out/bench_time_$tag.txt; tail -1 gpurun_out/bench_$tag.json | cut -c1-200
rm -f gpurun_out/fuzz_$tag.json
python scripts/fuzz_parity.py --seconds $secs --seed 6501 --pipeline --json gpurun_out/fuzz_$tag.json --head $head 2>&1 | tail -2
python scripts/fuzz_parity.py --seconds $secs --seed 6502 --focus --json gpurun_out/fuzz_$tag.json --head $head 2>&1 | tail -2
