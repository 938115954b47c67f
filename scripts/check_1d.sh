#!/bin/bash
# the 1-D parity tests and one open-ended 1-D fuzz session on the library in the tree: scripts/check_1d.sh TAG HEAD SECONDS
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
tag=$1; head=$2; secs=${3:-300}
rm -f gpurun_out/fuzz_$tag.json
python -m pytest tests/test_gpu_parity_1d.py tests/test_gpu_traces.py tests/test_gpu_batch_scale.py -m gpu -x -q 2>&1 | tail -3
python scripts/fuzz_parity.py --seconds $secs --seed 6401 --oned --json gpurun_out/fuzz_$tag.json --head $head 2>&1 | tail -3
