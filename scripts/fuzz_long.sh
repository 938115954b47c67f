#!/bin/bash
# two long open-ended sessions on the library in the tree (serial chain, closed form): scripts/fuzz_long.sh TAG HEAD SECONDS
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
tag=$1; head=$2; secs=${3:-600}
rm -f gpurun_out/fuzz_$tag.json gpurun_out/fuzz_${tag}_closed_form.json
python scripts/fuzz_parity.py --seconds $secs --seed 6301 --focus --json gpurun_out/fuzz_$tag.json --head $head 2>&1 | tail -2
PO_CHAIN_CLOSED=1 python scripts/fuzz_parity.py --seconds $secs --seed 6302 --focus --json gpurun_out/fuzz_${tag}_closed_form.json --head $head 2>&1 | tail -2
