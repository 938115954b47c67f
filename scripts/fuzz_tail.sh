#!/bin/bash
# three more short sessions on the library in the tree (all methods, 1-D, closed-form chain mode), appended to gpurun_out/fuzz_TAG*.json: scripts/fuzz_tail.sh TAG HEAD SECONDS
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
tag=$1; head=$2; secs=${3:-90}
python scripts/fuzz_parity.py --seconds $secs --seed 6503 --json gpurun_out/fuzz_$tag.json --head $head 2>&1 | tail -1
python scripts/fuzz_parity.py --seconds $secs --seed 6504 --oned --json gpurun_out/fuzz_$tag.json --head $head 2>&1 | tail -1
PO_CHAIN_CLOSED=1 python scripts/fuzz_parity.py --seconds $secs --seed 6505 --focus --json gpurun_out/fuzz_${tag}_closed_form.json --head $head 2>&1 | tail -1
