#!/bin/bash
# open-ended fuzz sessions of one gpurun call on the library in the tree: scripts/fuzz_round.sh TAG HEAD SECONDS
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
tag=$1; head=$2; secs=${3:-240}
python scripts/fuzz_parity.py --seconds $secs --seed 6101 --focus --json gpurun_out/fuzz_$tag.json --head $head 2>&1 | tail -3
python scripts/fuzz_parity.py --seconds $secs --seed 6102 --json gpurun_out/fuzz_$tag.json --head $head 2>&1 | tail -3
python scripts/fuzz_parity.py --seconds $secs --seed 6103 --pipeline --json gpurun_out/fuzz_$tag.json --head $head 2>&1 | tail -3
python scripts/fuzz_parity.py --seconds $((secs / 2)) --seed 6104 --oned --json gpurun_out/fuzz_$tag.json --head $head 2>&1 | tail -3
PO_CHAIN_CLOSED=1 python scripts/fuzz_parity.py --seconds $secs --seed 6105 --focus --json gpurun_out/fuzz_${tag}_closed_form.json --head $head 2>&1 | tail -3
