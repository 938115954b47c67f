cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r04_fuzz_final.json
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_g42_tests.log
timeout 600 python scripts/fuzz_parity.py --seconds 420 --seed 461 --route reg --focus --json gpurun_out/r04_fuzz_final.json --head 08d1e24 > gpurun_out/r04_g42_fuzz_focus.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 150 --seed 462 --json gpurun_out/r04_fuzz_final.json --head 08d1e24 > gpurun_out/r04_g42_fuzz_all.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 463 --pipeline --json gpurun_out/r04_fuzz_final.json --head 08d1e24 > gpurun_out/r04_g42_fuzz_pipeline.log 2>&1
timeout 200 python scripts/fuzz_parity.py --seconds 60 --seed 464 --route ring --focus --json gpurun_out/r04_fuzz_final.json --head 08d1e24 > gpurun_out/r04_g42_fuzz_ring.log 2>&1
timeout 200 python scripts/fuzz_parity.py --seconds 60 --seed 465 --oned --json gpurun_out/r04_fuzz_final.json --head 08d1e24 > gpurun_out/r04_g42_fuzz_oned.log 2>&1
exit 0
