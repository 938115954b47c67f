cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_g16_tests.log
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g16_small.log 2>&1
timeout 500 python scripts/fuzz_parity.py --seconds 300 --seed 411 --route reg --focus --json gpurun_out/r04_fuzz.json --head c783a7a > gpurun_out/r04_g16_fuzz_focus.log 2>&1
PO_REG_BOARD=1 timeout 300 python scripts/fuzz_parity.py --seconds 90 --seed 412 --route reg --focus --json gpurun_out/r04_fuzz_board.json --head c783a7a > gpurun_out/r04_g16_fuzz_board.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 413 --json gpurun_out/r04_fuzz.json --head c783a7a > gpurun_out/r04_g16_fuzz_all.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 414 --pipeline --json gpurun_out/r04_fuzz.json --head c783a7a > gpurun_out/r04_g16_fuzz_pipeline.log 2>&1
timeout 200 python scripts/fuzz_parity.py --seconds 60 --seed 415 --oned --json gpurun_out/r04_fuzz.json --head c783a7a > gpurun_out/r04_g16_fuzz_oned.log 2>&1
exit 0
