cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_g8_tests.log
timeout 400 python scripts/fuzz_parity.py --seconds 200 --seed 401 --route reg --focus --json gpurun_out/r04_fuzz_reg.json --head 0381665 > gpurun_out/r04_g8_fuzz_focus.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 402 --pipeline --json gpurun_out/r04_fuzz_reg.json --head 0381665 > gpurun_out/r04_g8_fuzz_pipeline.log 2>&1
timeout 900 python bench.py > gpurun_out/r04_g8_bench.json 2> gpurun_out/r04_g8_bench.err
exit 0
