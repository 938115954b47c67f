cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r04_fuzz_final5.json
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_g56_tests.log
timeout 300 python scripts/fuzz_parity.py --seconds 150 --seed 501 --oned --json gpurun_out/r04_fuzz_final5.json --head f4ca4fb > gpurun_out/r04_g56_fuzz_oned.log 2>&1
timeout 400 python scripts/fuzz_parity.py --seconds 240 --seed 502 --json gpurun_out/r04_fuzz_final5.json --head f4ca4fb > gpurun_out/r04_g56_fuzz_all.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 503 --pipeline --json gpurun_out/r04_fuzz_final5.json --head f4ca4fb > gpurun_out/r04_g56_fuzz_pipeline.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 504 --route reg --focus --json gpurun_out/r04_fuzz_final5.json --head f4ca4fb > gpurun_out/r04_g56_fuzz_focus.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 90 --seed 505 --route legacy --json gpurun_out/r04_fuzz_final5.json --head f4ca4fb > gpurun_out/r04_g56_fuzz_legacy.log 2>&1
timeout 900 python bench.py > gpurun_out/r04_g56_bench.json 2> gpurun_out/r04_g56_bench.err
exit 0
