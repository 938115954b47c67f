cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp profiles/r04_fuzz_final_binary.json gpurun_out/r04_fuzz_final6.json
timeout 900 python scripts/fuzz_parity.py --seconds 720 --seed 511 --route reg --focus --json gpurun_out/r04_fuzz_final6.json --head f4ca4fb > gpurun_out/r04_g59_fuzz_focus.log 2>&1
timeout 500 python scripts/fuzz_parity.py --seconds 360 --seed 512 --json gpurun_out/r04_fuzz_final6.json --head f4ca4fb > gpurun_out/r04_g59_fuzz_all.log 2>&1
timeout 400 python scripts/fuzz_parity.py --seconds 240 --seed 513 --pipeline --json gpurun_out/r04_fuzz_final6.json --head f4ca4fb > gpurun_out/r04_g59_fuzz_pipeline.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 180 --seed 514 --oned --json gpurun_out/r04_fuzz_final6.json --head f4ca4fb > gpurun_out/r04_g59_fuzz_oned.log 2>&1
exit 0
