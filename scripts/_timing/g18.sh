cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_g18_tests.log
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g18_small.log 2>&1
timeout 500 python scripts/fuzz_parity.py --seconds 240 --seed 411 --route reg --focus --dump gpurun_out/r04_g18_dump_focus.npz > gpurun_out/r04_g18_fuzz_focus.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 413 --dump gpurun_out/r04_g18_dump_all.npz > gpurun_out/r04_g18_fuzz_all.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 414 --pipeline > gpurun_out/r04_g18_fuzz_pipeline.log 2>&1
exit 0
