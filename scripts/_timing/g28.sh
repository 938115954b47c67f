cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g28_small.log 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_g28_tests.log
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 441 --route reg --focus --dump gpurun_out/r04_g28_dump.npz > gpurun_out/r04_g28_fuzz_focus.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 60 --seed 442 --pipeline > gpurun_out/r04_g28_fuzz_pipeline.log 2>&1
exit 0
