cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g12_small_noboard.log 2>&1
PO_REG_BOARD=1 PO_DEBUG_OCC=1 PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 7 1250 3584 4096 10000 > gpurun_out/r04_g12_small_board.log 2>&1
PO_REG_BOARD=1 timeout 300 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_batch_scale.py -x -q -k "reg or fuzz" 2>&1 | tail -4 > gpurun_out/r04_g12_tests_board.log
exit 0
