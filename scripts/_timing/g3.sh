cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PO_DEBUG_OCC=1 timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g3_small.log 2>&1
POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_regw3.so PO_ROUTES=reg PO_DEBUG_OCC=1 timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g3_small_w3.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_fuzz.py tests/test_gpu_batch_scale.py -x -q -k "reg or fuzz" 2>&1 | tail -8 > gpurun_out/r04_g3_tests.log
exit 0
