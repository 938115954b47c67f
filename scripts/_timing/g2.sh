cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04_g2_tests.log
python scripts/small_batch.py 1 1250 > gpurun_out/r04_g2_small.log 2>&1
POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_norun.so python scripts/small_batch.py 1 1250 > gpurun_out/r04_g2_small_norun.log 2>&1
PO_ROUTE=ring PO_RING_AUTO=1 python scripts/quick_time_2d.py 10000 5 > gpurun_out/r04_g2_ring10k.log 2>&1
POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_ringtiming.so PO_ROUTE=ring python scripts/quick_time_2d.py 1 5 > gpurun_out/r04_g2_ringtiming1.log 2>&1
exit 0
