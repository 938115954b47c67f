cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_regtiming.so PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 4096 2>&1 | grep -v "^\[po_reg_timing\]" | awk '/run loop/{c++} c<=1 || /^n=/ || c==4' > gpurun_out/r04_g5_regtiming.log
PO_ROUTES=reg,legacy timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g5_small.log 2>&1
exit 0
