cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_regtiming.so PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 4096 2>&1 | awk '/pair slot 0/{c++} c==1 || c==4 || /^n=/' > gpurun_out/r04_g57_regtiming.log
exit 0
