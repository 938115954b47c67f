cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g25_small.log 2>&1
POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_regtiming2.so PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 4096 2>&1 | awk '/pair slot 0/{c++} c==1 || c==4 || /^n=/' > gpurun_out/r04_g25_regtiming2.log
timeout 300 python scripts/fuzz_parity.py --seconds 100 --seed 431 --route reg --focus > gpurun_out/r04_g25_fuzz_focus.log 2>&1
exit 0
