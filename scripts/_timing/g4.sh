cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_regtiming.so PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 4096 > gpurun_out/r04_g4_regtiming.log 2>&1
PO_ROUTES=reg timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_g4 -o g4 -- python3 scripts/small_batch.py 10000 > gpurun_out/r04_g4_prof.log 2>&1
find gpurun_out/prof_g4 -name "*kernel_stats*" | head -3 >> gpurun_out/r04_g4_prof.log
exit 0
