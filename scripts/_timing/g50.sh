bash $GRAFT_REPO_ROOT/scripts/_timing/ab_bench.sh fields main fields main fields
cd $GRAFT_REPO_ROOT
POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_fields.so PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 10000 > gpurun_out/r04_g50_small.log 2>&1
exit 0
