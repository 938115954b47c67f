cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g35_small.log 2>&1
exit 0
