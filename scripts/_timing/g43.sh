bash $GRAFT_REPO_ROOT/scripts/_timing/ab_bench.sh flags main expect cfnounroll main expect
cd $GRAFT_REPO_ROOT
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g43_small.log 2>&1
exit 0
