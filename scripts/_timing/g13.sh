cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PO_REG_BOARD=1 PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 3584 10000 > gpurun_out/r04_g13_small_board.log 2>&1
exit 0
