cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in regbl regu2 regblu2; do
echo "== $v"; POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_$v.so PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 10000
done > gpurun_out/r04_g26_variants.log 2>&1
exit 0
