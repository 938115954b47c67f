cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== default (6)"; PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 10000
for v in regys3 regys10 regys1000 reghoist; do
echo "== $v"; POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_$v.so PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 10000
done
} > gpurun_out/r04_g37_yshift.log 2>&1
exit 0
