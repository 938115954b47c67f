cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in 1 1250 10000; do echo "== n=$n"; bash scripts/kstats.sh $n 5 poreover; done > gpurun_out/r04_g27_kstats.log 2>&1
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g27_small.log 2>&1
exit 0
