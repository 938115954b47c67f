cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g29_small.log 2>&1
for n in 1 10000; do echo "== n=$n"; bash scripts/kstats.sh $n 5 poreover; done > gpurun_out/r04_g29_kstats.log 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_g29_tests.log
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 451 --route reg --focus --dump gpurun_out/r04_g29_dump.npz > gpurun_out/r04_g29_fuzz_focus.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 60 --seed 452 --route ring --focus > gpurun_out/r04_g29_fuzz_ring.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 60 --seed 453 --pipeline > gpurun_out/r04_g29_fuzz_pipeline.log 2>&1
exit 0
