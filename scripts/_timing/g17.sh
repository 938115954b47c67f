cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g17_small.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 411 --route reg --focus --dump gpurun_out/r04_g17_dump_focus.npz > gpurun_out/r04_g17_fuzz_focus.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 120 --seed 413 --dump gpurun_out/r04_g17_dump_all.npz > gpurun_out/r04_g17_fuzz_all.log 2>&1
PO_REG_NEVER=1 timeout 300 python scripts/fuzz_parity.py --seconds 60 --seed 413 --dump gpurun_out/r04_g17_dump_noreg.npz > gpurun_out/r04_g17_fuzz_noreg.log 2>&1
exit 0
