cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r04_fuzz_final4.json
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_g51_tests.log
timeout 600 python scripts/fuzz_parity.py --seconds 240 --seed 491 --route reg --focus --json gpurun_out/r04_fuzz_final4.json --head a018658 > gpurun_out/r04_g51_fuzz_focus.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 90 --seed 492 --json gpurun_out/r04_fuzz_final4.json --head a018658 > gpurun_out/r04_g51_fuzz_all.log 2>&1
timeout 300 python scripts/fuzz_parity.py --seconds 90 --seed 493 --pipeline --json gpurun_out/r04_fuzz_final4.json --head a018658 > gpurun_out/r04_g51_fuzz_pipeline.log 2>&1
timeout 200 python scripts/fuzz_parity.py --seconds 60 --seed 464 --route ring --focus --json gpurun_out/r04_fuzz_final4.json --head a018658 > gpurun_out/r04_g51_fuzz_ring.log 2>&1
timeout 200 python scripts/fuzz_parity.py --seconds 60 --seed 465 --oned --json gpurun_out/r04_fuzz_final4.json --head a018658 > gpurun_out/r04_g51_fuzz_oned.log 2>&1

bash scripts/profile_round.sh r04e > gpurun_out/r04_g51_profile.log 2>&1
bash scripts/pmc_lds_round.sh r04e > gpurun_out/r04_g51_lds.log 2>&1
PO_ROUTES=reg timeout 300 python scripts/small_batch.py 1 1250 4096 10000 > gpurun_out/r04_g51_small.log 2>&1
exit 0
