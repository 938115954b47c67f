cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_counters.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04_g60_counters.log
exit 0
