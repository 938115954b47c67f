cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -60 > gpurun_out/r04_g62_tests.log
exit 0
