cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04_g63_tests.log
exit 0
