cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04_g11_tests.log
timeout 900 python bench.py > gpurun_out/r04_g11_bench.json 2> gpurun_out/r04_g11_bench.err
exit 0
