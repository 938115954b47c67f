cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r04_g61_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_g61_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/r04_g61_bench.json 2> gpurun_out/r04_g61_bench.err
exit 0
