cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r04_g30_bench.json 2> gpurun_out/r04_g30_bench.err
exit 0
