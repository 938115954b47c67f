cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_g53_smoke.log 2>&1
exit 0
