bash $GRAFT_REPO_ROOT/scripts/_timing/ab_bench.sh trim5 main trim5 main trim5
exit 0
