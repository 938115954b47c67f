bash $GRAFT_REPO_ROOT/scripts/_timing/ab_bench.sh trim4 main trim4 main trim4
exit 0
