bash $GRAFT_REPO_ROOT/scripts/_timing/ab_bench.sh flags2 expect expect2 expect expect2
exit 0
