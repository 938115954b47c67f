bash $GRAFT_REPO_ROOT/scripts/_timing/ab_bench.sh trim3 expect2 trim3 expect2 trim3
exit 0
