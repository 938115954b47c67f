bash $GRAFT_REPO_ROOT/scripts/_timing/ab_bench.sh nocount prev main prev main
exit 0
