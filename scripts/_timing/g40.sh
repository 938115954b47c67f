bash $GRAFT_REPO_ROOT/scripts/_timing/ab_bench.sh clamp base main base main
exit 0
