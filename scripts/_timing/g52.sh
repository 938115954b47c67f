bash $GRAFT_REPO_ROOT/scripts/_timing/ab_bench.sh fz main fz nornxt main fz nornxt
exit 0
