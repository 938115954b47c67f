bash $GRAFT_REPO_ROOT/scripts/_timing/ab_bench.sh groups main groups groups2fz main groups groups2fz
exit 0
