cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/profile_round.sh r04a > gpurun_out/r04_g9_profile.log 2>&1
exit 0
