cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/profile_round.sh r04b > gpurun_out/r04_g36_profile.log 2>&1
bash scripts/pmc_lds_round.sh r04b > gpurun_out/r04_g36_lds.log 2>&1
exit 0
