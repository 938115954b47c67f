cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== base 1d"; timeout 200 python scripts/quick_time_1d.py 1000; timeout 200 python scripts/quick_time_1d.py 1000 flipflop
echo "== early 1d"; POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_b1early.so timeout 200 python scripts/quick_time_1d.py 1000; POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_b1early.so timeout 200 python scripts/quick_time_1d.py 1000 flipflop
echo "== base 2d W10 / bonito W5 / legacy W5"; timeout 200 python scripts/quick_time_2d.py 10000 10; timeout 200 python scripts/quick_time_2d.py 10000 5 bonito; PO_ROUTE=legacy timeout 200 python scripts/quick_time_2d.py 10000 5
echo "== early 2d"; export POREOVER_HIP_LIB=scripts/_timing/libporeover_hip_b2early.so; timeout 200 python scripts/quick_time_2d.py 10000 10; timeout 200 python scripts/quick_time_2d.py 10000 5 bonito; PO_ROUTE=legacy timeout 200 python scripts/quick_time_2d.py 10000 5
} > gpurun_out/r04_g24_early.log 2>&1
exit 0
