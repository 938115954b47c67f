cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in "" b1t b1te "" b1t; do
echo "== 1d variant '$v'"; export POREOVER_HIP_LIB=${v:+scripts/_timing/libporeover_hip_$v.so}; [ -z "$v" ] && unset POREOVER_HIP_LIB
timeout 200 python scripts/quick_time_1d.py 1000 | grep beam1d; timeout 200 python scripts/quick_time_1d.py 1000 flipflop | grep beam1d
done
for v in "" b2t "" b2t; do
echo "== 2d variant '$v'"; export POREOVER_HIP_LIB=${v:+scripts/_timing/libporeover_hip_$v.so}; [ -z "$v" ] && unset POREOVER_HIP_LIB
timeout 200 python scripts/quick_time_2d.py 10000 10 | tail -1; timeout 200 python scripts/quick_time_2d.py 10000 5 bonito | tail -1; timeout 200 python scripts/quick_time_2d.py 10000 5 poreover row | tail -1; timeout 200 python scripts/quick_time_2d.py 1024 5 flipflop | tail -1
done
} > gpurun_out/r04_g55_variants.log 2>&1
exit 0
