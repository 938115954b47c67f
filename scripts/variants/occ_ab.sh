cd $GRAFT_REPO_ROOT
for v in main k1w4; do
  if [ $v = main ]; then unset POREOVER_HIP_LIB; else export POREOVER_HIP_LIB=$PWD/scripts/variants/libporeover_hip_$v.so; fi
  echo -n "$v W10 ctc: "; timeout 300 python scripts/quick_time_2d.py 10000 10 poreover 2>&1 | tail -1
done
for v in main k3n4 k3n2; do
  if [ $v = main ]; then unset POREOVER_HIP_LIB; else export POREOVER_HIP_LIB=$PWD/scripts/variants/libporeover_hip_$v.so; fi
  echo -n "$v W5 bonito: "; timeout 300 python scripts/quick_time_2d.py 10000 5 bonito 2>&1 | tail -1
  echo -n "$v W5 flipflop: "; timeout 300 python scripts/quick_time_2d.py 10000 5 flipflop 2>&1 | tail -1
done
