# A/B of two builds of the library inside one GPU call (scripts/exp/lib_prev.so vs the tree's): bench.py arguments in $Q
set -e
for v in prev new prev new; do
  if [ $v = prev ]; then export MI355_DECODE_LIB=$PWD/scripts/exp/lib_prev.so; else unset MI355_DECODE_LIB; fi
  timeout -k 10 300 python3 bench.py $Q 2>/dev/null | grep '^{' | sed "s/^/$v /" >> gpurun_out/ab_lib.txt
done
