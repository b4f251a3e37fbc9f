# A/B of two builds of the library inside one GPU call (scripts/exp/lib_prev.so vs the tree's) on the TP rank step
set -e
for v in prev new prev new; do
  if [ $v = prev ]; then export MI355_DECODE_LIB=$PWD/scripts/exp/lib_prev.so; else unset MI355_DECODE_LIB; fi
  timeout -k 10 300 python3 scripts/bench_tp_rank.py --tps 4,8 --only sharded-timed 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['tp'], d['batch'], d['rank_step_ms'], d['tiled_plans'], d['tuned_us_tiled_vs_library']['gu'][0])" >> gpurun_out/ab_lib_tp_rank.txt
done
Q="--batch-per-gpu 256 --no-cpu-baseline --no-served --steps 40 --warmup 5"
for v in prev new prev new; do
  if [ $v = prev ]; then export MI355_DECODE_LIB=$PWD/scripts/exp/lib_prev.so; else unset MI355_DECODE_LIB; fi
  timeout -k 10 300 python3 bench.py $Q 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v b256', d['value'], d['ms_per_step'])" >> gpurun_out/ab_lib_tp_rank.txt
done
