# decode attention alone (scripts/bench_attn_ctx.py) and the headline step: scripts/exp/lib_prev.so against the tree's build
set -e
for v in prev new prev new; do
  if [ $v = prev ]; then export MI355_DECODE_LIB=$PWD/scripts/exp/lib_prev.so; else unset MI355_DECODE_LIB; fi
  echo "== $v" >> gpurun_out/ab_attn_lib.txt
  timeout -k 10 200 python3 scripts/bench_attn_ctx.py 32 32 129,192,255,768 2>/dev/null >> gpurun_out/ab_attn_lib.txt
  timeout -k 10 200 python3 scripts/bench_attn_ctx.py 256 32 192 2>/dev/null >> gpurun_out/ab_attn_lib.txt
done
for v in prev new prev new; do
  if [ $v = prev ]; then export MI355_DECODE_LIB=$PWD/scripts/exp/lib_prev.so; else unset MI355_DECODE_LIB; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-served --steps 20 --warmup 5 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v step', d['value'], d['ms_per_step'], d['kernels']['paged_attn_decode_fused']['us'])" >> gpurun_out/ab_attn_lib.txt
done
