# the headline decode step: scripts/exp/lib_prev.so against the tree's build, alternating, in one GPU call
set -e
for v in prev new prev new prev new; do
  if [ $v = prev ]; then export MI355_DECODE_LIB=$PWD/scripts/exp/lib_prev.so; else unset MI355_DECODE_LIB; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-served --steps 20 --warmup 5 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d['kernels']; print('$v step', d['value'], d['ms_per_step'], 'norm', k['fused_add_rms_norm']['us'], 'attn', k['paged_attn_decode_fused']['us'], 'o', k['gemm_o']['us'], 'down', k['gemm_down']['us'], 'qkv', k['gemm_qkv']['us'])" >> gpurun_out/ab_lib_step.txt
done
