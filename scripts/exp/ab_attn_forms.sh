# decode attention: waves per (sequence, kv head, split) x register stages (MI355_ATTN_FORM = 42 / 41 / 12 / 11)
set -e
for f in 11 12; do
  MI355_ATTN_FORM=$f timeout -k 10 500 python3 -u -m pytest tests/test_ops_gpu.py -x -q -k "attn_decode or decode_fused or no_alloc" --timeout 120 --timeout-method thread > gpurun_out/attn_form_tests_$f.log 2>&1
  tail -1 gpurun_out/attn_form_tests_$f.log
done
for cfg in "32 32 192" "64 32 192,768" "128 32 192,768" "256 32 192,768"; do
  set -- $cfg
  for f in 42 41 12 11; do
    echo "== form=$f seqs=$1 heads=$2" >> gpurun_out/ab_attn_forms.txt
    MI355_ATTN_FORM=$f timeout -k 10 200 python3 scripts/bench_attn_ctx.py $1 $2 $3 2>/dev/null >> gpurun_out/ab_attn_forms.txt
  done
done
