# the decode step on one GPU at batch 64 / 128 / 256 under each decode-attention form (MI355_ATTN_FORM)
set -e
for b in 64 128 256; do
  for f in 41 12 11 41 11; do
    MI355_ATTN_FORM=$f timeout -k 10 400 python3 bench.py --batch-per-gpu $b --no-cpu-baseline --no-served --steps 20 --warmup 5 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d.get('kernels', {}); print('form $f b=$b', d['value'], d['ms_per_step'], {n: v for n, v in k.items() if 'attn' in n or 'attention' in n})" >> gpurun_out/ab_attn_forms_step.txt
  done
done
