# bench.py at batch 32 / 128 / 256 on one GPU: scripts/exp/lib_prev.so against the tree's build
set -e
for v in prev new; do
  if [ $v = prev ]; then export MI355_DECODE_LIB=$PWD/scripts/exp/lib_prev.so; else unset MI355_DECODE_LIB; fi
  for b in 32 128 256; do
    timeout -k 10 400 python3 bench.py --batch-per-gpu $b --no-cpu-baseline --no-served --steps 20 --warmup 5 2>/dev/null | grep '^{' | sed "s/^/$v b=$b /" >> gpurun_out/ab_attn_batch.txt
  done
done
