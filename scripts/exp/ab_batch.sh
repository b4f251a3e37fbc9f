# Llama-3-8B on one GPU at batch 128 / 256 (the harness's deployment default is max_num_seqs 256) with and without
# the row-block forms among the decode GEMM candidates
set -e
for b in 128 256; do
  for rb in 0 1; do
    MI355_ROW_BLOCKS=$rb timeout -k 10 400 python3 bench.py --batch-per-gpu $b --no-cpu-baseline --no-served --steps 20 --warmup 5 2>/dev/null | grep '^{' | sed "s/^/b=$b rb=$rb /" >> gpurun_out/ab_batch.txt
  done
done
