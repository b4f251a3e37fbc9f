# A/B of the row-block candidates inside one GPU call: Qwen2-7B batch 64 (fp8 and bf16) with and without them
set -e
Q="--model qwen2-7b --batch-per-gpu 64 --prompt-len 512 --gen-len 512 --no-cpu-baseline --no-served --steps 40 --warmup 5"
for rb in 0 1 0 1; do
  MI355_ROW_BLOCKS=$rb timeout -k 10 300 python3 bench.py $Q --quantization fp8 2>/dev/null | grep '^{' | sed "s/^/fp8 rb=$rb /" >> gpurun_out/ab_row_blocks.txt
  MI355_ROW_BLOCKS=$rb timeout -k 10 300 python3 bench.py $Q 2>/dev/null | grep '^{' | sed "s/^/bf16 rb=$rb /" >> gpurun_out/ab_row_blocks.txt
done
