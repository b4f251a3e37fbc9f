set -e
for v in prev new; do
  if [ $v = prev ]; then export MI355_DECODE_LIB=$PWD/scripts/exp/lib_prev.so; else unset MI355_DECODE_LIB; fi
  for cfg in "32 32 129,192,255,300,512,768,2080" "16 64 512,1024,2080" "2 8 2080" "128 32 192,768" ; do
    set -- $cfg
    echo "== $v seqs=$1 heads=$2" >> gpurun_out/ab_attn_ctx.txt
    timeout -k 10 200 python3 scripts/bench_attn_ctx.py $1 $2 $3 2>/dev/null >> gpurun_out/ab_attn_ctx.txt
  done
done
