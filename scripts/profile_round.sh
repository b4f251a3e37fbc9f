#!/bin/bash
# The round's profile set on one MI355X: kernel stats, HBM traffic (two PMC passes), MFMA-busy / wave-cycle mix.
# usage: scripts/profile_round.sh <tag>   (writes gpurun_out/<tag>_*; copy the summaries to profiles/)
set -u
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
common="--no-cpu-baseline --no-served --segments 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --steps 30 --warmup 3 $common > $out/${tag}_stats.log 2>&1 || exit 1
f=$(find $out/${tag}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats.csv
find $out/${tag}_stats -name "*kernel_trace.csv" -delete
export MI355_TILED_LINEARS=1     # counter collection serialises kernels: no timed choice of GEMM form on distorted timings
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_$c -- python3 bench.py --steps 4 --warmup 1 --no-kernel-timing $common > $out/${tag}_pmc_$c.log 2>&1 || exit 2
done
python3 scripts/pmc_summary.py $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_traffic.json > /dev/null || exit 3
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc_mfma -- python3 bench.py --steps 4 --warmup 1 --no-kernel-timing $common > $out/${tag}_pmc_mfma.log 2>&1 || exit 4
python3 scripts/pmc_kernel_mix.py $out/${tag}_pmc_mfma > $out/${tag}_pmc_mfma.json || exit 5
rm -rf $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_mfma $out/${tag}_stats
echo done
