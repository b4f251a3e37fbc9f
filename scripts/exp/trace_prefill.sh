# kernel trace of one prefill step of the headline workload: per-kernel time (scripts/prefill_timeline.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_prefill -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-served --segments 1 > gpurun_out/tl_prefill.log 2>&1 || exit 1
python3 scripts/prefill_timeline.py gpurun_out/tl_prefill gpurun_out/tl_prefill.json > /dev/null || exit 2
rm -rf gpurun_out/tl_prefill
echo done
