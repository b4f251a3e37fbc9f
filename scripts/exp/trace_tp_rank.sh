# kernel trace of one tensor-parallel rank's decode step (one GPU, collectives stubbed): per-kernel time of the last
# replayed step.  usage: trace_tp_rank.sh <tp> <tag> [--oneshot]   (--oneshot: a one-rank communicator of the one-shot kernel instead of stubs)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tp=${1:-4}; tag=${2:-tp4}; extra=${3:-}
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -- python3 scripts/bench_tp_rank.py --tps $tp --only sharded-timed $extra > gpurun_out/tl_$tag.log 2>&1 || exit 1
python3 scripts/step_timeline.py gpurun_out/tl_$tag gpurun_out/tl_$tag.json > /dev/null || exit 2
rm -rf gpurun_out/tl_$tag
echo done
