# bench.py --gpus 4 started by itself with all four ranks on ONE card (gloo process group, the hand-written
# collectives forced on): the production TP form (HIP graphs, no process-group call in a decode step) with the
# all-reduce + norm in one hop and in two
for two in ${FORMS:-0 1}; do
  MI355_ALLREDUCE_TWO_SHOT=$two MI355_DIST_BACKEND=gloo MI355_ONESHOT_ALLREDUCE=1 timeout -k 10 500 python3 bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline --no-served 2>gpurun_out/rehearse4_$two.err | grep '^{' > gpurun_out/rehearse4_$two.json
  [ -s gpurun_out/rehearse4_$two.json ] || { echo "two_shot=$two failed:"; grep -m2 Mi355Error gpurun_out/rehearse4_$two.err; continue; }
  python3 -c "
import json
d = json.load(open('gpurun_out/rehearse4_$two.json'))
print('two_shot=$two', d['value'], d['ms_per_step'], d['ranks_seen'], d['degraded']['allreduce'], d['degraded']['hip_graphs'])"
done
