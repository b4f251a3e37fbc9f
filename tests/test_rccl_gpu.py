"""RCCL as the engine uses it, as far as one GPU allows: a process group of ONE rank on the
`nccl` backend (the box has one card and RCCL refuses two ranks on one device), created the way
`mi355_decode.llm.init_distributed` creates it, then the three collectives of a tensor-parallel
decode step -- sum all-reduce of a 16-bit activation, max all-reduce of the 64-bit greedy keys,
all-gather of logit slices -- warmed up, captured into a HIP graph in thread-local mode between
two of the library's own kernels (the order `ModelRunner.capture` uses) and replayed.  With one
rank every collective is the identity, so the replayed step must equal the same kernels run
without it.  What this pins: the RCCL build on the box loads, builds a communicator with
`device_id`, and lets its collectives be captured and replayed next to hand-written kernels;
what it cannot: anything about a second rank or an xGMI link."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import datetime, os, sys
sys.path[:0] = [os.path.join(ROOT, "vllm-backend-public_amd"), ROOT]
import torch
import torch.distributed as dist
from mi355_decode import ops

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=60))
g0 = torch.Generator(device="cpu").manual_seed(5)
rows, H, Vl = 32, 4096, 2048
resid = torch.randn(rows, H, generator=g0).to(torch.bfloat16).to(dev)
w = (1 + 0.1 * torch.randn(H, generator=g0)).to(torch.bfloat16).to(dev)
logits = torch.randn(rows, Vl, generator=g0).to(torch.bfloat16).to(dev)
x = torch.empty_like(resid)
act = torch.empty_like(resid)
keys = torch.empty(rows, dtype=torch.int64, device=dev)
ids = torch.empty(rows, dtype=torch.int64, device=dev)
gathered = torch.empty(1, rows, Vl, dtype=torch.bfloat16, device=dev)


def body(collectives):
    ops.rms_norm(resid, w, 1e-5, out=x)
    act.copy_(x)
    if collectives:
        dist.all_reduce(act)
    ops.argmax_shard_key(logits, 0, out=keys)
    if collectives:
        dist.all_reduce(keys, op=dist.ReduceOp.MAX)
    ops.argmax_key_to_index(keys, out=ids)
    if collectives:
        dist.all_gather_into_tensor(gathered, logits)


body(False)
torch.cuda.synchronize()
want_act, want_ids = act.clone(), ids.clone()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        body(True)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
assert torch.equal(act, want_act) and torch.equal(ids, want_ids), "eager collectives changed the values"
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode="thread_local"):
    body(True)
for k in range(3):
    act.zero_(); ids.zero_(); gathered.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(act, want_act), f"replay {k}: all-reduced activation differs"
    assert torch.equal(ids, want_ids), f"replay {k}: greedy ids differ"
    assert torch.equal(gathered[0], logits), f"replay {k}: gathered logits differ"
assert torch.equal(ids.cpu(), logits.float().argmax(-1).cpu())
dist.destroy_process_group()
print("RCCL_CAPTURE_OK", flush=True)
"""


def test_rccl_single_rank_collectives_capture_and_replay():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], env=env, capture_output=True,
                       text=True, timeout=240)
    assert p.returncode == 0 and "RCCL_CAPTURE_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
