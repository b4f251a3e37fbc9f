"""One-shot all-reduce (mi355_allreduce_oneshot_*): two ranks -- two processes on the one GPU of
the test box, their buffers mapped into each other through HIP IPC -- against the sum formed on
the host in rank order (fp32, one rounding: what every rank must produce, bit for bit), eagerly
with changing message sizes and replayed from a HIP graph; and a single rank (sum of one).
Not run across xGMI links here: one-GPU boxes only."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import ops_ref as R  # noqa: E402

COUNTS = [8, 4096 * 16, 1000 * 8, 4096 * 128, 4096 * 32, 8, 4096 * 128]      # elements; 1 MiB = 4096 * 128 bf16


def _inputs(rank, i, count, dtype):
    rng = np.random.Generator(np.random.PCG64(1000 * i + rank))
    return R.round_to(rng.standard_normal(count).astype(np.float32) * (1 + rank), dtype)


def _worker(rank, world, conns, dtype, out_q):
    try:
        sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vllm-backend-public_amd")]
        import torch as T
        from mi355_decode.collectives import OneShotAllReduce
        T.cuda.set_device(0)
        tdt = {"bfloat16": T.bfloat16, "float16": T.float16}[dtype]

        def exchange(mine):
            # rank 0 collects and redistributes the handles over the pipes
            if rank == 0:
                hs = [mine] + [c.recv_bytes() for c in conns]
                for c in conns:
                    c.send(hs)
                return hs
            conns[0].send_bytes(mine)
            return conns[0].recv()

        ar = OneShotAllReduce(max_bytes=1 << 20, rank=rank, world=world, exchange=exchange)
        results = []
        for i, count in enumerate(COUNTS):                       # eager calls, sizes changing from call to call
            t = T.from_numpy(_inputs(rank, i, count, dtype)).to(tdt).cuda()
            ar.run(t)
            results.append(t.float().cpu().numpy())
        ar.check()
        # the same call replayed from a HIP graph: the round counter advances on the device
        count = 4096 * 32
        buf = T.zeros(count, dtype=tdt, device="cuda")
        ar.run(buf)
        T.cuda.synchronize()
        g = T.cuda.CUDAGraph()
        with T.cuda.graph(g):
            ar.run(buf)
        for k in range(3):
            buf.copy_(T.from_numpy(_inputs(rank, 100 + k, count, dtype)).to(tdt))
            g.replay()
            T.cuda.synchronize()
            results.append(buf.float().cpu().numpy())
        ar.check()
        ar.close()
        out_q.put((rank, "ok", results))
    except Exception as exc:                                      # noqa: BLE001
        import traceback
        out_q.put((rank, "error", traceback.format_exc() + repr(exc)))


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_two_ranks_on_one_gpu_match_the_host_sum(dtype):
    world = 2
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(0, world, [a], dtype, q)),
             ctx.Process(target=_worker, args=(1, world, [b], dtype, q))]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in procs:
            rank, status, payload = q.get(timeout=90)
            assert status == "ok", payload
            got[rank] = payload
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    cases = [(i, c) for i, c in enumerate(COUNTS)] + [(100 + k, 4096 * 32) for k in range(3)]
    for j, (i, count) in enumerate(cases):
        want = R.round_to(sum(_inputs(r, i, count, dtype).astype(np.float32) for r in range(world)), dtype)
        for r in range(world):
            np.testing.assert_array_equal(got[r][j], want, err_msg=f"rank {r}, call {j} ({count} elements)")


def test_single_rank_is_the_identity_and_sizes_are_checked():
    from mi355_decode._lib import Mi355Error
    from mi355_decode.collectives import OneShotAllReduce
    ar = OneShotAllReduce(max_bytes=1 << 16, rank=0, world=1)
    x = torch.randn(4096, device="cuda").to(torch.bfloat16)
    want = x.clone()
    ar.run(x)
    ar.check()
    assert torch.equal(x, want)
    assert not ar.fits(torch.zeros((1 << 15) + 8, dtype=torch.bfloat16, device="cuda"))   # beyond max_bytes
    assert not ar.fits(torch.zeros(12, dtype=torch.bfloat16, device="cuda"))               # not a multiple of 8
    with pytest.raises(ValueError):
        ar.run(torch.zeros(12, dtype=torch.bfloat16, device="cuda"))
    ar.close()
