"""One-shot all-reduce (mi355_allreduce_oneshot_*): two and four ranks -- processes on the one GPU
of the test box, their buffers mapped into each other through HIP IPC -- against the sum formed on
the host in rank order (fp32, one rounding: what every rank must produce, bit for bit), eagerly
with changing message sizes up to 2 MiB (256 rows of a 4096-wide model), from fp32 slabs (the slab
sum of a K-split projection formed inside the all-reduce launch) and replayed from a HIP graph; the
known-answer self-test the engine runs before it relies on the kernel; the error word copied out
with a step's results; and a single rank (sum of one).
Not run across xGMI links here: one-GPU boxes only."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import ops_ref as R  # noqa: E402

COUNTS = [8, 4096 * 16, 1000 * 8, 4096 * 128, 4096 * 256, 4096 * 32, 8, 4096 * 256]      # elements; 2 MiB = 4096 * 256 bf16
SLAB_CASES = [(2, 4096 * 16), (8, 4096 * 128), (3, 1000 * 8), (4, 4096 * 256)]            # (slabs, elements)
NORM_CASES = [(32, 4096, 0), (128, 4096, 4), (256, 4096, 0), (16, 8192, 2), (5, 3584, 0), (7, 1000, 1)]   # (rows, hidden, slabs)


def _sizes(world):
    """Eight ranks SHARE the test box's one GPU: a call's workgroups (one per 8 KB or per row, on every rank) spin
    until their chunk has arrived from every rank, so all ranks' workgroups of a call must be resident at once --
    on one card that caps the messages of an 8-rank call (256 KiB: 8 x 32 workgroups); a rank with a GPU of its
    own has room for the largest call (256 workgroups)."""
    if world < 8:
        return COUNTS, SLAB_CASES, NORM_CASES
    cap = 4096 * 32
    return ([min(c, cap) for c in COUNTS], [(n, min(c, cap)) for n, c in SLAB_CASES],
            [(min(r, 32), h, n) for r, h, n in NORM_CASES])


def _slabs(rank, i, n, count):
    rng = np.random.Generator(np.random.PCG64(77000 + 1000 * i + rank))
    return (rng.standard_normal((n, count)) * (1 + rank)).astype(np.float32)


def _slab_sum(s, dtype):
    acc = np.zeros(s.shape[1], np.float32)
    for g in range(s.shape[0]):
        acc = acc + s[g]
    return R.round_to(acc, dtype)


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

        ar = OneShotAllReduce(max_bytes=1 << 21, rank=rank, world=world, exchange=exchange)
        assert ar.self_test("cuda"), "the known-answer self-test failed"
        results = []
        COUNTS_W, SLAB_CASES_W, NORM_CASES_W = _sizes(world)
        for i, count in enumerate(COUNTS_W):                     # eager calls, sizes changing from call to call
            t = T.from_numpy(_inputs(rank, i, count, dtype)).to(tdt).cuda()
            ar.run(t)
            results.append(t.float().cpu().numpy())
        ar.check()
        for i, (n, count) in enumerate(SLAB_CASES_W):            # this rank's message formed from fp32 slabs in the launch
            sl = T.from_numpy(_slabs(rank, i, n + 1, count)).cuda()          # one slab more than is summed
            out = T.full((count,), 3.0, dtype=tdt, device="cuda")
            ar.run_slabs(sl, n, out)
            results.append(out.float().cpu().numpy())
        ar.queue_status(1)                                        # the error word, copied out with a step's results
        T.cuda.synchronize()
        ar.raise_if_failed(1)
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
        # the all-reduce with the residual add + RMSNorm behind it in one launch: bit-identical to the all-reduce
        # (from 16-bit rows or from slabs) followed by the add + norm kernel, eagerly and replayed from a graph
        from mi355_decode import ops
        for i, (rows, hidden, n) in enumerate(NORM_CASES_W):
            count = rows * hidden
            w = T.from_numpy(_inputs(0, 300 + i, hidden, dtype)).to(tdt).cuda()           # the same norm weight on every rank
            resid0 = T.from_numpy(_inputs(0, 400 + i, count, dtype)).to(tdt).cuda().view(rows, hidden)
            if n:
                sl = T.from_numpy(_slabs(rank, 50 + i, n, count)).cuda()
                ref = T.empty(rows, hidden, dtype=tdt, device="cuda")
                ar.run_slabs(sl, n, ref.view(-1))
            else:
                mine = T.from_numpy(_inputs(rank, 500 + i, count, dtype)).to(tdt).cuda().view(rows, hidden)
                ref = mine.clone()
                ar.run(ref.view(-1))
            resid_ref, x_ref = resid0.clone(), T.empty(rows, hidden, dtype=tdt, device="cuda")
            ops.fused_add_rms_norm(ref, resid_ref, w, 1e-5, out=x_ref)
            for two in (False, True):                            # one hop; two hops (rows dealt to the ranks)
                resid, x = resid0.clone(), T.full((rows, hidden), 7.0, dtype=tdt, device="cuda")
                if n:
                    ar.run_add_norm(resid, w, 1e-5, x, slabs=sl, num_slabs=n, two_shot=two)
                else:
                    ar.run_add_norm(resid, w, 1e-5, x, data=mine, two_shot=two)
                T.cuda.synchronize()
                assert T.equal(resid, resid_ref) and T.equal(x, x_ref), ("fused all-reduce + norm", rows, hidden, n, two)
            results.append(x.float().cpu().numpy().ravel())          # every rank must hold the same rows
        rows, hidden = 32, 4096
        mine = T.zeros(rows, hidden, dtype=tdt, device="cuda")
        resid, x = T.zeros(rows, hidden, dtype=tdt, device="cuda"), T.zeros(rows, hidden, dtype=tdt, device="cuda")
        w = T.from_numpy(_inputs(0, 399, hidden, dtype)).to(tdt).cuda()
        ar.run_add_norm(resid, w, 1e-5, x, data=mine)
        T.cuda.synchronize()
        g2 = T.cuda.CUDAGraph()
        with T.cuda.graph(g2):
            ar.run_add_norm(resid, w, 1e-5, x, data=mine)
        resid2, x2 = T.zeros_like(resid), T.zeros_like(x)        # the two-hop form, captured too
        ar.run_add_norm(resid2, w, 1e-5, x2, data=mine, two_shot=True)
        T.cuda.synchronize()
        g3 = T.cuda.CUDAGraph()
        with T.cuda.graph(g3):
            ar.run_add_norm(resid2, w, 1e-5, x2, data=mine, two_shot=True)
        for k in range(2):
            mine.copy_(T.from_numpy(_inputs(rank, 600 + k, rows * hidden, dtype)).to(tdt).view(rows, hidden))
            resid.copy_(T.from_numpy(_inputs(0, 700 + k, rows * hidden, dtype)).to(tdt).view(rows, hidden))
            resid2.copy_(resid)
            g2.replay()
            g3.replay()
            T.cuda.synchronize()
            assert T.equal(resid, resid2) and T.equal(x, x2), ("two-hop form replayed from a graph", k)
            results.append(x.float().cpu().numpy().ravel())
            results.append(resid.float().cpu().numpy().ravel())
        # the greedy pick over a vocabulary split across the ranks: keys max-reduced and turned into ids in one launch
        rows, vl = 37, 1000
        lg = T.from_numpy(_inputs(rank, 800, rows * vl, dtype)).to(tdt).cuda().view(rows, vl)
        lg[3] = 0.5                                               # a row tied across all ranks: the lowest index wins
        keys = ops.argmax_shard_key(lg, rank * vl)
        ids, ids2 = T.full((rows,), -1, dtype=T.int64, device="cuda"), T.full((rows,), -1, dtype=T.int64, device="cuda")
        ar.run_max_keys(keys, ids, ids2)
        T.cuda.synchronize()
        assert T.equal(ids, ids2)
        results.append(ids.cpu().numpy().astype(np.float32))
        ar.check()
        ar.close()
        out_q.put((rank, "ok", results))
    except Exception as exc:                                      # noqa: BLE001
        import traceback
        out_q.put((rank, "error", traceback.format_exc() + repr(exc)))


@pytest.mark.parametrize("dtype,world", [("bfloat16", 2), ("float16", 2), ("bfloat16", 4), ("float16", 4), ("bfloat16", 8)])
def test_ranks_on_one_gpu_match_the_host_sum(dtype, world):
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(world - 1)]
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(0, world, [a for a, _ in pipes], dtype, q))] + \
            [ctx.Process(target=_worker, args=(r, world, [pipes[r - 1][1]], dtype, q)) for r in range(1, world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in procs:
            rank, status, payload = q.get(timeout=90 if world < 8 else 150)
            assert status == "ok", payload
            got[rank] = payload
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    def rank_sum(parts):
        acc = np.zeros_like(parts[0], dtype=np.float32)
        for part in parts:                                       # rank order, fp32, one rounding
            acc = acc + part.astype(np.float32)
        return R.round_to(acc, dtype)

    counts, slab_cases, _ = _sizes(world)
    wants = [rank_sum([_inputs(r, i, c, dtype) for r in range(world)]) for i, c in enumerate(counts)]
    wants += [rank_sum([_slab_sum(_slabs(r, i, n + 1, c)[:n], dtype) for r in range(world)]) for i, (n, c) in enumerate(slab_cases)]
    wants += [rank_sum([_inputs(r, 100 + k, 4096 * 32, dtype) for r in range(world)]) for k in range(3)]
    extra = len(NORM_CASES) + 5                                  # the fused all-reduce + norm results: checked against the
    assert all(len(got[r]) == len(wants) + extra for r in range(world))   # unfused pair inside the workers; here: equal on every rank
    for j, want in enumerate(wants):
        for r in range(world):
            np.testing.assert_array_equal(got[r][j], want, err_msg=f"rank {r}, call {j} ({want.size} elements)")
    for j in range(len(wants), len(wants) + extra):
        for r in range(1, world):
            np.testing.assert_array_equal(got[r][j], got[0][j], err_msg=f"rank {r}, fused call {j}")
    full = np.concatenate([_inputs(r, 800, 37 * 1000, dtype).reshape(37, 1000) for r in range(world)], axis=1)
    full[3] = 0.5
    np.testing.assert_array_equal(got[0][len(wants) + len(NORM_CASES) + 4], full.argmax(axis=1).astype(np.float32))
    # the graph replays against the oracle: all-reduce in rank order, then the add + norm
    for k in range(2):
        summed = rank_sum([_inputs(r, 600 + k, 32 * 4096, dtype) for r in range(world)]).reshape(32, 4096)
        x_want, resid_want = R.fused_add_rms_norm(summed, _inputs(0, 700 + k, 32 * 4096, dtype).reshape(32, 4096),
                                                  _inputs(0, 399, 4096, dtype), 1e-5, dtype)
        j = len(wants) + len(NORM_CASES) + 2 * k
        np.testing.assert_array_equal(got[0][j + 1].reshape(32, 4096), resid_want)      # the residual add is exact
        bad = got[0][j].reshape(32, 4096) != x_want
        assert bad.mean() <= 0.005, bad.mean()                                          # the norm's bar of the ops tests


def test_single_rank_is_the_identity_and_sizes_are_checked():
    from mi355_decode._lib import Mi355Error
    from mi355_decode.collectives import OneShotAllReduce
    ar = OneShotAllReduce(max_bytes=1 << 16, rank=0, world=1)
    x = torch.randn(4096, device="cuda").to(torch.bfloat16)
    want = x.clone()
    ar.run(x)
    ar.check()
    assert torch.equal(x, want)
    slabs = torch.randn(3, 4096, device="cuda")
    out = torch.empty(4096, dtype=torch.bfloat16, device="cuda")
    ar.run_slabs(slabs, 3, out)                                   # one rank: the rounded slab sum itself
    assert torch.equal(out, ((slabs[0] + slabs[1]) + slabs[2]).to(torch.bfloat16))
    with pytest.raises(Mi355Error, match="num_slabs"):
        ar.run_slabs(torch.randn(9, 4096, device="cuda"), 9, out)
    from mi355_decode import ops
    w = torch.randn(4096, device="cuda").to(torch.bfloat16)
    part, resid = torch.randn(8, 4096, device="cuda").to(torch.bfloat16), torch.randn(8, 4096, device="cuda").to(torch.bfloat16)
    resid_ref, x_ref = resid.clone(), torch.empty_like(resid)
    ops.fused_add_rms_norm(part.clone(), resid_ref, w, 1e-5, out=x_ref)
    x = torch.empty_like(resid)
    ar.run_add_norm(resid, w, 1e-5, x, data=part)                 # one rank: the add + norm of its own rows
    assert torch.equal(resid, resid_ref) and torch.equal(x, x_ref)
    with pytest.raises(ValueError):
        ar.run_add_norm(resid, w, 1e-5, x)                        # neither data nor slabs
    assert not ar.fits_add_norm(257, 64, torch.bfloat16) and not ar.fits_add_norm(8, 8200, torch.bfloat16)
    assert ar.self_test("cuda")
    assert not ar.fits(torch.zeros((1 << 15) + 8, dtype=torch.bfloat16, device="cuda"))   # beyond max_bytes
    assert not ar.fits(torch.zeros(12, dtype=torch.bfloat16, device="cuda"))               # not a multiple of 8
    with pytest.raises(ValueError):
        ar.run(torch.zeros(12, dtype=torch.bfloat16, device="cuda"))
    ar.close()
