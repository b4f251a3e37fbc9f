"""Tensor parallelism end to end on ONE GPU: two engine processes (rank 0 and 1 of a
TP=2 group) share the card and exchange their partial sums over gloo -- RCCL refuses two
ranks on one device, and gloo cannot be graph-captured, so the steps run eagerly.  What is
pinned: the head, MLP and vocabulary sharding of the weights, both all-reduce call sites,
(over gloo, and through the hand-written one-shot all-reduce over IPC-mapped buffers), the
key max-reduce of greedy sampling and the gather of whole rows for random sampling, the
lock-step of the two schedulers, and the generated tokens against the numpy oracle of the
unsharded model (a sharded sum rounds differently: a sequence may leave the oracle at a
near-tie of the oracle's own logits, as in test_model_gpu)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = 16
LENGTHS = [5, 16, 17, 33]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, dtype, shard_mlp, q, oneshot=False):
    import faulthandler
    faulthandler.dump_traceback_later(80, exit=True)      # a rank stuck in a collective says where, then leaves
    try:
        for p in (ROOT, os.path.join(ROOT, "vllm-backend-public_amd"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK="0", MI355_DIST_BACKEND="gloo", MI355_TP_MLP="1" if shard_mlp else "0",
                          MI355_TP_VOCAB="1" if shard_mlp else "0", MI355_ONESHOT_ALLREDUCE="1" if oneshot else "0",
                          MI355_ALLREDUCE_TWO_SHOT="1" if oneshot == "graphs-two-hop" else "0")
        import torch
        import torch.distributed as dist
        from test_model_gpu import build, engine_generate
        from oracle.tiny_models import make_prompts
        graphs = oneshot in ("graphs", "graphs-two-hop")     # with every decode collective on the one-shot kernel the step is captured, gloo or not
        eng, cfg, w = build(name, dtype, tensor_parallel_size=world, enforce_eager=not graphs)
        r = eng.runner
        assert r.use_graphs == graphs
        assert r.tp_size == world and r.mlp_sharded == shard_mlp == r.vocab_sharded
        assert r.F * (world if shard_mlp else 1) == cfg["intermediate_size"]
        calls, nested = [0], [0]
        inner, inner_slabs = r._allreduce, r._allreduce_slabs

        def counted(t):
            if not nested[0]:
                calls[0] += 1
            inner(t)

        def counted_slabs(partial, num_slabs, out):       # the all-reduce of a K-split projection: one collective too
            calls[0] += 1
            nested[0] += 1
            try:
                inner_slabs(partial, num_slabs, out)
            finally:
                nested[0] -= 1
        r._allreduce, r._allreduce_slabs = counted, counted_slabs
        toks = engine_generate(eng, make_prompts(name, len(LENGTHS), LENGTHS), GEN)
        steps = eng.stats["decode_steps"] + eng.stats["prefill_steps"]
        per_step = r.L * (2 if shard_mlp else 1)               # a step launched ahead may not be counted yet
        # with the one-shot kernel a decode step's all-reduces run together with the add + norm behind them
        # (one launch, past the two wrappers counted above)
        total = calls[0] + r.fused_allreduce_norms
        if graphs:                                              # replays are not counted on the host
            assert r.oneshot is not None and r.graphs and r.fused_allreduce_norms > 0 and r.oneshot_key_reductions > 0
            r.oneshot.check()
        elif oneshot:                                           # every message of these tiny models fits the one-shot kernel
            assert total % per_step == 0 and total >= steps * per_step, (calls[0], r.fused_allreduce_norms, steps)
            assert r.oneshot is not None and r.oneshot_calls == total, (r.oneshot_calls, total)
            assert r.fused_allreduce_norms >= eng.stats["decode_steps"] * per_step > 0
            r.oneshot.check()
        else:
            assert total % per_step == 0 and total >= steps * per_step, (calls[0], r.fused_allreduce_norms, steps)
            assert r.oneshot is None and r.fused_allreduce_norms == 0
        # random sampling and log-probabilities need whole rows: gathered from the ranks' slices
        from mi355_decode.sampling_params import SamplingParams
        sampled = {}
        for i, p in enumerate(make_prompts(name, 3, [7, 19, 30])):
            eng.add_request(f"s{i}", [int(t) for t in p], SamplingParams(max_tokens=6, temperature=0.8, top_k=20,
                                                                       top_p=0.9, seed=5 + i, logprobs=2,
                                                                       prompt_logprobs=1, ignore_eos=True))
            sampled[f"s{i}"] = ([], [], [len(p)])
        while eng.has_unfinished():
            for o in eng.step():
                sampled[o.request_id][0].extend(o.new_token_ids)
                sampled[o.request_id][1].extend(o.logprobs or [])
                if o.prompt_logprobs is not None:
                    sampled[o.request_id][2].append([None if e is None else [e[0], e[1]] for e in o.prompt_logprobs])
        for ids, lps, (plen, plp) in sampled.values():
            assert len(ids) == 6 and len(lps) == 6 and len(plp) == plen and plp[0] is None
            assert all(e[0] <= 0.0 and len(e[1]) == 1 for e in plp[1:])
            for tok, (lp, top) in zip(ids, lps):
                assert lp <= 0.0 and len(top) == 2 and top[0][1] >= top[1][1] and top[0][1] >= lp - 1e-6
                assert 0 <= tok < cfg["vocab_size"] and all(0 <= t < cfg["vocab_size"] for t, _ in top)
        dist.barrier()                                    # both ranks got here: nobody is left inside a collective
        q.put((rank, "ok", [toks.tolist(), {k: [v[0], [[lp, top] for lp, top in v[1]], v[2]] for k, v in sampled.items()}]))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), None))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("shard_mlp,oneshot", [(True, False), (False, False), (True, True), (True, "graphs"),
                                               (True, "graphs-two-hop")],
                         ids=["mlp_sharded", "mlp_replicated", "mlp_sharded_oneshot_allreduce",
                              "mlp_sharded_oneshot_allreduce_in_hip_graphs", "mlp_sharded_two_hop_allreduce_in_hip_graphs"])
@pytest.mark.parametrize("name,dtype", [("llama_d128", "bfloat16"), ("qwen2_d128", "float16")])
def test_two_ranks_generate_like_the_unsharded_oracle(name, dtype, shard_mlp, oneshot):
    import torch.multiprocessing as mp
    from oracle import ops_ref as R
    from oracle.tiny_models import make_prompts, make_weights
    from test_model_gpu import LOGIT_TOL, compare_tokens, oracle_run
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, dtype, shard_mlp, q, oneshot)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=100 if not res or res[-1][1] == "ok" else 5))
    except queue.Empty:
        pass                                             # a rank that failed leaves its peer waiting in a collective
    finally:
        for p in procs:
            p.join(5 if len(res) < len(procs) else 60)
            if p.is_alive():
                p.kill()
    for rank, msg, _ in res:
        assert msg == "ok", f"rank {rank}: {msg}"
    assert len(res) == len(procs), "a rank produced no result within the time limit"
    res.sort()
    t0, t1 = np.array(res[0][2][0]), np.array(res[1][2][0])
    assert np.array_equal(t0, t1), "the two ranks sampled different tokens"
    assert res[0][2][1] == res[1][2][1], "the two ranks drew different random samples or log-probabilities"
    _, cfg, w = make_weights(name)
    w = {k: R.round_to(v, dtype) for k, v in w.items()}
    ref_tokens, ref_logits = oracle_run(cfg, w, dtype, make_prompts(name, len(LENGTHS), LENGTHS), GEN)
    scale = np.abs(ref_logits[0]).max()
    exact = compare_tokens(t0, ref_tokens, ref_logits, 2 * LOGIT_TOL[dtype] * scale)
    print(f"{name} {dtype} tp2 {'mlp sharded' if shard_mlp else 'mlp replicated'}: "
          f"{exact}/{ref_tokens.size} tokens equal to the unsharded oracle")
    # the bar is compare_tokens' margin rule alone: a sequence may leave the oracle's only at a step where
    # the oracle's own margin is inside the logit-error bound (after which it is a different generation)
