"""Tensor-parallel path on CPU, world_size 2 over gloo: head sharding of the
weights (column-parallel qkv, row-parallel o_proj whose partial sums are
all-reduced) and the replicated-scheduler control channel of the serving loop."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "vllm-backend-public_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _shard_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        from mi355_decode import weights as W
        from mi355_decode.config import ModelConfig
        from oracle.tiny_models import make_weights
        arch, cfg, w = make_weights("llama_d128")
        mc = ModelConfig.from_dict({**cfg, "architectures": [arch]})
        sd = {k: torch.from_numpy(v) for k, v in w.items()}
        full = W.from_state_dict(mc, sd, torch.float32, "cpu")
        mine = W.from_state_dict(mc, sd, torch.float32, "cpu", tp_rank=rank, tp_size=world)
        nh, nkv, d, H = mc.num_attention_heads // world, mc.num_key_value_heads // world, mc.head_dim, mc.hidden_size
        lw, fw = mine.layers[0], full.layers[0]
        assert lw.qkv_w.shape == ((nh + 2 * nkv) * d, H) and lw.o_w.shape == (H, nh * d)
        assert torch.equal(lw.gate_up_w, fw.gate_up_w) and torch.equal(mine.lm_head, full.lm_head)   # replicated
        # attention block on this rank's heads, then all-reduce of the o_proj partial sums
        g = torch.Generator().manual_seed(0)
        x = torch.randn(5, H, generator=g)
        qkv_full = x @ fw.qkv_w.t()
        qkv_mine = x @ lw.qkv_w.t()
        NH, NKV = mc.num_attention_heads, mc.num_key_value_heads
        q_full = qkv_full[:, : NH * d]
        assert torch.allclose(qkv_mine[:, : nh * d], q_full[:, rank * nh * d:(rank + 1) * nh * d])
        k_full = qkv_full[:, NH * d:(NH + NKV) * d]
        assert torch.allclose(qkv_mine[:, nh * d:(nh + nkv) * d], k_full[:, rank * nkv * d:(rank + 1) * nkv * d])
        attn_full = torch.tanh(q_full)                      # any per-head function stands in for attention
        attn_mine = attn_full[:, rank * nh * d:(rank + 1) * nh * d]
        partial = attn_mine @ lw.o_w.t()
        dist.all_reduce(partial)
        assert torch.allclose(partial, attn_full @ fw.o_w.t(), atol=1e-4)
        # sharded MLP: rows of gate/up and columns of down; SwiGLU is elementwise over the
        # intermediate dimension, so the all-reduced partial sums are the full MLP output
        sh = W.from_state_dict(mc, sd, torch.float32, "cpu", tp_rank=rank, tp_size=world, shard_mlp=True)
        F = mc.intermediate_size
        Fl = F // world
        sw = sh.layers[0]
        assert sw.gate_up_w.shape == (2 * Fl, H) and sw.down_w.shape == (H, Fl)
        assert torch.equal(sw.gate_up_w[:Fl], fw.gate_up_w[rank * Fl:(rank + 1) * Fl])
        assert torch.equal(sw.gate_up_w[Fl:], fw.gate_up_w[F + rank * Fl:F + (rank + 1) * Fl])
        assert torch.equal(sw.qkv_w, lw.qkv_w) and torch.equal(sh.lm_head, full.lm_head)

        def mlp(wl, fl):
            gu = x @ wl.gate_up_w.t()
            return (torch.nn.functional.silu(gu[:, :fl]) * gu[:, fl:]) @ wl.down_w.t()
        part = mlp(sw, Fl)
        dist.all_reduce(part)
        assert torch.allclose(part, mlp(fw, F), atol=1e-4)
        # vocabulary-parallel lm_head: each rank's best-token key, max-reduced, is the global argmax
        from oracle import ops_ref as R
        vw = W.from_state_dict(mc, sd, torch.float32, "cpu", tp_rank=rank, tp_size=world, shard_vocab=True)
        Vl = mc.vocab_size // world
        assert vw.lm_head.shape == (Vl, H) and torch.equal(vw.lm_head, full.lm_head[rank * Vl:(rank + 1) * Vl])
        logits_full = (x @ full.lm_head.t()).numpy()
        logits_full[0, [3, Vl + 1]] = logits_full.max() + 1.0            # a tie across the two slices
        keys = torch.from_numpy(R.argmax_shard_key(logits_full[:, rank * Vl:(rank + 1) * Vl], rank * Vl))
        dist.all_reduce(keys, op=dist.ReduceOp.MAX)
        assert np.array_equal(R.argmax_key_to_index(keys.numpy()), R.argmax_lowest(logits_full))
        q.put((rank, "ok"))
    except Exception as exc:  # noqa: BLE001
        q.put((rank, repr(exc)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _serving_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        from test_engine_cpu import FakeRunner, reference
        from mi355_decode.engine import LLMEngine
        from mi355_decode.sampling_params import SamplingParams
        from mi355_decode.serving import EngineThread
        import threading
        group = dist.new_group(backend="gloo")
        th = EngineThread(lambda: LLMEngine(FakeRunner(64, 4, 256), 4, 64, 256), control_group=group, rank=rank)
        got, done = {}, threading.Event()
        prompts = [list(range(3 + i, 20 + 2 * i)) for i in range(6)]

        def sink_for(i):
            def sink(o):
                got.setdefault(i, []).extend(o.new_token_ids)
                if o.finished and sum(1 for k in got if len(got[k]) == 10) == len(prompts):
                    done.set()
            return sink

        if rank == 0:
            th.start()
            assert th.ready.wait(30)
            for i, p in enumerate(prompts):
                th.submit(f"r{i}", p, SamplingParams(max_tokens=10, temperature=0.0), sink_for(i))
            assert done.wait(60), "generation did not finish"
            th.shutdown()
            th.thread.join(30)
            for i, p in enumerate(prompts):
                assert got[i] == reference(p, 10)
            steps = th.engine.stats["decode_steps"] + th.engine.stats["prefill_steps"]
        else:
            th._run()                                        # follows rank 0 until shutdown
            steps = th.engine.stats["decode_steps"] + th.engine.stats["prefill_steps"]
        assert th.failed is None, repr(th.failed)
        allsteps = [None] * world
        dist.all_gather_object(allsteps, (steps, th.engine.stats["decode_tokens"]), group=group)
        assert allsteps[0] == allsteps[1], f"ranks diverged: {allsteps}"     # identical state machines
        q.put((rank, "ok"))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(worker):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_head_sharding_and_allreduce_world2():
    _run(_shard_worker)


def test_replicated_scheduler_control_channel_world2():
    _run(_serving_worker)
