#!/usr/bin/env python3
"""Decode-throughput benchmark of the MI355X engine on BASELINE.json's headline
workload: Llama-3-8B geometry, bf16, batch 32 per GPU, 128 tokens in / 128 out,
synthetic weights and token-id prompts.

    python bench.py --gpus 1 --steps 127 --warmup 10
    python bench.py --gpus 4 --steps 127 --warmup 10          # starts its own 4 ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 4 --steps 127 --warmup 10

The tensor-parallel degree reaches the engine as a bare integer and nothing upstream
starts ranks (benchmark-vllm/vllm_templates.py:55), so `--gpus N` outside a launcher
starts N rank processes itself, like the server does: the parent touches no GPU, gives
each child RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, relays rank 0's JSON line and
exits non-zero if any rank did.

A "step" is one engine decode step: every running sequence advances one token
(scheduler bookkeeping, two small host-to-device copies, one HIP-graph replay of
the whole model step including greedy sampling, token ids back to the host).
The timed region starts with all prompts prefilled (K/V resident in HBM) and
covers exactly K steps; contexts grow from prompt_len+1 as in a real generation,
and a fresh batch is prefilled (untimed) whenever the sequences reach their
output length.  With N > 1 GPUs the model runs tensor-parallel over attention
heads and the MLP's intermediate dimension (one process per GPU, RCCL all-reduce
of the attention output projection and of the down projection; MI355_TP_MLP=0
keeps the MLP replicated) and the batch grows to 32*N sequences (weak scaling, BASELINE.json configs[2]).

Rank 0 prints ONE JSON line: the contract fields plus
  ttft_ms_p50   time to first token of the batch's prefill (prompt_len*batch tokens)
  roofline      the dominant kernel of the step (the gate/up projection GEMM):
                algorithmic bytes / average launch time measured here with HIP events
  step_roofline the whole decode step against the HBM roofline (bytes per step from
                the model geometry and mean context / measured step time)
  kernels       the hand-written HIP kernels, each timed with HIP events
  cpu_baseline  the numpy oracle timed on this host on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
HBM_READ_FLOOR_GBS = 6600.0          # measured: read-only streaming kernel, buffer >> caches
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak


def baseline_config_label(a, fp8_weights) -> str:
    """Which entry of BASELINE.json's `configs` the run is, if any."""
    shape = (a.model, a.batch_per_gpu, a.prompt_len, a.gen_len, bool(fp8_weights))
    if shape == ("llama-3-8b", 32, 128, 128, False):
        return f"BASELINE.json configs[{1 if a.gpus == 1 else 2}]"
    if shape == ("qwen2-7b", 64, 512, 512, True) and a.gpus == 1:
        return "BASELINE.json configs[4]"
    if a.model == "llama-3-70b" and (a.prompt_len, a.gen_len) == (2048, 256) and not fp8_weights:
        return ("BASELINE.json configs[3]" if a.gpus == 8 and a.batch_per_gpu == 2 else
                f"the model and request shape of BASELINE.json configs[3] on {a.gpus} GPU(s), not that config's TP=8")
    return "not one of BASELINE.json's configs"


def launch_ranks(n, cmd, env=None, timeout=None):
    """Start `cmd` n times as rank processes of one job on this node (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR=127.0.0.1, a free MASTER_PORT) and wait for all of them.
    The caller must not have initialised a GPU: the children are fresh processes, nothing is
    exec'ed over this one.  Rank 0's stdout is captured and returned; the other ranks' stdout
    goes to this process's stderr, everyone's stderr is inherited.  A rank that exits non-zero
    ends the others (they would wait in a collective for it).  Returns (exit code, rank 0's
    stdout): the first non-zero exit code of any rank, else 0."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ if env is None else env, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(list(cmd), env=dict(base, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=subprocess.PIPE if r == 0 else 2, text=True))   # 2: this process's stderr
    out0, rc, t_end = [], 0, (time.time() + timeout if timeout else None)
    import threading
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                print(f"bench: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
        if live and (rc != 0 or (t_end and time.time() > t_end)):
            rc = rc or 124
            for r in live:
                procs[r].terminate()
            deadline = time.time() + 10
            for r in live:
                try:
                    procs[r].wait(timeout=max(deadline - time.time(), 0.1))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        if live:
            time.sleep(0.05)
    reader.join(timeout=10)
    return rc, (out0[0] if out0 else "")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=127)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--gen-len", type=int, default=128)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--quantization", default=None, help="fp8: weight-only fp8 decode GEMMs (BASELINE.json configs[4])")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (result is marked invalid)")
    ap.add_argument("--decode-splits", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--eager", action="store_true", help="debug only: no HIP graphs")
    ap.add_argument("--oneshot-allreduce", action="store_true",
                    help="N > 1: force the hand-written one-shot all-reduce for the decode-sized messages under any "
                         "backend (MI355_ONESHOT_ALLREDUCE=1; under RCCL it is the default, =0 switches it off)")
    ap.add_argument("--segments", type=int, default=5,
                    help="timed regions of exactly --steps steps each; the median one is reported")
    ap.add_argument("--no-served", action="store_true", help="skip the measurement through the HTTP server")
    ap.add_argument("--served-runs", type=int, default=5)
    return ap.parse_args()


def model_bytes(cfg, tp, batch, mean_ctx, elem=2, layer_w_elem=None, mlp_tp=1, vocab_tp=1):
    """Algorithmic HBM bytes of one decode step per GPU: every weight once
    (attention projections sharded over tp, the MLP over mlp_tp, lm_head over vocab_tp),
    the K/V pages of the mean context, the new K/V rows, and the logits."""
    H, F, V, L, d = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.num_hidden_layers, cfg.head_dim
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    attn_w = ((nh + 2 * nkv) * d * H + nh * d * H) // tp
    mlp_w = 3 * F * H // mlp_tp
    norms = 2 * H
    lw = elem if layer_w_elem is None else layer_w_elem      # fp8: 1 byte per layer weight + fp32 row scales
    rows = ((nh + 2 * nkv) * d + H) // tp + 2 * F // mlp_tp + H
    w_bytes = L * ((attn_w + mlp_w) * lw + norms * elem + (rows * 4 if lw != elem else 0)) + (V // vocab_tp * H + H) * elem
    kv_tok = 2 * (nkv // tp) * d * elem * L
    return w_bytes + batch * mean_ctx * kv_tok + batch * kv_tok + batch * V // vocab_tp * elem, w_bytes


def make_prompts(vocab, batch, plen, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    lo, hi = min(1000, vocab // 4), max(vocab - 1000, vocab // 2)
    return torch.randint(lo, hi, (batch, plen), generator=g).tolist()


def time_kernels(engine, batch, ctx_len):
    """Average launch time of each kernel of the decode step.  Each kernel is
    captured into a HIP graph that walks over the layers (consecutive launches
    touch different weights / KV pages, so nothing is served from the Infinity
    Cache that a real step would fetch from HBM) and the replay is timed with HIP
    events on the stream it is launched on: the same back-to-back launch
    conditions as inside the decode-step graph, without host launch gaps."""
    import torch
    from mi355_decode import ops
    r = engine.runner
    b = batch
    L = r.L
    reps = max(1, 64 // L)
    out = {}

    def timed(fn, bytes_per_launch):
        for i in range(min(L, 4)):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(reps):
                for i in range(L):
                    fn(i)
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            g.replay()
        e.record()
        e.synchronize()
        us = s.elapsed_time(e) * 1000.0 / (3 * reps * L)
        return {"us": round(us, 2), "bytes": int(bytes_per_launch),
                "gbs": round(bytes_per_launch / us / 1e3, 1), "frac": round(bytes_per_launch / us / 1e3 / HBM_PEAK_GBS, 4)}

    es = 2
    x, qkv, attn, o, gu, act, dn, resid = r.x[:b], r.qkv[:b], r.attn[:b], r.o[:b], r.gu[:b], r.act[:b], r.dn[:b], r.resid[:b]
    W = r.w.layers
    wes = 1 if r.weight_fp8 else es                      # bytes per layer-weight element
    gemm = lambda m, n, k, out_es=es, w_es=None: n * k * (wes if w_es is None else w_es) + m * k * es + m * n * out_es   # noqa: E731
    mode = r._mode(b)
    plans = r._tiled_plan(r.bucket_for(b)) if any(mode.values()) else {}
    tq, gq = plans.get("qkv", (0, 1))
    to, go = plans.get("o", (0, 1))
    tp_one = r.tp_size == 1
    td, gd = plans.get("down", (0, 1))
    T, P, lt = r.tiled, ops.EPI_PARTIAL, r._linear_tiled
    name = {"qkv": "qkv", "o": "o", "gu": "gate_up", "down": "down", "lm_head": "lm_head"}
    out["impl"] = ("per projection, the faster of the hand-written weight-streaming GEMM on fragment-tiled weights "
                   "(linear_tiled_kernel, geometry tiles_per_wg/k_groups) and hipBLASLt through torch.mm: "
                   + ", ".join(f"{name[k]} " + (f"tiled {plans[k][0]}/{plans[k][1]}" if mode[k] else "hipBLASLt")
                               for k in r.KINDS)
                   + ("; SwiGLU in the gate_up epilogue" if mode["gu"] and plans["gu"][1] == 1 else ""))
    if getattr(r, "tune_us", None):
        out["tuned_us_tiled_vs_library"] = {name[k]: r.tune_us[(r.bucket_for(b), k)] for k in r.KINDS
                                            if (r.bucket_for(b), k) in r.tune_us}
    tg, gg = plans.get("gu", (0, 1))
    if mode["gu"] and gg > 1:      # K split over workgroups: fp32 slabs, the SwiGLU gate from the slabs
        out["gemm_gate_up"] = timed(lambda i: lt(x, T[i]["gu"], 2 * r.F, partial=r.part_gu, tiles_per_wg=tg, k_groups=gg,
                                                 epilogue=P), gemm(b, 2 * r.F, r.H, 4 * gg))
        out["silu_and_mul_slabs"] = timed(lambda i: ops.silu_and_mul_slabs(r.part_gu, gg, act),
                                          b * r.F * (8 * gg + es))
    elif mode["gu"]:
        out["gemm_gate_up"] = timed(lambda i: lt(x, T[i]["gu"], 2 * r.F, out=act, tiles_per_wg=tg,
                                                 epilogue=ops.EPI_SILU_MUL),
                                    2 * r.F * r.H * wes + b * r.H * es + b * r.F * es)
    else:
        out["gemm_gate_up"] = timed(lambda i: torch.mm(x, r._rm(i, "gu").t(), out=gu), gemm(b, 2 * r.F, r.H, w_es=es))
        out["silu_and_mul"] = timed(lambda i: ops.silu_and_mul(gu, out=act), 3 * b * r.F * es)
    # o / down leave fp32 slabs for the add + norm kernel on one GPU; a projection whose output is
    # all-reduced (o under tensor parallelism, down when the MLP is sharded) leaves slabs only when K
    # is split over workgroups (mi355_sum_slabs then forms the 16-bit tensor), else stores 16 bits
    slabs_o = mode["o"] and (tp_one or go > 1)
    slabs_dn = mode["down"] and (not r.mlp_sharded or gd > 1)
    if slabs_dn:
        out["gemm_down"] = timed(lambda i: lt(act, T[i]["down"], r.H, partial=r.part_dn, tiles_per_wg=td,
                                              k_groups=gd, epilogue=P), gemm(b, r.H, r.F, 4 * gd))
        if r.mlp_sharded:
            out["sum_slabs_down"] = timed(lambda i: ops.sum_slabs(r.part_dn, gd, dn), b * r.H * (4 * gd + es))
    elif mode["down"]:
        out["gemm_down"] = timed(lambda i: lt(act, T[i]["down"], r.H, out=dn, tiles_per_wg=td), gemm(b, r.H, r.F))
    else:
        out["gemm_down"] = timed(lambda i: torch.mm(act, r._rm(i, "down").t(), out=dn), gemm(b, r.H, r.F, w_es=es))
    if mode["qkv"]:
        out["gemm_qkv"] = timed(lambda i: lt(x, T[i]["qkv"], r.qw, partial=r.part_qkv, tiles_per_wg=tq,
                                             k_groups=gq, epilogue=P), gemm(b, r.qw, r.H, 4 * gq))
    else:
        out["gemm_qkv"] = timed(lambda i: torch.mm(x, r._rm(i, "qkv").t(), out=qkv), gemm(b, r.qw, r.H, w_es=es))
    if slabs_o:
        out["gemm_o"] = timed(lambda i: lt(attn, T[i]["o"], r.H, partial=r.part_o, tiles_per_wg=to,
                                           k_groups=go, epilogue=P), gemm(b, r.H, r.nh * r.d, 4 * go))
        if not tp_one:
            out["sum_slabs_o"] = timed(lambda i: ops.sum_slabs(r.part_o, go, o), b * r.H * (4 * go + es))
    elif mode["o"]:
        out["gemm_o"] = timed(lambda i: lt(attn, T[i]["o"], r.H, out=o, tiles_per_wg=to), gemm(b, r.H, r.nh * r.d))
    else:
        out["gemm_o"] = timed(lambda i: torch.mm(attn, r._rm(i, "o").t(), out=o), gemm(b, r.H, r.nh * r.d, w_es=es))
    lm = r.w.lm_head
    out["gemm_lm_head"] = timed(lambda i: r.lm_head_logits(x, r.logits[:b]), gemm(b, r.Vl, r.H, w_es=es))

    # attention over the live cache, at the context lengths of the last timed step
    splits = r._splits_for(b)
    ctx_tokens = int(r.ctx[:b].sum().item())              # the device buffers still hold the last step's inputs
    kv_bytes = ctx_tokens * 2 * r.nkv * r.d * es + 2 * b * r.nh * r.d * es
    rope_bytes = b * ((r.nh + r.nkv) * r.d * 2 + r.nkv * r.d + 2 * r.nkv * r.d) * es
    if slabs_dn and not r.mlp_sharded:
        out["fused_add_rms_norm"] = timed(
            lambda i: ops.fused_add_rms_norm_partial(r.part_dn, gd, resid, W[i].post_norm, r.eps, x),
            b * r.H * (4 * gd + 3 * es))
    else:
        out["fused_add_rms_norm"] = timed(lambda i: ops.fused_add_rms_norm(dn, resid, W[i].post_norm, r.eps, out=x),
                                          5 * b * r.H * es)
    if mode["qkv"]:
        out["paged_attn_decode_fused"] = timed(
            lambda i: ops.paged_attn_decode_fused(None, r.cos_sin, r.slots[:b], r.k_cache[i], r.v_cache[i],
                                                  r.block_tables[:b], r.ctx[:b], r.nh, r.nkv, r.d, r.scale,
                                                  num_splits=splits, workspace=r.attn_ws, out=attn, partial=r.part_qkv,
                                                  num_slabs=gq, bias=W[i].qkv_b),
            kv_bytes + b * r.qw * 4 * gq + b * 2 * r.nkv * r.d * es)
    else:
        out["paged_attn_decode_fused"] = timed(
            lambda i: ops.paged_attn_decode_fused(qkv, r.cos_sin, r.slots[:b], r.k_cache[i], r.v_cache[i],
                                                  r.block_tables[:b], r.ctx[:b], r.nh, r.nkv, r.d, r.scale,
                                                  num_splits=splits, workspace=r.attn_ws, out=attn),
            kv_bytes + rope_bytes)
    out["paged_attn_decode_fused"]["splits"] = splits
    if b <= r.keys.numel():      # the many-workgroup form the decode step runs
        out["argmax"] = timed(lambda i: ops.argmax_chunked(r.logits[:b], r.keys, r.next_ids[:b], r.ids[:b]), b * r.Vl * es)
    else:
        out["argmax"] = timed(lambda i: ops.argmax(r.logits[:b], out=r.next_ids[:b]), b * r.Vl * es)
    return out


def cpu_baseline_twin(cfg, batch, ctx_len, sample_layers=4, steps=3):
    """Whole decode steps built from the CPU twin of the operator ABI (libmi355_decode_cpu.so: the same
    entry points as the GPU library with host pointers, OpenMP) timed on this host: `batch`
    sequences at context `ctx_len` through ALL the model's layers plus the lm_head and the greedy
    pick -- measured, not scaled.  The layers cycle through `sample_layers` distinct real-shape
    weight sets (0.44 GB each for Llama-3-8B: far beyond the host caches, so every layer streams
    its weights from DRAM as 32 distinct ones would, without 16 GB of host memory to fill); every
    layer has its own KV pages.  Synthetic bf16 weights (finite, small, random signs: the
    arithmetic does not depend on the values)."""
    import ctypes

    import numpy as np
    lib = ctypes.CDLL(os.path.join(ROOT, "vllm-backend-public_amd", "mi355_decode", "libmi355_decode_cpu.so"))
    P, I32, I64, F = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    sig = {"mi355_cpu_rms_norm": [P, P, P, I64, I32, I64, I64, F, I32],
           "mi355_cpu_fused_add_rms_norm": [P, P, P, P, I64, I32, I64, I64, I64, F, I32],
           "mi355_cpu_rope_kv_write": [P, I64, P, P, I32, P, P, P, I32, I64, I32, I32, I32, I32],
           "mi355_cpu_paged_attn_decode": [P, P, I64, P, P, I32, P, I32, P, I32, I32, I32, I32, F, I32],
           "mi355_cpu_silu_and_mul": [P, P, I64, I64, I32, I32],
           "mi355_cpu_argmax": [P, P, I64, I32, I32, I32],
           "mi355_cpu_linear": [P, I64, P, I64, P, P, I32, I32, I32, I32]}
    for name, args in sig.items():
        getattr(lib, name).restype, getattr(lib, name).argtypes = I32, args
    lib.mi355_cpu_last_error.restype = ctypes.c_char_p
    H, Fi, V, d = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.head_dim
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    qw = (nh + 2 * nkv) * d
    rng = np.random.Generator(np.random.PCG64(0))

    def buf(*shape):                       # 16-byte aligned uint16 words
        n = int(np.prod(shape))
        raw = np.empty(n * 2 + 16, np.uint8)
        off = (-raw.ctypes.data) % 16
        return raw[off:off + n * 2].view(np.uint16).reshape(shape)

    def mat(*shape):                       # bf16 words of magnitude 2^-7 .. 2^-6, random sign and mantissa
        a = buf(*shape)
        a[...] = (rng.integers(0, 1 << 16, size=shape, dtype=np.uint16) & np.uint16(0x807F)) | np.uint16(0x3C00)
        return a

    def ones(n):
        a = buf(n)
        a[...] = 0x3F80
        return a

    layers = [dict(qkv=mat(qw, H), o=mat(H, nh * d), gu=mat(2 * Fi, H), down=mat(H, Fi), n1=ones(H), n2=ones(H))
              for _ in range(sample_layers)]
    embed_rows, lm_head, final_norm = mat(batch, H), mat(V, H), ones(H)
    max_pos = ctx_len + steps + 8
    blocks_per_seq = (max_pos + 15) // 16
    nblk = batch * blocks_per_seq
    L = cfg.num_hidden_layers
    kc = [mat(nblk, nkv, 16, d) for _ in range(L)]
    vc = [mat(nblk, nkv, d, 16) for _ in range(L)]
    cos_sin = mat(max_pos, d)
    bt = np.arange(nblk, dtype=np.int32).reshape(batch, blocks_per_seq)
    resid, x, qkv, attn, o, gu, act, dn = (buf(batch, w) for w in (H, H, qw, nh * d, H, 2 * Fi, Fi, H))
    logits, ids = buf(batch, V), np.zeros(batch, np.int64)
    ptr = lambda a: a.ctypes.data   # noqa: E731

    def ok(rc):
        if rc != 0:
            raise RuntimeError(lib.mi355_cpu_last_error().decode())

    def step(k, with_head):
        pos = np.full(batch, ctx_len - 1 + k, dtype=np.int64)
        slots = bt[np.arange(batch), pos // 16].astype(np.int64) * 16 + pos % 16
        ctx = (pos + 1).astype(np.int32)
        t0 = time.perf_counter()
        resid[...] = embed_rows
        for i in range(L):
            lw = layers[i % sample_layers]
            if i == 0:
                ok(lib.mi355_cpu_rms_norm(ptr(x), ptr(resid), ptr(lw["n1"]), batch, H, H, H, cfg.rms_norm_eps, 0))
            else:
                ok(lib.mi355_cpu_fused_add_rms_norm(ptr(x), ptr(dn), ptr(resid), ptr(lw["n1"]), batch, H, H, H, H,
                                                    cfg.rms_norm_eps, 0))
            ok(lib.mi355_cpu_linear(ptr(qkv), qw, ptr(x), H, ptr(lw["qkv"]), None, batch, qw, H, 0))
            ok(lib.mi355_cpu_rope_kv_write(ptr(qkv), qw, ptr(pos), ptr(cos_sin), max_pos, ptr(slots), ptr(kc[i]), ptr(vc[i]),
                                           nblk, batch, nh, nkv, d, 0))
            ok(lib.mi355_cpu_paged_attn_decode(ptr(attn), ptr(qkv), qw, ptr(kc[i]), ptr(vc[i]), nblk, ptr(bt), blocks_per_seq,
                                               ptr(ctx), batch, nh, nkv, d, d ** -0.5, 0))
            ok(lib.mi355_cpu_linear(ptr(o), H, ptr(attn), nh * d, ptr(lw["o"]), None, batch, H, nh * d, 0))
            ok(lib.mi355_cpu_fused_add_rms_norm(ptr(x), ptr(o), ptr(resid), ptr(lw["n2"]), batch, H, H, H, H, cfg.rms_norm_eps, 0))
            ok(lib.mi355_cpu_linear(ptr(gu), 2 * Fi, ptr(x), H, ptr(lw["gu"]), None, batch, 2 * Fi, H, 0))
            ok(lib.mi355_cpu_silu_and_mul(ptr(act), ptr(gu), 2 * Fi, batch, Fi, 0))
            ok(lib.mi355_cpu_linear(ptr(dn), H, ptr(act), Fi, ptr(lw["down"]), None, batch, H, Fi, 0))
        t_layers = time.perf_counter() - t0
        if with_head:
            ok(lib.mi355_cpu_fused_add_rms_norm(ptr(x), ptr(dn), ptr(resid), ptr(final_norm), batch, H, H, H, H,
                                                cfg.rms_norm_eps, 0))
            ok(lib.mi355_cpu_linear(ptr(logits), V, ptr(x), H, ptr(lm_head), None, batch, V, H, 0))
            ok(lib.mi355_cpu_argmax(ptr(ids), ptr(logits), V, batch, V, 0))
        return t_layers, time.perf_counter() - t0

    step(0, True)                                              # warm the threads and the page cache
    runs = [step(1 + k, True) for k in range(steps)]
    step_s = min(r[1] for r in runs)
    t_layers = min(r[0] for r in runs)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    return {"value": round(batch / step_s, 2), "unit": "tok/s", "cores": threads, "kind": "port",
            "sample": (f"CPU twin of the operator ABI (libmi355_decode_cpu.so, C++/OpenMP), {steps} whole decode steps (best "
                       f"taken) of batch {batch} at context {ctx_len}: all {L} layers (cycling through {sample_layers} distinct "
                       f"real-shape weight sets, each far beyond the host caches) + lm_head + argmax, measured: layers "
                       f"{t_layers * 1e3:.0f} ms + head {(step_s - t_layers) * 1e3:.0f} ms = {step_s:.2f} s per step"),
            "stock_vllm_cpu": "unavailable offline (vllm is not installed; nothing can be fetched)"}


def cpu_baseline(cfg, batch, ctx_len, sample_layers=2, steps=2):
    """The numpy oracle (oracle/model_ref.py) timed on this host: one decode step of
    `batch` sequences at context `ctx_len` through `sample_layers` real-shape layers
    plus the lm_head, scaled to the model's layer count."""
    import numpy as np
    from oracle import model_ref as M
    from oracle.ops_ref import round_to
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    H, F, V, d = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.head_dim
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    rng = np.random.Generator(np.random.PCG64(0))

    def mat(*shape):
        return round_to(rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02), "bfloat16")

    w = {"model.embed_tokens.weight": mat(V, H), "lm_head.weight": mat(V, H), "model.norm.weight": np.ones(H, np.float32)}
    for i in range(sample_layers):
        p = f"model.layers.{i}."
        w[p + "self_attn.q_proj.weight"], w[p + "self_attn.k_proj.weight"] = mat(nh * d, H), mat(nkv * d, H)
        w[p + "self_attn.v_proj.weight"], w[p + "self_attn.o_proj.weight"] = mat(nkv * d, H), mat(H, nh * d)
        w[p + "mlp.gate_proj.weight"], w[p + "mlp.up_proj.weight"], w[p + "mlp.down_proj.weight"] = mat(F, H), mat(F, H), mat(H, F)
        w[p + "input_layernorm.weight"] = w[p + "post_attention_layernorm.weight"] = np.ones(H, np.float32)
    rc = M.RefConfig(vocab_size=V, hidden_size=H, intermediate_size=F, num_hidden_layers=sample_layers,
                     num_attention_heads=nh, num_key_value_heads=nkv, head_dim=d, rope_theta=cfg.rope_theta,
                     rms_norm_eps=cfg.rms_norm_eps, max_position_embeddings=ctx_len + 2 * steps + 8)
    blocks_per_seq = (ctx_len + 2 * steps + 8 + 15) // 16
    model = M.RefModel(rc, w, "bfloat16", num_blocks=batch * blocks_per_seq, layers=sample_layers)
    for kc, vc in zip(model.k_cache, model.v_cache):
        kc[:] = mat(*kc.shape)
        vc[:] = mat(*vc.shape)
    bt = np.arange(batch * blocks_per_seq, dtype=np.int32).reshape(batch, blocks_per_seq)
    ids = rng.integers(0, V, batch)

    def one_step(k, layers_only):
        pos = np.full(batch, ctx_len - 1 + k, dtype=np.int64)
        slots = bt[np.arange(batch), pos // 16].astype(np.int64) * 16 + pos % 16
        t0 = time.perf_counter()
        if layers_only:
            saved = model.w["lm_head.weight"]
            model.w["lm_head.weight"] = saved[:8]             # negligible lm_head: isolates the layer cost
            model.forward(ids, pos, slots, bt, (pos + 1).astype(np.int32))
            model.w["lm_head.weight"] = saved
        else:
            model.forward(ids, pos, slots, bt, (pos + 1).astype(np.int32))
        return time.perf_counter() - t0

    one_step(0, False)                                        # warm the BLAS threads
    t_full = min(one_step(1 + k, False) for k in range(steps))
    t_layers = min(one_step(1 + steps + k, True) for k in range(steps))
    per_layer = t_layers / sample_layers
    t_head = max(t_full - t_layers, 0.0)
    step_s = per_layer * cfg.num_hidden_layers + t_head
    return {"value": round(batch / step_s, 2), "unit": "tok/s", "cores": threads, "kind": "port",
            "sample": (f"numpy oracle, {steps} decode steps of batch {batch} at context {ctx_len} through "
                       f"{sample_layers} of {cfg.num_hidden_layers} real-shape layers + lm_head; per-layer time "
                       f"{per_layer * 1e3:.0f} ms x {cfg.num_hidden_layers} + lm_head {t_head * 1e3:.0f} ms = "
                       f"{step_s:.2f} s per step"),
            "stock_vllm_cpu": "unavailable offline (vllm is not installed; nothing can be fetched)"}


def served_path(engine, cfg, a, batch):
    """The same workload through the served path the harness drives: the engine behind the HTTP
    server of this process (engine thread + uvicorn, as `api_server` runs them), `batch` concurrent
    streaming POST /v1/completions token-id requests from a separate client process
    (mi355_decode/bench_client.py), 1 warm-up run + `--served-runs` measured runs, medians; the
    client then posts the numbers as an EvaluationPayload to a local stand-in of benchmark-results
    that applies the accept rules of its request model (benchmark-results/models.py:6-19)."""
    import http.server
    import socket
    import subprocess
    import threading

    import uvicorn
    from mi355_decode.entrypoints.openai.api_server import build_app
    from mi355_decode.serving import EngineThread
    from mi355_decode.tokenizer import load_tokenizer

    def free_port():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            return s.getsockname()[1]

    stored = []

    class Results(http.server.BaseHTTPRequestHandler):
        def do_POST(self):
            body = json.loads(self.rfile.read(int(self.headers.get("Content-Length", "0"))) or b"{}")
            need = ("benchmark_name", "data", "timestamp", "model_id", "tokenizer_id", "source")
            ok = self.path in ("/raw_input", "/standardized_output") and all(
                k in body and body[k] is not None for k in need) and all(
                isinstance(body[k], str) for k in need if k != "data")
            if ok:
                stored.append((self.path, f"{body['timestamp']}-{body['benchmark_name']}-{body.get('run_id')}"))
            self.send_response(200 if ok else 422)
            self.send_header("Content-Type", "application/json")
            self.end_headers()
            self.wfile.write(b'{"status": "success"}' if ok else b'{"detail": "invalid payload"}')

        def log_message(self, *args):
            pass

    rport, sport = free_port(), free_port()
    results = http.server.ThreadingHTTPServer(("127.0.0.1", rport), Results)
    threading.Thread(target=results.serve_forever, daemon=True).start()

    name = "bench-model"
    sys.setswitchinterval(float(os.environ.get("MI355_GIL_SWITCH_S", "0.0005")))   # as api_server.main does
    thread = EngineThread(lambda: engine)
    app = build_app(thread, name, load_tokenizer(a.model, cfg.vocab_size), cfg.vocab_size, a.prompt_len + a.gen_len)
    thread.start()
    server = uvicorn.Server(uvicorn.Config(app, host="127.0.0.1", port=sport, log_level="warning", access_log=False))
    st = threading.Thread(target=server.run, daemon=True)
    st.start()
    try:
        t0 = time.time()
        while not server.started:
            if time.time() - t0 > 30:
                raise RuntimeError("server did not start")
            time.sleep(0.02)
        env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "vllm-backend-public_amd") + os.pathsep
                   + os.environ.get("PYTHONPATH", ""))
        cmd = [sys.executable, "-m", "mi355_decode.bench_client", "--url", f"http://127.0.0.1:{sport}", "--model", name,
               "--batch", str(batch), "--prompt-len", str(a.prompt_len), "--gen-len", str(a.gen_len),
               "--vocab", str(cfg.vocab_size), "--runs", str(a.served_runs), "--warmup", "1",
               "--results-url", f"http://127.0.0.1:{rport}"]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            raise RuntimeError(f"bench client failed: {p.stderr[-500:]}")
        out = json.loads(p.stdout.strip().splitlines()[-1])
    finally:
        server.should_exit = True
        st.join(timeout=10)
        thread.shutdown()
        thread.thread.join(timeout=10)
        results.shutdown()
    out["results_stored"] = [f"{path} pk={pk}" for path, pk in stored]
    full = [b for b in thread.bursts if b[0] == batch][-a.served_runs:]
    if full:
        import statistics
        out["ttft_breakdown_ms_p50"] = {
            "requests_arrive_over": round(statistics.median(b[1] for b in full), 2),
            "engine_waits_for_stragglers": round(statistics.median(b[2] for b in full), 2),
            "longest_pauses_between_arrivals": [[round(g, 2), round(g2, 2)] for n_, g, g2 in thread.burst_gaps if n_ == batch],
            "note": "server side, per burst admitted in one piece: first to last request handed to the engine (HTTP "
                    "parsing on one thread), then the admission window; the rest of TTFT is the prefill step and the "
                    "first SSE event's way back"}
        out["bursts_admitted_whole"] = f"{len(full)} of {a.served_runs}"
        # the whole way of a burst, client and server stamps on one clock (both read the host's monotonic clock):
        # medians over the runs whose burst was admitted in one piece, ms
        stamps = out.pop("stamps", None) or []
        legs = []
        for st in stamps:
            b = next((b for b in full if b[6] is not None and st["submit_first"] <= b[3] <= st["first_token_last"]), None)
            if b is None:
                continue
            legs.append({"client_submits_over": (st["submit_last"] - st["submit_first"]) * 1e3,
                         "first_submit_to_first_arrival": (b[3] - st["submit_first"]) * 1e3,
                         "arrivals_over": (b[4] - b[3]) * 1e3, "admission_window": (b[5] - b[4]) * 1e3,
                         "admitted_to_step_start": (b[7] - b[5]) * 1e3, "prefill_step": (b[6] - b[7]) * 1e3,
                         "first_tokens_exist_to_first_client_event": (st["first_token_first"] - b[6]) * 1e3,
                         "to_median_client_event": (st["first_token_p50"] - b[6]) * 1e3,
                         "to_last_client_event": (st["first_token_last"] - b[6]) * 1e3})
        if legs:
            out["ttft_timeline_ms_p50"] = {k: round(statistics.median(l[k] for l in legs), 2) for k in legs[0]}
    out.pop("stamps", None)
    out["path"] = ("client process -> HTTP/SSE (uvicorn) -> engine thread -> HIP-graph decode steps; tokenizer: "
                   "byte-wise stand-in (no tokenizer.json for synthetic weights), one SSE event per token")
    return out


def pmc_traffic(tiled):
    """(HBM bytes per launch of the gate/up GEMM, source) from the committed rocprofv3
    PMC summary of this workload, or (None, None)."""
    import re
    name = "r03_pmc_traffic.json" if tiled else "r01_b_pmc_traffic.json"
    # the gate/up GEMM is the instantiation with the SwiGLU epilogue (5th template argument == 1)
    pat = re.compile(r"linear_tiled_kernel<\w+, \d+, \d+, \d+, 1[,>]" if tiled else r"gemm_gate_up")
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            for key, rec in json.load(f)["kernels"].items():
                if pat.match(key):
                    return rec["hbm_bytes_per_launch"], f"profiles/{name}"
    except Exception:
        pass
    return None, None


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher above us: become the parent of a.gpus rank processes (nothing GPU-related has
        # been imported yet) and relay rank 0's line
        rc, out = launch_ranks(a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        # stdout carries the result line only: what a collective library printed there (gloo's
        # connection notes, RCCL's debug output) goes to stderr
        for line in out.splitlines():
            print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
        raise SystemExit(rc)
    import torch
    import torch.distributed as dist
    from mi355_decode.config import EngineArgs, ModelConfig
    from mi355_decode.llm import build_engine
    from mi355_decode.sampling_params import SamplingParams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} under a launcher that started {world} rank(s): the two must agree")
    tp = a.gpus
    batch = a.batch_per_gpu * a.gpus
    P, G = a.prompt_len, a.gen_len

    if a.oneshot_allreduce:
        os.environ["MI355_ONESHOT_ALLREDUCE"] = "1"
    cfg = ModelConfig.from_pretrained(a.model)
    args = EngineArgs(model=a.model, dtype=a.dtype, max_num_seqs=batch, max_model_len=P + G,
                      max_num_batched_tokens=max(8192, 1), tensor_parallel_size=tp, load_format="dummy",
                      gpu_memory_utilization=0.9, enforce_eager=a.eager, num_layers_override=a.layers, seed=0,
                      quantization=a.quantization)
    engine = build_engine(args, model_config=cfg, decode_splits=a.decode_splits)
    params = SamplingParams(max_tokens=G, temperature=0.0, ignore_eos=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    round_no = [0]

    def start_batch():
        """Admit and prefill a fresh batch; returns the prefill wall time (s)."""
        prompts = make_prompts(cfg.vocab_size, batch, P, 1234 + round_no[0])
        round_no[0] += 1
        barrier()
        t0 = time.perf_counter()
        for i, p in enumerate(prompts):
            engine.add_request(f"b{round_no[0]}-{i}", p, params)
        while any(not s.output_ids for s in list(engine.sched.waiting) + engine.sched.running):
            engine.step()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def drain():
        for rid in list(engine.seqs):
            engine.abort_request(rid)

    # ---- warm-up: prefill shapes, graph capture, W decode steps
    start_batch()
    for _ in range(max(a.warmup, 1)):
        engine.step()
    drain()

    # ---- TTFT: wall time from submitting the batch to its first tokens (one prefill pass)
    ttfts = []
    for _ in range(3):
        ttfts.append(start_batch())
        drain()
    ttft_ms = sorted(ttfts)[len(ttfts) // 2] * 1e3

    # ---- timed decode: a region is exactly K steps between two barriers (pieces of at most G-1
    # steps per batch, fresh batches prefilled outside the clock); `segments` such regions are
    # timed back to back and the median one is reported, so that a K of 20 steps (0.07 s) is not
    # at the mercy of the clocks still ramping up after the warm-up
    def timed_region(steps):
        left, total_t, ctx_sum = steps, 0.0, 0
        while left > 0:
            have = min((s.params.max_tokens - len(s.output_ids) for s in engine.sched.running), default=0)
            if have < 1 or not engine.sched.running:
                drain()
                start_batch()
                have = G - 1
            seg = min(left, have)
            done_before = len(engine.sched.running[0].output_ids)
            barrier()
            t0 = time.perf_counter()
            ahead = engine.run_ahead
            for k in range(seg):
                if k == seg - 1:
                    # the engine launches step n+1 before it reads step n's tokens: not at the last
                    # step of the region, or the barrier below would wait for a step nobody counted
                    engine.run_ahead = False
                engine.step()
            engine.run_ahead = ahead
            barrier()
            # RCCL reduces device tensors; any other backend (a rehearsal over gloo) gets a host tensor
            on_dev = world > 1 and dist.get_backend() == "nccl"
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=engine.device if on_dev else "cpu")
            if world > 1:
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            total_t += float(dt.item())
            ctx_sum += sum(P + done_before + k for k in range(seg))
            left -= seg
        return total_t, ctx_sum

    regions = [timed_region(a.steps) for _ in range(max(a.segments, 1))]
    drain()
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    total_t, ctx_sum = regions[order[len(order) // 2]]
    mean_ctx = ctx_sum / a.steps
    ms_per_step = total_t / a.steps * 1e3
    tok_s = batch * a.steps / total_t

    lwe = 1 if a.quantization == "fp8" else None
    mlp_tp = tp if engine.runner.mlp_sharded else 1
    vocab_tp = tp if engine.runner.vocab_sharded else 1
    step_bytes, w_bytes = model_bytes(cfg, tp, batch, mean_ctx, layer_w_elem=lwe, mlp_tp=mlp_tp, vocab_tp=vocab_tp)
    if a.layers:
        import copy
        c2 = copy.copy(cfg)
        c2.num_hidden_layers = a.layers
        step_bytes, w_bytes = model_bytes(c2, tp, batch, mean_ctx, layer_w_elem=lwe, mlp_tp=mlp_tp, vocab_tp=vocab_tp)
    step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9

    result = {
        "metric": "decode_tok_s", "value": round(tok_s, 1), "unit": "tok/s", "n_gpus": a.gpus, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": ("bf16" if a.dtype == "bfloat16" else a.dtype) + ("+fp8w" if lwe else ""),
        "data": "synthetic",
        "config": {"workload": f"{a.model} decode, batch {batch} ({a.batch_per_gpu}/GPU), {P}-in/{G}-out, greedy, "
                               f"synthetic weights and token-id prompts; " + baseline_config_label(a, lwe),
                   "global_batch": batch, "prompt_len": P, "gen_len": G,
                   "parallelism": (f"tp{tp} (attention heads and MLP sharded, two all-reduces per layer; lm_head "
                                   + ("sharded over the vocabulary, greedy keys max-reduced)" if vocab_tp > 1 else "replicated)")
                                   if mlp_tp > 1 else
                                   f"tp{tp} (attention heads sharded, MLP+lm_head replicated)") if tp > 1 else "tp1",
                   "mean_context": round(mean_ctx, 1), "hip_graphs": bool(engine.runner.use_graphs),
                   "weight_copies": ("tiled only (MI355_SINGLE_WEIGHT_COPY=1)"
                                     if getattr(engine.runner, "single_weight_copy", False) else "row-major + tiled"),
                   "gpu_memory_allocated_gb": round(torch.cuda.memory_allocated() / 1e9, 2)},
        "segments": {"count": len(regions), "reported": "median", "steps_each": a.steps,
                     "ms_per_step_each": [round(t / a.steps * 1e3, 4) for t, _ in regions]},
        "ttft_ms_p50": round(ttft_ms, 2),
        "step_roofline": {"bound": "hbm", "achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(step_gbs / HBM_PEAK_GBS, 4), "bytes_per_step": int(step_bytes),
                          "weight_bytes": int(w_bytes), "roofline_tok_s": round(batch / (step_bytes / (HBM_PEAK_GBS * 1e9)), 1),
                          # the chip's measured streaming-read rate (a read-only kernel over a buffer far beyond the
                          # caches reaches ~6.6 TB/s of the 8 TB/s data-sheet figure): what a perfect step could reach
                          "read_floor_gbs": HBM_READ_FLOOR_GBS, "frac_of_read_floor": round(step_gbs / HBM_READ_FLOOR_GBS, 4)},
    }
    if a.layers:
        result["config"]["INVALID_reduced_layers"] = a.layers
    # prefill of the batch (TTFT): dense GEMM + attention FLOPs per GPU against the bf16 MFMA peak
    Lc = a.layers or cfg.num_hidden_layers
    lin = Lc * ((cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim * cfg.hidden_size // tp
                + cfg.num_attention_heads * cfg.head_dim * cfg.hidden_size // tp
                + 3 * cfg.intermediate_size * cfg.hidden_size // mlp_tp)
    pf_flops = 2.0 * lin * batch * P + 2.0 * (cfg.vocab_size // vocab_tp) * cfg.hidden_size * batch \
        + 4.0 * batch * (P * (P + 1) / 2) * (cfg.num_attention_heads // tp) * cfg.head_dim * Lc
    result["prefill_roofline"] = {"bound": "mfma", "achieved": round(pf_flops / (ttft_ms * 1e-3) / 1e12, 1),
                                  "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(pf_flops / (ttft_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                  "flops": int(pf_flops), "tokens": batch * P,
                                  "note": "wall time of the batch's prefill step (host scheduling included); "
                                          "linear layers on hipBLASLt, attention on the hand-written kernel"}
    r_ = engine.runner
    result["degraded"] = {"hip_graphs": bool(r_.use_graphs), "graph_capture_failures": r_.capture_failures,
                          "gemm_tune_failures": r_.tune_failures,
                          "allreduce": ("none (1 GPU)" if tp == 1 else
                                        "one-shot kernel over IPC-mapped buffers (self-test passed on every rank); "
                                        + ("the decode step's all-reduce + add + norm in two hops (rows dealt to the ranks); "
                                           if r_.oneshot is not None and r_.oneshot.two_shot(batch, cfg.hidden_size) else "")
                                        + "larger messages on " + dist.get_backend() if r_.oneshot is not None else
                                        ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend() + " (rehearsal)")
                                        + (" -- the one-shot kernel's set-up or self-test failed, fell back"
                                           if r_.oneshot_fallbacks else "")),
                          "oneshot_allreduce_fallbacks": r_.oneshot_fallbacks,
                          "gemm_form": {k: ("tiled" if v else "library") for k, v in r_._mode(batch).items()},
                          # (tiles_per_wg code, K groups) of the tiled forms: + 1000 / + 2000 = row blocks of 64 / 32
                          "gemm_plan": {k: list(v) for k, v in r_._tiled_plan(r_.bucket_for(batch)).items()}}

    # what the job really ran on: ranks the process group saw, its backend, every rank's device
    mine = {"rank": rank, "device": torch.cuda.get_device_name(engine.device), "index": engine.device.index,
            "pid": os.getpid()}
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        result["ranks_seen"] = dist.get_world_size()
        result["dist_backend"] = dist.get_backend()
    else:
        seen = [mine]
        result["ranks_seen"] = 1
        result["dist_backend"] = None
    result["rank_devices"] = seen
    result["launched_by"] = "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else (
        "bench.py itself" if world > 1 else "single process")

    kern = None
    if rank == 0 and not a.no_kernel_timing:
        try:
            kern = time_kernels(engine, batch, int(mean_ctx))
        except Exception as exc:                          # never lose the step measurement to the per-kernel table
            result["kernels"] = {"error": repr(exc)}
    if kern is not None:
        result["kernels"] = kern
        dom = kern["gemm_gate_up"]
        tiled = engine.runner._mode(batch)["gu"]
        # the committed PMC passes were of the default 1-GPU workload: any other one reports no traffic figure
        measured = (tp == 1 and a.model == "llama-3-8b" and batch == 32 and a.dtype == "bfloat16" and
                    not a.quantization and not a.layers)
        traffic, src = pmc_traffic(tiled) if measured else (None, None)
        result["roofline"] = {"bound": "hbm",
                              "kernel": ("gate/up projection GEMM%s (M=%d, N=%d, K=%d): %s" % (
                                  " + SwiGLU epilogue" if tiled else "", batch, 2 * engine.runner.F,
                                  cfg.hidden_size, "linear_tiled_kernel" if tiled else "hipBLASLt")),
                              "achieved": dom["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
                              "traffic": traffic,
                              "traffic_source": (src + " (rocprofv3 PMC passes of this workload; not re-measured in "
                                                 "this run)") if src else "none (no PMC passes of this workload)",
                              "bytes_per_launch": dom["bytes"], "us_per_launch": dom["us"]}
    if world > 1:
        dist.barrier()
    if rank == 0 and a.gpus == 1 and not a.no_served:
        try:
            result["served"] = served_path(engine, cfg, a, batch)
        except Exception as exc:                          # never lose the engine-loop measurement
            result["served"] = {"error": repr(exc)}
    if rank == 0:
        if not a.no_cpu_baseline and a.gpus == 1:
            try:
                result["cpu_baseline"] = cpu_baseline_twin(cfg, a.batch_per_gpu, int(mean_ctx))
            except Exception as exc:                      # the twin is missing or failed: the numpy oracle instead
                try:
                    result["cpu_baseline"] = cpu_baseline(cfg, a.batch_per_gpu, int(mean_ctx))
                    result["cpu_baseline"]["twin_error"] = repr(exc)
                except Exception as exc2:                 # never lose the GPU measurement
                    result["cpu_baseline"] = {"value": None, "error": repr(exc2)}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
