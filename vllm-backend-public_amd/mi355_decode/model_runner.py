"""One model step on one GPU: embed -> L x [norm, qkv, rope + kv-store + paged attention (one
kernel at decode), o-proj (+all-reduce under TP), norm, gate_up + SwiGLU, down (+all-reduce when
the MLP is sharded)] -> norm -> lm_head -> greedy token (a key max-reduce when the lm_head is
sharded over the vocabulary).

The hot operators go through the C ABI (mi355_decode.ops -> libmi355_decode.so).  Decode steps of
up to 128 sequences can run every linear layer through the weight-streaming GEMM on fragment-tiled
weight copies (K-split layers hand fp32 partial sums to the norm / attention kernels, the SwiGLU
gate is applied in the gate_up epilogue); which projections do, and in which geometry, is timed
against the library GEMM once per batch-size class (tune).  Larger batches and prefill use library
GEMMs on the row-major weights.  PyTorch only owns memory, streams and the process group.  The
decode step works on statically allocated buffers and is captured once per batch-size bucket into
a HIP graph (a decode step is ~230 short kernels; replay removes the per-launch host cost);
prefill runs eagerly.
"""
from __future__ import annotations

import math
import os

import numpy as np
from typing import Optional

import torch

from . import ops
from ._lib import BLOCK_SIZE, Mi355Error
from .config import ModelConfig
from .weights import ModelWeights


def build_cos_sin_table(cfg: ModelConfig, max_pos: int, dtype: torch.dtype, device) -> torch.Tensor:
    """[max_pos, head_dim] = cos | sin halves, computed in fp32 and rounded to the
    model dtype (the table a 16-bit model rotates with)."""
    d = cfg.head_dim
    if cfg.family == "opt":
        # learned positions: no rotation.  One row, cos = 1 | sin = 0 -- the kernels clamp every
        # position to the table's last row, and rotating by it returns the input bit for bit
        return torch.cat([torch.ones(1, d // 2), torch.zeros(1, d // 2)], dim=1).to(dtype).contiguous().to(device)
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    sc = cfg.rope_scaling
    if sc is not None:  # "llama3" frequency-dependent scaling
        factor, low, high = sc["factor"], sc["low_freq_factor"], sc["high_freq_factor"]
        old = sc["original_max_position_embeddings"]
        wavelen = 2 * math.pi / inv_freq
        scaled = torch.where(wavelen > old / low, inv_freq / factor, inv_freq)
        smooth = (old / wavelen - low) / (high - low)
        smoothed = (1 - smooth) * scaled / factor + smooth * scaled
        mid = (wavelen >= old / high) & (wavelen <= old / low)
        inv_freq = torch.where(mid, smoothed, scaled)
    freqs = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv_freq.float())
    return torch.cat([freqs.cos(), freqs.sin()], dim=1).to(dtype).contiguous().to(device)


def batch_buckets(max_num_seqs: int):
    """Padded batch sizes a decode graph is captured for."""
    sizes = [1, 2, 4] + list(range(8, 257, 8)) + list(range(288, 1025, 32))
    out = [s for s in sizes if s < max_num_seqs]
    out.append(max_num_seqs)
    return out


class ModelRunner:
    NUM_STAGES = 2
    MAX_LOGPROBS = 20

    def __init__(self, cfg: ModelConfig, weights: ModelWeights, dtype: torch.dtype, device,
                 num_blocks: int, max_num_seqs: int, max_model_len: int, tp_rank: int = 0,
                 tp_size: int = 1, tp_group=None, use_graphs: bool = True,
                 decode_splits: Optional[int] = None, tiled_linears: Optional[bool] = None,
                 weight_fp8: bool = False, single_weight_copy: bool = False):
        self.cfg, self.w, self.dtype, self.device = cfg, weights, dtype, torch.device(device)
        self.tp_rank, self.tp_size, self.tp_group = tp_rank, tp_size, tp_group
        self.nh = cfg.num_attention_heads // tp_size
        self.nkv = cfg.num_key_value_heads // tp_size
        self.d = cfg.head_dim
        self.H, self.V = cfg.hidden_size, cfg.vocab_size
        # the weights say whether this rank holds the whole MLP or its 1/tp_size slice of the
        # intermediate dimension (then down_proj's partial sums are all-reduced like o_proj's)
        self.F = int(weights.layers[0].down_w.shape[1]) if weights.layers else cfg.intermediate_size
        self.mlp_sharded = self.F != cfg.intermediate_size
        if self.mlp_sharded and self.F * tp_size != cfg.intermediate_size:
            raise ValueError(f"MLP slice of {self.F} columns does not match intermediate_size "
                             f"{cfg.intermediate_size} / tensor_parallel_size {tp_size}")
        # ... and whether it holds the whole lm_head or the rows of its slice of the vocabulary
        self.Vl = int(weights.lm_head.shape[0])
        self.vocab_sharded = self.Vl != self.V
        if self.vocab_sharded and self.Vl * tp_size != self.V:
            raise ValueError(f"lm_head slice of {self.Vl} rows does not match vocab_size {self.V} / "
                             f"tensor_parallel_size {tp_size}")
        self.vocab_offset = tp_rank * self.Vl if self.vocab_sharded else 0
        self.L = len(weights.layers)
        self.qw = (self.nh + 2 * self.nkv) * self.d
        self.scale = self.d ** -0.5
        self.eps = cfg.rms_norm_eps
        self.num_blocks = int(num_blocks)
        self.max_num_seqs = int(max_num_seqs)
        self.max_model_len = int(max_model_len)
        self.max_blocks = (self.max_model_len + BLOCK_SIZE - 1) // BLOCK_SIZE
        self.use_graphs = use_graphs
        # A decode step that cannot be captured (a collective the RCCL build refuses to capture) may
        # run as eager launches instead -- only under tensor parallelism, where that can happen;
        # on one GPU a failed capture or a failed timing run is an error and is raised.
        self.allow_eager_fallback = tp_size > 1
        self.capture_failures = 0                     # exported on /metrics and in the bench line
        self.tune_failures = 0
        self._device_collectives = None               # decided at the first collective (see _collective)
        self.oneshot = None                           # collectives.OneShotAllReduce, set by build_engine (opt-in)
        self.oneshot_calls = 0
        self.fused_allreduce_norms = 0                # ... of which all-reduce + residual add + norm in one launch
        self.oneshot_key_reductions = 0               # greedy picks of a vocabulary-sharded step reduced by the one-shot kernel
        self.oneshot_fallbacks = 0                    # set-ups that ended on the process group's all-reduce instead
        self._no_collectives = False                  # True while GEMM forms are timed (tune)
        self.decode_splits = decode_splits
        self.weight_fp8 = bool(weight_fp8)
        # keep only the fragment-tiled copy of every decode weight (see _init_tiled / _rm)
        self.single_weight_copy = bool(single_weight_copy)
        self._rm_scratch = {}

        dev = self.device
        self.cos_sin = build_cos_sin_table(cfg, self.max_model_len, dtype, dev)
        # paged KV cache, zero-filled so that every slot always holds finite values
        self.k_cache = [torch.zeros(self.num_blocks, self.nkv, BLOCK_SIZE, self.d, dtype=dtype, device=dev)
                        for _ in range(self.L)]
        self.v_cache = [torch.zeros(self.num_blocks, self.nkv, self.d, BLOCK_SIZE, dtype=dtype, device=dev)
                        for _ in range(self.L)]

        B = self.max_num_seqs
        i64 = dict(dtype=torch.int64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        f = dict(dtype=dtype, device=dev)
        # step inputs (static addresses: the decode graphs read them).  Packed so the
        # engine refreshes them with two host-to-device copies per step:
        #   in64 = ids | positions | slots            (3 * B int64)
        #   in32 = context_lens | block_tables        (B + B * max_blocks int32)
        # one allocation (int32 words): the 64-bit part first, so that a step's inputs are ONE copy
        self.inall = torch.zeros(6 * B + B + B * self.max_blocks, **i32)
        self.in64 = self.inall[:6 * B].view(torch.int64)
        self.in32 = self.inall[6 * B:]
        self.ids, self.positions, self.slots = self.in64[:B], self.in64[B:2 * B], self.in64[2 * B:]
        self.slots.fill_(-1)
        self.ctx = self.in32[:B]
        self.block_tables = self.in32[B:].view(B, self.max_blocks)
        # activations
        self.resid = torch.zeros(B, self.H, **f)
        self.x = torch.zeros(B, self.H, **f)
        self.qkv = torch.zeros(B, self.qw, **f)
        self.attn = torch.zeros(B, self.nh * self.d, **f)
        self.o = torch.zeros(B, self.H, **f)
        self.gu = torch.zeros(B, 2 * self.F, **f)
        self.act = torch.zeros(B, self.F, **f)
        self.dn = torch.zeros(B, self.H, **f)
        self.logits = torch.zeros(B, self.Vl, **f)          # this rank's slice of the vocabulary
        self.next_ids = torch.zeros(B, **i64)
        self.keys = torch.zeros(B, **i64)             # argmax merge keys: zero between launches
        # True while self.ids already holds the tokens the last decode step sampled (its argmax wrote
        # them there too): a step launched ahead of the host then needs no copy of its input tokens
        self._ids_synced = False
        self._max_splits = 16
        self.attn_ws = torch.empty(max(ops.decode_workspace_bytes(B, self.nh, self.d, self._max_splits), 16),
                                   dtype=torch.uint8, device=dev)
        self.graphs = {}
        self.buckets = batch_buckets(B)
        self._init_tiled(tiled_linears)
        if self.weight_fp8:
            self._fake_quantize_row_major()
        # pinned host staging for the packed inputs and the sampled tokens: two sets, so
        # that the host can stage step n+1 while step n is still running
        self.h_alls = [torch.zeros(self.inall.numel(), dtype=torch.int32).pin_memory() for _ in range(self.NUM_STAGES)]
        self.h64s = [h[:6 * B].view(torch.int64) for h in self.h_alls]
        self.h32s = [h[6 * B:] for h in self.h_alls]
        self.h_nexts = [torch.zeros(B, dtype=torch.int64).pin_memory() for _ in range(self.NUM_STAGES)]
        self.done = [torch.cuda.Event() for _ in range(self.NUM_STAGES)]
        # log-probabilities of a step (only when a request asked for them): chosen token, top-k ids / values
        self.h_lp = [torch.zeros(B, 1 + 2 * self.MAX_LOGPROBS, dtype=torch.float32).pin_memory()
                     for _ in range(self.NUM_STAGES)]
        self.h64, self.h32, self.h_next = self.h64s[0], self.h32s[0], self.h_nexts[0]
        # sampling parameters of a decode step, packed so that one copy ships them:
        # seeds i64 | steps i64 | temperature f32 | top_p f32 | top_k i32, B entries each
        self.h_samp = [torch.zeros(28 * B, dtype=torch.uint8).pin_memory() for _ in range(self.NUM_STAGES)]
        self.d_samp = torch.zeros(28 * B, dtype=torch.uint8, device=self.device)
        cut = lambda buf: (buf[:8 * B].view(torch.int64), buf[8 * B:16 * B].view(torch.int64),      # noqa: E731
                           buf[16 * B:20 * B].view(torch.float32), buf[20 * B:24 * B].view(torch.float32),
                           buf[24 * B:].view(torch.int32))
        self.h_samp_np = [tuple(v.numpy() for v in cut(h)) for h in self.h_samp]
        self.d_samp_views = cut(self.d_samp)

    # ------------------------------------------------------------------ #
    KINDS = ("qkv", "o", "gu", "down", "lm_head")
    ROW_CLASSES = (16, 32, 64, 128, 256)              # row counts the tiled kernels are instantiated for
    # the fused attention kernel sums up to 4 slabs of the qkv projection with straight-line loads
    # (more would not fit its registers); deeper K splits of qkv would take its slow generic prologue
    QKV_MAX_SLABS = 4

    def _init_tiled(self, want: Optional[bool]) -> None:
        """Fragment-tiled copies of the weights for decode batches of up to 128 rows.
        A projection whose output is all-reduced (o under tensor parallelism, down when the
        MLP is sharded) has to hand the collective a 16-bit tensor: it cannot split K across
        workgroups into fp32 slabs, and runs the tiled kernel with the store epilogue (for the
        narrow slices of a rank: K split across the waves of a workgroup) or a library GEMM,
        whichever the timing picks.

        `want` True: every projection that has a tiled kernel for the step's row count uses it;
        False: library GEMMs only; None (the serving default, MI355_TILED_LINEARS=1/0 forces
        either): the first step of each batch-size bucket times both forms of every projection
        on this GPU and keeps the faster one (wide batches on narrow tensor-parallel slices
        launch too few workgroups for the streaming kernel to win)."""
        import os
        self.tiled_policy = {True: "on", False: "off", None: "auto"}[want]
        if want is None and os.environ.get("MI355_TILED_LINEARS") in ("0", "1"):
            self.tiled_policy = "on" if os.environ["MI355_TILED_LINEARS"] == "1" else "off"
        self.tiled = None
        self._rm_attr = {"qkv": "qkv_w", "o": "o_w", "gu": "gate_up_w", "down": "down_w"}
        self._rm_shape = {"qkv": (self.qw, self.H), "o": (self.H, self.nh * self.d), "gu": (2 * self.F, self.H),
                          "down": (self.H, self.F)}
        self._modes = {}
        self._plan_cache = {}
        self._tuned_plans = {}                        # bucket -> {projection: (tiles_per_wg, k_groups)} picked by tune()
        self._tuned_classes = {}                      # kernel row class (16/32/64/128) -> (mode, plans) already timed
        self.tiled_shapes = {}
        if self.tiled_policy == "off" or self.dtype not in (torch.bfloat16, torch.float16) or (2 * self.F) % 32:
            return
        # third field: K may be split across workgroups into fp32 slabs that the next kernel adds up
        # (attention for qkv; the add + norm kernel, or -- under tensor parallelism, where an
        # all-reduce sits in between -- mi355_sum_slabs for o and down; the SwiGLU-from-slabs
        # kernel for gate_up); the lm_head's logits are consumed as they are
        self.tiled_shapes = {"qkv": (self.qw, self.H, True), "gu": (2 * self.F, self.H, True),
                             "lm_head": (self.Vl, self.H, False), "o": (self.H, self.nh * self.d, True),
                             "down": (self.H, self.F, True)}
        ms = []                                       # the kernels' row-count classes a bucket can fall in
        for m in self.ROW_CLASSES:
            ms.append(m)
            if m >= self.max_num_seqs:
                break
        # projections with no kernel at any row count are not tiled at all
        self.tiled_shapes = {k: v for k, v in self.tiled_shapes.items() if any(k in self._tiled_plan(m) for m in ms)}
        if not self.tiled_shapes:
            return
        self.tiled = []
        # fp8: (bytes, row scales) per matrix instead of a 16-bit tiled tensor; the lm_head stays 16-bit
        tile = ops.tile_weights_fp8 if self.weight_fp8 else ops.tile_weights
        src = {"qkv": "qkv_w", "o": "o_w", "gu": "gate_up_w", "down": "down_w"}
        for lw in self.w.layers:
            self.tiled.append({k: tile(getattr(lw, src[k]), interleave_halves=(k == "gu"))
                               for k in src if k in self.tiled_shapes})
            if self.single_weight_copy:
                # The tiled copy becomes the only one: the row-major matrix a library GEMM needs
                # (prefill, batches beyond 256 rows) is rebuilt from it into a scratch buffer on
                # demand (_rm).  Freed layer by layer, so the two copies never coexist for more
                # than one layer.
                for k in src:
                    if k in self.tiled_shapes:
                        setattr(lw, src[k], None)
        if self.single_weight_copy:                       # one scratch matrix per projection, before any graph capture
            for k in src:
                if k in self.tiled_shapes:
                    self._rm_scratch[k] = torch.empty(*self._rm_shape[k], dtype=self.dtype, device=self.device)
        self.lm_head_tiled = ops.tile_weights(self.w.lm_head) if "lm_head" in self.tiled_shapes else None
        f32 = dict(dtype=torch.float32, device=self.device)
        # room for the deepest K split any plan or timed candidate of a row class asks for
        size = {k: 0 for k in ("qkv", "o", "gu", "down")}
        for m in ms:
            rows = min(m, max(self.max_num_seqs, 16))
            for k in size:
                if k not in self._tiled_plan(m):
                    continue
                n, kk, split = self.tiled_shapes[k]
                deepest = max([g for _, g in self._candidates(k, m)] + [self._tiled_plan(m)[k][1]])
                size[k] = max(size[k], deepest * rows * n)
        self.part_qkv = torch.empty(max(size["qkv"], 4), **f32)
        self.part_o = torch.empty(max(size["o"], 4), **f32)
        self.part_dn = torch.empty(max(size["down"], 4), **f32)
        self.part_gu = torch.empty(max(size["gu"], 4), **f32)

    def _candidates(self, kind: str, m: int):
        """(tiles_per_wg, k_groups) forms of projection `kind` worth timing at m rows."""
        n, kk, split = self.tiled_shapes[kind]
        out = []
        for cand in ops.linear_tiled_candidates(n, kk, split, m, swiglu=(kind == "gu")):
            if kind == "qkv" and cand[1] > self.QKV_MAX_SLABS:
                continue
            # gate_up pays a kernel of its own for taking slabs: worth timing only where the unsplit
            # launch leaves most of the chip idle (a tensor-parallel rank's slice)
            if kind == "gu" and cand[1] > 1 and -(-((n + 15) // 16) // 8) >= 160:
                continue
            out.append(cand)
        return out

    def _fake_quantize_row_major(self) -> None:
        """--quantization fp8: the row-major matrices (prefill, large batches) are replaced by what
        the fp8 copies decode to -- round(fp8(w / scale) * scale), scale = max|row| / 448 -- so that
        every path computes with the same quantised model.  A projection that has an fp8 tiled copy
        is decoded from it (the same bits the decode GEMM multiplies by); one without is quantised
        with the library's conversion."""
        def fq(w: torch.Tensor) -> None:
            f = w.float()
            amax = f.abs().amax(dim=1, keepdim=True)
            scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
            w.copy_(((f / scale).to(torch.float8_e4m3fn).float() * scale).to(w.dtype))
        for i, lw in enumerate(self.w.layers):
            for kind, attr in self._rm_attr.items():
                t = getattr(lw, attr)
                if t is None:                           # freed: rebuilt from the fp8 copy on demand
                    continue
                tw = self.tiled[i].get(kind) if self.tiled is not None else None
                if isinstance(tw, tuple):
                    ops.untile_weights(tw, t.shape[0], t.shape[1], t.dtype, interleave_halves=(kind == "gu"), out=t)
                else:
                    fq(t)

    def _linear_tiled(self, x, w, N, **kw):
        """linear_tiled on a tiled weight of this runner (16-bit tensor, or (fp8 bytes, scales))."""
        if isinstance(w, tuple):
            return ops.linear_tiled_w8(x, w[0], w[1], N, **kw)
        return ops.linear_tiled(x, w, N, **kw)

    def _rm(self, i: int, kind: str) -> torch.Tensor:
        """Row-major [N, K] weight of projection `kind` (qkv / o / gu / down) of layer i for a library
        GEMM.  Normally the stored matrix.  With a single weight copy it is rebuilt from the layer's
        tiled (or fp8) copy into one scratch buffer per projection, in stream order: the GEMM that
        follows reads it before the next layer's rebuild overwrites it."""
        w = getattr(self.w.layers[i], self._rm_attr[kind])
        if w is not None:
            return w
        n, k = self._rm_shape[kind]
        buf = self._rm_scratch.get(kind)
        if buf is None:
            buf = self._rm_scratch[kind] = torch.empty(n, k, dtype=self.dtype, device=self.device)
        return ops.untile_weights(self.tiled[i][kind], n, k, self.dtype, interleave_halves=(kind == "gu"), out=buf)

    def _tiled_plan(self, m: int):
        """{projection: (tiles_per_wg, k_groups)} of the projections that have a tiled kernel for m rows."""
        if m in self._tuned_plans:
            return self._tuned_plans[m]
        if m not in self._plan_cache:
            plans = {}
            if m <= ops.LINEAR_TILED_MAX_M:
                plans = {k: ops.plan_linear_tiled(n, kk, split, m) for k, (n, kk, split) in self.tiled_shapes.items()}
            plans = {k: p for k, p in plans.items() if p is not None}
            if "qkv" in plans:                          # see QKV_MAX_SLABS
                plans["qkv"] = (plans["qkv"][0], min(plans["qkv"][1], self.QKV_MAX_SLABS))
            self._plan_cache[m] = plans
        return self._plan_cache[m]

    def _mode(self, b: int) -> dict:
        """{projection: True if its decode GEMM at b rows is the tiled kernel}.  Fixed per
        batch-size bucket, so a bucket's graph and its eager steps agree."""
        m = self.bucket_for(b) if b <= self.max_num_seqs else b
        if m not in self._modes:
            avail = self._tiled_plan(m) if self.tiled is not None else {}
            # until a bucket is tuned (or when tuning is off) every available kernel is used
            return {k: k in avail for k in self.KINDS}
        return self._modes[m]

    def _use_tiled(self, b: int) -> bool:
        """True if any projection of a b-row decode step runs on the tiled kernels."""
        return any(self._mode(b).values())

    def _time_launches(self, fn) -> float:
        """Average GPU time (us) of fn(layer), replayed from a graph that walks the layers."""
        reps = max(1, 32 // max(self.L, 1))
        fn(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(reps):
                for i in range(self.L):
                    fn(i)
        g.replay()
        best = float("inf")                               # fastest of five replays: least disturbed by other work
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            e.synchronize()
            best = min(best, s.elapsed_time(e))
        return best * 1e3 / (reps * self.L)

    def _time_after_layers(self, fn, b: int) -> float:
        """GPU time (us) of fn() when it runs where the lm_head runs in a step: right after the
        weight streams of all layers.  Timed back to back, a 131-525 MB lm_head is served from the
        256 MB memory-side cache its own previous launch filled, which flatters whichever form is
        less bound by HBM (the library GEMM at 128+ rows) -- in the step the layers' weights have
        passed through that cache in between.  Here every call of fn is preceded by the gate_up
        weight stream of every layer, and the same launches without fn are timed and subtracted."""
        x, act, gu = self.x[:b], self.act[:b], self.gu[:b]

        def layers():
            for i in range(self.L):
                if self.tiled is not None and "gu" in self.tiled[i]:
                    t_, g_ = self._tiled_plan(self.bucket_for(b))["gu"]
                    if g_ > 1:
                        self._linear_tiled(x, self.tiled[i]["gu"], 2 * self.F, partial=self.part_gu, tiles_per_wg=t_,
                                           k_groups=g_, epilogue=ops.EPI_PARTIAL)
                    else:
                        self._linear_tiled(x, self.tiled[i]["gu"], 2 * self.F, out=act, tiles_per_wg=t_,
                                           epilogue=ops.EPI_SILU_MUL)
                else:
                    torch.mm(x, self._rm(i, "gu").t(), out=gu)

        def run(with_fn: bool) -> float:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(2):
                    layers()
                    if with_fn:
                        fn()
            g.replay()
            best = float("inf")
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                g.replay()
                e.record()
                e.synchronize()
                best = min(best, s.elapsed_time(e))
            return best * 1e3 / 2

        layers()
        fn()
        torch.cuda.synchronize()
        return max(run(True) - run(False), 0.0)

    @staticmethod
    def _row_class(m: int) -> int:
        return 16 if m <= 16 else 32 if m <= 32 else 64 if m <= 64 else 128 if m <= 128 else 256

    def tune(self, bucket: int) -> dict:
        """Decides, for one batch-size bucket, how every projection runs: times the tiled kernel in
        each workgroup shape / K split it is built for and the library GEMM (each with the consumer
        kernel whose input format it changes) on scratch activations and neutral step inputs, and
        keeps the fastest.  Buckets that fall in the same row class of the kernels (16/32/64/128)
        share one timing.  Under tensor parallelism rank 0's decisions are broadcast: the replicated
        projections must round the same way on every rank, or the ranks' greedy tokens could part."""
        if bucket in self._modes:
            return self._modes[bucket]
        avail = dict(self._tiled_plan(bucket)) if self.tiled is not None else {}
        mode = {k: k in avail for k in self.KINDS}
        cls = self._row_class(bucket)
        if self.tiled_policy == "auto" and avail and cls in self._tuned_classes:
            mode, avail = self._tuned_classes[cls]
        elif self.tiled_policy == "auto" and avail:
            b = bucket
            # neutral step inputs while timing (no KV store, empty contexts); the staged ones come back after
            saved64, saved32 = self.in64.clone(), self.in32.clone()
            self.slots.fill_(-1)
            self.in32.zero_()
            slots, ctx, bt, splits = self.slots[:b], self.ctx[:b], self.block_tables[:b], self._splits_for(b)
            resid, x, attn = self.resid[:b], self.x[:b], self.attn[:b]
            W = self.w.layers

            def attend(i, **kw):
                ops.paged_attn_decode_fused(kw.pop("qkv", None), self.cos_sin, slots, self.k_cache[i], self.v_cache[i],
                                            bt, ctx, self.nh, self.nkv, self.d, self.scale, num_splits=splits,
                                            workspace=self.attn_ws, out=attn, **kw)

            # projection -> (tiled form for a (tiles_per_wg, k_groups) plan, library form), each with the
            # kernel(s) that take its output format to the next 16-bit tensor of the step
            def form(kind):
                return (lambda t_, g_: lambda i: self._project(kind, i, b, W[i], attend, plan=(t_, g_), tiled=True),
                        lambda i: self._project(kind, i, b, W[i], attend, tiled=False))
            forms = {k: form(k) for k in ("qkv", "o", "gu")}
            forms["down"] = (lambda t_, g_: lambda i: (self._project("down", i, b, W[i], attend, plan=(t_, g_), tiled=True),
                                                       self._add_mlp_and_norm(b, W[i].input_norm, plan=(t_, g_), tiled=True)),
                             lambda i: (self._project("down", i, b, W[i], attend, tiled=False),
                                        self._add_mlp_and_norm(b, W[i].input_norm, tiled=False)))
            forms["lm_head"] = (lambda t_, g_: lambda i: ops.linear_tiled(x, self.lm_head_tiled, self.Vl,
                                                                          out=self.logits[:b], tiles_per_wg=t_),
                                lambda i: torch.mm(x, self.w.lm_head.t(), out=self.logits[:b]))
            self.tune_us = getattr(self, "tune_us", {})
            self._no_collectives = True                     # the all-reduce costs every form the same: not timed
            try:
                for k in [k for k in self.KINDS if k in avail]:
                    tiled_form, lib_fn = forms[k]
                    n, kk, split = self.tiled_shapes[k]
                    best, best_us = avail[k], None
                    # the lm_head is timed where it runs: behind the layers' weight streams (_time_after_layers)
                    clock = (lambda f: self._time_after_layers(lambda: f(0), b)) if k == "lm_head" else self._time_launches
                    for cand in self._candidates(k, b):
                        try:
                            us = clock(tiled_form(*cand))
                        except Mi355Error:                  # no kernel of this form for the code (e.g. fp8 with 108)
                            continue
                        # the planner's own choice unless beaten by more than the timing's repeatability
                        # (replays of one graph agree to ~0.3 %; a 2 % margin here left a form that is
                        # 2 % faster -- down_proj split over 8 K groups, 1.8 % of the whole step -- to chance)
                        if best_us is None or us < 0.99 * best_us:
                            best, best_us = cand, us
                    if best_us is None:
                        best_us = clock(tiled_form(*best))
                    lib_fn(0)                               # first library call may pick its algorithm
                    tl = clock(lib_fn)
                    best_us, tl = round(best_us, 2), round(tl, 2)
                    self.tune_us[(bucket, k)] = (best_us, tl)
                    avail[k] = best
                    mode[k] = best_us <= 1.05 * tl          # near-ties stay tiled: the choice should not flip from run to run
            except RuntimeError as exc:                     # timing replays graphs: without them, no choice is made
                if isinstance(exc, Mi355Error) or not self.allow_eager_fallback:
                    raise
                self.tune_failures += 1
                import logging
                logging.getLogger("mi355_decode").warning(
                    "timing the GEMM forms for batch bucket %d failed (%s); using the tiled kernels", bucket, exc)
                torch.cuda.synchronize()
                avail = dict(self._tiled_plan(bucket))
                mode = {k: k in avail for k in self.KINDS}
            finally:
                self._no_collectives = False
            resid.zero_()                                   # the timed norms accumulated into it
            self.in64.copy_(saved64)
            self.in32.copy_(saved32)
        if self.tp_size > 1 and self.tp_group is not None:
            flat = [v for k in self.KINDS for v in (int(mode[k]), *avail.get(k, (0, 0)))]
            flags = torch.tensor(flat, dtype=torch.int32, device=self.device)
            self._collective(lambda c: torch.distributed.broadcast(c, src=0, group=self.tp_group), flags)
            flat = flags.tolist()
            mode = {k: bool(flat[3 * i]) for i, k in enumerate(self.KINDS)}
            avail = {k: (flat[3 * i + 1], flat[3 * i + 2]) for i, k in enumerate(self.KINDS) if k in avail}
        self._tuned_classes[cls] = (mode, avail)
        self._tuned_plans[bucket] = avail
        self._modes[bucket] = mode
        return mode

    def warm_up(self) -> None:
        """Makes the timed choices (tune) for every kernel row class a decode batch of this runner
        can fall in, at start-up instead of inside the first request that reaches the class.  The
        step inputs are left as they were; HIP graphs are still captured on first use."""
        if self.tiled is None or self.tiled_policy != "auto":
            return
        done = set()
        for bucket in reversed(self.buckets):             # the largest bucket of each class stands for it
            cls = self._row_class(bucket)
            if bucket <= ops.LINEAR_TILED_MAX_M and cls not in done:
                done.add(cls)
                self.tune(bucket)

    def _project(self, kind: str, i: int, b: int, lw, attend, plan=None, tiled=None) -> None:
        """Projection `kind` (qkv / o / gu / down) of layer i over b rows, together with the kernels
        that take its result to the next 16-bit tensor of the step:
          qkv   x -> [fused rope + KV store + attention] -> attn
          o     attn -> (all-reduce under tensor parallelism) + residual add + norm -> x
          gu    x -> SwiGLU -> act
          down  act -> (all-reduce when the MLP is sharded) -> dn, or slabs for the next add + norm
        The GEMM is the tiled kernel in the bucket's plan (`plan` / `tiled` override it: the timing
        runs) or a library GEMM.  A tiled GEMM with K split over workgroups leaves fp32 slabs that
        the next kernel adds up: attention, the add + norm kernel, mi355_silu_and_mul_slabs, or --
        where an all-reduce needs a 16-bit tensor first -- mi355_sum_slabs."""
        P = ops.EPI_PARTIAL
        if tiled is None:
            tiled = self._mode(b)[kind]
        if tiled and plan is None:
            plan = self._tiled_plan(self.bucket_for(b))[kind]
        t_, g_ = plan if tiled else (0, 1)
        tw = self.tiled[i] if self.tiled is not None else None
        resid, x, attn, act = self.resid[:b], self.x[:b], self.attn[:b], self.act[:b]
        if kind == "qkv":
            if tiled:
                self._linear_tiled(x, tw["qkv"], self.qw, partial=self.part_qkv, tiles_per_wg=t_, k_groups=g_, epilogue=P)
                attend(i, partial=self.part_qkv, num_slabs=g_, bias=lw.qkv_b)
            else:
                qkv = self.qkv[:b]
                if lw.qkv_b is None:
                    torch.mm(x, self._rm(i, "qkv").t(), out=qkv)
                else:
                    torch.addmm(lw.qkv_b, x, self._rm(i, "qkv").t(), out=qkv)
                attend(i, qkv=qkv)
        elif kind == "o":
            o = self.o[:b]
            if tiled and self.tp_size == 1:               # the add + norm kernel sums the slabs (one slab too)
                self._linear_tiled(attn, tw["o"], self.H, partial=self.part_o, tiles_per_wg=t_, k_groups=g_, epilogue=P)
                ops.fused_add_rms_norm_partial(self.part_o, g_, resid, lw.post_norm, self.eps, x)
                return
            fused = self._allreduce_norm_fused(b)         # the all-reduce, the residual add and the norm in one launch
            if tiled and g_ > 1:
                self._linear_tiled(attn, tw["o"], self.H, partial=self.part_o, tiles_per_wg=t_, k_groups=g_, epilogue=P)
                if fused:
                    self.oneshot.run_add_norm(resid, lw.post_norm, self.eps, x, slabs=self.part_o, num_slabs=g_)
                    self.oneshot_calls += 1
                    self.fused_allreduce_norms += 1
                    return
                self._allreduce_slabs(self.part_o, g_, o)
            else:
                if tiled:
                    self._linear_tiled(attn, tw["o"], self.H, out=o, tiles_per_wg=t_)
                else:
                    torch.mm(attn, self._rm(i, "o").t(), out=o)
                if fused:
                    self.oneshot.run_add_norm(resid, lw.post_norm, self.eps, x, data=o)
                    self.oneshot_calls += 1
                    self.fused_allreduce_norms += 1
                    return
                self._allreduce(o)
            ops.fused_add_rms_norm(o, resid, lw.post_norm, self.eps, out=x)
        elif kind == "gu":
            if tiled and g_ > 1:
                self._linear_tiled(x, tw["gu"], 2 * self.F, partial=self.part_gu, tiles_per_wg=t_, k_groups=g_, epilogue=P)
                ops.silu_and_mul_slabs(self.part_gu, g_, act)
            elif tiled:
                self._linear_tiled(x, tw["gu"], 2 * self.F, out=act, tiles_per_wg=t_, epilogue=ops.EPI_SILU_MUL)
            else:
                gu = self.gu[:b]
                torch.mm(x, self._rm(i, "gu").t(), out=gu)
                ops.silu_and_mul(gu, out=act)
        elif kind == "down":
            dn = self.dn[:b]
            if tiled and not self.mlp_sharded:            # slabs for the add + norm kernel that follows (_add_mlp_and_norm)
                self._linear_tiled(act, tw["down"], self.H, partial=self.part_dn, tiles_per_wg=t_, k_groups=g_, epilogue=P)
                return
            fused = self.mlp_sharded and self._allreduce_norm_fused(b)    # then _add_mlp_and_norm reduces what is left here
            if tiled and g_ > 1:
                self._linear_tiled(act, tw["down"], self.H, partial=self.part_dn, tiles_per_wg=t_, k_groups=g_, epilogue=P)
                if not fused:
                    self._allreduce_slabs(self.part_dn, g_, dn)
            else:
                if tiled:
                    self._linear_tiled(act, tw["down"], self.H, out=dn, tiles_per_wg=t_)
                else:
                    torch.mm(act, self._rm(i, "down").t(), out=dn)
                if self.mlp_sharded and not fused:
                    self._allreduce(dn)
        else:
            raise ValueError(kind)

    def _add_mlp_and_norm(self, b: int, weight, plan=None, tiled=None) -> None:
        """Residual add of the MLP output + the norm that follows it: from the down projection's slabs
        when it left slabs (tiled, MLP not sharded), from its 16-bit result otherwise."""
        if tiled is None:
            tiled = self._mode(b)["down"]
        g_ = (plan or self._tiled_plan(self.bucket_for(b))["down"])[1] if tiled else 1
        if tiled and not self.mlp_sharded:
            ops.fused_add_rms_norm_partial(self.part_dn, g_, self.resid[:b], weight, self.eps, self.x[:b])
        elif self.mlp_sharded and self._allreduce_norm_fused(b):
            # the down projection left this rank's partial result (16-bit rows, or slabs when its K was split):
            # summed over the ranks, added to the residual and normed by one launch
            if g_ > 1:
                self.oneshot.run_add_norm(self.resid[:b], weight, self.eps, self.x[:b], slabs=self.part_dn, num_slabs=g_)
            else:
                self.oneshot.run_add_norm(self.resid[:b], weight, self.eps, self.x[:b], data=self.dn[:b])
            self.oneshot_calls += 1
            self.fused_allreduce_norms += 1
        else:
            ops.fused_add_rms_norm(self.dn[:b], self.resid[:b], weight, self.eps, out=self.x[:b])

    def _decode_layers(self, b: int, slots, bt, ctx, splits) -> None:
        """All decoder layers + the final norm for b decode tokens held in self.resid[:b]; leaves the
        normed hidden state in self.x[:b].  Each projection runs as the tiled kernel or as a library
        GEMM as self._mode(b) says (_project)."""
        attn = self.attn[:b]

        def attend(i, qkv=None, **kw):
            ops.paged_attn_decode_fused(qkv, self.cos_sin, slots, self.k_cache[i], self.v_cache[i], bt, ctx,
                                        self.nh, self.nkv, self.d, self.scale, num_splits=splits,
                                        workspace=self.attn_ws, out=attn, **kw)

        for i, lw in enumerate(self.w.layers):
            if i > 0:           # layer 0's input norm ran with the embedding gather
                self._add_mlp_and_norm(b, lw.input_norm)
            for kind in ("qkv", "o", "gu", "down"):
                self._project(kind, i, b, lw, attend)
        self._add_mlp_and_norm(b, self.w.final_norm)

    def lm_head_logits(self, x: torch.Tensor, out: torch.Tensor) -> None:
        """Decode-step logits of the rows in x."""
        b = x.shape[0]
        if self._mode(b)["lm_head"]:
            ops.linear_tiled(x, self.lm_head_tiled, self.Vl, out=out,
                             tiles_per_wg=self._tiled_plan(self.bucket_for(b))["lm_head"][0])
        else:
            torch.mm(x, self.w.lm_head.t(), out=out)

    def greedy(self, logits: torch.Tensor, out: torch.Tensor, ids_out: Optional[torch.Tensor] = None) -> None:
        """Greedy token of every row (`ids_out`: a second destination, the next step's input tokens).  With the vocabulary split over the ranks each rank reduces its
        slice to one 64-bit (logit, global index) key per row and the keys are max-reduced: 8 bytes
        per row cross the links instead of the logits, and the result equals the argmax over the
        whole vocabulary (lowest index on ties) on every rank."""
        if not self.vocab_sharded:
            if logits.shape[0] <= self.keys.numel():
                ops.argmax_chunked(logits, self.keys, out, ids_out)
            else:
                ops.argmax(logits, out=out)
                if ids_out is not None:
                    ids_out.copy_(out)
            return
        keys = self.keys[: logits.shape[0]] if logits.shape[0] <= self.keys.numel() else \
            torch.empty(logits.shape[0], dtype=torch.int64, device=self.device)
        ops.argmax_shard_key(logits, self.vocab_offset, out=keys)
        if (self.oneshot is not None and self.tp_size > 1 and not self._no_collectives
                and logits.shape[0] <= self.oneshot.MAX_KEY_ROWS):
            # the keys' max over the ranks and the winners' ids in one launch of the one-shot kernel
            self.oneshot.run_max_keys(keys, out, ids_out)
            self.oneshot_key_reductions += 1
            return
        self._allreduce_max(keys)
        ops.argmax_key_to_index(keys, out=out)
        if ids_out is not None:
            ids_out.copy_(out)

    def _allreduce_max(self, t: torch.Tensor) -> None:
        if self.tp_size > 1:
            self._collective(lambda c: torch.distributed.all_reduce(c, op=torch.distributed.ReduceOp.MAX,
                                                                    group=self.tp_group), t)

    def _collective(self, fn, t: torch.Tensor) -> None:
        """Runs the in-place collective fn on t.  RCCL works on the device tensor (and is captured
        into the decode graph).  Any other backend -- the gloo rehearsal of several ranks on one GPU,
        eager only -- gets an explicit host copy: gloo's own device-tensor path was seen to stall
        with both ranks inside the same collective."""
        if self._device_collectives is None:
            self._device_collectives = torch.distributed.get_backend(self.tp_group) == "nccl"
        if self._device_collectives:
            fn(t)
        else:
            c = t.cpu()
            fn(c)
            t.copy_(c)

    def full_logits(self, logits: torch.Tensor) -> torch.Tensor:
        """[rows, vocab_size] logits from this rank's slice (sampling and log-probabilities need the
        whole row; greedy steps never gather)."""
        if not self.vocab_sharded:
            return logits
        return self._gather_columns(logits.contiguous())

    def _gather_columns(self, t: torch.Tensor) -> torch.Tensor:
        """Concatenation over the ranks of their [rows, Vl] slices, on every rank.  RCCL: one
        all-gather.  Other backends (the gloo rehearsal of several ranks on one GPU): a sum all-reduce of the slice placed in a zero row
        of full width -- x + 0 is exact, so the values are the same."""
        rows = t.shape[0]
        if torch.distributed.get_backend(self.tp_group) == "nccl":
            buf = torch.empty(self.tp_size, rows, self.Vl, dtype=t.dtype, device=t.device)
            torch.distributed.all_gather_into_tensor(buf, t, group=self.tp_group)
            return buf.permute(1, 0, 2).reshape(rows, self.V)
        full = torch.zeros(rows, self.V, dtype=t.dtype, device=t.device)
        full[:, self.vocab_offset:self.vocab_offset + self.Vl] = t
        self._allreduce(full)
        return full

    def kv_cache_bytes_per_block(self) -> int:
        return 2 * self.L * self.nkv * BLOCK_SIZE * self.d * self.k_cache[0].element_size()

    def _splits_for(self, batch: int) -> int:
        if self.decode_splits:
            return int(self.decode_splits)
        wgs = batch * self.nkv
        # a split repeats the query prologue and adds a merge pass: it pays only while the unsplit
        # launch would leave CUs idle (measured at 256 workgroups, contexts 512-1024: 1 split 27.8 us,
        # 2 splits 37.0, 4 splits 42.0)
        want = max(1, 256 // max(wgs, 1))
        by_len = max(1, self.max_model_len // 256)      # keep >= 2 wave-steps per workgroup
        return max(1, min(want, by_len, self._max_splits))

    def decode_collectives_one_shot(self) -> bool:
        """True if every collective of a decode step of this runner, at any batch size it takes, runs on the
        one-shot kernel: the all-reduces (fused with the add + norm behind them) and, with a sharded vocabulary,
        the greedy pick's key reduction."""
        return (self.tp_size > 1 and self.oneshot is not None and self._allreduce_norm_fused(self.max_num_seqs)
                and (not self.vocab_sharded or self.max_num_seqs <= self.oneshot.MAX_KEY_ROWS))

    def _allreduce_norm_fused(self, b: int) -> bool:
        """True if the all-reduce behind o_proj / down_proj of a b-row decode step runs together with the residual
        add + norm that follows it (the one-shot kernel's add + norm form: one launch instead of two)."""
        return (self.tp_size > 1 and self.oneshot is not None and not self._no_collectives
                and os.environ.get("MI355_ALLREDUCE_NORM", "1") != "0"
                and self.oneshot.fits_add_norm(b, self.H, self.dtype))

    def _allreduce(self, t: torch.Tensor) -> None:
        """Sum of the ranks' partial results, in place.  Decode-sized 16-bit messages take the one-shot
        kernel over peer-mapped buffers when the engine was built with one (every rank adds in rank
        order: bit-identical results); everything else the process group's all-reduce."""
        if self.tp_size > 1 and not self._no_collectives:
            if self.oneshot is not None and self.oneshot.fits(t):
                self.oneshot.run(t)
                self.oneshot_calls += 1
            else:
                self._collective(lambda c: torch.distributed.all_reduce(c, group=self.tp_group), t)

    def _allreduce_slabs(self, partial: torch.Tensor, num_slabs: int, out: torch.Tensor) -> None:
        """All-reduce of a K-split projection's result: the slabs are added up and rounded into `out`
        (the 16-bit partial result of this rank), which is then summed over the ranks in place."""
        if self.oneshot is not None and self.oneshot.fits(out) and not self._no_collectives:
            # the slab sum is formed inside the all-reduce launch (no launch of its own)
            self.oneshot.run_slabs(partial, num_slabs, out)
            self.oneshot_calls += 1
            return
        ops.sum_slabs(partial, num_slabs, out)
        self._allreduce(out)

    def _layers(self, n: int, resid, x, qkv, attn, o, gu, act, dn, attend) -> torch.Tensor:
        """Runs all decoder layers over n packed tokens held in `resid`.  Returns the
        buffer holding the last MLP output (to be added to `resid` by the final norm)."""
        w = self.w
        for i, lw in enumerate(w.layers):
            if i == 0:
                ops.rms_norm(resid, lw.input_norm, self.eps, out=x)
            else:
                ops.fused_add_rms_norm(dn, resid, lw.input_norm, self.eps, out=x)
            if lw.qkv_b is None:
                torch.mm(x, self._rm(i, "qkv").t(), out=qkv)
            else:
                torch.addmm(lw.qkv_b, x, self._rm(i, "qkv").t(), out=qkv)
            attend(i, qkv, attn)
            torch.mm(attn, self._rm(i, "o").t(), out=o)
            self._allreduce(o)
            ops.fused_add_rms_norm(o, resid, lw.post_norm, self.eps, out=x)
            torch.mm(x, self._rm(i, "gu").t(), out=gu)
            ops.silu_and_mul(gu, out=act)
            torch.mm(act, self._rm(i, "down").t(), out=dn)
            if self.mlp_sharded:
                self._allreduce(dn)
        return dn

    # ------------------------------------------------------------------ #
    # decode: one token per sequence, static buffers, HIP-graph replay
    # ------------------------------------------------------------------ #
    def _decode_body(self, b: int) -> None:
        ids, pos, slots, ctx, bt = self.ids[:b], self.positions[:b], self.slots[:b], self.ctx[:b], self.block_tables[:b]
        if self.L > 0:          # embedding gather + the first layer's input norm in one launch
            ops.embed_rms_norm(self.w.embed, ids, self.w.layers[0].input_norm, self.eps, self.resid[:b], self.x[:b])
        else:
            torch.index_select(self.w.embed, 0, ids, out=self.resid[:b])
        self._decode_layers(b, slots, bt, ctx, self._splits_for(b))
        self.lm_head_logits(self.x[:b], self.logits[:b])
        self.greedy(self.logits[:b], self.next_ids[:b], ids_out=ids)

    def bucket_for(self, batch: int) -> int:
        for s in self.buckets:
            if s >= batch:
                return s
        raise ValueError(f"batch {batch} exceeds max_num_seqs {self.max_num_seqs}")

    def capture(self, bucket: int) -> None:
        """Warm up and capture the decode step for one padded batch size."""
        if bucket in self.graphs:
            return
        # Warm-up and capture run on neutral inputs (no KV store, empty contexts) so
        # they cannot disturb live pages; the caller's staged inputs are put back after.
        saved64, saved32 = self.in64.clone(), self.in32.clone()
        self.slots.fill_(-1)
        self.ctx.zero_()
        try:
            self.tune(bucket)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._decode_body(bucket)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # Only calls made by the capturing thread may end the capture.  In the default ("global")
            # mode a HIP call from ANY other thread of the process -- a process group's watchdog, a
            # profiler's or a monitoring hook's sampler, another engine thread -- invalidates it, and
            # the error then surfaces while the half-built graph is torn down.
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._decode_body(bucket)
        finally:                              # whatever happened, the caller's staged inputs come back
            self.in64.copy_(saved64)
            self.in32.copy_(saved32)
        self.graphs[bucket] = g

    def decode(self, batch: int) -> torch.Tensor:
        """Runs one decode step over the first `batch` rows of the static input
        buffers (ids, positions, slots, ctx, block_tables must be filled; rows
        beyond `batch` up to the bucket size must be neutral: slot -1, ctx 0).
        Returns a view of the sampled (greedy) token ids, on the device."""
        bucket = self.bucket_for(batch)
        if not self.use_graphs:
            if bucket not in self._modes:
                self.tune(bucket)                         # scratch activations only: staged inputs are untouched
            self._decode_body(batch)
            return self.next_ids[:batch]
        if bucket not in self.graphs:
            try:
                self.capture(bucket)
            except RuntimeError as exc:
                # e.g. a collective that cannot be captured on this RCCL build: every rank takes the
                # same exception at the same step, so all fall back to eager launches together.
                # A kernel or argument error (Mi355Error) is never a capture problem.
                if isinstance(exc, Mi355Error) or not self.allow_eager_fallback:
                    raise
                self.capture_failures += 1
                import logging
                logging.getLogger("mi355_decode").warning(
                    "HIP-graph capture of the decode step failed (%s); continuing with eager launches", exc)
                self.use_graphs = False
                torch.cuda.synchronize()
                self._decode_body(batch)
                return self.next_ids[:batch]
        self.graphs[bucket].replay()
        return self.next_ids[:batch]

    def host_inputs(self, stage: int = 0):
        """numpy views of one set of pinned staging buffers, laid out like in64 / in32."""
        return self.h64s[stage].numpy(), self.h32s[stage].numpy()

    def _sample(self, logits: torch.Tensor, sampling: dict, out: torch.Tensor, stage: Optional[int] = None) -> None:
        """Draw every row's token per its sampling parameters (greedy where temperature <= 0).
        With `stage` (a decode step) the parameters travel in that staging set's packed pinned
        buffer, one copy; without (a prefill, which waits for its result anyway) array by array."""
        n = logits.shape[0]
        tk, tpp, tmp = (np.asarray(sampling[k]) for k in ("top_k", "top_p", "temperature"))
        cut = bool(((tmp > 0) & (((tk > 0) & (tk < logits.shape[1])) | (tpp < 1))).any())   # any row with an active top-k / top-p
        if stage is not None:
            seeds, steps, temp, top_p, top_k = self.h_samp_np[stage]
            seeds[:n], steps[:n] = sampling["seeds"], sampling["steps"]
            temp[:n], top_p[:n], top_k[:n] = sampling["temperature"], sampling["top_p"], sampling["top_k"]
            self.d_samp.copy_(self.h_samp[stage], non_blocking=True)
            d_seeds, d_steps, d_temp, d_top_p, d_top_k = self.d_samp_views
            ops.sample(logits, d_temp[:n], d_top_k[:n], d_top_p[:n], d_seeds[:n], d_steps[:n], out=out, cut_rows=cut)
            return
        dev = self.device
        t = lambda a, dt: torch.from_numpy(a).to(dev, dtype=dt, non_blocking=True)  # noqa: E731
        ops.sample(logits, t(sampling["temperature"], torch.float32), t(sampling["top_k"], torch.int32),
                   t(sampling["top_p"], torch.float32), t(sampling["seeds"], torch.int64),
                   t(sampling["steps"], torch.int64), out=out, cut_rows=cut)

    def _logprobs(self, logits: torch.Tensor, chosen: torch.Tensor, k: int) -> torch.Tensor:
        """[rows, 1 + 2k] float32 on the device: log-probability of the chosen token, then the k
        most likely token ids (as floats: exact below 2^24) and their log-probabilities (most likely first,
        ties to the lower id)."""
        lp, lse = ops.token_logprobs(logits, chosen)
        cols = [lp[:, None]]
        if k > 0:
            topi, topv = ops.topk_logits(logits, k, lse=lse)
            cols += [topi.float(), topv]
        return torch.cat(cols, dim=1)

    @staticmethod
    def _unpack_logprobs(arr, k: int):
        """Rows of _logprobs -> per row (chosen logprob, [(token id, logprob)] * k)."""
        return [(float(r[0]), [(int(r[1 + j]), float(r[1 + k + j])) for j in range(k)]) for r in arr]

    def launch_decode(self, batch: int, stage: int = 0, sampling: Optional[dict] = None,
                      ids_from_device: bool = False, logprobs: int = -1) -> None:
        """Enqueues one decode step on the current stream and returns at once: ships
        staging set `stage`, replays the step, copies the sampled ids to the set's
        pinned result buffer.  With `ids_from_device` the input token of row i is
        the token row i sampled in the step launched before this one (it never
        leaves the GPU), and only positions, slots, context lengths and block tables
        come from the host: the host can launch step n+1 before it has seen the
        tokens of step n.  `sampling` (numpy arrays temperature/top_k/top_p/seeds/
        steps, one entry per row) is given only when some row is not greedy; `logprobs` >= 0 also
        computes the log-probability of every row's token and of its `logprobs` most likely ones."""
        B = self.max_num_seqs
        if ids_from_device:
            self.inall[2 * B:].copy_(self.h_alls[stage][2 * B:], non_blocking=True)   # everything but the token ids
            if not self._ids_synced:
                self.ids[:batch].copy_(self.next_ids[:batch], non_blocking=True)
        else:
            self.inall.copy_(self.h_alls[stage], non_blocking=True)
        nxt = self.decode(batch)
        if sampling is not None or logprobs >= 0:
            logits = self.full_logits(self.logits[:batch])
        if sampling is not None:
            self._sample(logits, sampling, nxt, stage)
        self._ids_synced = sampling is None          # re-drawn rows are in next_ids only
        self.h_nexts[stage][:batch].copy_(nxt, non_blocking=True)
        if self.oneshot is not None:
            self.oneshot.queue_status(stage)              # the communicator's error word travels with the step's tokens
        if logprobs >= 0:
            self.h_lp[stage][:batch, : 1 + 2 * logprobs].copy_(self._logprobs(logits, nxt, logprobs),
                                                               non_blocking=True)
        self.done[stage].record()

    def finish_decode(self, batch: int, stage: int = 0, logprobs: int = -1):
        """Waits for the step launched on `stage`; returns its sampled ids as a list (and, with
        `logprobs` >= 0 as at launch, the per-row log-probabilities from _unpack_logprobs)."""
        self.done[stage].synchronize()
        if self.oneshot is not None:
            # a peer that never arrived left this rank's activations un-reduced: raise in the very step
            # that hit it (the engine thread then fails, /health turns 500 and the process exits non-zero)
            self.oneshot.raise_if_failed(stage)
        toks = self.h_nexts[stage][:batch].tolist()
        if logprobs < 0:
            return toks
        return toks, self._unpack_logprobs(self.h_lp[stage][:batch, : 1 + 2 * logprobs].numpy(), logprobs)

    def run_decode(self, batch: int, sampling: Optional[dict] = None):
        """One synchronous decode step over staging set 0."""
        self.launch_decode(batch, 0, sampling)
        return self.finish_decode(batch, 0)

    def run_prefill(self, token_ids, positions, slot_mapping, block_tables, cu_seqlens, context_lens,
                    max_q_len: int, sampling: Optional[dict] = None, logprobs: int = -1, prompt_logprobs=None):
        """Host-facing prefill step over numpy inputs; returns next-token ids (list), and with
        `logprobs` >= 0 also their log-probabilities (as finish_decode).  `prompt_logprobs` =
        (k, rows, targets) additionally scores packed rows `rows` against the tokens `targets` that
        follow them in their prompts: the result becomes (the usual result, [(logprob of the target,
        [(token id, logprob)] * k)] per scored row)."""
        dev = self.device

        def t(a, dt):
            return torch.from_numpy(a).to(dev, dtype=dt, non_blocking=True)

        score = None
        if prompt_logprobs is not None:
            pk, rows, targets = prompt_logprobs
            score = (int(pk), t(np.asarray(rows), torch.int64), t(np.asarray(targets), torch.int64))
        nxt, logits = self.prefill(t(token_ids, torch.int64), t(positions, torch.int64), t(slot_mapping, torch.int64),
                                   t(block_tables, torch.int32), t(cu_seqlens, torch.int32),
                                   t(context_lens, torch.int32), int(max_q_len),
                                   want_logits=sampling is not None or logprobs >= 0, score=score)
        if sampling is not None:
            self._sample(logits, sampling, nxt)
        if self.oneshot is not None:
            self.oneshot.check()                          # synchronises (a prefill waits for its tokens anyway)
        if logprobs >= 0:
            lp = self._logprobs(logits, nxt, logprobs).cpu().numpy()
            res = nxt.cpu().tolist(), self._unpack_logprobs(lp, logprobs)
        else:
            res = nxt.cpu().tolist()
        if score is not None:
            return res, self._unpack_logprobs(self.prompt_scores.cpu().numpy(), score[0])
        return res

    # ------------------------------------------------------------------ #
    # prefill: packed variable-length prompts (or prompt chunks), eager
    # ------------------------------------------------------------------ #
    def prefill(self, token_ids: torch.Tensor, positions: torch.Tensor, slot_mapping: torch.Tensor,
                block_tables: torch.Tensor, cu_seqlens: torch.Tensor, context_lens: torch.Tensor,
                max_q_len: int, want_logits: bool = True, score=None):
        """token_ids/positions/slot_mapping: int64 [T] on the device; block_tables
        int32 [S, max_blocks]; cu_seqlens int32 [S+1]; context_lens int32 [S].
        Returns (greedy next token per sequence [S], logits [S, V] or None).  `score` = (k, rows,
        targets) leaves in self.prompt_scores the _logprobs rows of packed rows `rows` scored
        against `targets` (the lm_head runs over those rows too, 1024 at a time)."""
        T = token_ids.numel()
        f = dict(dtype=self.dtype, device=self.device)
        resid = torch.index_select(self.w.embed, 0, token_ids)
        x = torch.empty(T, self.H, **f)
        qkv = torch.empty(T, self.qw, **f)
        attn = torch.empty(T, self.nh * self.d, **f)
        o = torch.empty(T, self.H, **f)
        gu = torch.empty(T, 2 * self.F, **f)
        act = torch.empty(T, self.F, **f)
        dn = torch.empty(T, self.H, **f)

        def attend(i, qkv_, attn_):
            ops.rope_kv_write(qkv_, positions, self.cos_sin, slot_mapping, self.k_cache[i], self.v_cache[i],
                              self.nh, self.nkv, self.d)
            ops.paged_attn_prefill(qkv_[:, : self.nh * self.d], self.k_cache[i], self.v_cache[i], block_tables,
                                   cu_seqlens, context_lens, max_q_len, self.nh, self.nkv, self.d, self.scale,
                                   out=attn_)

        last = self._layers(T, resid, x, qkv, attn, o, gu, act, dn, attend)
        rows = (cu_seqlens[1:] - 1).to(torch.int64)
        last_sel = last.index_select(0, rows)
        resid_sel = resid.index_select(0, rows)
        xs = torch.empty_like(last_sel)
        ops.fused_add_rms_norm(last_sel, resid_sel, self.w.final_norm, self.eps, out=xs)
        if score is not None:
            k, srows, targets = score
            parts = []
            for a in range(0, srows.numel(), 1024):
                sel = srows[a:a + 1024]
                hs = torch.empty(sel.numel(), self.H, **f)
                # the final norm adds into its residual argument: work on copies of the selected rows
                ops.fused_add_rms_norm(last.index_select(0, sel), resid.index_select(0, sel), self.w.final_norm,
                                       self.eps, out=hs)
                lg = self.full_logits(torch.mm(hs, self.w.lm_head.t()))
                parts.append(self._logprobs(lg, targets[a:a + 1024], k))
            self.prompt_scores = torch.cat(parts, dim=0)
        logits = torch.mm(xs, self.w.lm_head.t())
        nxt = torch.empty(logits.shape[0], dtype=torch.int64, device=self.device)
        self.greedy(logits, nxt)
        return nxt, (self.full_logits(logits) if want_logits else None)
