"""One model step on one GPU: embed -> L x [norm, qkv, rope+kv-store, paged
attention, o-proj (+all-reduce under TP), norm, gate_up, SwiGLU, down] -> norm ->
lm_head -> greedy token.

The hot operators go through the C ABI (mi355_decode.ops -> libmi355_decode.so).
Decode steps of up to 32 sequences on one GPU run their per-layer linear layers
through the weight-streaming GEMM on fragment-tiled weight copies (K-split layers
hand fp32 partial sums to the norm / rotary kernels, the SwiGLU gate is applied in
the gate_up epilogue); larger batches, prefill and the lm_head use library GEMMs on
the row-major weights.  PyTorch only owns memory, streams and the process group.  The decode step works on statically allocated
buffers and is captured once per batch-size bucket into a HIP graph (a decode
step is ~300 short kernels; replay removes the per-launch host cost); prefill
runs eagerly.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import ops
from ._lib import BLOCK_SIZE
from .config import ModelConfig
from .weights import ModelWeights


def build_cos_sin_table(cfg: ModelConfig, max_pos: int, dtype: torch.dtype, device) -> torch.Tensor:
    """[max_pos, head_dim] = cos | sin halves, computed in fp32 and rounded to the
    model dtype (the table a 16-bit model rotates with)."""
    d = cfg.head_dim
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
                 weight_fp8: bool = False):
        self.cfg, self.w, self.dtype, self.device = cfg, weights, dtype, torch.device(device)
        self.tp_rank, self.tp_size, self.tp_group = tp_rank, tp_size, tp_group
        self.nh = cfg.num_attention_heads // tp_size
        self.nkv = cfg.num_key_value_heads // tp_size
        self.d = cfg.head_dim
        self.H, self.F, self.V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
        self.L = len(weights.layers)
        self.qw = (self.nh + 2 * self.nkv) * self.d
        self.scale = self.d ** -0.5
        self.eps = cfg.rms_norm_eps
        self.num_blocks = int(num_blocks)
        self.max_num_seqs = int(max_num_seqs)
        self.max_model_len = int(max_model_len)
        self.max_blocks = (self.max_model_len + BLOCK_SIZE - 1) // BLOCK_SIZE
        self.use_graphs = use_graphs
        self.decode_splits = decode_splits
        self.weight_fp8 = bool(weight_fp8)

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
        self.in64 = torch.zeros(3 * B, **i64)
        self.in32 = torch.zeros(B + B * self.max_blocks, **i32)
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
        self.logits = torch.zeros(B, self.V, **f)
        self.next_ids = torch.zeros(B, **i64)
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
        self.h64s = [torch.zeros(3 * B, dtype=torch.int64).pin_memory() for _ in range(self.NUM_STAGES)]
        self.h32s = [torch.zeros(B + B * self.max_blocks, dtype=torch.int32).pin_memory()
                     for _ in range(self.NUM_STAGES)]
        self.h_nexts = [torch.zeros(B, dtype=torch.int64).pin_memory() for _ in range(self.NUM_STAGES)]
        self.done = [torch.cuda.Event() for _ in range(self.NUM_STAGES)]
        # log-probabilities of a step (only when a request asked for them): chosen token, top-k ids / values
        self.h_lp = [torch.zeros(B, 1 + 2 * self.MAX_LOGPROBS, dtype=torch.float32).pin_memory()
                     for _ in range(self.NUM_STAGES)]
        self.h64, self.h32, self.h_next = self.h64s[0], self.h32s[0], self.h_nexts[0]

    # ------------------------------------------------------------------ #
    def _init_tiled(self, want: Optional[bool]) -> None:
        """Fragment-tiled copies of the weights for decode batches of up to 128 rows.
        Under tensor parallelism the o projection stays a library GEMM: its output has to
        be a 16-bit tensor for the all-reduce, and the K-split kernel's strength is handing
        fp32 partial sums to the next kernel."""
        import os
        if want is None:
            want = os.environ.get("MI355_TILED_LINEARS", "1") != "0"
        self.tiled = None
        if not want or self.dtype not in (torch.bfloat16, torch.float16) or (2 * self.F) % 32:
            return
        self.tiled_shapes = {"qkv": (self.qw, self.H, True), "gu": (2 * self.F, self.H, False),
                             "down": (self.H, self.F, True), "lm_head": (self.V, self.H, False)}
        if self.tp_size == 1:
            self.tiled_shapes["o"] = (self.H, self.nh * self.d, True)
        self._plan_cache = {}
        self.tiled_max_m = 0
        for m in (16, 32, 64, 128):                   # largest row count every layer has a kernel for
            if self._tiled_plan(m) is None:
                break
            self.tiled_max_m = m
        self.tiled_max_m = min(self.tiled_max_m, self.max_num_seqs)
        if self.tiled_max_m == 0:
            return
        self.tiled = []
        # fp8: (bytes, row scales) per matrix instead of a 16-bit tiled tensor; the lm_head stays 16-bit
        tile = ops.tile_weights_fp8 if self.weight_fp8 else ops.tile_weights
        for lw in self.w.layers:
            tw = {"qkv": tile(lw.qkv_w), "gu": tile(lw.gate_up_w, interleave_halves=True), "down": tile(lw.down_w)}
            if self.tp_size == 1:
                tw["o"] = tile(lw.o_w)
            self.tiled.append(tw)
        self.lm_head_tiled = ops.tile_weights(self.w.lm_head)
        f32 = dict(dtype=torch.float32, device=self.device)
        size = {k: 0 for k in ("qkv", "o", "down")}
        for m in (16, 32, 64, 128):
            m = min(m, self.tiled_max_m)
            for k in size:
                if k in self.tiled_shapes:
                    size[k] = max(size[k], self._tiled_plan(m)[k][1] * m * self.tiled_shapes[k][0])
        self.part_qkv = torch.empty(size["qkv"], **f32)
        self.part_o = torch.empty(max(size["o"], 4), **f32)
        self.part_dn = torch.empty(size["down"], **f32)

    def _fake_quantize_row_major(self) -> None:
        """--quantization fp8: the row-major matrices (prefill, large batches, the TP o projection) are
        replaced by what the fp8 copies decode to -- round(fp8(w / scale) * scale), scale = max|row| / 448
        -- so that every path computes with the same quantised model."""
        def fq(w: torch.Tensor) -> None:
            f = w.float()
            amax = f.abs().amax(dim=1, keepdim=True)
            scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
            w.copy_(((f / scale).to(torch.float8_e4m3fn).float() * scale).to(w.dtype))
        for lw in self.w.layers:
            for t in (lw.qkv_w, lw.o_w, lw.gate_up_w, lw.down_w):
                fq(t)

    def _linear_tiled(self, x, w, N, **kw):
        """linear_tiled on a tiled weight of this runner (16-bit tensor, or (fp8 bytes, scales))."""
        if isinstance(w, tuple):
            return ops.linear_tiled_w8(x, w[0], w[1], N, **kw)
        return ops.linear_tiled(x, w, N, **kw)

    def _tiled_plan(self, m: int):
        """{layer: (tiles_per_wg, k_groups)} for m rows, or None if some layer has no kernel."""
        if m not in self._plan_cache:
            plans = {k: ops.plan_linear_tiled(n, kk, split, m) for k, (n, kk, split) in self.tiled_shapes.items()}
            self._plan_cache[m] = None if any(p is None for p in plans.values()) else plans
        return self._plan_cache[m]

    def _use_tiled(self, b: int) -> bool:
        return self.tiled is not None and b <= self.tiled_max_m and self._tiled_plan(b) is not None

    def _decode_layers_tiled(self, b: int, pos, slots, bt, ctx, splits) -> None:
        """All decoder layers + the final norm for b <= 128 decode tokens on the tiled
        weights; leaves the normed hidden state in self.x[:b]."""
        resid, x, qkv, attn, act, o = self.resid[:b], self.x[:b], self.qkv[:b], self.attn[:b], self.act[:b], self.o[:b]
        P = ops.EPI_PARTIAL
        plans = self._tiled_plan(b)
        (tq, gq), (tg, _), (td, gd) = (plans[k] for k in ("qkv", "gu", "down"))
        for i, (lw, tw) in enumerate(zip(self.w.layers, self.tiled)):
            if i == 0:
                ops.rms_norm(resid, lw.input_norm, self.eps, out=x)
            else:
                ops.fused_add_rms_norm_partial(self.part_dn, gd, resid, lw.input_norm, self.eps, x)
            self._linear_tiled(x, tw["qkv"], self.qw, partial=self.part_qkv, tiles_per_wg=tq, k_groups=gq, epilogue=P)
            ops.paged_attn_decode_fused(None, self.cos_sin, slots, self.k_cache[i], self.v_cache[i], bt, ctx,
                                        self.nh, self.nkv, self.d, self.scale, num_splits=splits,
                                        workspace=self.attn_ws, out=attn, partial=self.part_qkv, num_slabs=gq,
                                        bias=lw.qkv_b)
            if self.tp_size == 1:
                to, go = plans["o"]
                self._linear_tiled(attn, tw["o"], self.H, partial=self.part_o, tiles_per_wg=to, k_groups=go, epilogue=P)
                ops.fused_add_rms_norm_partial(self.part_o, go, resid, lw.post_norm, self.eps, x)
            else:
                torch.mm(attn, lw.o_w.t(), out=o)
                self._allreduce(o)
                ops.fused_add_rms_norm(o, resid, lw.post_norm, self.eps, out=x)
            self._linear_tiled(x, tw["gu"], 2 * self.F, out=act, tiles_per_wg=tg, epilogue=ops.EPI_SILU_MUL)
            self._linear_tiled(act, tw["down"], self.H, partial=self.part_dn, tiles_per_wg=td, k_groups=gd, epilogue=P)
        ops.fused_add_rms_norm_partial(self.part_dn, gd, resid, self.w.final_norm, self.eps, x)

    def lm_head_logits(self, x: torch.Tensor, out: torch.Tensor) -> None:
        """Decode-step logits of the rows in x."""
        if self._use_tiled(x.shape[0]):
            ops.linear_tiled(x, self.lm_head_tiled, self.V, out=out,
                             tiles_per_wg=self._tiled_plan(x.shape[0])["lm_head"][0])
        else:
            torch.mm(x, self.w.lm_head.t(), out=out)

    def kv_cache_bytes_per_block(self) -> int:
        return 2 * self.L * self.nkv * BLOCK_SIZE * self.d * self.k_cache[0].element_size()

    def _splits_for(self, batch: int) -> int:
        if self.decode_splits:
            return int(self.decode_splits)
        wgs = batch * self.nkv
        want = max(1, 512 // max(wgs, 1))
        by_len = max(1, self.max_model_len // 256)      # keep >= 2 wave-steps per workgroup
        return max(1, min(want, by_len, self._max_splits))

    def _allreduce(self, t: torch.Tensor) -> None:
        if self.tp_size > 1:
            torch.distributed.all_reduce(t, group=self.tp_group)

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
                torch.mm(x, lw.qkv_w.t(), out=qkv)
            else:
                torch.addmm(lw.qkv_b, x, lw.qkv_w.t(), out=qkv)
            attend(i, qkv, attn)
            torch.mm(attn, lw.o_w.t(), out=o)
            self._allreduce(o)
            ops.fused_add_rms_norm(o, resid, lw.post_norm, self.eps, out=x)
            torch.mm(x, lw.gate_up_w.t(), out=gu)
            ops.silu_and_mul(gu, out=act)
            torch.mm(act, lw.down_w.t(), out=dn)
        return dn

    # ------------------------------------------------------------------ #
    # decode: one token per sequence, static buffers, HIP-graph replay
    # ------------------------------------------------------------------ #
    def _decode_body(self, b: int) -> None:
        ids, pos, slots, ctx, bt = self.ids[:b], self.positions[:b], self.slots[:b], self.ctx[:b], self.block_tables[:b]
        resid, x, qkv, attn = self.resid[:b], self.x[:b], self.qkv[:b], self.attn[:b]
        o, gu, act, dn = self.o[:b], self.gu[:b], self.act[:b], self.dn[:b]
        splits = self._splits_for(b)
        torch.index_select(self.w.embed, 0, ids, out=resid)
        if self._use_tiled(b):
            self._decode_layers_tiled(b, pos, slots, bt, ctx, splits)
            self.lm_head_logits(x, self.logits[:b])
            ops.argmax(self.logits[:b], out=self.next_ids[:b])
            return

        def attend(i, qkv_, attn_):
            ops.paged_attn_decode_fused(qkv_, self.cos_sin, slots, self.k_cache[i], self.v_cache[i], bt, ctx,
                                        self.nh, self.nkv, self.d, self.scale, num_splits=splits,
                                        workspace=self.attn_ws, out=attn_)

        last = self._layers(b, resid, x, qkv, attn, o, gu, act, dn, attend)
        ops.fused_add_rms_norm(last, resid, self.w.final_norm, self.eps, out=x)
        self.lm_head_logits(x, self.logits[:b])
        ops.argmax(self.logits[:b], out=self.next_ids[:b])

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
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._decode_body(bucket)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # with a process group alive, its watchdog thread may touch the HIP runtime while this
            # thread captures: only calls made by the capturing thread may end the capture
            mode = "thread_local" if self.tp_size > 1 else "global"
            with torch.cuda.graph(g, capture_error_mode=mode):
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
        if not self.use_graphs:
            self._decode_body(batch)
            return self.next_ids[:batch]
        bucket = self.bucket_for(batch)
        if bucket not in self.graphs:
            try:
                self.capture(bucket)
            except RuntimeError as exc:
                # e.g. a collective that cannot be captured on this RCCL build: every rank takes the
                # same exception at the same step, so all fall back to eager launches together
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

    def _sample(self, logits: torch.Tensor, sampling: dict, out: torch.Tensor) -> None:
        """Re-draw the rows that asked for random sampling (greedy rows keep argmax)."""
        dev = self.device
        t = lambda a, dt: torch.from_numpy(a).to(dev, dtype=dt, non_blocking=True)  # noqa: E731
        ops.sample(logits, t(sampling["temperature"], torch.float32), t(sampling["top_k"], torch.int32),
                   t(sampling["top_p"], torch.float32), t(sampling["seeds"], torch.int64),
                   t(sampling["steps"], torch.int64), out=out)

    def _logprobs(self, logits: torch.Tensor, chosen: torch.Tensor, k: int) -> torch.Tensor:
        """[rows, 1 + 2k] float32 on the device: log-probability of the chosen token, then the k
        most likely token ids (as floats: exact below 2^24) and their log-probabilities.  The
        log-sum-exp is the HIP kernel's; picking the top k is a selection, not arithmetic."""
        lp, lse = ops.token_logprobs(logits, chosen)
        cols = [lp[:, None]]
        if k > 0:
            topv, topi = torch.topk(logits.float(), k, dim=1)
            cols += [topi.float(), topv - lse[:, None]]
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
            self.in64[B:].copy_(self.h64s[stage][B:], non_blocking=True)
            self.ids[:batch].copy_(self.next_ids[:batch], non_blocking=True)
        else:
            self.in64.copy_(self.h64s[stage], non_blocking=True)
        self.in32.copy_(self.h32s[stage], non_blocking=True)
        nxt = self.decode(batch)
        if sampling is not None:
            self._sample(self.logits[:batch], sampling, nxt)
        self.h_nexts[stage][:batch].copy_(nxt, non_blocking=True)
        if logprobs >= 0:
            self.h_lp[stage][:batch, : 1 + 2 * logprobs].copy_(self._logprobs(self.logits[:batch], nxt, logprobs),
                                                               non_blocking=True)
        self.done[stage].record()

    def finish_decode(self, batch: int, stage: int = 0, logprobs: int = -1):
        """Waits for the step launched on `stage`; returns its sampled ids as a list (and, with
        `logprobs` >= 0 as at launch, the per-row log-probabilities from _unpack_logprobs)."""
        self.done[stage].synchronize()
        toks = self.h_nexts[stage][:batch].tolist()
        if logprobs < 0:
            return toks
        return toks, self._unpack_logprobs(self.h_lp[stage][:batch, : 1 + 2 * logprobs].numpy(), logprobs)

    def run_decode(self, batch: int, sampling: Optional[dict] = None):
        """One synchronous decode step over staging set 0."""
        self.launch_decode(batch, 0, sampling)
        return self.finish_decode(batch, 0)

    def run_prefill(self, token_ids, positions, slot_mapping, block_tables, cu_seqlens, context_lens,
                    max_q_len: int, sampling: Optional[dict] = None, logprobs: int = -1):
        """Host-facing prefill step over numpy inputs; returns next-token ids (list), and with
        `logprobs` >= 0 also their log-probabilities (as finish_decode)."""
        dev = self.device

        def t(a, dt):
            return torch.from_numpy(a).to(dev, dtype=dt, non_blocking=True)

        nxt, logits = self.prefill(t(token_ids, torch.int64), t(positions, torch.int64), t(slot_mapping, torch.int64),
                                   t(block_tables, torch.int32), t(cu_seqlens, torch.int32),
                                   t(context_lens, torch.int32), int(max_q_len),
                                   want_logits=sampling is not None or logprobs >= 0)
        if sampling is not None:
            self._sample(logits, sampling, nxt)
        if logprobs >= 0:
            lp = self._logprobs(logits, nxt, logprobs).cpu().numpy()
            return nxt.cpu().tolist(), self._unpack_logprobs(lp, logprobs)
        return nxt.cpu().tolist()

    # ------------------------------------------------------------------ #
    # prefill: packed variable-length prompts (or prompt chunks), eager
    # ------------------------------------------------------------------ #
    def prefill(self, token_ids: torch.Tensor, positions: torch.Tensor, slot_mapping: torch.Tensor,
                block_tables: torch.Tensor, cu_seqlens: torch.Tensor, context_lens: torch.Tensor,
                max_q_len: int, want_logits: bool = True):
        """token_ids/positions/slot_mapping: int64 [T] on the device; block_tables
        int32 [S, max_blocks]; cu_seqlens int32 [S+1]; context_lens int32 [S].
        Returns (greedy next token per sequence [S], logits [S, V] or None)."""
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
        logits = torch.mm(xs, self.w.lm_head.t())
        nxt = ops.argmax(logits)
        return nxt, (logits if want_logits else None)
