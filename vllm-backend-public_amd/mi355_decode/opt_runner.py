"""One model step on one GPU for OPT-style decoders (facebook/opt-125m is the model the harness
launches when a request names none, benchmark-vllm/models.py:7): embedding + learned position ->
L x [LayerNorm, qkv (+bias), KV store + paged attention, out-proj (+bias), residual add + LayerNorm,
fc1 (+bias), ReLU, fc2 (+bias)] -> residual add + LayerNorm -> lm_head (tied to the embedding) ->
greedy token.

Shares the step machinery of ModelRunner (static step inputs, HIP-graph capture per batch bucket,
run-ahead, sampling, log-probabilities) and replaces the model-specific parts.  Every operator but
the prefill GEMMs is a hand-written kernel behind the C ABI: decode-step projections (up to 128
rows) stream their row-major weights through mi355_linear_decode, which takes any K that is a
multiple of 32 (OPT's 768 is not a multiple of the tiled kernel's 512-wide slices); the fused decode
attention kernel stores K/V and attends in one launch, its rotation made the identity by a one-row
cos = 1 / sin = 0 table (positions enter an OPT model through the learned table, added to the
embedding, not through the attention).  One GPU only: tensor parallelism is refused for this family.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .config import ModelConfig
from .model_runner import ModelRunner
from .weights import ModelWeights


class OPTRunner(ModelRunner):
    def __init__(self, cfg: ModelConfig, weights: ModelWeights, dtype: torch.dtype, device, num_blocks: int,
                 max_num_seqs: int, max_model_len: int, tp_rank: int = 0, tp_size: int = 1, tp_group=None,
                 use_graphs: bool = True, decode_splits: Optional[int] = None,
                 tiled_linears: Optional[bool] = None, weight_fp8: bool = False):
        if cfg.family != "opt":
            raise ValueError("OPTRunner runs OPT-style models")
        if tp_size != 1:
            raise ValueError("OPT models run on one GPU (tensor_parallel_size 1)")
        if weight_fp8:
            raise ValueError("--quantization fp8 is not available for OPT models")
        if weights.pos_embed is None or weights.final_norm_b is None:
            raise ValueError("OPT weights need the learned position table and the final norm's bias")
        limit = weights.pos_embed.shape[0] - cfg.position_offset
        if max_model_len > limit:
            raise ValueError(f"max_model_len {max_model_len} exceeds the model's {limit} learned positions")
        super().__init__(cfg, weights, dtype, device, num_blocks, max_num_seqs, max_model_len, tp_rank=0, tp_size=1,
                         tp_group=None, use_graphs=use_graphs, decode_splits=decode_splits, tiled_linears=False,
                         weight_fp8=False)
        self.pos_offset = int(cfg.position_offset)

    # ------------------------------------------------------------------ #
    def _linear(self, x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], out: torch.Tensor) -> None:
        """out = x @ w.T + b: the weight-streaming kernel for decode-sized row counts, a library GEMM beyond."""
        if x.shape[0] <= ops.LINEAR_DECODE_MAX_M:
            ops.linear_decode(x, w, b, out=out)
        elif b is None:
            torch.mm(x, w.t(), out=out)
        else:
            torch.addmm(b, x, w.t(), out=out)

    def _run_layers(self, x, resid, qkv, attn, o, f1, dn, attend) -> None:
        """The decoder layers and the final norm over the rows whose embedding sum is in `resid` and
        whose first-layer norm is in `x`; leaves the normed last hidden state in `x`."""
        for i, lw in enumerate(self.w.layers):
            if i > 0:
                ops.fused_add_layer_norm(dn, resid, lw.attn_norm_w, lw.attn_norm_b, self.eps, out=x)
            self._linear(x, lw.qkv_w, lw.qkv_b, qkv)
            attend(i, qkv, attn)
            self._linear(attn, lw.o_w, lw.o_b, o)
            ops.fused_add_layer_norm(o, resid, lw.mlp_norm_w, lw.mlp_norm_b, self.eps, out=x)
            self._linear(x, lw.fc1_w, lw.fc1_b, f1)
            ops.relu(f1)
            self._linear(f1, lw.fc2_w, lw.fc2_b, dn)
        if self.L > 0:
            ops.fused_add_layer_norm(dn, resid, self.w.final_norm, self.w.final_norm_b, self.eps, out=x)

    def _embed(self, ids, positions, resid, x) -> None:
        first = self.w.layers[0] if self.L > 0 else None
        nw, nb = (first.attn_norm_w, first.attn_norm_b) if first is not None else (self.w.final_norm, self.w.final_norm_b)
        ops.embed_pos_layer_norm(self.w.embed, ids, self.w.pos_embed, positions, self.pos_offset, nw, nb, self.eps,
                                 resid, x)

    # ------------------------------------------------------------------ #
    def _decode_body(self, b: int) -> None:
        ids, pos, slots, ctx, bt = self.ids[:b], self.positions[:b], self.slots[:b], self.ctx[:b], self.block_tables[:b]
        resid, x, qkv, attn, o, f1, dn = (t[:b] for t in (self.resid, self.x, self.qkv, self.attn, self.o, self.act,
                                                           self.dn))
        splits = self._splits_for(b)
        self._embed(ids, pos, resid, x)

        def attend(i, qkv_, attn_):
            ops.paged_attn_decode_fused(qkv_, self.cos_sin, slots, self.k_cache[i], self.v_cache[i], bt, ctx, self.nh,
                                        self.nkv, self.d, self.scale, num_splits=splits, workspace=self.attn_ws,
                                        out=attn_)

        self._run_layers(x, resid, qkv, attn, o, f1, dn, attend)
        self.lm_head_logits(x, self.logits[:b])
        self.greedy(self.logits[:b], self.next_ids[:b], ids_out=ids)

    def lm_head_logits(self, x: torch.Tensor, out: torch.Tensor) -> None:
        self._linear(x, self.w.lm_head, None, out)

    def prefill(self, token_ids: torch.Tensor, positions: torch.Tensor, slot_mapping: torch.Tensor,
                block_tables: torch.Tensor, cu_seqlens: torch.Tensor, context_lens: torch.Tensor,
                max_q_len: int, want_logits: bool = True, score=None):
        """As ModelRunner.prefill: packed prompts (or prompt chunks), eager."""
        T = token_ids.numel()
        f = dict(dtype=self.dtype, device=self.device)
        resid, x, o, dn = (torch.empty(T, self.H, **f) for _ in range(4))
        qkv = torch.empty(T, self.qw, **f)
        attn = torch.empty(T, self.nh * self.d, **f)
        f1 = torch.empty(T, self.F, **f)
        self._embed(token_ids, positions, resid, x)

        def attend(i, qkv_, attn_):
            ops.rope_kv_write(qkv_, positions, self.cos_sin, slot_mapping, self.k_cache[i], self.v_cache[i],
                              self.nh, self.nkv, self.d)          # identity rotation: a plain K/V store
            ops.paged_attn_prefill(qkv_[:, : self.nh * self.d], self.k_cache[i], self.v_cache[i], block_tables,
                                   cu_seqlens, context_lens, max_q_len, self.nh, self.nkv, self.d, self.scale,
                                   out=attn_)

        self._run_layers(x, resid, qkv, attn, o, f1, dn, attend)
        rows = (cu_seqlens[1:] - 1).to(torch.int64)
        xs = x.index_select(0, rows)
        if score is not None:
            k, srows, targets = score
            parts = []
            for a in range(0, srows.numel(), 1024):
                sel = srows[a:a + 1024]
                lg = torch.mm(x.index_select(0, sel), self.w.lm_head.t())
                parts.append(self._logprobs(lg, targets[a:a + 1024], k))
            self.prompt_scores = torch.cat(parts, dim=0)
        logits = torch.mm(xs, self.w.lm_head.t())
        nxt = torch.empty(logits.shape[0], dtype=torch.int64, device=self.device)
        self.greedy(logits, nxt)
        return nxt, (logits if want_logits else None)
