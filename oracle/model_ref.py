"""CPU restatement (numpy) of one engine step for Llama / Qwen2 style decoders (RefModel) and
for OPT (RefOPT) over the paged KV cache.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows HF transformers 5.15.0 models/llama/modeling_llama.py:
LlamaDecoderLayer.forward (:295-326), LlamaAttention.forward (:243-281),
LlamaMLP.forward (:174-176), LlamaModel.forward (:367-418, final norm) and
LlamaForCausalLM.forward (:438-492, lm_head); Qwen2 differs by a bias on the
q/k/v projections (models/qwen2/modeling_qwen2.py).  The reference repository
itself only launches the engine that runs this arithmetic
(benchmark-vllm/vllm_manager.py:371-375) -- parity with stock vLLM is unpinned;
this restatement is pinned by tests/golden/tiny_llama_*.npz.
"""
from __future__ import annotations

import numpy as np

from . import ops_ref as R


class RefConfig:
    def __init__(self, **kw):
        self.vocab_size = kw["vocab_size"]
        self.hidden_size = kw["hidden_size"]
        self.intermediate_size = kw["intermediate_size"]
        self.num_hidden_layers = kw["num_hidden_layers"]
        self.num_attention_heads = kw["num_attention_heads"]
        self.num_key_value_heads = kw.get("num_key_value_heads", self.num_attention_heads)
        self.head_dim = kw.get("head_dim") or self.hidden_size // self.num_attention_heads
        self.rope_theta = float(kw.get("rope_theta", 10000.0))
        self.rope_scaling = kw.get("rope_scaling")
        self.rms_norm_eps = float(kw.get("rms_norm_eps", 1e-6))
        self.max_position_embeddings = kw.get("max_position_embeddings", 2048)
        self.attention_bias = bool(kw.get("attention_bias", False))
        self.tie_word_embeddings = bool(kw.get("tie_word_embeddings", False))


class RefModel:
    """weights: dict of HF parameter name -> float32 array already on the
    ``dtype`` grid.  The cache is allocated here ([L] pairs of K/V arrays)."""

    def __init__(self, cfg: RefConfig, weights: dict, dtype: str, num_blocks: int,
                 block_size: int = 16, layers: int | None = None):
        self.cfg, self.w, self.dtype = cfg, weights, dtype
        self.block_size = block_size
        self.L = layers if layers is not None else cfg.num_hidden_layers
        kvh, d = cfg.num_key_value_heads, cfg.head_dim
        self.k_cache = [np.zeros((num_blocks, kvh, block_size, d), np.float32) for _ in range(self.L)]
        self.v_cache = [np.zeros((num_blocks, kvh, d, block_size), np.float32) for _ in range(self.L)]
        self.table = R.rope_cos_sin_table(cfg.max_position_embeddings, d, cfg.rope_theta, dtype,
                                          cfg.rope_scaling)
        self.scale = 1.0 / np.sqrt(d)
        self._fused = {}

    def _qkv(self, i):
        if i not in self._fused:
            p = f"model.layers.{i}.self_attn."
            w = np.concatenate([self.w[p + "q_proj.weight"], self.w[p + "k_proj.weight"],
                                self.w[p + "v_proj.weight"]], axis=0)
            b = None
            if self.cfg.attention_bias:
                b = np.concatenate([self.w[p + "q_proj.bias"], self.w[p + "k_proj.bias"],
                                    self.w[p + "v_proj.bias"]], axis=0)
            m = f"model.layers.{i}.mlp."
            gu = np.concatenate([self.w[m + "gate_proj.weight"], self.w[m + "up_proj.weight"]], axis=0)
            self._fused[i] = (w, b, gu)
        return self._fused[i]

    def forward(self, token_ids, positions, slot_mapping, block_tables, context_lens,
                cu_seqlens=None, logits_rows=None):
        """One step over T packed tokens.
        decode (cu_seqlens None): T == B, one token per sequence.
        prefill: sequence i owns rows [cu_seqlens[i], cu_seqlens[i+1]).
        Returns fp32 logits for ``logits_rows`` (default: all rows for decode,
        last row of each sequence for prefill)."""
        c, dt = self.cfg, self.dtype
        nh, kvh, d = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        h = self.w["model.embed_tokens.weight"][np.asarray(token_ids, dtype=np.int64)]
        T = h.shape[0]
        residual = None
        for i in range(self.L):
            p = f"model.layers.{i}."
            wqkv, bqkv, wgu = self._qkv(i)
            if residual is None:
                residual = h
                x = R.rms_norm(h, self.w[p + "input_layernorm.weight"], c.rms_norm_eps, dt)
            else:
                x, residual = R.fused_add_rms_norm(h, residual, self.w[p + "input_layernorm.weight"],
                                                   c.rms_norm_eps, dt)
            qkv = R.linear(x, wqkv, bqkv, dt)
            qkv = R.rope_kv_write(qkv, positions, self.table, slot_mapping, self.k_cache[i],
                                  self.v_cache[i], nh, kvh, d, dt)
            q = qkv[:, :nh * d].reshape(T, nh, d)
            if cu_seqlens is None:
                a = R.paged_attn_decode(q, self.k_cache[i], self.v_cache[i], block_tables,
                                        context_lens, self.scale, dt)
            else:
                a = R.paged_attn_prefill(q, self.k_cache[i], self.v_cache[i], block_tables,
                                         cu_seqlens, context_lens, self.scale, dt)
            h = R.linear(a.reshape(T, nh * d), self.w[p + "self_attn.o_proj.weight"], None, dt)
            x, residual = R.fused_add_rms_norm(h, residual, self.w[p + "post_attention_layernorm.weight"],
                                               c.rms_norm_eps, dt)
            gu = R.linear(x, wgu, None, dt)
            act = R.silu_and_mul(gu, dt)
            h = R.linear(act, self.w[p + "mlp.down_proj.weight"], None, dt)
        x, _ = R.fused_add_rms_norm(h, residual, self.w["model.norm.weight"], c.rms_norm_eps, dt)
        if logits_rows is None:
            if cu_seqlens is None:
                logits_rows = np.arange(T)
            else:
                logits_rows = np.asarray(cu_seqlens[1:], dtype=np.int64) - 1
        lm = self.w["model.embed_tokens.weight"] if c.tie_word_embeddings else self.w["lm_head.weight"]
        x = x[logits_rows]
        return x.astype(np.float32) @ lm.astype(np.float32).T


class RefOPT:
    """OPT decoder (pre-LayerNorm form: do_layer_norm_before, as opt-125m / 1.3b / ... 175b) over
    the same paged KV cache and with the same forward() signature as RefModel.

    Follows HF transformers 5.15.0 models/opt/modeling_opt.py: OPTDecoderLayer.forward (:203-262:
    LayerNorm -> attention -> residual add -> LayerNorm -> fc1 -> ReLU -> fc2 -> residual add),
    OPTAttention.forward (:135-181: biased q/k/v/out projections; HF multiplies q by head_dim^-0.5
    and rounds it before QK^T, the engine this replaces applies the same factor to the fp32 scores
    inside its attention kernel -- done here as the engine does; for head_dim 64 the factor is
    0.125 and the two are the same numbers), OPTLearnedPositionalEmbedding (:44-70, offset 2),
    OPTDecoder.forward (embeddings + positions, final_layer_norm) and OPTForCausalLM (lm_head tied to
    the embeddings).  Pinned by tests/golden/tiny_opt_*.npz."""

    POS_OFFSET = 2

    def __init__(self, cfg: dict, weights: dict, dtype: str, num_blocks: int, block_size: int = 16,
                 layers: int | None = None):
        self.cfg, self.w, self.dtype, self.block_size = cfg, weights, dtype, block_size
        self.H, self.nh = cfg["hidden_size"], cfg["num_attention_heads"]
        self.d = self.H // self.nh
        if cfg.get("word_embed_proj_dim", self.H) != self.H or not cfg.get("do_layer_norm_before", True):
            raise ValueError("only the pre-LayerNorm OPT form without embedding projections is restated")
        self.L = layers if layers is not None else cfg["num_hidden_layers"]
        self.k_cache = [np.zeros((num_blocks, self.nh, block_size, self.d), np.float32) for _ in range(self.L)]
        self.v_cache = [np.zeros((num_blocks, self.nh, self.d, block_size), np.float32) for _ in range(self.L)]
        self.scale = 1.0 / np.sqrt(self.d)
        self.eps = 1e-5
        self._fused = {}

    def _qkv(self, i):
        if i not in self._fused:
            p = f"model.decoder.layers.{i}.self_attn."
            self._fused[i] = (np.concatenate([self.w[p + f"{n}_proj.weight"] for n in "qkv"], axis=0),
                              np.concatenate([self.w[p + f"{n}_proj.bias"] for n in "qkv"], axis=0))
        return self._fused[i]

    def forward(self, token_ids, positions, slot_mapping, block_tables, context_lens,
                cu_seqlens=None, logits_rows=None):
        dt, nh, d, w = self.dtype, self.nh, self.d, self.w
        D = "model.decoder."
        h = R.embed_positions(w[D + "embed_tokens.weight"], w[D + "embed_positions.weight"], token_ids, positions,
                              self.POS_OFFSET, dt)
        T = h.shape[0]
        residual = None
        for i in range(self.L):
            p = f"{D}layers.{i}."
            wqkv, bqkv = self._qkv(i)
            ln1 = (w[p + "self_attn_layer_norm.weight"], w[p + "self_attn_layer_norm.bias"])
            if residual is None:
                residual = h
                x = R.layer_norm(h, *ln1, self.eps, dt)
            else:
                x, residual = R.fused_add_layer_norm(h, residual, *ln1, self.eps, dt)
            qkv = R.linear(x, wqkv, bqkv, dt)
            q = qkv[:, :nh * d].reshape(T, nh, d)
            R.kv_write(qkv[:, nh * d:2 * nh * d].reshape(T, nh, d), qkv[:, 2 * nh * d:].reshape(T, nh, d),
                       slot_mapping, self.k_cache[i], self.v_cache[i])
            if cu_seqlens is None:
                a = R.paged_attn_decode(q, self.k_cache[i], self.v_cache[i], block_tables, context_lens,
                                        self.scale, dt)
            else:
                a = R.paged_attn_prefill(q, self.k_cache[i], self.v_cache[i], block_tables, cu_seqlens,
                                         context_lens, self.scale, dt)
            h = R.linear(a.reshape(T, nh * d), w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"], dt)
            x, residual = R.fused_add_layer_norm(h, residual, w[p + "final_layer_norm.weight"],
                                                 w[p + "final_layer_norm.bias"], self.eps, dt)
            f1 = R.relu(R.linear(x, w[p + "fc1.weight"], w[p + "fc1.bias"], dt))
            h = R.linear(f1, w[p + "fc2.weight"], w[p + "fc2.bias"], dt)
        x, _ = R.fused_add_layer_norm(h, residual, w[D + "final_layer_norm.weight"], w[D + "final_layer_norm.bias"],
                                      self.eps, dt)
        if logits_rows is None:
            logits_rows = np.arange(T) if cu_seqlens is None else np.asarray(cu_seqlens[1:], dtype=np.int64) - 1
        return x[logits_rows].astype(np.float32) @ w[D + "embed_tokens.weight"].astype(np.float32).T


def make_ref(arch: str, cfg: dict, weights: dict, dtype: str, num_blocks: int, **kw):
    """The restatement of `arch` (an HF architecture name) over `weights`."""
    if arch == "OPTForCausalLM":
        return RefOPT(cfg, weights, dtype, num_blocks, **kw)
    return RefModel(RefConfig(**cfg), weights, dtype, num_blocks, **kw)


def greedy_generate(model, prompts, max_new_tokens: int):
    """Prefill all prompts in one packed step, then decode greedily.
    Block tables are assigned contiguously.  Returns (tokens [B, G], logits of
    every step as a list of [B, V] arrays)."""
    bs = model.block_size
    B = len(prompts)
    lens = [len(p) for p in prompts]
    max_total = max(lens) + max_new_tokens
    blocks_per_seq = (max_total + bs - 1) // bs
    block_tables = np.arange(B * blocks_per_seq, dtype=np.int32).reshape(B, blocks_per_seq)
    assert block_tables.max() < model.k_cache[0].shape[0], "cache too small"

    tok = np.concatenate([np.asarray(p, dtype=np.int64) for p in prompts])
    pos = np.concatenate([np.arange(n, dtype=np.int64) for n in lens])
    slots = np.concatenate([block_tables[b, np.arange(n) // bs] * bs + np.arange(n) % bs
                            for b, n in enumerate(lens)]).astype(np.int64)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ctx = np.asarray(lens, dtype=np.int32)
    logits = model.forward(tok, pos, slots, block_tables, ctx, cu_seqlens=cu)
    all_logits = [logits]
    nxt = R.argmax_lowest(logits)
    out = [nxt]
    for _ in range(max_new_tokens - 1):
        pos = ctx.astype(np.int64)
        slots = (block_tables[np.arange(B), pos // bs] * bs + pos % bs).astype(np.int64)
        ctx = ctx + 1
        logits = model.forward(nxt, pos, slots, block_tables, ctx)
        all_logits.append(logits)
        nxt = R.argmax_lowest(logits)
        out.append(nxt)
    return np.stack(out, axis=1), all_logits
