"""Tensor-level wrappers over the C ABI (libmi355_decode.so).

PyTorch is used only as the owner of device memory and streams: each wrapper
checks dtype/layout, then passes raw device pointers, sizes and the current HIP
stream to the library.  Every call is asynchronous on
``torch.cuda.current_stream()`` and allocates nothing unless it has to return a
new tensor (pass ``out=`` to avoid that, e.g. under HIP graph capture).
"""
from __future__ import annotations

import os

import torch

from . import _lib
from ._lib import BLOCK_SIZE, check, lib

_DT = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16, torch.float32: _lib.DTYPE_F32}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device})")


def _dt16(t: torch.Tensor, name: str) -> int:
    _dev(t, name)
    if t.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError(f"{name}: dtype {t.dtype} not supported (bfloat16/float16)")
    return _DT[t.dtype]


def _rows(t: torch.Tensor, name: str) -> int:
    """Row stride (elements) of a 2-D view whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name} must be 2-D with a contiguous last dim (shape {tuple(t.shape)}, "
                         f"strides {t.stride()})")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: torch.Tensor | None = None):
    dt = _dt16(x, "x")
    _dev(weight, "weight")
    if weight.dtype != x.dtype or weight.numel() != x.shape[1] or not weight.is_contiguous():
        raise ValueError("weight must be contiguous [hidden] in the dtype of x")
    if out is None:
        out = torch.empty_like(x, memory_format=torch.contiguous_format)
    check(lib.mi355_rms_norm(out.data_ptr(), x.data_ptr(), weight.data_ptr(), x.shape[0], x.shape[1],
                             _rows(x, "x"), _rows(out, "out"), float(eps), dt, _stream()))
    return out


def fused_add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                       out: torch.Tensor | None = None):
    """residual <- x + residual (in place); returns rms_norm(residual).  ``out`` defaults to ``x``."""
    dt = _dt16(x, "x")
    _dev(residual, "residual"), _dev(weight, "weight")
    if residual.dtype != x.dtype or residual.shape != x.shape or weight.dtype != x.dtype:
        raise ValueError("residual must match x; weight must have the dtype of x")
    if weight.numel() != x.shape[1] or not weight.is_contiguous():
        raise ValueError("weight must be contiguous [hidden]")
    if out is None:
        out = x
    check(lib.mi355_fused_add_rms_norm(out.data_ptr(), x.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                                       x.shape[0], x.shape[1], _rows(x, "x"), _rows(out, "out"),
                                       _rows(residual, "residual"), float(eps), dt, _stream()))
    return out


def _check_partial(partial: torch.Tensor, num_slabs: int, rows: int, width: int) -> None:
    _dev(partial, "partial")
    if (partial.dtype != torch.float32 or not partial.is_contiguous() or num_slabs < 1
            or partial.numel() < num_slabs * rows * width):
        raise ValueError("partial must be contiguous float32 holding [num_slabs, rows, width]")


def fused_add_rms_norm_partial(partial: torch.Tensor, num_slabs: int, residual: torch.Tensor,
                               weight: torch.Tensor, eps: float, out: torch.Tensor,
                               bias: torch.Tensor | None = None):
    """fused_add_rms_norm whose x is round(sum of the fp32 slabs partial[:num_slabs] (+ bias)):
    the partial-sum output of linear_tiled(..., epilogue=EPI_PARTIAL) over the same rows."""
    dt = _dt16(residual, "residual")
    _dev(weight, "weight"), _dev(out, "out")
    T, H = residual.shape
    _check_partial(partial, num_slabs, T, H)
    if weight.dtype != residual.dtype or weight.numel() != H or not weight.is_contiguous():
        raise ValueError("weight must be contiguous [hidden] in the dtype of residual")
    if out.dtype != residual.dtype or tuple(out.shape) != (T, H):
        raise ValueError("out must match residual")
    if bias is not None and (bias.dtype != residual.dtype or bias.numel() != H or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous [hidden] in the dtype of residual")
    check(lib.mi355_fused_add_rms_norm_partial(out.data_ptr(), partial.data_ptr(), num_slabs, T * H,
                                               bias.data_ptr() if bias is not None else 0, residual.data_ptr(),
                                               weight.data_ptr(), T, H, _rows(out, "out"),
                                               _rows(residual, "residual"), float(eps), dt, _stream()))
    return out


def rope_kv_write(qkv: torch.Tensor, positions: torch.Tensor, cos_sin: torch.Tensor,
                  slot_mapping: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                  num_heads: int, num_kv_heads: int, head_dim: int, partial: torch.Tensor | None = None,
                  num_slabs: int = 0, bias: torch.Tensor | None = None) -> None:
    """In-place rotary embedding on the q|k parts of ``qkv`` + paged store of k, v.
    k_cache [blocks, kv_heads, 16, head_dim]; v_cache [blocks, kv_heads, head_dim, 16].
    With ``partial`` (fp32 [num_slabs, tokens, width], the partial-sum output of
    linear_tiled) the rows are first formed as round(sum of slabs (+ bias)) and
    written to ``qkv`` in full."""
    dt = _dt16(qkv, "qkv")
    T = qkv.shape[0]
    for t, n in ((positions, "positions"), (slot_mapping, "slot_mapping")):
        _dev(t, n)
        if t.dtype != torch.int64 or t.numel() != T or not t.is_contiguous():
            raise ValueError(f"{n} must be contiguous int64 [num_tokens]")
    nb = k_cache.shape[0]
    if (tuple(k_cache.shape) != (nb, num_kv_heads, BLOCK_SIZE, head_dim)
            or tuple(v_cache.shape) != (nb, num_kv_heads, head_dim, BLOCK_SIZE)
            or not k_cache.is_contiguous() or not v_cache.is_contiguous()
            or k_cache.dtype != qkv.dtype or v_cache.dtype != qkv.dtype):
        raise ValueError("k_cache/v_cache must be contiguous [blocks, kv_heads, 16, d] / [blocks, kv_heads, d, 16]")
    if cos_sin.dtype != qkv.dtype or cos_sin.dim() != 2 or cos_sin.shape[1] != head_dim or not cos_sin.is_contiguous():
        raise ValueError("cos_sin must be contiguous [max_position, head_dim] in the dtype of qkv")
    if qkv.shape[1] != (num_heads + 2 * num_kv_heads) * head_dim:
        raise ValueError("qkv width must be (num_heads + 2*num_kv_heads) * head_dim")
    if partial is not None:
        _check_partial(partial, num_slabs, T, qkv.shape[1])
        if bias is not None and (bias.dtype != qkv.dtype or bias.numel() != qkv.shape[1] or not bias.is_contiguous()):
            raise ValueError("bias must be contiguous [width] in the dtype of qkv")
        check(lib.mi355_rope_kv_write_partial(
            qkv.data_ptr(), _rows(qkv, "qkv"), partial.data_ptr(), num_slabs, T * qkv.shape[1],
            bias.data_ptr() if bias is not None else 0, positions.data_ptr(), cos_sin.data_ptr(), cos_sin.shape[0],
            slot_mapping.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), nb, T, num_heads, num_kv_heads,
            head_dim, dt, _stream()))
        return
    check(lib.mi355_rope_kv_write(qkv.data_ptr(), _rows(qkv, "qkv"), positions.data_ptr(), cos_sin.data_ptr(),
                                  cos_sin.shape[0], slot_mapping.data_ptr(), k_cache.data_ptr(),
                                  v_cache.data_ptr(), nb, T, num_heads, num_kv_heads, head_dim, dt, _stream()))


def decode_workspace_bytes(num_seqs: int, num_heads: int, head_dim: int, num_splits: int) -> int:
    return int(lib.mi355_paged_attn_decode_workspace(num_seqs, num_heads, head_dim, num_splits))


def _check_cache(q, k_cache, v_cache, block_tables, num_kv_heads, head_dim):
    nb = k_cache.shape[0]
    if (tuple(k_cache.shape) != (nb, num_kv_heads, BLOCK_SIZE, head_dim)
            or tuple(v_cache.shape) != (nb, num_kv_heads, head_dim, BLOCK_SIZE)
            or not k_cache.is_contiguous() or not v_cache.is_contiguous()
            or k_cache.dtype != q.dtype or v_cache.dtype != q.dtype):
        raise ValueError("k_cache/v_cache must be contiguous [blocks, kv_heads, 16, d] / [blocks, kv_heads, d, 16]")
    _dev(block_tables, "block_tables")
    if block_tables.dtype != torch.int32 or block_tables.dim() != 2 or not block_tables.is_contiguous():
        raise ValueError("block_tables must be contiguous int32 [num_seqs, max_blocks]")
    return nb


def paged_attn_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                      block_tables: torch.Tensor, context_lens: torch.Tensor, num_heads: int,
                      num_kv_heads: int, head_dim: int, scale: float, num_splits: int = 1,
                      workspace: torch.Tensor | None = None, out: torch.Tensor | None = None):
    """q: 2-D view [num_seqs, >= num_heads*head_dim] (e.g. the q part of the qkv rows)."""
    dt = _dt16(q, "q")
    B = q.shape[0]
    nb = _check_cache(q, k_cache, v_cache, block_tables, num_kv_heads, head_dim)
    _dev(context_lens, "context_lens")
    if context_lens.dtype != torch.int32 or context_lens.numel() != B or not context_lens.is_contiguous():
        raise ValueError("context_lens must be contiguous int32 [num_seqs]")
    if block_tables.shape[0] != B:
        raise ValueError("block_tables rows must equal num_seqs")
    if out is None:
        out = torch.empty((B, num_heads * head_dim), dtype=q.dtype, device=q.device)
    elif not out.is_contiguous() or out.numel() != B * num_heads * head_dim or out.dtype != q.dtype:
        raise ValueError("out must be contiguous [num_seqs, num_heads*head_dim]")
    ws_ptr, ws_bytes = 0, 0
    if num_splits > 1:
        need = decode_workspace_bytes(B, num_heads, head_dim, num_splits)
        if workspace is None:
            workspace = torch.empty(need, dtype=torch.uint8, device=q.device)
        ws_ptr, ws_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    check(lib.mi355_paged_attn_decode(out.data_ptr(), q.data_ptr(), _rows(q, "q"), k_cache.data_ptr(),
                                      v_cache.data_ptr(), nb, block_tables.data_ptr(), block_tables.shape[1],
                                      context_lens.data_ptr(), B, num_heads, num_kv_heads, head_dim,
                                      float(scale), num_splits, ws_ptr, ws_bytes, dt, _stream()))
    return out


STATIC_SCRATCH_BYTES = 16 * 1024 * 1024       # MI355_PAGED_ATTN_DECODE_STATIC_MAX_BYTES


def paged_attn_decode_static(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                             block_tables: torch.Tensor, context_lens: torch.Tensor, num_heads: int,
                             num_kv_heads: int, head_dim: int, scale: float, num_splits: int, out: torch.Tensor):
    """paged_attn_decode through the entry point that allocates nothing: the split-KV partials live in
    the library's built-in device array, `out` must be given (this wrapper creates no tensor either)."""
    dt = _dt16(q, "q")
    B = q.shape[0]
    nb = _check_cache(q, k_cache, v_cache, block_tables, num_kv_heads, head_dim)
    _dev(context_lens, "context_lens")
    if context_lens.dtype != torch.int32 or context_lens.numel() != B or not context_lens.is_contiguous():
        raise ValueError("context_lens must be contiguous int32 [num_seqs]")
    if block_tables.shape[0] != B:
        raise ValueError("block_tables rows must equal num_seqs")
    if not out.is_contiguous() or out.numel() != B * num_heads * head_dim or out.dtype != q.dtype:
        raise ValueError("out must be contiguous [num_seqs, num_heads*head_dim]")
    check(lib.mi355_paged_attn_decode_static(out.data_ptr(), q.data_ptr(), _rows(q, "q"), k_cache.data_ptr(),
                                             v_cache.data_ptr(), nb, block_tables.data_ptr(), block_tables.shape[1],
                                             context_lens.data_ptr(), B, num_heads, num_kv_heads, head_dim,
                                             float(scale), num_splits, dt, _stream()))
    return out


def paged_attn_decode_fused(qkv: torch.Tensor | None, cos_sin: torch.Tensor, slot_mapping: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                            block_tables: torch.Tensor, context_lens: torch.Tensor, num_heads: int,
                            num_kv_heads: int, head_dim: int, scale: float, num_splits: int = 1,
                            workspace: torch.Tensor | None = None, out: torch.Tensor | None = None,
                            partial: torch.Tensor | None = None, num_slabs: int = 0,
                            bias: torch.Tensor | None = None):
    """rope_kv_write + paged_attn_decode in one launch; the new token of a sequence is
    the one at position context_lens - 1.  The step's projection is either
    ``qkv`` (packed 16-bit rows [num_seqs, (nh + 2 nkv) d], left untouched) or ``partial``
    (fp32 [num_slabs, num_seqs, width] from linear_tiled, + ``bias``)."""
    B = slot_mapping.numel()
    width = (num_heads + 2 * num_kv_heads) * head_dim
    if partial is None:
        dt = _dt16(qkv, "qkv")
        if qkv.shape != (B, width):
            raise ValueError("qkv must be [num_seqs, (num_heads + 2*num_kv_heads) * head_dim]")
        ref = qkv
    else:
        dt = _dt16(k_cache, "k_cache")
        _check_partial(partial, num_slabs, B, width)
        if bias is not None and (bias.dtype != k_cache.dtype or bias.numel() != width or not bias.is_contiguous()):
            raise ValueError("bias must be contiguous [width] in the cache dtype")
        ref = k_cache
    nb = _check_cache(ref, k_cache, v_cache, block_tables, num_kv_heads, head_dim)
    _dev(slot_mapping, "slot_mapping")
    if slot_mapping.dtype != torch.int64 or not slot_mapping.is_contiguous():
        raise ValueError("slot_mapping must be contiguous int64 [num_seqs]")
    _dev(context_lens, "context_lens")
    if context_lens.dtype != torch.int32 or context_lens.numel() != B or not context_lens.is_contiguous():
        raise ValueError("context_lens must be contiguous int32 [num_seqs]")
    if block_tables.shape[0] != B:
        raise ValueError("block_tables rows must equal num_seqs")
    if cos_sin.dtype != ref.dtype or cos_sin.dim() != 2 or cos_sin.shape[1] != head_dim or not cos_sin.is_contiguous():
        raise ValueError("cos_sin must be contiguous [max_position, head_dim] in the model dtype")
    if out is None:
        out = torch.empty((B, num_heads * head_dim), dtype=ref.dtype, device=ref.device)
    elif not out.is_contiguous() or out.numel() != B * num_heads * head_dim or out.dtype != ref.dtype:
        raise ValueError("out must be contiguous [num_seqs, num_heads*head_dim]")
    ws_ptr, ws_bytes = 0, 0
    if num_splits > 1:
        need = decode_workspace_bytes(B, num_heads, head_dim, num_splits)
        if workspace is None:
            workspace = torch.empty(need, dtype=torch.uint8, device=ref.device)
        ws_ptr, ws_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    check(lib.mi355_paged_attn_decode_fused(
        out.data_ptr(), qkv.data_ptr() if qkv is not None else 0, _rows(qkv, "qkv") if qkv is not None else 0,
        partial.data_ptr() if partial is not None else 0, num_slabs, B * width,
        bias.data_ptr() if bias is not None else 0, cos_sin.data_ptr(), cos_sin.shape[0],
        slot_mapping.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), nb, block_tables.data_ptr(),
        block_tables.shape[1], context_lens.data_ptr(), B, num_heads, num_kv_heads, head_dim, float(scale),
        num_splits, ws_ptr, ws_bytes, dt, _stream()))
    return out


def paged_attn_prefill(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                       block_tables: torch.Tensor, cu_seqlens: torch.Tensor, context_lens: torch.Tensor,
                       max_q_len: int, num_heads: int, num_kv_heads: int, head_dim: int, scale: float,
                       out: torch.Tensor | None = None):
    dt = _dt16(q, "q")
    T = q.shape[0]
    nb = _check_cache(q, k_cache, v_cache, block_tables, num_kv_heads, head_dim)
    S = block_tables.shape[0]
    for t, n, cnt in ((cu_seqlens, "cu_seqlens", S + 1), (context_lens, "context_lens", S)):
        _dev(t, n)
        if t.dtype != torch.int32 or t.numel() != cnt or not t.is_contiguous():
            raise ValueError(f"{n} must be contiguous int32 [{cnt}]")
    if out is None:
        out = torch.empty((T, num_heads * head_dim), dtype=q.dtype, device=q.device)
    check(lib.mi355_paged_attn_prefill(out.data_ptr(), _rows(out, "out"), q.data_ptr(), _rows(q, "q"),
                                       k_cache.data_ptr(), v_cache.data_ptr(), nb, block_tables.data_ptr(),
                                       block_tables.shape[1], cu_seqlens.data_ptr(), context_lens.data_ptr(),
                                       S, int(max_q_len), num_heads, num_kv_heads, head_dim, float(scale), dt,
                                       _stream()))
    return out


def silu_and_mul(x: torch.Tensor, out: torch.Tensor | None = None):
    dt = _dt16(x, "x")
    T, two_f = x.shape
    if two_f % 2:
        raise ValueError("x must be [num_tokens, 2*inter]")
    inter = two_f // 2
    if out is None:
        out = torch.empty((T, inter), dtype=x.dtype, device=x.device)
    elif not out.is_contiguous() or tuple(out.shape) != (T, inter):
        raise ValueError("out must be contiguous [num_tokens, inter]")
    check(lib.mi355_silu_and_mul(out.data_ptr(), x.data_ptr(), _rows(x, "x"), T, inter, dt, _stream()))
    return out


def sum_slabs(partial: torch.Tensor, num_slabs: int, out: torch.Tensor, bias: torch.Tensor | None = None):
    """out[M, N] = round(sum of the first num_slabs fp32 slabs of `partial` ([num_slabs, M, N], flat is fine) + bias),
    slabs added in slab order: the 16-bit tensor an all-reduce needs from a K-split projection."""
    dt = _dt16(out, "out")
    M, N = out.shape
    _check_partial(partial, num_slabs, M, N)
    if bias is not None and (bias.dtype != out.dtype or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous [N] in the dtype of out")
    check(lib.mi355_sum_slabs(out.data_ptr(), _rows(out, "out"), partial.data_ptr(), num_slabs,
                              bias.data_ptr() if bias is not None else 0, M, N, dt, _stream()))
    return out


def silu_and_mul_slabs(partial: torch.Tensor, num_slabs: int, out: torch.Tensor):
    """SwiGLU from the slabs of a K-split gate_up projection on weights tiled with interleave_halves
    ([num_slabs, M, 2F], 16-column tiles alternating gate / up): out[M, F], the values the GEMM's own
    SwiGLU epilogue would have produced from the same sums."""
    dt = _dt16(out, "out")
    M, F = out.shape
    _check_partial(partial, num_slabs, M, 2 * F)
    check(lib.mi355_silu_and_mul_slabs(out.data_ptr(), _rows(out, "out"), partial.data_ptr(), num_slabs, M, 2 * F, dt,
                                       _stream()))
    return out


def argmax(logits: torch.Tensor, out: torch.Tensor | None = None):
    _dev(logits, "logits")
    if logits.dtype not in _DT:
        raise TypeError(f"logits dtype {logits.dtype} not supported")
    B, V = logits.shape
    if out is None:
        out = torch.empty(B, dtype=torch.int64, device=logits.device)
    check(lib.mi355_argmax(out.data_ptr(), logits.data_ptr(), _rows(logits, "logits"), B, V,
                           _DT[logits.dtype], _stream()))
    return out


def argmax_chunked(logits: torch.Tensor, keys: torch.Tensor, out: torch.Tensor, out2: torch.Tensor | None = None):
    """argmax for a decode batch: rows cut into column ranges over many workgroups.  `keys` int64
    [>= rows] must be zero before the first call (every call leaves it zero); `out2` gets the ids too."""
    _dev(logits, "logits"), _dev(keys, "keys"), _dev(out, "out")
    if logits.dtype not in _DT:
        raise TypeError(f"logits dtype {logits.dtype} not supported")
    B, V = logits.shape
    for t, n in ((keys, "keys"), (out, "out")) + (((out2, "out2"),) if out2 is not None else ()):
        if t.dtype != torch.int64 or t.numel() < B or not t.is_contiguous():
            raise ValueError(f"{n} must be contiguous int64 with at least one element per row")
    check(lib.mi355_argmax_chunked(out.data_ptr(), out2.data_ptr() if out2 is not None else 0, keys.data_ptr(),
                                   logits.data_ptr(), _rows(logits, "logits"), B, V, _DT[logits.dtype], _stream()))
    return out


def embed_rms_norm(embed: torch.Tensor, token_ids: torch.Tensor, weight: torch.Tensor, eps: float,
                   residual: torch.Tensor, out: torch.Tensor):
    """residual <- embed[token_ids]; out <- rms_norm(residual): the first launch of a decode step."""
    dt = _dt16(embed, "embed")
    _dev(token_ids, "token_ids"), _dev(weight, "weight"), _dev(residual, "residual"), _dev(out, "out")
    T = token_ids.numel()
    H = embed.shape[1]
    if token_ids.dtype != torch.int64 or not token_ids.is_contiguous():
        raise ValueError("token_ids must be contiguous int64")
    if weight.dtype != embed.dtype or weight.numel() != H or not weight.is_contiguous():
        raise ValueError("weight must be contiguous [hidden] in the dtype of embed")
    for t, n in ((residual, "residual"), (out, "out")):
        if t.dtype != embed.dtype or tuple(t.shape) != (T, H):
            raise ValueError(f"{n} must be [num_tokens, hidden] in the dtype of embed")
    check(lib.mi355_embed_rms_norm(out.data_ptr(), residual.data_ptr(), embed.data_ptr(), token_ids.data_ptr(),
                                   embed.shape[0], weight.data_ptr(), T, H, _rows(embed, "embed"), _rows(out, "out"),
                                   _rows(residual, "residual"), float(eps), dt, _stream()))
    return out


def _affine(weight: torch.Tensor, bias: torch.Tensor, hidden: int, dtype) -> None:
    for t, n in ((weight, "weight"), (bias, "bias")):
        _dev(t, n)
        if t.dtype != dtype or t.numel() != hidden or not t.is_contiguous():
            raise ValueError(f"{n} must be contiguous [hidden] in the dtype of the rows")


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float,
               out: torch.Tensor | None = None):
    """LayerNorm with weight and bias over the rows of x (OPT-style decoders)."""
    dt = _dt16(x, "x")
    _affine(weight, bias, x.shape[1], x.dtype)
    if out is None:
        out = torch.empty_like(x, memory_format=torch.contiguous_format)
    check(lib.mi355_layer_norm(out.data_ptr(), x.data_ptr(), weight.data_ptr(), bias.data_ptr(), x.shape[0], x.shape[1],
                               _rows(x, "x"), _rows(out, "out"), float(eps), dt, _stream()))
    return out


def fused_add_layer_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float,
                         out: torch.Tensor | None = None):
    """residual <- x + residual (in place); returns layer_norm(residual).  ``out`` defaults to ``x``."""
    dt = _dt16(x, "x")
    _dev(residual, "residual")
    if residual.dtype != x.dtype or residual.shape != x.shape:
        raise ValueError("residual must match x")
    _affine(weight, bias, x.shape[1], x.dtype)
    if out is None:
        out = x
    check(lib.mi355_fused_add_layer_norm(out.data_ptr(), x.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                                         bias.data_ptr(), x.shape[0], x.shape[1], _rows(x, "x"), _rows(out, "out"),
                                         _rows(residual, "residual"), float(eps), dt, _stream()))
    return out


def embed_pos_layer_norm(embed: torch.Tensor, token_ids: torch.Tensor, pos_embed: torch.Tensor,
                         positions: torch.Tensor, position_offset: int, weight: torch.Tensor, bias: torch.Tensor,
                         eps: float, residual: torch.Tensor, out: torch.Tensor):
    """residual <- embed[token_ids] + pos_embed[positions + position_offset]; out <- layer_norm(residual):
    the first launch of an OPT-style step."""
    dt = _dt16(embed, "embed")
    _dev(token_ids, "token_ids"), _dev(positions, "positions"), _dev(residual, "residual"), _dev(out, "out")
    _dev(pos_embed, "pos_embed")
    T, H = token_ids.numel(), embed.shape[1]
    for t, n in ((token_ids, "token_ids"), (positions, "positions")):
        if t.dtype != torch.int64 or t.numel() != T or not t.is_contiguous():
            raise ValueError(f"{n} must be contiguous int64 [num_tokens]")
    if pos_embed.dtype != embed.dtype or pos_embed.dim() != 2 or pos_embed.shape[1] != H:
        raise ValueError("pos_embed must be [num_positions, hidden] in the dtype of embed")
    _affine(weight, bias, H, embed.dtype)
    for t, n in ((residual, "residual"), (out, "out")):
        if t.dtype != embed.dtype or tuple(t.shape) != (T, H):
            raise ValueError(f"{n} must be [num_tokens, hidden] in the dtype of embed")
    check(lib.mi355_embed_pos_layer_norm(out.data_ptr(), residual.data_ptr(), embed.data_ptr(), token_ids.data_ptr(),
                                         embed.shape[0], _rows(embed, "embed"), pos_embed.data_ptr(),
                                         positions.data_ptr(), pos_embed.shape[0], int(position_offset),
                                         _rows(pos_embed, "pos_embed"), weight.data_ptr(), bias.data_ptr(), T, H,
                                         _rows(out, "out"), _rows(residual, "residual"), float(eps), dt, _stream()))
    return out


def relu(x: torch.Tensor, out: torch.Tensor | None = None):
    """x < 0 ? 0 : x (NaN and -0.0 kept); in place by default."""
    dt = _dt16(x, "x")
    if out is None:
        out = x
    _dev(out, "out")
    if not x.is_contiguous() or not out.is_contiguous() or out.dtype != x.dtype or out.numel() != x.numel():
        raise ValueError("x and out must be contiguous tensors of one size and dtype")
    check(lib.mi355_relu(out.data_ptr(), x.data_ptr(), x.numel(), dt, _stream()))
    return out


def argmax_shard_key(logits: torch.Tensor, index_offset: int, out: torch.Tensor | None = None):
    """Key of the best token of a vocabulary slice whose first row is token `index_offset`:
    a signed 64-bit max over the slices' keys is the key of the argmax of the whole vocabulary
    (argmax_key_to_index turns it into the token id)."""
    _dev(logits, "logits")
    if logits.dtype not in _DT:
        raise TypeError(f"logits dtype {logits.dtype} not supported")
    B, V = logits.shape
    if out is None:
        out = torch.empty(B, dtype=torch.int64, device=logits.device)
    check(lib.mi355_argmax_shard_key(out.data_ptr(), logits.data_ptr(), _rows(logits, "logits"), B, V,
                                     _DT[logits.dtype], int(index_offset), _stream()))
    return out


def argmax_key_to_index(keys: torch.Tensor, out: torch.Tensor | None = None):
    _dev(keys, "keys")
    if keys.dtype != torch.int64 or not keys.is_contiguous():
        raise TypeError("keys must be contiguous int64")
    if out is None:
        out = torch.empty_like(keys)
    check(lib.mi355_argmax_key_to_index(out.data_ptr(), keys.data_ptr(), keys.numel(), _stream()))
    return out


LINEAR_DECODE_MAX_M = 128


def token_logprobs(logits: torch.Tensor, token_ids: torch.Tensor):
    """(logprob of token_ids[b] in row b, logsumexp of row b), both float32 [rows]."""
    _dev(logits, "logits"), _dev(token_ids, "token_ids")
    ldt = _DT.get(logits.dtype)
    if ldt is None or logits.dim() != 2 or logits.stride(1) != 1:
        raise TypeError("logits must be a 2-D bf16/fp16/fp32 tensor with unit column stride")
    B, V = logits.shape
    if token_ids.dtype != torch.int64 or token_ids.numel() != B or not token_ids.is_contiguous():
        raise ValueError("token_ids must be contiguous int64 [rows]")
    lp = torch.empty(B, dtype=torch.float32, device=logits.device)
    lse = torch.empty(B, dtype=torch.float32, device=logits.device)
    check(lib.mi355_token_logprobs(lp.data_ptr(), lse.data_ptr(), logits.data_ptr(), _rows(logits, "logits"), B, V,
                                   ldt, token_ids.data_ptr(), _stream()))
    return lp, lse


TOPK_MAX = 64


def topk_logits(logits: torch.Tensor, k: int, lse: torch.Tensor | None = None):
    """(token ids int64 [rows, k], values float32 [rows, k]) of the k largest logits of every row, most likely
    first, ties to the lower id; with `lse` (float32 [rows]) the values are log-probabilities."""
    _dev(logits, "logits")
    ldt = _DT.get(logits.dtype)
    if ldt is None or logits.dim() != 2 or logits.stride(1) != 1:
        raise TypeError("logits must be a 2-D bf16/fp16/fp32 tensor with unit column stride")
    B, V = logits.shape
    if lse is not None and (lse.dtype != torch.float32 or lse.numel() != B or not lse.is_contiguous() or not lse.is_cuda):
        raise ValueError("lse must be contiguous float32 [rows] on the device")
    ids = torch.empty((B, k), dtype=torch.int64, device=logits.device)
    vals = torch.empty((B, k), dtype=torch.float32, device=logits.device)
    check(lib.mi355_topk_logits(ids.data_ptr(), vals.data_ptr(), logits.data_ptr(), _rows(logits, "logits"), B, V, ldt,
                                int(k), lse.data_ptr() if lse is not None else 0, _stream()))
    return ids, vals


def linear_decode(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None,
                  out: torch.Tensor | None = None, k_slices: int = 0):
    """y = x @ weight.T (+ bias) for decode batches (M <= 128): weights streamed once."""
    dt = _dt16(x, "x")
    _dev(weight, "weight")
    M, K = x.shape
    if weight.dtype != x.dtype or weight.dim() != 2 or weight.shape[1] != K or not weight.is_contiguous():
        raise ValueError("weight must be contiguous [N, K] in the dtype of x")
    N = weight.shape[0]
    if bias is not None and (bias.dtype != x.dtype or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous [N] in the dtype of x")
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    elif out.dtype != x.dtype or tuple(out.shape) != (M, N):
        raise ValueError("out must be [M, N] in the dtype of x")
    check(lib.mi355_linear_decode(out.data_ptr(), _rows(out, "out"), x.data_ptr(), _rows(x, "x"), weight.data_ptr(),
                                  bias.data_ptr() if bias is not None else 0, M, N, K, int(k_slices), dt, _stream()))
    return out


def sample(logits: torch.Tensor, temperature: torch.Tensor, top_k: torch.Tensor, top_p: torch.Tensor,
           seeds: torch.Tensor, steps: torch.Tensor, out: torch.Tensor | None = None, cut_rows: bool = True):
    """Per-row sampling (greedy where temperature <= 0).  temperature/top_p float32 [B], top_k int32 [B],
    seeds/steps int64 [B], all on the device.  cut_rows=False: the caller knows that no row has an active
    top-k / top-p, so the follow-up launches that serve rows with huge kept sets are left out (same tokens)."""
    _dev(logits, "logits")
    if logits.dtype not in _DT:
        raise TypeError(f"logits dtype {logits.dtype} not supported")
    B, V = logits.shape
    for t, n, dt in ((temperature, "temperature", torch.float32), (top_k, "top_k", torch.int32),
                     (top_p, "top_p", torch.float32), (seeds, "seeds", torch.int64), (steps, "steps", torch.int64)):
        _dev(t, n)
        if t.dtype != dt or t.numel() != B or not t.is_contiguous():
            raise ValueError(f"{n} must be contiguous {dt} [num_rows]")
    if out is None:
        out = torch.empty(B, dtype=torch.int64, device=logits.device)
    check(lib.mi355_sample_topk_topp_hinted(out.data_ptr(), logits.data_ptr(), _rows(logits, "logits"), B, V,
                                            _DT[logits.dtype], temperature.data_ptr(), top_k.data_ptr(),
                                            top_p.data_ptr(), seeds.data_ptr(), steps.data_ptr(),
                                            1 if cut_rows else 0, _stream()))
    return out


EPI_STORE, EPI_SILU_MUL, EPI_PARTIAL, EPI_REDUCE_STORE, EPI_REDUCE_RESID = 0, 1, 2, 3, 4
SUMSQ_ROWS = 32                      # row pitch of the sum-of-squares tables
LINEAR_TILED_MAX_M = 256             # beyond 128 rows: two row blocks of one launch


KW_CODES = (302, 304, 308, 402, 404)   # K split across 2 / 4 / 8 waves of an 8-wave workgroup (4xx: 512-wide x slices)


def row_block_rows(tiles_per_wg: int) -> int:
    """Rows of a row block (one workgroup's rows) for a tiles_per_wg code: 128, + 1000: 64, + 2000: 32."""
    return (128, 64, 32)[tiles_per_wg // 1000]


def tiled_slice_width(M: int, tiles_per_wg: int) -> int:
    """K slice of x a workgroup of linear_tiled stages at a time."""
    M = min(M, row_block_rows(tiles_per_wg))
    tiles_per_wg %= 1000
    mt = 1 if M <= 16 else 2 if M <= 32 else 4 if M <= 64 else 8
    narrow = tiles_per_wg in (104, 108, 208, 402, 404)      # 210 .. 216 (two tiles per wave, 5 .. 8 waves) stage the wide slice
    return (512 if narrow else 1024) // (1 if mt <= 2 else mt // 2)


def tiles_per_workgroup(tiles_per_wg: int) -> int:
    """Weight tiles one workgroup covers for a tiles_per_wg code (208 .. 216: two tiles per wave;
    30x / 40x: 8 waves, x waves per tile; + 1000 / + 2000: the same in row blocks of 64 / 32)."""
    tiles_per_wg %= 1000
    if tiles_per_wg in KW_CODES:
        return 8 // (tiles_per_wg % 100)
    return tiles_per_wg % 100


def plan_linear_tiled(N: int, K: int, split_k: bool, M: int = 32):
    """(tiles_per_wg, k_groups) for linear_tiled on an [N, K] weight and M rows, or None when
    the kernel does not cover the shape.  One workgroup per CU (>= ~192 workgroups) is what
    saturates the weight stream: wide layers take 8 tiles per workgroup, unsplit; narrower
    ones take 4 tiles and the narrow x slice (two workgroups per CU; 8 tiles beyond 64 rows)
    and, when the caller can consume or reduce fp32 partial sums, split K over workgroups
    until there are enough of them."""
    if M > LINEAR_TILED_MAX_M:
        return None
    tiles_total = (N + 15) // 16
    if M > 64:
        candidates = [8]
    elif tiles_total >= 8 * 160:
        candidates = [8, 104]
    else:
        candidates = [104]
    for tiles in candidates:
        ksl = tiled_slice_width(M, tiles)
        if K % ksl:
            continue
        units = K // ksl
        wgs = -(-tiles_total // (tiles % 100)) * (-(-M // 128))          # beyond 128 rows: two row blocks
        groups = 1
        if split_k and wgs < 160:
            # at most 8 slabs: the consumers (attention, norm) sum up to 8 with straight-line loads
            groups = max(1, min(-(-192 // wgs), units, 8))
            # an even deal of the slices if one is near; otherwise K just does not divide evenly
            for g in range(groups, min(2 * groups, units, 8) + 1):
                if units % g == 0:
                    groups = g
                    break
        return tiles, groups
    return None


def linear_tiled_candidates(N: int, K: int, split_k: bool, M: int = 32, swiglu: bool = False):
    """Every (tiles_per_wg, k_groups) worth timing for linear_tiled on an [N, K] weight and M rows:
    both workgroup shapes that every kernel form (fp8, fused epilogues) is built for, K split into
    1, 2, 4 or 8 groups when the caller can take partial sums, beyond 32 rows also the forms with
    two tiles per wave (208 .. 216: 4 .. 8 waves, so that a tile count that 16 per workgroup would
    spread over too few CUs finds a better deal), kept if the launch has between 64 and 1024 workgroups; up to 32
    rows and few tiles also the forms that split K across the waves of a workgroup (`swiglu`: the
    gate / up pair must share a workgroup, at most 4 waves per tile); beyond 64 / 32 rows also the forms of
    the 64- / 32-row class run in row blocks of 64 / 32 (tiles_per_wg + 1000 / + 2000).  Not
    every form exists for every code (fp8 at up to 32 rows: 8 and 104 only): the caller skips a
    candidate the library refuses.  plan_linear_tiled's choice comes first."""
    first = plan_linear_tiled(N, K, split_k, M)
    if first is None:
        return []
    out = [first]
    tiles_total = (N + 15) // 16
    row_blocks = -(-M // 128)
    for tiles in ((8, 104, 108) if M <= 32 else (8, 104, 108, 208, 216, 210, 212, 214) if M <= 64
                  else (8, 216, 210, 212, 214)):
        ksl = tiled_slice_width(M, tiles)
        if K % ksl:
            continue
        wgs = -(-tiles_total // tiles_per_workgroup(tiles)) * row_blocks
        for g in ((1, 2, 4, 8) if split_k else (1,)):
            if g <= K // ksl and 64 <= wgs * g <= 1024 and (tiles, g) not in out:
                out.append((tiles, g))
    # beyond 64 / 32 rows: the forms of the 64- / 32-row class in row blocks of 64 / 32 rows (tiles_per_wg + 1000 / + 2000)
    for rb_code, rb, forms in ((1000, 64, (8, 104, 108, 208, 216, 210, 212, 214, 302, 304, 402)),
                               (2000, 32, (8, 104, 108, 302, 304, 402))):
        if M <= rb or os.environ.get("MI355_ROW_BLOCKS", "1") == "0":      # (the switch: for A/B timing)
            continue
        blocks = -(-M // rb)
        for tiles in forms:
            code = rb_code + tiles
            ksl = tiled_slice_width(M, code)
            if K % ksl or (tiles in KW_CODES and ((swiglu and tiles % 100 > 4) or (tiles % 100) * 128 > ksl)):
                continue
            wgs = -(-tiles_total // tiles_per_workgroup(code)) * blocks
            for g in ((1, 2, 4, 8) if split_k else (1,)):
                if g <= K // ksl and 96 <= wgs * g <= 1024 and (code, g) not in out:
                    out.append((code, g))
    if tiles_total < 8 * 160:
        # few tiles (a tensor-parallel rank's slice): K split across the waves of a workgroup, the
        # only way to more waves when the output must be 16-bit or gated (no slabs)
        for tiles in KW_CODES:
            ksl = tiled_slice_width(M, tiles)
            if K % ksl or (swiglu and tiles % 100 > 4) or (tiles % 100) * 128 > ksl:   # a wave needs a register group of the slice
                continue
            wgs = -(-tiles_total // tiles_per_workgroup(tiles)) * row_blocks
            for g in ((1, 2, 4, 8) if split_k else (1,)):
                if g <= K // ksl and 48 <= wgs * g <= 1024 and (tiles, g) not in out:
                    out.append((tiles, g))
    return out


def tile_weights(weight: torch.Tensor, interleave_halves: bool = False) -> torch.Tensor:
    """Row-major [N, K] -> fragment-tiled flat tensor for linear_tiled (done once at load)."""
    _dt16(weight, "weight")
    if weight.dim() != 2 or not weight.is_contiguous():
        raise ValueError("weight must be contiguous [N, K]")
    N, K = weight.shape
    n = int(lib.mi355_tiled_weight_elems(N, K))
    if n == 0:
        raise ValueError(f"K {K} must be a multiple of 128")
    out = torch.empty(n, dtype=weight.dtype, device=weight.device)
    check(lib.mi355_tile_weights(out.data_ptr(), weight.data_ptr(), N, K, int(bool(interleave_halves)), _stream()))
    return out


def untile_weights(w_tiled, N: int, K: int, dtype: torch.dtype, interleave_halves: bool = False,
                   out: torch.Tensor | None = None) -> torch.Tensor:
    """Row-major [N, K] matrix from a tiled weight: a 16-bit tensor of tile_weights (the same bits
    back), or the (fp8 bytes, row scales) pair of tile_weights_fp8 (decoded: round(fp8 * scale))."""
    fp8 = isinstance(w_tiled, tuple)
    src = w_tiled[0] if fp8 else w_tiled
    _dev(src, "w_tiled")
    if dtype not in _DT:
        raise TypeError(f"dtype {dtype} not supported")
    want = int(lib.mi355_tiled_weight_fp8_bytes(N, K)) if fp8 else int(lib.mi355_tiled_weight_elems(N, K))
    if want == 0 or src.numel() != want or (not fp8 and src.dtype != dtype) or (fp8 and src.dtype != torch.uint8):
        raise ValueError("w_tiled does not match (N, K) / dtype")
    if out is None:
        out = torch.empty((N, K), dtype=dtype, device=src.device)
    elif out.dtype != dtype or tuple(out.shape) != (N, K) or not out.is_contiguous():
        raise ValueError("out must be contiguous [N, K] in the model dtype")
    if fp8:
        check(lib.mi355_untile_weights_fp8(out.data_ptr(), src.data_ptr(), w_tiled[1].data_ptr(), N, K,
                                           int(bool(interleave_halves)), _DT[dtype], _stream()))
    else:
        check(lib.mi355_untile_weights(out.data_ptr(), src.data_ptr(), N, K, int(bool(interleave_halves)), _stream()))
    return out


def linear_tiled(x: torch.Tensor, w_tiled: torch.Tensor, N: int, bias: torch.Tensor | None = None,
                 out: torch.Tensor | None = None, partial: torch.Tensor | None = None, tiles_per_wg: int = 0,
                 k_groups: int = 1, epilogue: int = EPI_STORE):
    """y = x @ W.T on fragment-tiled weights (M <= 256).  Returns `out`
    ([M, N], or [M, N/2] with the SwiGLU epilogue) or `partial` ([k_groups, M, N] fp32)."""
    dt = _dt16(x, "x")
    _dev(w_tiled, "w_tiled")
    M, K = x.shape
    if w_tiled.dtype != x.dtype or w_tiled.numel() != int(lib.mi355_tiled_weight_elems(N, K)):
        raise ValueError("w_tiled does not match (N, K) / dtype")
    tiles_total = (N + 15) // 16
    if tiles_per_wg == 0:                         # 8 tiles (waves) per workgroup once that still fills the chip
        tiles_per_wg = 8 if M > 64 or -(-tiles_total // 8) * k_groups >= 160 else (4 if M <= 32 else 104)
    y_ptr = p_ptr = 0
    y_stride = 0
    if epilogue == EPI_PARTIAL:
        if partial is None:
            partial = torch.empty((k_groups, M, N), dtype=torch.float32, device=x.device)
        elif partial.dtype != torch.float32 or partial.numel() < k_groups * M * N or not partial.is_contiguous():
            raise ValueError("partial must be contiguous float32 [k_groups, M, N]")
        p_ptr = partial.data_ptr()
        ret = partial
    else:
        width = N // 2 if epilogue == EPI_SILU_MUL else N
        if out is None:
            out = torch.empty((M, width), dtype=x.dtype, device=x.device)
        elif out.dtype != x.dtype or tuple(out.shape) != (M, width):
            raise ValueError(f"out must be [M, {width}] in the dtype of x")
        y_ptr, y_stride, ret = out.data_ptr(), _rows(out, "out"), out
    if bias is not None and (bias.dtype != x.dtype or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous [N] in the dtype of x")
    check(lib.mi355_linear_tiled(y_ptr, y_stride, p_ptr, x.data_ptr(), _rows(x, "x"), w_tiled.data_ptr(),
                                 bias.data_ptr() if bias is not None else 0, M, N, K, tiles_per_wg, k_groups,
                                 epilogue, dt, _stream()))
    return ret


def linear_tiled_fused(x: torch.Tensor, w_tiled: torch.Tensor, N: int, *, tiles_per_wg: int, k_groups: int = 1,
                       epilogue: int = EPI_STORE, bias: torch.Tensor | None = None, out: torch.Tensor | None = None,
                       partial: torch.Tensor | None = None, norm_weight: torch.Tensor | None = None,
                       norm_sumsq: torch.Tensor | None = None, norm_parts: int = 0, eps: float = 0.0,
                       residual: torch.Tensor | None = None, sumsq_out: torch.Tensor | None = None,
                       counters: torch.Tensor | None = None):
    """linear_tiled with the RMSNorm before it (norm_weight + norm_sumsq: x is the residual
    stream, normalised while it is staged) and/or the K-split reduction, residual add and
    sum of squares after it (epilogues EPI_REDUCE_STORE / EPI_REDUCE_RESID) in the same launch."""
    dt = _dt16(x, "x")
    _dev(w_tiled, "w_tiled")
    M, K = x.shape
    if w_tiled.dtype != x.dtype or w_tiled.numel() != int(lib.mi355_tiled_weight_elems(N, K)):
        raise ValueError("w_tiled does not match (N, K) / dtype")
    nw = tiles_per_wg % 100
    groups = -(-((N + 15) // 16) // max(nw, 1))
    y_ptr, y_stride = 0, 0
    if epilogue in (EPI_STORE, EPI_SILU_MUL, EPI_REDUCE_STORE):
        width = N // 2 if epilogue == EPI_SILU_MUL else N
        if out is None:
            out = torch.empty((M, width), dtype=x.dtype, device=x.device)
        elif out.dtype != x.dtype or tuple(out.shape) != (M, width):
            raise ValueError(f"out must be [M, {width}] in the dtype of x")
        y_ptr, y_stride = out.data_ptr(), _rows(out, "out")
    if epilogue in (EPI_PARTIAL, EPI_REDUCE_STORE, EPI_REDUCE_RESID):
        if partial is None:
            partial = torch.empty((k_groups, M, N), dtype=torch.float32, device=x.device)
        _check_partial(partial, k_groups, M, N)
    if epilogue in (EPI_REDUCE_STORE, EPI_REDUCE_RESID):
        if counters is None or counters.dtype != torch.int32 or counters.numel() < groups or not counters.is_cuda:
            raise ValueError(f"counters must be a zero-initialised int32 device tensor with >= {groups} elements")
    if epilogue == EPI_REDUCE_RESID:
        if residual is None or residual.dtype != x.dtype or tuple(residual.shape) != (M, N):
            raise ValueError("residual must be [M, N] in the dtype of x")
        if (sumsq_out is None or sumsq_out.dtype != torch.float32 or sumsq_out.numel() < groups * SUMSQ_ROWS
                or not sumsq_out.is_contiguous()):
            raise ValueError(f"sumsq_out must be contiguous float32 with >= {groups} x {SUMSQ_ROWS} elements")
    if norm_weight is not None:
        if norm_weight.dtype != x.dtype or norm_weight.numel() != K or not norm_weight.is_contiguous():
            raise ValueError("norm_weight must be contiguous [K] in the dtype of x")
        if (norm_sumsq is None or norm_sumsq.dtype != torch.float32 or norm_parts < 1
                or norm_sumsq.numel() < norm_parts * SUMSQ_ROWS or not norm_sumsq.is_contiguous()):
            raise ValueError("norm_sumsq must be contiguous float32 [norm_parts, 32]")
    if bias is not None and (bias.dtype != x.dtype or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous [N] in the dtype of x")
    ptr = lambda t: t.data_ptr() if t is not None else 0  # noqa: E731
    check(lib.mi355_linear_tiled_fused(
        y_ptr, y_stride, ptr(partial), x.data_ptr(), _rows(x, "x"), w_tiled.data_ptr(), ptr(bias), M, N, K,
        tiles_per_wg, k_groups, epilogue, ptr(norm_weight), ptr(norm_sumsq), norm_parts, float(eps), ptr(residual),
        _rows(residual, "residual") if residual is not None else 0, ptr(sumsq_out), ptr(counters), dt, _stream()))
    if epilogue == EPI_PARTIAL:
        return partial
    if epilogue == EPI_REDUCE_RESID:
        return residual
    return out


def tile_weights_fp8(weight: torch.Tensor, interleave_halves: bool = False):
    """Row-major 16-bit [N, K] -> (fp8 e4m3fn fragment-tiled bytes, fp32 row scales in tiled row order)."""
    dt = _dt16(weight, "weight")
    if weight.dim() != 2 or not weight.is_contiguous():
        raise ValueError("weight must be contiguous [N, K]")
    N, K = weight.shape
    n = int(lib.mi355_tiled_weight_fp8_bytes(N, K))
    if n == 0:
        raise ValueError(f"K {K} must be a multiple of 128")
    q = torch.empty(n, dtype=torch.uint8, device=weight.device)
    scale = torch.empty((N + 15) // 16 * 16, dtype=torch.float32, device=weight.device)
    check(lib.mi355_tile_weights_fp8(q.data_ptr(), scale.data_ptr(), weight.data_ptr(), N, K,
                                     int(bool(interleave_halves)), dt, _stream()))
    return q, scale


def linear_tiled_w8(x: torch.Tensor, w_fp8: torch.Tensor, w_scale: torch.Tensor, N: int,
                    bias: torch.Tensor | None = None, out: torch.Tensor | None = None,
                    partial: torch.Tensor | None = None, tiles_per_wg: int = 0, k_groups: int = 1,
                    epilogue: int = EPI_STORE):
    """linear_tiled on fp8 weights from tile_weights_fp8 (x stays 16-bit; M <= 128)."""
    dt = _dt16(x, "x")
    _dev(w_fp8, "w_fp8"), _dev(w_scale, "w_scale")
    M, K = x.shape
    if w_fp8.dtype != torch.uint8 or w_fp8.numel() != int(lib.mi355_tiled_weight_fp8_bytes(N, K)):
        raise ValueError("w_fp8 does not match (N, K)")
    if w_scale.dtype != torch.float32 or w_scale.numel() != (N + 15) // 16 * 16 or not w_scale.is_contiguous():
        raise ValueError("w_scale must be contiguous float32 [ceil(N/16)*16]")
    if tiles_per_wg == 0:
        tiles_per_wg = 8 if M > 64 or -(-((N + 15) // 16) // 8) * k_groups >= 160 else 104
    y_ptr = p_ptr = y_stride = 0
    if epilogue == EPI_PARTIAL:
        if partial is None:
            partial = torch.empty((k_groups, M, N), dtype=torch.float32, device=x.device)
        _check_partial(partial, k_groups, M, N)
        p_ptr, ret = partial.data_ptr(), partial
    else:
        width = N // 2 if epilogue == EPI_SILU_MUL else N
        if out is None:
            out = torch.empty((M, width), dtype=x.dtype, device=x.device)
        elif out.dtype != x.dtype or tuple(out.shape) != (M, width):
            raise ValueError(f"out must be [M, {width}] in the dtype of x")
        y_ptr, y_stride, ret = out.data_ptr(), _rows(out, "out"), out
    if bias is not None and (bias.dtype != x.dtype or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous [N] in the dtype of x")
    check(lib.mi355_linear_tiled_w8(y_ptr, y_stride, p_ptr, x.data_ptr(), _rows(x, "x"), w_fp8.data_ptr(),
                                    w_scale.data_ptr(), bias.data_ptr() if bias is not None else 0, M, N, K,
                                    tiles_per_wg, k_groups, epilogue, dt, _stream()))
    return ret


def quant_rows_fp8(x: torch.Tensor, out: torch.Tensor | None = None, scale: torch.Tensor | None = None):
    """16-bit rows [M, K] -> (e4m3fn rows as uint8 [M, K], one fp32 scale per row): the activation
    input of linear_tiled_w8a8."""
    dt = _dt16(x, "x")
    M, K = x.shape
    if out is None:
        out = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    elif out.dtype != torch.uint8 or tuple(out.shape) != (M, K):
        raise ValueError("out must be uint8 [M, K]")
    if scale is None:
        scale = torch.empty(M, dtype=torch.float32, device=x.device)
    elif scale.dtype != torch.float32 or scale.numel() != M or not scale.is_contiguous():
        raise ValueError("scale must be contiguous float32 [M]")
    check(lib.mi355_quant_rows_fp8(out.data_ptr(), _rows(out, "out"), scale.data_ptr(), x.data_ptr(), _rows(x, "x"),
                                   M, K, dt, _stream()))
    return out, scale


def linear_tiled_w8a8(xq: torch.Tensor, x_scale: torch.Tensor, w_fp8: torch.Tensor, w_scale: torch.Tensor, N: int,
                      dtype: torch.dtype = torch.bfloat16, bias: torch.Tensor | None = None,
                      out: torch.Tensor | None = None, partial: torch.Tensor | None = None, tiles_per_wg: int = 0,
                      k_groups: int = 1, epilogue: int = EPI_STORE):
    """fp8 activations (quant_rows_fp8) x fp8 weights (tile_weights_fp8) on the fp8 matrix instruction;
    `dtype`: the 16-bit type of the output and the bias.  M <= 128."""
    _dev(xq, "xq"), _dev(x_scale, "x_scale"), _dev(w_fp8, "w_fp8"), _dev(w_scale, "w_scale")
    if xq.dtype != torch.uint8 or xq.dim() != 2:
        raise ValueError("xq must be uint8 [M, K]")
    M, K = xq.shape
    dt = _DT.get(dtype)
    if dt is None:
        raise ValueError("dtype must be bfloat16 or float16")
    if x_scale.dtype != torch.float32 or x_scale.numel() != M or not x_scale.is_contiguous():
        raise ValueError("x_scale must be contiguous float32 [M]")
    if w_fp8.dtype != torch.uint8 or w_fp8.numel() != int(lib.mi355_tiled_weight_fp8_bytes(N, K)):
        raise ValueError("w_fp8 does not match (N, K)")
    if w_scale.dtype != torch.float32 or w_scale.numel() != (N + 15) // 16 * 16 or not w_scale.is_contiguous():
        raise ValueError("w_scale must be contiguous float32 [ceil(N/16)*16]")
    if tiles_per_wg == 0:
        tiles_per_wg = 216 if M <= 32 or -(-((N + 15) // 16) // 16) * k_groups >= 200 else 210
    y_ptr = p_ptr = y_stride = 0
    if epilogue == EPI_PARTIAL:
        if partial is None:
            partial = torch.empty((k_groups, M, N), dtype=torch.float32, device=xq.device)
        _check_partial(partial, k_groups, M, N)
        p_ptr, ret = partial.data_ptr(), partial
    else:
        width = N // 2 if epilogue == EPI_SILU_MUL else N
        if out is None:
            out = torch.empty((M, width), dtype=dtype, device=xq.device)
        elif out.dtype != dtype or tuple(out.shape) != (M, width):
            raise ValueError(f"out must be [M, {width}] in `dtype`")
        y_ptr, y_stride, ret = out.data_ptr(), _rows(out, "out"), out
    if bias is not None and (bias.dtype != dtype or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous [N] in `dtype`")
    check(lib.mi355_linear_tiled_w8a8(y_ptr, y_stride, p_ptr, xq.data_ptr(), _rows(xq, "xq"), x_scale.data_ptr(),
                                      w_fp8.data_ptr(), w_scale.data_ptr(), bias.data_ptr() if bias is not None else 0,
                                      M, N, K, tiles_per_wg, k_groups, epilogue, dt, _stream()))
    return ret
