"""ctypes binding of libmi355_decode.so (C ABI declared in include/mi355_decode.h).

The library is the product's only compute path for the hot operators; there is
no CPU or framework fallback.  If the shared object is missing or a symbol
cannot be bound, importing this module raises immediately.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int32, c_int64, c_void_p

# PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be the
# one already mapped when libmi355_decode.so resolves its libamdhip64 dependency:
# with two runtimes in one process the second one finds no device.  Hence torch first.
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_DECODE_LIB", os.path.join(_HERE, "libmi355_decode.so"))

ABI_VERSION = 1
BLOCK_SIZE = 16
DTYPE_BF16, DTYPE_F16, DTYPE_F32 = 0, 1, 2


class Mi355Error(RuntimeError):
    """A C-ABI call returned a negative status."""


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C vllm-backend-public_amd/csrc` "
            "(or __graft_entry__.build()); the engine has no fallback path")
    lib = ctypes.CDLL(LIB_PATH)
    P, I32, I64, F = c_void_p, c_int32, c_int64, c_float
    sigs = {
        "mi355_abi_version": (I32, []),
        "mi355_last_error": (c_char_p, []),
        "mi355_target_arch": (c_char_p, []),
        "mi355_bounds_enabled": (I32, []),
        "mi355_bounds_set_ranges": (I32, [P, I32]),
        "mi355_bounds_report": (I32, [P, P, I32]),
        "mi355_rms_norm": (I32, [P, P, P, I64, I32, I64, I64, F, I32, P]),
        "mi355_fused_add_rms_norm": (I32, [P, P, P, P, I64, I32, I64, I64, I64, F, I32, P]),
        "mi355_rope_kv_write": (I32, [P, I64, P, P, I32, P, P, P, I32, I64, I32, I32, I32, I32, P]),
        "mi355_paged_attn_decode_workspace": (I64, [I32, I32, I32, I32]),
        "mi355_paged_attn_decode": (I32, [P, P, I64, P, P, I32, P, I32, P, I32, I32, I32, I32, F, I32, P,
                                          I64, I32, P]),
        "mi355_paged_attn_decode_static": (I32, [P, P, I64, P, P, I32, P, I32, P, I32, I32, I32, I32, F, I32, I32, P]),
        "mi355_paged_attn_prefill": (I32, [P, I64, P, I64, P, P, I32, P, I32, P, P, I32, I32, I32, I32, I32,
                                           F, I32, P]),
        "mi355_silu_and_mul": (I32, [P, P, I64, I64, I32, I32, P]),
        "mi355_sum_slabs": (I32, [P, I64, P, I32, P, I32, I32, I32, P]),
        "mi355_silu_and_mul_slabs": (I32, [P, I64, P, I32, I32, I32, I32, P]),
        "mi355_argmax": (I32, [P, P, I64, I32, I32, I32, P]),
        "mi355_argmax_chunked": (I32, [P, P, P, P, I64, I32, I32, I32, P]),
        "mi355_embed_rms_norm": (I32, [P, P, P, P, I64, P, I64, I32, I64, I64, I64, F, I32, P]),
        "mi355_layer_norm": (I32, [P, P, P, P, I64, I32, I64, I64, F, I32, P]),
        "mi355_fused_add_layer_norm": (I32, [P, P, P, P, P, I64, I32, I64, I64, I64, F, I32, P]),
        "mi355_embed_pos_layer_norm": (I32, [P, P, P, P, I64, I64, P, P, I64, I64, I64, P, P, I64, I32, I64, I64, F,
                                             I32, P]),
        "mi355_relu": (I32, [P, P, I64, I32, P]),
        "mi355_argmax_shard_key": (I32, [P, P, I64, I32, I32, I32, I64, P]),
        "mi355_argmax_key_to_index": (I32, [P, P, I32, P]),
        "mi355_tiled_weight_elems": (I64, [I32, I32]),
        "mi355_tile_weights": (I32, [P, P, I32, I32, I32, P]),
        "mi355_untile_weights": (I32, [P, P, I32, I32, I32, P]),
        "mi355_untile_weights_fp8": (I32, [P, P, P, I32, I32, I32, I32, P]),
        "mi355_linear_tiled": (I32, [P, I64, P, P, I64, P, P, I32, I32, I32, I32, I32, I32, I32, P]),
        "mi355_token_logprobs": (I32, [P, P, P, I64, I32, I32, I32, P, P]),
        "mi355_topk_logits": (I32, [P, P, P, I64, I32, I32, I32, I32, P, P]),
        "mi355_tiled_weight_fp8_bytes": (I64, [I32, I32]),
        "mi355_tile_weights_fp8": (I32, [P, P, P, I32, I32, I32, I32, P]),
        "mi355_linear_tiled_w8": (I32, [P, I64, P, P, I64, P, P, P, I32, I32, I32, I32, I32, I32, I32, P]),
        "mi355_quant_rows_fp8": (I32, [P, I64, P, P, I64, I32, I32, I32, P]),
        "mi355_linear_tiled_w8a8": (I32, [P, I64, P, P, I64, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, P]),
        "mi355_paged_attn_decode_fused": (I32, [P, P, I64, P, I32, I64, P, P, I32, P, P, P, I32, P, I32, P, I32, I32,
                                                I32, I32, F, I32, P, I64, I32, P]),
        "mi355_linear_tiled_fused": (I32, [P, I64, P, P, I64, P, P, I32, I32, I32, I32, I32, I32, P, P, I32, F, P, I64,
                                           P, P, I32, P]),
        "mi355_fused_add_rms_norm_partial": (I32, [P, P, I32, I64, P, P, P, I64, I32, I64, I64, F, I32, P]),
        "mi355_rope_kv_write_partial": (I32, [P, I64, P, I32, I64, P, P, P, I32, P, P, P, I32, I64, I32, I32, I32,
                                              I32, P]),
        "mi355_sample_topk_topp": (I32, [P, P, I64, I32, I32, I32, P, P, P, P, P, P]),
        "mi355_sample_topk_topp_hinted": (I32, [P, P, I64, I32, I32, I32, P, P, P, P, P, I32, P]),
        "mi355_allreduce_oneshot_handle_bytes": (I64, []),
        "mi355_allreduce_oneshot_init": (I32, [P, I32, I32, I64, P]),
        "mi355_allreduce_oneshot_connect": (I32, [P, P]),
        "mi355_allreduce_oneshot_run": (I32, [P, P, I64, I32, P]),
        "mi355_allreduce_oneshot_run_slabs": (I32, [P, P, P, I32, I64, I32, P]),
        "mi355_allreduce_oneshot_run_add_norm": (I32, [P, P, P, I32, P, I64, P, P, I64, I32, I32, F, I32, P]),
        "mi355_allreduce_twoshot_run_add_norm": (I32, [P, P, P, I32, P, I64, P, P, I64, I32, I32, F, I32, P]),
        "mi355_allreduce_oneshot_run_max_keys": (I32, [P, P, P, P, I32, P]),
        "mi355_allreduce_oneshot_status": (I32, [P]),
        "mi355_allreduce_oneshot_status_async": (I32, [P, P, P]),
        "mi355_allreduce_oneshot_destroy": (I32, [P]),
        "mi355_linear_decode": (I32, [P, I64, P, I64, P, P, I32, I32, I32, I32, I32, P]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    got = lib.mi355_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {got}, host expects {ABI_VERSION}")
    return lib


lib = _load()
EXPORTS = ("mi355_abi_version", "mi355_last_error", "mi355_target_arch",
           "mi355_bounds_enabled", "mi355_bounds_set_ranges", "mi355_bounds_report", "mi355_rms_norm",
           "mi355_fused_add_rms_norm", "mi355_rope_kv_write", "mi355_paged_attn_decode_workspace",
           "mi355_paged_attn_decode", "mi355_paged_attn_decode_static", "mi355_paged_attn_prefill", "mi355_silu_and_mul", "mi355_sum_slabs", "mi355_silu_and_mul_slabs", "mi355_argmax",
           "mi355_argmax_chunked", "mi355_embed_rms_norm",
           "mi355_layer_norm", "mi355_fused_add_layer_norm", "mi355_embed_pos_layer_norm", "mi355_relu",
           "mi355_argmax_shard_key", "mi355_argmax_key_to_index",
           "mi355_linear_decode", "mi355_sample_topk_topp", "mi355_sample_topk_topp_hinted",
           "mi355_tiled_weight_elems", "mi355_tile_weights", "mi355_untile_weights", "mi355_untile_weights_fp8",
           "mi355_linear_tiled",
           "mi355_fused_add_rms_norm_partial", "mi355_rope_kv_write_partial",
           "mi355_linear_tiled_fused", "mi355_paged_attn_decode_fused",
           "mi355_tiled_weight_fp8_bytes", "mi355_tile_weights_fp8", "mi355_linear_tiled_w8",
           "mi355_quant_rows_fp8", "mi355_linear_tiled_w8a8",
           "mi355_token_logprobs", "mi355_topk_logits",
           "mi355_allreduce_oneshot_handle_bytes", "mi355_allreduce_oneshot_init", "mi355_allreduce_oneshot_connect",
           "mi355_allreduce_oneshot_run", "mi355_allreduce_oneshot_run_slabs", "mi355_allreduce_oneshot_run_add_norm", "mi355_allreduce_twoshot_run_add_norm", "mi355_allreduce_oneshot_run_max_keys",
           "mi355_allreduce_oneshot_status",
           "mi355_allreduce_oneshot_status_async", "mi355_allreduce_oneshot_destroy")


def check(status: int) -> None:
    if status != 0:
        raise Mi355Error(f"status {status}: {lib.mi355_last_error().decode()}")
