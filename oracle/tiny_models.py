"""Deterministic tiny decoder checkpoints for parity tests.  TEST INFRASTRUCTURE
(see oracle/__init__.py).

Weights are drawn from numpy's PCG64 stream (stable across numpy versions),
rounded to the bfloat16 grid, and named with HF parameter names, so the same
tensors can be loaded into ``transformers`` (oracle/gen_golden.py), the numpy
oracle (oracle/model_ref.py) and the HIP engine without being stored on disk.
"""
from __future__ import annotations

import numpy as np

from .ops_ref import round_to

# name -> (HF architecture, config kwargs, weight seed)
TINY = {
    # GQA group 2, head_dim 64
    "llama_d64": ("LlamaForCausalLM", dict(
        vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
        num_attention_heads=4, num_key_value_heads=2, head_dim=64, rope_theta=500000.0,
        rms_norm_eps=1e-5, max_position_embeddings=512, tie_word_embeddings=False,
        attention_bias=False), 11),
    # GQA group 4, head_dim 128 (the Llama-3-8B attention geometry)
    "llama_d128": ("LlamaForCausalLM", dict(
        vocab_size=512, hidden_size=1024, intermediate_size=1536, num_hidden_layers=2,
        num_attention_heads=8, num_key_value_heads=2, head_dim=128, rope_theta=500000.0,
        rms_norm_eps=1e-5, max_position_embeddings=512, tie_word_embeddings=False,
        attention_bias=False), 12),
    # Qwen2: bias on q/k/v, group 7 like Qwen2-7B (28 / 4), theta 1e6, eps 1e-6
    "qwen2_d64": ("Qwen2ForCausalLM", dict(
        vocab_size=384, hidden_size=448, intermediate_size=640, num_hidden_layers=2,
        num_attention_heads=7, num_key_value_heads=1, head_dim=64, rope_theta=1000000.0,
        rms_norm_eps=1e-6, max_position_embeddings=512, tie_word_embeddings=False,
        attention_bias=True), 13),
    # Qwen2 with every matrix wide enough (K a multiple of 512) for the fragment-tiled
    # decode GEMMs: bias on q/k/v through the K-split path, group 2, head_dim 128
    "qwen2_d128": ("Qwen2ForCausalLM", dict(
        vocab_size=448, hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
        num_attention_heads=4, num_key_value_heads=2, head_dim=128, rope_theta=1000000.0,
        rms_norm_eps=1e-6, max_position_embeddings=512, tie_word_embeddings=False,
        attention_bias=True), 14),
    # OPT (pre-LayerNorm, learned positions with offset 2, biased projections, ReLU MLP, tied
    # lm_head): the head geometry of opt-125m (head_dim 64, where the HF and the engine placement of
    # the softmax scale are the same numbers) ...
    "opt_d64": ("OPTForCausalLM", dict(
        vocab_size=320, hidden_size=384, ffn_dim=1024, num_hidden_layers=2, num_attention_heads=6,
        max_position_embeddings=512, word_embed_proj_dim=384, do_layer_norm_before=True,
        activation_function="relu", tie_word_embeddings=True), 15),
    # ... and of opt-6.7b (head_dim 128: the scale 128^-0.5 is not a power of two)
    "opt_d128": ("OPTForCausalLM", dict(
        vocab_size=384, hidden_size=512, ffn_dim=1536, num_hidden_layers=2, num_attention_heads=4,
        max_position_embeddings=512, word_embed_proj_dim=512, do_layer_norm_before=True,
        activation_function="relu", tie_word_embeddings=True), 16),
}


def opt_param_shapes(cfg: dict) -> dict:
    H, F, V = cfg["hidden_size"], cfg["ffn_dim"], cfg["vocab_size"]
    D = "model.decoder."
    shapes = {D + "embed_tokens.weight": (V, H), D + "embed_positions.weight": (cfg["max_position_embeddings"] + 2, H)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"{D}layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            shapes[p + f"self_attn.{n}.weight"] = (H, H)
            shapes[p + f"self_attn.{n}.bias"] = (H,)
        shapes[p + "self_attn_layer_norm.weight"] = shapes[p + "self_attn_layer_norm.bias"] = (H,)
        shapes[p + "fc1.weight"], shapes[p + "fc1.bias"] = (F, H), (F,)
        shapes[p + "fc2.weight"], shapes[p + "fc2.bias"] = (H, F), (H,)
        shapes[p + "final_layer_norm.weight"] = shapes[p + "final_layer_norm.bias"] = (H,)
    shapes[D + "final_layer_norm.weight"] = shapes[D + "final_layer_norm.bias"] = (H,)
    return shapes


def param_shapes(cfg: dict) -> dict:
    if "ffn_dim" in cfg:
        return opt_param_shapes(cfg)
    H, F, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    nh, kvh, d = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    shapes = {"model.embed_tokens.weight": (V, H)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        shapes[p + "self_attn.q_proj.weight"] = (nh * d, H)
        shapes[p + "self_attn.k_proj.weight"] = (kvh * d, H)
        shapes[p + "self_attn.v_proj.weight"] = (kvh * d, H)
        if cfg.get("attention_bias"):
            shapes[p + "self_attn.q_proj.bias"] = (nh * d,)
            shapes[p + "self_attn.k_proj.bias"] = (kvh * d,)
            shapes[p + "self_attn.v_proj.bias"] = (kvh * d,)
        shapes[p + "self_attn.o_proj.weight"] = (H, nh * d)
        shapes[p + "mlp.gate_proj.weight"] = (F, H)
        shapes[p + "mlp.up_proj.weight"] = (F, H)
        shapes[p + "mlp.down_proj.weight"] = (H, F)
        shapes[p + "input_layernorm.weight"] = (H,)
        shapes[p + "post_attention_layernorm.weight"] = (H,)
    shapes["model.norm.weight"] = (H,)
    if not cfg.get("tie_word_embeddings"):
        shapes["lm_head.weight"] = (V, H)
    return shapes


def make_weights(name: str) -> tuple[str, dict, dict]:
    """Returns (architecture, config kwargs, {param name: float32 array on the
    bf16 grid}).  Matrices ~ N(0, (2/sqrt(in))^2) so logits have clear margins;
    norm weights ~ 1 + 0.1 N(0,1); biases ~ 0.1 N(0,1); learned positions ~ 0.5 N(0,1)."""
    arch, cfg, seed = TINY[name]
    rng = np.random.Generator(np.random.PCG64(seed))
    w = {}
    for pname, shape in param_shapes(cfg).items():
        x = rng.standard_normal(shape, dtype=np.float32)
        if pname.endswith(("layernorm.weight", "layer_norm.weight")) or pname == "model.norm.weight":
            x = 1.0 + 0.1 * x
        elif pname.endswith(".bias"):
            x = 0.1 * x
        elif pname.endswith("embed_tokens.weight"):
            x = x
        elif pname.endswith("embed_positions.weight"):
            x = 0.5 * x
        else:
            x = x * (2.0 / np.sqrt(shape[1]))
        w[pname] = round_to(x.astype(np.float32), "bfloat16")
    return arch, dict(cfg), w


def make_prompts(name: str, batch: int, lengths, seed: int = 1):
    """Ragged token-id prompts from the PCG64 stream."""
    _, cfg, _ = TINY[name]
    rng = np.random.Generator(np.random.PCG64(seed))
    return [rng.integers(0, cfg["vocab_size"], size=int(n)).astype(np.int64) for n in lengths][:batch]
