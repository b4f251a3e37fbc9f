"""OPT support without a GPU: the config of the harness's default model (benchmark-vllm/models.py:7
model_name "facebook/opt-125m") and of HF-style OPT config.json files, the variants that are
refused, and the mapping of HF OPT parameter names onto the runner's layer weights."""
import numpy as np
import pytest
import torch

from mi355_decode import weights as W
from mi355_decode.config import ModelConfig
from oracle.tiny_models import make_weights


def test_default_model_of_the_harness_resolves_to_opt_125m():
    cfg = ModelConfig.from_pretrained("facebook/opt-125m")
    assert (cfg.family, cfg.architecture) == ("opt", "OPTForCausalLM")
    assert (cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads) == (768, 3072, 12, 12)
    assert (cfg.head_dim, cfg.num_key_value_heads, cfg.vocab_size) == (64, 12, 50272)
    assert cfg.position_offset == 2 and cfg.max_position_embeddings == 2048 and cfg.tie_word_embeddings
    assert cfg.rms_norm_eps == 1e-5                     # LayerNorm's eps travels in the same field
    # published parameter count of opt-125m: 125 239 296
    assert cfg.num_params() == 125_239_296


def test_llama_configs_keep_their_family():
    assert ModelConfig.from_pretrained("llama-3-8b").family == "llama"


@pytest.mark.parametrize("change,match", [
    (dict(word_embed_proj_dim=512), "word_embed_proj_dim"),
    (dict(do_layer_norm_before=False), "do_layer_norm_before"),
    (dict(activation_function="gelu"), "ReLU"),
    (dict(enable_bias=False), "biases"),
])
def test_unsupported_opt_variants_are_refused_with_the_reason(change, match):
    _, cfg, _ = make_weights("opt_d64")
    with pytest.raises(ValueError, match=match):
        ModelConfig.from_dict({**cfg, "architectures": ["OPTForCausalLM"], **change})


def test_hf_parameter_names_map_onto_layer_weights():
    arch, cfg, w = make_weights("opt_d64")
    mc = ModelConfig.from_dict({**cfg, "architectures": [arch]})
    mw = W.from_state_dict(mc, w, torch.bfloat16, "cpu")
    H, F = cfg["hidden_size"], cfg["ffn_dim"]
    assert len(mw.layers) == 2 and mw.lm_head is mw.embed                      # tied
    assert tuple(mw.pos_embed.shape) == (cfg["max_position_embeddings"] + 2, H)
    lw = mw.layers[1]
    p = "model.decoder.layers.1."
    qkv = np.concatenate([w[p + f"self_attn.{n}_proj.weight"] for n in "qkv"])
    np.testing.assert_array_equal(lw.qkv_w.float().numpy(), qkv)
    np.testing.assert_array_equal(lw.qkv_b.float().numpy(), np.concatenate([w[p + f"self_attn.{n}_proj.bias"] for n in "qkv"]))
    np.testing.assert_array_equal(lw.o_w.float().numpy(), w[p + "self_attn.out_proj.weight"])
    np.testing.assert_array_equal(lw.fc1_b.float().numpy(), w[p + "fc1.bias"])
    assert tuple(lw.fc1_w.shape) == (F, H) and tuple(lw.fc2_w.shape) == (H, F) and lw.down_w is lw.fc2_w
    np.testing.assert_array_equal(lw.attn_norm_w.float().numpy(), w[p + "self_attn_layer_norm.weight"])
    np.testing.assert_array_equal(lw.mlp_norm_b.float().numpy(), w[p + "final_layer_norm.bias"])
    np.testing.assert_array_equal(mw.final_norm_b.float().numpy(), w["model.decoder.final_layer_norm.bias"])
    assert mw.nbytes() == 2 * sum(v.size for v in w.values())


def test_tensor_parallel_opt_is_refused():
    arch, cfg, w = make_weights("opt_d64")
    mc = ModelConfig.from_dict({**cfg, "architectures": [arch]})
    with pytest.raises(ValueError, match="one GPU"):
        W.from_state_dict(mc, w, torch.bfloat16, "cpu", tp_rank=0, tp_size=2)
    with pytest.raises(ValueError, match="one GPU"):
        W.synthetic(mc, torch.bfloat16, "cpu", tp_size=2)


def test_synthetic_opt_weights_have_the_real_shapes():
    mc = ModelConfig.from_pretrained("opt-125m")
    mw = W.synthetic(mc, torch.bfloat16, "cpu", num_layers=1)
    lw = mw.layers[0]
    assert tuple(lw.qkv_w.shape) == (2304, 768) and tuple(lw.fc1_w.shape) == (3072, 768)
    assert tuple(mw.pos_embed.shape) == (2050, 768) and mw.lm_head is mw.embed
