"""Model and engine configuration.

``EngineArgs`` mirrors the engine CLI the reference harness generates
(benchmark-vllm/vllm_templates.py:50-87, vllm_manager.py:399-416): same flag
names and defaults as the engine it launches, so a `VLLMConfig`
(benchmark-vllm/models.py:5-31) maps onto it field by field.
``ModelConfig`` is read from the HF-style ``config.json`` of a local model
directory (weights are served from a local path, benchmark-mlflow/vllm_processor.py:29).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Any, Optional

SUPPORTED_ARCHS = ("LlamaForCausalLM", "Qwen2ForCausalLM", "MistralForCausalLM", "OPTForCausalLM")

# Well-known geometries, used when --model names one of them and no local
# config.json exists (synthetic-weight benchmarking; there is no network).
KNOWN_MODELS = {
    "llama-3-8b": dict(architectures=["LlamaForCausalLM"], vocab_size=128256, hidden_size=4096,
                       intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                       num_key_value_heads=8, rope_theta=500000.0, rms_norm_eps=1e-5,
                       max_position_embeddings=8192, tie_word_embeddings=False, torch_dtype="bfloat16"),
    "llama-3-70b": dict(architectures=["LlamaForCausalLM"], vocab_size=128256, hidden_size=8192,
                        intermediate_size=28672, num_hidden_layers=80, num_attention_heads=64,
                        num_key_value_heads=8, rope_theta=500000.0, rms_norm_eps=1e-5,
                        max_position_embeddings=8192, tie_word_embeddings=False, torch_dtype="bfloat16"),
    # the model the harness launches when a request names none (benchmark-vllm/models.py:7)
    "opt-125m": dict(architectures=["OPTForCausalLM"], vocab_size=50272, hidden_size=768, ffn_dim=3072,
                     num_hidden_layers=12, num_attention_heads=12, max_position_embeddings=2048,
                     word_embed_proj_dim=768, do_layer_norm_before=True, activation_function="relu",
                     bos_token_id=2, eos_token_id=2, torch_dtype="float16"),      # published in float16
    "qwen2-7b": dict(architectures=["Qwen2ForCausalLM"], vocab_size=152064, hidden_size=3584,
                     intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
                     num_key_value_heads=4, rope_theta=1000000.0, rms_norm_eps=1e-6,
                     max_position_embeddings=32768, tie_word_embeddings=False, torch_dtype="bfloat16"),
}


def _known_key(name: str) -> Optional[str]:
    n = os.path.basename(name.rstrip("/")).lower().replace("meta-", "").replace("_", "-")
    for key in KNOWN_MODELS:
        if key in n or key.replace("-", "") in n.replace("-", ""):
            return key
    return None


@dataclass
class ModelConfig:
    architecture: str
    vocab_size: int
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    rope_theta: float
    rope_scaling: Optional[dict]
    rms_norm_eps: float
    max_position_embeddings: int
    tie_word_embeddings: bool
    attention_bias: bool
    bos_token_id: Optional[int] = None
    eos_token_id: Any = None
    # the decoder family: "llama" (RMSNorm, rotary positions, SwiGLU MLP) or "opt" (LayerNorm with
    # bias, learned positions stored at an offset, biased projections, fc1 -> ReLU -> fc2)
    family: str = "llama"
    position_offset: int = 0
    # the type the checkpoint's weights are stored in (config.json's torch_dtype / dtype), if it says
    torch_dtype: Optional[str] = None

    @classmethod
    def _from_opt_dict(cls, d: dict) -> "ModelConfig":
        H, nh = d["hidden_size"], d["num_attention_heads"]
        if d.get("word_embed_proj_dim", H) != H:
            raise ValueError("OPT checkpoints whose embeddings are projected (word_embed_proj_dim != hidden_size, "
                             "opt-350m) are not supported")
        if not d.get("do_layer_norm_before", True):
            raise ValueError("OPT checkpoints that normalise after the sub-layers (do_layer_norm_before false, "
                             "opt-350m) are not supported")
        if d.get("activation_function", "relu") != "relu" or not d.get("enable_bias", True) \
                or not d.get("layer_norm_elementwise_affine", True):
            raise ValueError("OPT variants without biases / affine norms or with another activation than ReLU are "
                             "not supported")
        return cls(architecture="OPTForCausalLM", vocab_size=d["vocab_size"], hidden_size=H,
                   intermediate_size=d["ffn_dim"], num_hidden_layers=d["num_hidden_layers"], num_attention_heads=nh,
                   num_key_value_heads=nh, head_dim=H // nh, rope_theta=0.0, rope_scaling=None,
                   rms_norm_eps=1e-5,                     # torch.nn.LayerNorm's default, which OPT keeps
                   max_position_embeddings=int(d.get("max_position_embeddings", 2048)),
                   tie_word_embeddings=bool(d.get("tie_word_embeddings", True)), attention_bias=True,
                   bos_token_id=d.get("bos_token_id"), eos_token_id=d.get("eos_token_id"), family="opt",
                   position_offset=2)

    @classmethod
    def from_dict(cls, d: dict) -> "ModelConfig":
        archs = d.get("architectures") or ["LlamaForCausalLM"]
        arch = archs[0]
        if arch not in SUPPORTED_ARCHS:
            raise ValueError(f"architecture {arch!r} is not supported (supported: {SUPPORTED_ARCHS})")
        stored = str(d.get("torch_dtype") or d.get("dtype") or "").replace("torch.", "").lower()
        stored = _DTYPES.get(stored) if stored and stored != "auto" else None
        if arch == "OPTForCausalLM":
            cfg = cls._from_opt_dict(d)
            cfg.torch_dtype = stored
            return cfg
        nh = d["num_attention_heads"]
        rope_params = d.get("rope_parameters") or {}
        rope_scaling = d.get("rope_scaling")
        if rope_scaling is None and rope_params.get("rope_type", "default") != "default":
            rope_scaling = rope_params
        if rope_scaling is not None:
            kind = rope_scaling.get("rope_type", rope_scaling.get("type", "default"))
            if kind == "default":
                rope_scaling = None
            elif kind != "llama3":
                raise ValueError(f"rope scaling {kind!r} is not supported (default, llama3)")
        return cls(
            architecture=arch,
            vocab_size=d["vocab_size"],
            hidden_size=d["hidden_size"],
            intermediate_size=d["intermediate_size"],
            num_hidden_layers=d["num_hidden_layers"],
            num_attention_heads=nh,
            num_key_value_heads=d.get("num_key_value_heads") or nh,
            head_dim=d.get("head_dim") or d["hidden_size"] // nh,
            rope_theta=float(d.get("rope_theta") or rope_params.get("rope_theta") or 10000.0),
            rope_scaling=rope_scaling,
            rms_norm_eps=float(d.get("rms_norm_eps", 1e-6)),
            max_position_embeddings=int(d.get("max_position_embeddings", 4096)),
            tie_word_embeddings=bool(d.get("tie_word_embeddings", False)),
            # Qwen2 always carries a bias on q/k/v; Llama only if attention_bias is set
            attention_bias=bool(d.get("attention_bias", arch == "Qwen2ForCausalLM")),
            bos_token_id=d.get("bos_token_id"),
            eos_token_id=d.get("eos_token_id"),
            torch_dtype=stored,
        )

    @classmethod
    def from_pretrained(cls, model: str) -> "ModelConfig":
        """`model` is a local directory with config.json, or the name of a known
        geometry (then weights can only be synthetic)."""
        path = os.path.join(model, "config.json")
        if os.path.isfile(path):
            with open(path) as f:
                cfg = cls.from_dict(json.load(f))
            # generation_config.json may name more end-of-sequence ids than config.json (Llama-3
            # instruct: <|end_of_text|> in one, <|eot_id|> too in the other): generation stops on any
            gen = os.path.join(model, "generation_config.json")
            if os.path.isfile(gen):
                with open(gen) as f:
                    more = json.load(f).get("eos_token_id")
                cfg.eos_token_id = merge_eos(cfg.eos_token_id, more)
            return cfg
        key = _known_key(model)
        if key is not None:
            return cls.from_dict(KNOWN_MODELS[key])
        raise FileNotFoundError(
            f"{model!r}: no config.json there and not a built-in geometry ({sorted(KNOWN_MODELS)}); "
            "models are loaded from a local directory, nothing is downloaded")

    def num_params(self) -> int:
        H, F, V, L = self.hidden_size, self.intermediate_size, self.vocab_size, self.num_hidden_layers
        qkv = (self.num_attention_heads + 2 * self.num_key_value_heads) * self.head_dim * H
        o = self.num_attention_heads * self.head_dim * H
        if self.family == "opt":
            per_layer = qkv + o + 2 * F * H + (qkv // H + 2 * H + F) + 4 * H
            return L * per_layer + V * H + (self.max_position_embeddings + self.position_offset) * H + 2 * H
        per_layer = qkv + o + 3 * F * H + 2 * H
        return L * per_layer + V * H * (1 if self.tie_word_embeddings else 2) + H


def merge_eos(*values) -> Any:
    """Union of end-of-sequence ids given as None, an int or a list, first-seen order; None if empty."""
    out = []
    for v in values:
        for t in ([] if v is None else (list(v) if isinstance(v, (list, tuple)) else [v])):
            if isinstance(t, int) and not isinstance(t, bool) and t not in out:
                out.append(t)
    return None if not out else (out[0] if len(out) == 1 else out)


_DTYPES = {"auto": "bfloat16", "bfloat16": "bfloat16", "bf16": "bfloat16", "half": "float16",
           "float16": "float16", "fp16": "float16", "float": "float32", "float32": "float32"}


@dataclass
class EngineArgs:
    model: str
    served_model_name: Optional[str] = None
    dtype: str = "auto"
    max_model_len: Optional[int] = None
    max_num_seqs: int = 256
    max_num_batched_tokens: int = 8192
    block_size: int = 16
    gpu_memory_utilization: float = 0.9
    tensor_parallel_size: int = 1
    pipeline_parallel_size: int = 1
    quantization: Optional[str] = None
    load_format: str = "auto"
    seed: int = 0
    enforce_eager: bool = False
    trust_remote_code: bool = False
    device: str = "auto"
    num_layers_override: Optional[int] = None      # testing / bounded benchmarks only
    extra: dict = field(default_factory=dict)      # accepted-and-ignored engine flags

    def resolved_dtype(self, checkpoint_dtype: Optional[str] = None) -> str:
        """The 16-bit arithmetic type the kernels run in; `checkpoint_dtype`: the type the weights are
        stored in (ModelConfig.torch_dtype), if known.  An explicit 16-bit --dtype is taken as is.
        `auto` follows the rule of the engine this one replaces: a bfloat16 checkpoint is served in
        bfloat16, a float16 or float32 one in float16.  `float32` (the harness's default,
        benchmark-vllm/models.py:14) cannot be had -- the kernels have no fp32 path --: a 16-bit
        checkpoint is then served in its own type (every stored weight exact; OPT is float16), a
        float32 or unlabelled one in bfloat16 (same exponent range: nothing overflows)."""
        key = (self.dtype or "auto").lower()
        if key not in _DTYPES:
            raise ValueError(f"--dtype {self.dtype!r} not understood")
        if key == "auto":
            return {"bfloat16": "bfloat16", "float16": "float16", "float32": "float16"}.get(checkpoint_dtype, "bfloat16")
        dt = _DTYPES[key]
        if dt == "float32":
            return checkpoint_dtype if checkpoint_dtype in ("bfloat16", "float16") else "bfloat16"
        return dt

    def validate(self) -> None:
        if self.pipeline_parallel_size != 1:
            raise ValueError("--pipeline-parallel-size > 1 is not supported: this engine shards attention "
                             "heads across GPUs (tensor parallel) and does not pipeline layers")
        if self.tensor_parallel_size < 1:
            raise ValueError("--tensor-parallel-size must be >= 1")
        if self.quantization not in (None, "", "none", "fp8"):
            raise ValueError(f"--quantization {self.quantization!r} is not supported (fp8 or none)")
        if self.max_num_seqs < 1:
            raise ValueError("--max-num-seqs must be >= 1")
        self.resolved_dtype()
