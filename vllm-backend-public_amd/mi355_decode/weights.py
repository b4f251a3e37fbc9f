"""Model weights in the layout the kernels and GEMMs consume.

Per layer the q/k/v projections are concatenated into one [ (nh + 2*nkv) * d, H ]
matrix and gate/up into one [2F, H] matrix, all row-major [out, in] in the model
dtype.  Under tensor parallelism (one process per GPU) rank r keeps query heads
[r*nh/P, (r+1)*nh/P), the matching kv heads and the matching COLUMNS of o_proj
(row-parallel: partial sums are all-reduced).  With ``shard_mlp`` it also keeps
only rows [r*F/P, (r+1)*F/P) of gate_proj and up_proj and the matching columns of
down_proj (a second all-reduce per layer); otherwise the MLP is replicated.  With
``shard_vocab`` it keeps rows [r*V/P, (r+1)*V/P) of lm_head (greedy sampling then
max-reduces one key per sequence; full rows are gathered only for random sampling
and log-probabilities).  The embeddings are always replicated.

Sources: a local HF directory with *.safetensors (the harness serves models
from a local path, benchmark-mlflow/vllm_processor.py:29), an in-memory state
dict (tests), or seeded synthetic weights of the right shapes
(``--load-format dummy``; also the fallback when the directory holds no weight
files, because nothing can be downloaded).
"""
from __future__ import annotations

import glob
import json
import os
from dataclasses import dataclass, field
from typing import Callable, Optional

import torch

from .config import ModelConfig


@dataclass
class LayerWeights:
    qkv_w: torch.Tensor
    qkv_b: Optional[torch.Tensor]
    o_w: torch.Tensor
    gate_up_w: torch.Tensor
    down_w: torch.Tensor
    input_norm: torch.Tensor
    post_norm: torch.Tensor


@dataclass
class OPTLayerWeights:
    """One OPT decoder layer: q/k/v concatenated like LayerWeights; every projection has a bias,
    both norms a weight and a bias."""
    qkv_w: torch.Tensor
    qkv_b: torch.Tensor
    o_w: torch.Tensor
    o_b: torch.Tensor
    fc1_w: torch.Tensor
    fc1_b: torch.Tensor
    fc2_w: torch.Tensor
    fc2_b: torch.Tensor
    attn_norm_w: torch.Tensor
    attn_norm_b: torch.Tensor
    mlp_norm_w: torch.Tensor
    mlp_norm_b: torch.Tensor

    @property
    def down_w(self) -> torch.Tensor:          # [hidden, intermediate], the name the runner sizes its buffers by
        return self.fc2_w

    def tensors(self):
        return (self.qkv_w, self.qkv_b, self.o_w, self.o_b, self.fc1_w, self.fc1_b, self.fc2_w, self.fc2_b,
                self.attn_norm_w, self.attn_norm_b, self.mlp_norm_w, self.mlp_norm_b)


@dataclass
class ModelWeights:
    embed: torch.Tensor
    layers: list = field(default_factory=list)
    final_norm: torch.Tensor = None
    lm_head: torch.Tensor = None
    synthetic: bool = False
    pos_embed: Optional[torch.Tensor] = None     # learned positions (OPT): [max_positions + offset, hidden]
    final_norm_b: Optional[torch.Tensor] = None

    def nbytes(self) -> int:
        total = self.embed.numel() * self.embed.element_size()
        if self.lm_head.untyped_storage().data_ptr() != self.embed.untyped_storage().data_ptr():
            total += self.lm_head.numel() * self.lm_head.element_size()
        total += self.final_norm.numel() * self.final_norm.element_size()
        for t in (self.pos_embed, self.final_norm_b):
            if t is not None:
                total += t.numel() * t.element_size()
        for lw in self.layers:
            ts = lw.tensors() if isinstance(lw, OPTLayerWeights) else \
                (lw.qkv_w, lw.qkv_b, lw.o_w, lw.gate_up_w, lw.down_w, lw.input_norm, lw.post_norm)
            for t in ts:
                if t is not None:
                    total += t.numel() * t.element_size()
        return total


def _shard_heads(cfg: ModelConfig, tp_rank: int, tp_size: int):
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    if nh % tp_size or nkv % tp_size:
        raise ValueError(f"tensor_parallel_size {tp_size} must divide num_attention_heads {nh} and "
                         f"num_key_value_heads {nkv}")
    nh_l, nkv_l = nh // tp_size, nkv // tp_size
    return nh_l, nkv_l, slice(tp_rank * nh_l * cfg.head_dim, (tp_rank + 1) * nh_l * cfg.head_dim), \
        slice(tp_rank * nkv_l * cfg.head_dim, (tp_rank + 1) * nkv_l * cfg.head_dim)


def _shard_mlp(cfg: ModelConfig, tp_rank: int, tp_size: int) -> slice:
    F = cfg.intermediate_size
    if F % tp_size:
        raise ValueError(f"tensor_parallel_size {tp_size} must divide intermediate_size {F}")
    return slice(tp_rank * (F // tp_size), (tp_rank + 1) * (F // tp_size))


def _shard_vocab(cfg: ModelConfig, tp_rank: int, tp_size: int) -> slice:
    V = cfg.vocab_size
    if V % (16 * tp_size):
        raise ValueError(f"16 * tensor_parallel_size {tp_size} must divide vocab_size {V} to shard lm_head")
    return slice(tp_rank * (V // tp_size), (tp_rank + 1) * (V // tp_size))


def from_getter(cfg: ModelConfig, get: Callable[[str], torch.Tensor], dtype: torch.dtype, device,
                tp_rank: int = 0, tp_size: int = 1, num_layers: Optional[int] = None,
                shard_mlp: bool = False, shard_vocab: bool = False) -> ModelWeights:
    """``get(name)`` returns the full (unsharded) CPU tensor of an HF parameter."""
    if cfg.family == "opt":
        return _opt_from_getter(cfg, get, dtype, device, tp_size, num_layers)
    _, _, qs, ks = _shard_heads(cfg, tp_rank, tp_size)
    fs = _shard_mlp(cfg, tp_rank, tp_size) if shard_mlp else slice(None)

    def put(t: torch.Tensor) -> torch.Tensor:
        return t.to(dtype).contiguous().to(device)

    L = num_layers if num_layers is not None else cfg.num_hidden_layers
    w = ModelWeights(embed=put(get("model.embed_tokens.weight")))
    for i in range(L):
        p = f"model.layers.{i}."
        q, k, v = (get(p + f"self_attn.{n}_proj.weight") for n in "qkv")
        qkv_w = torch.cat([q[qs], k[ks], v[ks]], dim=0)
        qkv_b = None
        if cfg.attention_bias:
            qb, kb, vb = (get(p + f"self_attn.{n}_proj.bias") for n in "qkv")
            qkv_b = put(torch.cat([qb[qs], kb[ks], vb[ks]], dim=0))
        o = get(p + "self_attn.o_proj.weight")[:, qs]
        gu = torch.cat([get(p + "mlp.gate_proj.weight")[fs], get(p + "mlp.up_proj.weight")[fs]], dim=0)
        w.layers.append(LayerWeights(
            qkv_w=put(qkv_w), qkv_b=qkv_b, o_w=put(o), gate_up_w=put(gu),
            down_w=put(get(p + "mlp.down_proj.weight")[:, fs]),
            input_norm=put(get(p + "input_layernorm.weight")),
            post_norm=put(get(p + "post_attention_layernorm.weight"))))
    w.final_norm = put(get("model.norm.weight"))
    vs = _shard_vocab(cfg, tp_rank, tp_size) if shard_vocab else slice(None)
    w.lm_head = w.embed[vs] if cfg.tie_word_embeddings else put(get("lm_head.weight")[vs])
    return w


def _opt_only_one_gpu(tp_size: int) -> None:
    if tp_size != 1:
        raise ValueError("OPT models run on one GPU (tensor_parallel_size 1): the harness deploys them as its "
                         "single-process plumbing case")


def _opt_from_getter(cfg: ModelConfig, get, dtype, device, tp_size: int, num_layers: Optional[int]) -> ModelWeights:
    """HF OPT parameter names (model.decoder.*; transformers models/opt/modeling_opt.py).  Some
    published OPT checkpoints store the decoder without the `model.` prefix (`decoder.*`) next to a
    separate, tied `lm_head.weight`: both namings load, the lm_head copy is not read."""
    _opt_only_one_gpu(tp_size)
    get_as_named = get

    def get(name):                                       # noqa: F811  model.decoder.X, else decoder.X
        try:
            return get_as_named(name)
        except KeyError:
            if not name.startswith("model."):
                raise
            try:
                return get_as_named(name[len("model."):])
            except KeyError:
                raise KeyError(f"{name} (or {name[len('model.'):]}) not found in the checkpoint") from None

    def put(t: torch.Tensor) -> torch.Tensor:
        return t.to(dtype).contiguous().to(device)

    D = "model.decoder."
    L = num_layers if num_layers is not None else cfg.num_hidden_layers
    w = ModelWeights(embed=put(get(D + "embed_tokens.weight")), pos_embed=put(get(D + "embed_positions.weight")))
    for i in range(L):
        p = f"{D}layers.{i}."
        w.layers.append(OPTLayerWeights(
            qkv_w=put(torch.cat([get(p + f"self_attn.{n}_proj.weight") for n in "qkv"], dim=0)),
            qkv_b=put(torch.cat([get(p + f"self_attn.{n}_proj.bias") for n in "qkv"], dim=0)),
            o_w=put(get(p + "self_attn.out_proj.weight")), o_b=put(get(p + "self_attn.out_proj.bias")),
            fc1_w=put(get(p + "fc1.weight")), fc1_b=put(get(p + "fc1.bias")),
            fc2_w=put(get(p + "fc2.weight")), fc2_b=put(get(p + "fc2.bias")),
            attn_norm_w=put(get(p + "self_attn_layer_norm.weight")), attn_norm_b=put(get(p + "self_attn_layer_norm.bias")),
            mlp_norm_w=put(get(p + "final_layer_norm.weight")), mlp_norm_b=put(get(p + "final_layer_norm.bias"))))
    w.final_norm = put(get(D + "final_layer_norm.weight"))
    w.final_norm_b = put(get(D + "final_layer_norm.bias"))
    w.lm_head = w.embed                                  # tied (OPTForCausalLM ties lm_head to embed_tokens)
    return w


def _opt_synthetic(cfg: ModelConfig, dtype, device, seed: int, tp_size: int, num_layers: Optional[int],
                   std: float) -> ModelWeights:
    _opt_only_one_gpu(tp_size)
    H, F, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    gen = torch.Generator(device=device).manual_seed(seed)

    def mat(*shape):
        return torch.empty(shape, dtype=dtype, device=device).normal_(0.0, std, generator=gen)

    def ones():
        return torch.ones(H, dtype=dtype, device=device)

    L = num_layers if num_layers is not None else cfg.num_hidden_layers
    w = ModelWeights(embed=mat(V, H), synthetic=True,
                     pos_embed=mat(cfg.max_position_embeddings + cfg.position_offset, H))
    for _ in range(L):
        w.layers.append(OPTLayerWeights(
            qkv_w=mat(3 * H, H), qkv_b=mat(3 * H), o_w=mat(H, H), o_b=mat(H), fc1_w=mat(F, H), fc1_b=mat(F),
            fc2_w=mat(H, F), fc2_b=mat(H), attn_norm_w=ones(), attn_norm_b=mat(H), mlp_norm_w=ones(),
            mlp_norm_b=mat(H)))
    w.final_norm, w.final_norm_b = ones(), mat(H)
    w.lm_head = w.embed
    return w


def from_state_dict(cfg, state: dict, dtype, device, **kw) -> ModelWeights:
    def get(name):
        t = state[name]
        return t if isinstance(t, torch.Tensor) else torch.from_numpy(t)
    return from_getter(cfg, get, dtype, device, **kw)


def has_checkpoint(model_dir: str) -> bool:
    return os.path.isdir(model_dir) and bool(glob.glob(os.path.join(model_dir, "*.safetensors")))


def from_safetensors(cfg, model_dir: str, dtype, device, **kw) -> ModelWeights:
    from safetensors import safe_open

    index = os.path.join(model_dir, "model.safetensors.index.json")
    files = sorted(glob.glob(os.path.join(model_dir, "*.safetensors")))
    where = {}
    if os.path.isfile(index):
        with open(index) as f:
            where = {k: os.path.join(model_dir, v) for k, v in json.load(f)["weight_map"].items()}
    handles = {}

    def get(name):
        path = where.get(name)
        if path is None:
            for fpath in files:
                h = handles.setdefault(fpath, safe_open(fpath, framework="pt", device="cpu"))
                if name in h.keys():
                    path = fpath
                    break
            else:
                raise KeyError(f"{name} not found in {model_dir}")
        h = handles.setdefault(path, safe_open(path, framework="pt", device="cpu"))
        return h.get_tensor(name)

    return from_getter(cfg, get, dtype, device, **kw)


def synthetic(cfg: ModelConfig, dtype, device, seed: int = 0, tp_rank: int = 0, tp_size: int = 1,
              num_layers: Optional[int] = None, std: float = 0.02, shard_mlp: bool = False,
              shard_vocab: bool = False) -> ModelWeights:
    """Seeded N(0, std^2) matrices of the real shapes, generated on the device
    (an 8B model is 15 GB: too slow to draw on the host).  Every rank draws the
    replicated tensors from the same stream so they are bit-identical across ranks."""
    if cfg.family == "opt":
        return _opt_synthetic(cfg, dtype, device, seed, tp_size, num_layers, std)
    nh_l, nkv_l, _, _ = _shard_heads(cfg, tp_rank, tp_size)
    H, F, V, d = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.head_dim
    if shard_mlp:
        fs = _shard_mlp(cfg, tp_rank, tp_size)
        F = fs.stop - fs.start
    shared = torch.Generator(device=device).manual_seed(seed)
    local = torch.Generator(device=device).manual_seed(seed * 1000003 + 17 + tp_rank)

    def mat(gen, *shape):
        return torch.empty(shape, dtype=dtype, device=device).normal_(0.0, std, generator=gen)

    L = num_layers if num_layers is not None else cfg.num_hidden_layers
    w = ModelWeights(embed=mat(shared, V, H), synthetic=True)
    for _ in range(L):
        w.layers.append(LayerWeights(
            qkv_w=mat(local, (nh_l + 2 * nkv_l) * d, H),
            qkv_b=mat(local, (nh_l + 2 * nkv_l) * d) if cfg.attention_bias else None,
            o_w=mat(local, H, nh_l * d),
            gate_up_w=mat(local if shard_mlp else shared, 2 * F, H),
            down_w=mat(local if shard_mlp else shared, H, F),
            input_norm=torch.ones(H, dtype=dtype, device=device),
            post_norm=torch.ones(H, dtype=dtype, device=device)))
    w.final_norm = torch.ones(H, dtype=dtype, device=device)
    if shard_vocab:
        vs = _shard_vocab(cfg, tp_rank, tp_size)
        w.lm_head = w.embed[vs] if cfg.tie_word_embeddings else mat(local, vs.stop - vs.start, H)
    else:
        w.lm_head = w.embed if cfg.tie_word_embeddings else mat(shared, V, H)
    return w
