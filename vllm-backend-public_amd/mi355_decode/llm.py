"""Engine construction from EngineArgs and an offline ``LLM.generate`` front end
(same role and names as the offline entry point of the engine the harness
launches; the HTTP server in entrypoints/ drives the same LLMEngine)."""
from __future__ import annotations

import logging
import os
from typing import List, Optional

import torch

from ._lib import BLOCK_SIZE
from .config import EngineArgs, ModelConfig
from .engine import LLMEngine
from .model_runner import ModelRunner
from .sampling_params import SamplingParams
from . import weights as W

log = logging.getLogger("mi355_decode")

TORCH_DTYPE = {"bfloat16": torch.bfloat16, "float16": torch.float16}


def init_distributed(tp_size: int):
    """One process per GPU.  Returns (rank, world, device, group).  MI355_DIST_BACKEND=gloo swaps
    RCCL for gloo (collectives staged through the host, not graph-capturable: with
    --enforce-eager it lets several ranks rehearse on one GPU, which RCCL refuses)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if tp_size > 1:
        if world != tp_size:
            raise RuntimeError(f"--tensor-parallel-size {tp_size} needs {tp_size} processes "
                               f"(WORLD_SIZE={world}); launch with torch.distributed.run")
        backend = os.environ.get("MI355_DIST_BACKEND", "nccl")
        if backend != "nccl":                          # rehearsal: the ranks share whatever cards there are
            local %= max(torch.cuda.device_count(), 1)
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
        if not torch.distributed.is_initialized():
            # a peer that died must surface as an error on the others, not as a collective that
            # never returns (rank 0 talks to the workers at least every 50 ms, idle or not)
            import datetime
            timeout = datetime.timedelta(seconds=int(os.environ.get("MI355_DIST_TIMEOUT_S", "300")))
            if backend == "nccl":
                torch.distributed.init_process_group("nccl", device_id=device, timeout=timeout)
            else:
                torch.distributed.init_process_group(backend, timeout=timeout)
        return rank, world, device, torch.distributed.group.WORLD
    device = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(device)
    return 0, 1, device, None


def plan_num_blocks(args: EngineArgs, cfg: ModelConfig, tp_size: int, num_layers: int, max_model_len: int,
                    elem_size: int, device, extra_reserve: int = 0) -> int:
    """KV pages to allocate: what max_num_seqs sequences of max_model_len tokens
    can ever use, capped by the memory budget (gpu_memory_utilization of the card
    minus what is already resident and an activation reserve).  The harness's
    default utilisation is 0.0 (benchmark-vllm/models.py:8): then only the cap by
    free memory applies."""
    per_block = 2 * num_layers * (cfg.num_key_value_heads // tp_size) * BLOCK_SIZE * cfg.head_dim * elem_size
    blocks_per_seq = (max_model_len + BLOCK_SIZE - 1) // BLOCK_SIZE
    want = args.max_num_seqs * blocks_per_seq + 1
    free, total = torch.cuda.mem_get_info(device)
    reserve = (6 << 30) + int(extra_reserve)
    util = args.gpu_memory_utilization if args.gpu_memory_utilization and args.gpu_memory_utilization > 0 else 0.9
    budget = min(free - reserve, int(total * util) - (total - free) - reserve)
    fit = max(int(budget // per_block), 0)
    n = min(want, fit)
    if n < blocks_per_seq:
        raise RuntimeError(f"not enough GPU memory for the KV cache: {fit} pages fit, one sequence of "
                           f"max_model_len {max_model_len} needs {blocks_per_seq}")
    return n


def build_engine(args: EngineArgs, state_dict: Optional[dict] = None, model_config: Optional[ModelConfig] = None,
                 decode_splits: Optional[int] = None) -> LLMEngine:
    args.validate()
    rank, world, device, group = init_distributed(args.tensor_parallel_size)
    cfg = model_config or ModelConfig.from_pretrained(args.model)
    dtype = TORCH_DTYPE[args.resolved_dtype(cfg.torch_dtype)]
    if (args.dtype or "").lower() in ("float32", "float"):
        log.warning("--dtype float32 requested: the gfx950 kernels compute in 16 bits, serving in %s (checkpoint "
                    "stored in %s)", args.resolved_dtype(cfg.torch_dtype), cfg.torch_dtype or "an unlabelled type")
    if args.block_size != BLOCK_SIZE:
        log.warning("--block-size %d requested: this engine always uses %d-token KV pages",
                    args.block_size, BLOCK_SIZE)
    max_model_len = args.max_model_len or min(cfg.max_position_embeddings, 8192)
    if cfg.family == "opt" and max_model_len > cfg.max_position_embeddings:
        log.warning("--max-model-len %d exceeds the model's %d learned positions: serving %d", max_model_len,
                    cfg.max_position_embeddings, cfg.max_position_embeddings)
        max_model_len = cfg.max_position_embeddings
    nl = args.num_layers_override
    tp = world if args.tensor_parallel_size > 1 else 1
    # MI355_TP_MLP=0 keeps the MLP replicated under tensor parallelism (one all-reduce per layer
    # instead of two, at N times the MLP weight bytes per GPU)
    shard_mlp = tp > 1 and os.environ.get("MI355_TP_MLP", "1") != "0" and cfg.intermediate_size % (32 * tp) == 0
    # MI355_TP_VOCAB=0 keeps a full lm_head on every rank
    shard_vocab = tp > 1 and os.environ.get("MI355_TP_VOCAB", "1") != "0" and cfg.vocab_size % (16 * tp) == 0
    kw = dict(tp_rank=rank, tp_size=tp, num_layers=nl, shard_mlp=shard_mlp, shard_vocab=shard_vocab)
    if state_dict is not None:
        weights = W.from_state_dict(cfg, state_dict, dtype, device, **kw)
    elif args.load_format != "dummy" and W.has_checkpoint(args.model):
        weights = W.from_safetensors(cfg, args.model, dtype, device, **kw)
    else:
        if args.load_format != "dummy":
            log.warning("no *.safetensors under %r: using seeded synthetic weights (nothing is downloaded)",
                        args.model)
        weights = W.synthetic(cfg, dtype, device, seed=args.seed, **kw)
    # the runner keeps a second, fragment-tiled copy of the decode-time matrices
    fp8 = (args.quantization or "").lower() == "fp8"
    if cfg.family == "opt":
        tiled_bytes = 0                               # its decode GEMMs stream the row-major weights
    else:
        tiled_bytes = sum(t.numel() * (1 if fp8 else t.element_size()) for lw in weights.layers
                          for t in (lw.qkv_w, lw.gate_up_w, lw.o_w, lw.down_w))
        tiled_bytes += weights.lm_head.numel() * weights.lm_head.element_size()
    # MI355_SINGLE_WEIGHT_COPY=1: the tiled copy replaces the row-major one (half the weight memory:
    # what lets a model whose weights fill more than half the card be served at all); prefill then
    # rebuilds each layer's row-major matrices into scratch buffers, ~10 % more time to first token
    single = os.environ.get("MI355_SINGLE_WEIGHT_COPY", "0") == "1" and cfg.family != "opt"
    if single and weights.layers:
        lw0 = weights.layers[0]
        scratch = sum(t.numel() * t.element_size() for t in (lw0.qkv_w, lw0.gate_up_w, lw0.o_w, lw0.down_w))
        tiled_bytes = scratch + weights.lm_head.numel() * weights.lm_head.element_size()
    num_blocks = plan_num_blocks(args, cfg, tp, len(weights.layers), max_model_len,
                                 torch.empty(0, dtype=dtype).element_size(), device, extra_reserve=tiled_bytes)
    # only RCCL collectives can be captured into the decode graph
    graphs = not args.enforce_eager and (tp == 1 or torch.distributed.get_backend(group) == "nccl")
    runner_cls = ModelRunner
    if cfg.family == "opt":
        from .opt_runner import OPTRunner
        runner_cls = OPTRunner
    extra = {"single_weight_copy": True} if single else {}
    runner = runner_cls(cfg, weights, dtype, device, num_blocks, args.max_num_seqs, max_model_len,
                         tp_rank=rank, tp_size=tp, tp_group=group, use_graphs=graphs,
                         decode_splits=decode_splits, weight_fp8=(args.quantization or "").lower() == "fp8", **extra)
    # The decode-sized all-reduces of a tensor-parallel step (64-160 per step, each latency-bound) take
    # the hand-written one-shot kernel over peer-mapped buffers when the process group is RCCL
    # (MI355_ONESHOT_ALLREDUCE=0: RCCL for everything; =1: also under another backend, the one-GPU
    # rehearsals).  Every rank maps the peers' buffers and runs a known-answer all-reduce through
    # them first; if any rank fails either, all of them stay on the process group -- counted in
    # runner.oneshot_fallbacks (/metrics, the bench line's degraded.allreduce).
    want = os.environ.get("MI355_ONESHOT_ALLREDUCE", "auto")
    if tp > 1 and (want == "1" or (want != "0" and torch.distributed.get_backend(group) == "nccl")):
        from .collectives import connect_or_none
        rows = max(args.max_num_seqs, 16)
        max_bytes = min(max(1 << 20, rows * cfg.hidden_size * 2), 16 << 20)
        runner.oneshot = connect_or_none(group, max_bytes, device)
        if runner.oneshot is None:
            runner.oneshot_fallbacks += 1
            log.warning("rank %d: one-shot all-reduce unavailable (buffer mapping or self-test failed on some rank); "
                        "using the process group's all-reduce", rank)
        else:
            log.info("rank %d: one-shot all-reduce for messages up to %d bytes (self-test passed on all %d ranks)",
                     rank, runner.oneshot.max_bytes, tp)
    if (not graphs and not args.enforce_eager and tp > 1 and runner.oneshot is not None
            and runner.decode_collectives_one_shot()):
        # no process-group call is left in a decode step (every all-reduce and the greedy pick's key reduction
        # run on the one-shot kernel): the step can be captured whatever the process group's backend is
        runner.use_graphs = graphs = True
    if os.environ.get("MI355_LAZY_TUNE", "0") != "1":
        runner.warm_up()                              # every rank, same order: the choices are broadcast
    log.info("tensor-parallel rank %d of %d on %s: %d layers, MLP %s, lm_head %s, %d KV pages, HIP graphs %s, %s",
             rank, tp, device, len(weights.layers), "sharded" if shard_mlp else "whole",
             "sharded" if shard_vocab else "whole", num_blocks, "on" if graphs else "off",
             "tiled weight copy only" if single else "row-major + tiled weight copies")
    eos = cfg.eos_token_id
    eos = [] if eos is None else (list(eos) if isinstance(eos, (list, tuple)) else [eos])
    return LLMEngine(runner, args.max_num_seqs, max(args.max_num_batched_tokens, 1), max_model_len, eos,
                     seed=args.seed)


class LLM:
    """Offline batch generation over token-id prompts."""

    def __init__(self, engine: LLMEngine):
        self.engine = engine

    @classmethod
    def from_engine_args(cls, args: EngineArgs, **kw) -> "LLM":
        return cls(build_engine(args, **kw))

    def generate(self, prompt_token_ids: List[List[int]], params: SamplingParams) -> List[List[int]]:
        eng = self.engine
        out = {}
        for i, p in enumerate(prompt_token_ids):
            eng.add_request(f"offline-{i}", p, params)
            out[f"offline-{i}"] = []
        while eng.has_unfinished():
            for o in eng.step():
                out[o.request_id].extend(o.new_token_ids)
        return [out[f"offline-{i}"] for i in range(len(prompt_token_ids))]
