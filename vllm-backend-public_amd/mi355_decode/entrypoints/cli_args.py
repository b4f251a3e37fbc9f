"""Engine command line.

Accepts, without error, every flag the reference harness puts on the engine
container: the manifest templates (benchmark-vllm/vllm_templates.py:50-87,
221-258), the Helm values builder (benchmark-vllm/vllm_manager.py:399-416,
451-456) and the extra keys its config fixtures carry
(benchmark-vllm/configs/test_small_model.yaml:21-30).  The harness repeats some
flags (its defaults are appended after it already hard-coded them): the last
occurrence wins.  Flags this engine has no use for are parsed and recorded,
never rejected, so a manifest written for the stock engine starts this one
unchanged; unknown ``--key [value]`` pairs are tolerated the same way.
"""
from __future__ import annotations

import argparse
import logging
import os
from typing import List, Optional, Tuple

from ..config import EngineArgs

log = logging.getLogger("mi355_decode")

# flags taken for compatibility and ignored (name -> takes a value)
_IGNORED = {
    "--device": True,                         # templates force "cpu"; GPUs are used when present
    "--max-parallel-loading-workers": True,
    "--swap-space": True,
    "--cpu-offload-gb": True,
    "--worker-use-ray": False,
    "--disable-custom-all-reduce": False,
    "--disable-log-stats": False,
    "--disable-log-requests": False,
    "--distributed-executor-backend": True,
    "--kv-cache-dtype": True,
    "--tokenizer": True,
    "--tokenizer-mode": True,
    "--revision": True,
    "--download-dir": True,
    "--uvicorn-log-level": True,
    "--api-key": True,
    "--chat-template": True,
    "--response-role": True,
    "--enable-prefix-caching": False,
    "--enable-chunked-prefill": False,
}


def make_arg_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="mi355_decode.entrypoints.openai.api_server", allow_abbrev=False,
                                description="MI355X-native OpenAI-compatible completion server")
    p.add_argument("--model", required=True)
    p.add_argument("--served-model-name", default=None)
    p.add_argument("--host", default="0.0.0.0")
    p.add_argument("--port", type=int, default=8000)
    p.add_argument("--dtype", default="auto")
    p.add_argument("--max-model-len", type=int, default=None)
    p.add_argument("--max-num-seqs", type=int, default=256)
    p.add_argument("--max-num-batched-tokens", type=int, default=8192)
    p.add_argument("--block-size", type=int, default=16)
    p.add_argument("--gpu-memory-utilization", type=float, default=0.9)
    p.add_argument("--tensor-parallel-size", "-tp", type=int, default=1)
    p.add_argument("--pipeline-parallel-size", "-pp", type=int, default=1)
    p.add_argument("--quantization", "-q", default=None)
    p.add_argument("--load-format", default="auto")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--enforce-eager", action="store_true")
    p.add_argument("--trust-remote-code", action="store_true")
    for flag, has_value in _IGNORED.items():
        if has_value:
            p.add_argument(flag, default=None)
        else:
            p.add_argument(flag, action="store_true")
    return p


def parse_args(argv: Optional[List[str]] = None) -> Tuple[argparse.Namespace, EngineArgs]:
    parser = make_arg_parser()
    ns, unknown = parser.parse_known_args(argv)
    extra = {}
    i = 0
    while i < len(unknown):                      # tolerate unknown "--key [value]" pairs
        tok = unknown[i]
        if tok.startswith("--"):
            if i + 1 < len(unknown) and not unknown[i + 1].startswith("--"):
                extra[tok[2:]] = unknown[i + 1]
                i += 2
            else:
                extra[tok[2:]] = True
                i += 1
        else:
            raise SystemExit(f"unexpected positional argument {tok!r}")
    for flag in _IGNORED:
        v = getattr(ns, flag[2:].replace("-", "_"))
        if v not in (None, False):
            extra[flag[2:]] = v
    if ns.enforce_eager:
        # The harness always passes --enforce-eager (vllm_templates.py:61, models.py:26).  It
        # does not change results, only disables graph replay; it is advisory here
        # (set MI355_ENFORCE_EAGER=1 to really launch kernel by kernel).
        extra["enforce-eager"] = "advisory"
    if extra:
        log.info("accepted and ignored engine flags: %s", extra)
    quant = ns.quantization if ns.quantization not in ("None", "none", "") else None
    args = EngineArgs(
        model=ns.model, served_model_name=ns.served_model_name, dtype=ns.dtype,
        max_model_len=ns.max_model_len, max_num_seqs=ns.max_num_seqs,
        max_num_batched_tokens=ns.max_num_batched_tokens, block_size=ns.block_size,
        gpu_memory_utilization=ns.gpu_memory_utilization, tensor_parallel_size=ns.tensor_parallel_size,
        pipeline_parallel_size=ns.pipeline_parallel_size, quantization=quant, load_format=ns.load_format,
        seed=ns.seed, enforce_eager=os.environ.get("MI355_ENFORCE_EAGER") == "1",
        trust_remote_code=ns.trust_remote_code,
        device=extra.get("device", "auto") or "auto", extra=extra)
    args.validate()
    return ns, args


def sanitize_env(environ=os.environ) -> dict:
    """The harness's manifests export CPU-forcing / NVIDIA-specific variables
    (vllm_templates.py:132-181: CUDA_VISIBLE_DEVICES="", VLLM_TARGET_DEVICE=cpu ...;
    vllm_manager.py:434-437: CUDA_VISIBLE_DEVICES=0, NVIDIA_VISIBLE_DEVICES=all).
    HIP honours CUDA_VISIBLE_DEVICES, so left alone the server would see zero (or
    one) GPU.  Call before the first HIP/torch import: removes the variable unless
    MI355_KEEP_CUDA_VISIBLE_DEVICES=1; the GPUs to use are whatever Kubernetes
    allocated (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES are left untouched)."""
    removed = {}
    if environ.get("MI355_KEEP_CUDA_VISIBLE_DEVICES") != "1" and "CUDA_VISIBLE_DEVICES" in environ:
        removed["CUDA_VISIBLE_DEVICES"] = environ.pop("CUDA_VISIBLE_DEVICES")
    return removed
