"""Restatement of the pure functions on the harness side of the path.  TEST
INFRASTRUCTURE (see oracle/__init__.py).

The modules that hold these functions cannot be imported in the build container
(ordinary Python errors: ``kubernetes`` / ``motor`` missing, a SyntaxError in
benchmark-results/database.py, pydantic-settings missing), so they are restated
from the cited lines and pinned by hand-derived known answers in
tests/test_harness_contract.py.  Functions whose modules DO import
(vllm_templates.py, models.py) are not restated: their outputs are golden
fixtures (tests/golden/harness_launch.json).
"""
from __future__ import annotations

import hashlib
import json
from typing import Any, Dict, List, Optional


def helm_args(model_name: str, served_model_name: Optional[str], host: str, port: int,
              quantization: Optional[str], additional_args: Optional[Dict[str, Any]],
              gpu_resource_type: str, gpu_resource_count: int) -> List[str]:
    """Engine args of the Helm values (benchmark-vllm/vllm_manager.py:399-416, 451-456)."""
    args = ["--model", model_name, "--served-model-name", served_model_name or model_name,
            "--host", host, "--port", str(port)]
    if quantization:
        args += ["--quantization", quantization]
    for key, value in (additional_args or {}).items():
        if isinstance(value, bool) and value:
            args.append(f"--{key}")
        elif not isinstance(value, bool):
            args += [f"--{key}", str(value)]
    if not (gpu_resource_type != "cpu" and gpu_resource_count > 0):
        args += ["--device", "cpu", "--enforce-eager", "--disable-custom-all-reduce"]
    return args


def helm_env(gpu_resource_type: str, gpu_resource_count: int) -> List[Dict[str, str]]:
    """Env of the Helm values (benchmark-vllm/vllm_manager.py:434-437, 458-461)."""
    if gpu_resource_type != "cpu" and gpu_resource_count > 0:
        return [{"name": "CUDA_VISIBLE_DEVICES", "value": "0"}, {"name": "NVIDIA_VISIBLE_DEVICES", "value": "all"}]
    return [{"name": "VLLM_TARGET_DEVICE", "value": "cpu"}, {"name": "CUDA_VISIBLE_DEVICES", "value": ""}]


def release_name(model_name: str, gpu_resource_type: str, gpu_resource_count: int,
                 served_model_name: Optional[str], values_hash: Optional[str] = None) -> str:
    """benchmark-vllm/vllm_manager.py:339-361."""
    safe = model_name.lower().replace("/", "-").replace("_", "-")
    gpu = "gpu" if gpu_resource_type != "cpu" else "cpu"
    if values_hash:
        ident = values_hash[:8]
    else:
        ident = hashlib.md5(f"{model_name}-{gpu}-{gpu_resource_count}-{served_model_name}".encode()).hexdigest()[:8]
    return f"vllm-{safe}-{ident}-{gpu}-{gpu_resource_count}"


def result_pk(timestamp: str, benchmark_name: str, run_id: Optional[str]) -> str:
    """benchmark-results/results_manager.py:21-22 (run_id None renders as 'None')."""
    return f"{timestamp}-{benchmark_name}-{run_id or None}"


def stored_document(payload: Dict[str, Any]) -> Dict[str, Any]:
    """What save_raw_result / save_standardized_result upsert
    (benchmark-results/results_manager.py:24-40): string data that parses as JSON is stored parsed."""
    data = payload["data"]
    if isinstance(data, str):
        try:
            data = json.loads(data)
        except json.JSONDecodeError:
            pass
    return {"pk": result_pk(payload["timestamp"], payload["benchmark_name"], payload.get("run_id")),
            "benchmark_name": payload["benchmark_name"], "data": data, "model_id": payload["model_id"],
            "tokenizer_id": payload["tokenizer_id"], "source": payload["source"], "timestamp": payload["timestamp"]}


def payload_accepted(payload: Dict[str, Any]) -> bool:
    """EvaluationPayload validation (benchmark-results/models.py:6-19): run_id optional str,
    data anything, the other five fields required strings."""
    for key in ("benchmark_name", "timestamp", "model_id", "tokenizer_id", "source"):
        if not isinstance(payload.get(key), str):
            return False
    return payload.get("run_id") is None or isinstance(payload["run_id"], str)


def process_template(template: str, model_name: str, inference_engine_url: str) -> str:
    """benchmark-eval/main.py:55-58."""
    return (template.replace("{model_name}", model_name)
            .replace("{inference_engine_url}", inference_engine_url)
            .replace("{model_name_lower}", model_name.lower()))


def evaluation_model_name(model_name: str) -> str:
    """benchmark-eval/main.py:113: the name substituted into the job template."""
    return model_name.replace("_", "-").replace(".", "-")
