"""Engine-plugin seam of the harness's model-registry service: a processor for this
engine with the interface of the reference's per-engine YAML processors
(benchmark-mlflow/base_processor.py:6-66; the vLLM one is vllm_processor.py:8-53,
registered by engine name in github_client.py:34-37 and listed in config.py:92-102).

The chart values this engine is deployed with are the vLLM chart's (same section
names, same keys): the processor fills them exactly as VLLMProcessor does and, in
addition, points the release at this engine's image and at AMD GPUs.  Registering it
is one line in the reference (``self.processors['mi355-decode'] = Mi355Processor(url)``)
plus the engine name in ``get_engines_to_process``.
"""
from __future__ import annotations

import logging
from datetime import datetime
from typing import Any, Dict, Optional

try:                                   # inside benchmark-mlflow: derive from its base class
    from base_processor import BaseYAMLProcessor
except ImportError:                    # anywhere else: same contract, restated
    class BaseYAMLProcessor:           # benchmark-mlflow/base_processor.py:6-66
        def __init__(self, benchmark_eval_url: str):
            self.benchmark_eval_url = benchmark_eval_url

        def _get_model_name_k8s(self, model_name: str) -> str:
            return model_name.replace("/", "-").replace(".", "-").replace("_", "-").lower()

        def _update_global_section(self, yaml_data: Dict[str, Any], model_name: str, run_id: str,
                                   experiment_id: str, model_id: str, version: str) -> None:
            g = yaml_data.setdefault("global", {})
            g["experimentId"] = experiment_id
            g["runid"] = run_id
            g["modelid"] = model_id
            g["timestamp"] = datetime.now().strftime("%Y%m%d-%H%M%S")
            g["modelName"] = model_name
            g["modelVersion"] = version

logger = logging.getLogger(__name__)

ENGINE_TYPE = "mi355-decode"
DEFAULT_IMAGE = {"repository": "registry.local/mi355-decode", "tag": "0.1.0", "pullPolicy": "IfNotPresent"}
GPU_RESOURCE = "amd.com/gpu"


class Mi355Processor(BaseYAMLProcessor):
    """YAML processor for the mi355-decode engine (sibling of VLLMProcessor)."""

    def __init__(self, benchmark_eval_url: str, image: Optional[Dict[str, str]] = None, gpus: int = 1):
        super().__init__(benchmark_eval_url)
        self.image = dict(image or DEFAULT_IMAGE)
        self.gpus = int(gpus)

    def process_yaml_data(self, yaml_data: Dict[str, Any], model_name: str, run_id: str, experiment_id: str,
                          model_id: str, version: str) -> Dict[str, Any]:
        self._update_global_section(yaml_data, model_name, run_id, experiment_id, model_id, version)
        k8s = self._get_model_name_k8s(model_name)
        if "vllm" not in yaml_data:
            logger.warning("no vllm section in the values for %s", model_name)
            return yaml_data
        sec = yaml_data["vllm"]
        if "vllm" in sec:                                  # vllm_processor.py:26-29
            sec["vllm"]["evalEndpoint"] = self.benchmark_eval_url
            sec["vllm"]["model"] = f"/data/local_models/{k8s}"
            if self.gpus > 1:
                sec["vllm"]["tensor_parallel_size"] = self.gpus
        sec["fullnameOverride"] = f"vllm-{k8s}"            # vllm_processor.py:32
        if "serviceAccount" in sec:                        # vllm_processor.py:35-36
            sec["serviceAccount"]["name"] = f"vllm-{k8s}-sa"
        # what makes the release this engine: its image, and AMD GPUs instead of nvidia.com/gpu
        sec["image"] = dict(self.image)
        res = sec.setdefault("resources", {})
        for kind in ("limits", "requests"):
            r = res.setdefault(kind, {})
            r.pop("nvidia.com/gpu", None)
            r[GPU_RESOURCE] = self.gpus
        return yaml_data

    def get_application_name(self, model_name: str) -> str:
        return f"vllm-{self._get_model_name_k8s(model_name)}"  # vllm_processor.py:48-53: same release name
