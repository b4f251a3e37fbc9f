"""Launch, drive and ingest contracts with the reference harness (CPU only).

* Launch: the container args the reference's own templates produce for the
  BASELINE configs and its YAML fixtures (tests/golden/harness_launch.json,
  captured from benchmark-vllm/vllm_templates.py + models.py) must parse in this
  engine's CLI to the same settings, with no error.
* Helm path / release naming / result keys / template substitution: restated in
  oracle/harness_ref.py from the cited lines, pinned by hand-derived known answers.
* Ingest: payloads built by mi355_decode.harness pass the reference's
  EvaluationPayload validation (golden accept/reject table)."""
import json
import os

import pytest

from conftest import GOLDEN
from mi355_decode.entrypoints.cli_args import parse_args, sanitize_env
from mi355_decode.harness import evaluation_payload, result_pk
from oracle import harness_ref as HR

LAUNCH = json.load(open(os.path.join(GOLDEN, "harness_launch.json")))
RESULTS = json.load(open(os.path.join(GOLDEN, "harness_results.json")))


@pytest.mark.parametrize("case", sorted(LAUNCH["cases"]))
@pytest.mark.parametrize("kind", ["deployment", "statefulset"])
def test_reference_generated_args_parse(case, kind):
    rec = LAUNCH["cases"][case]
    cfg, args = rec["config"], rec[kind]["args"]
    if cfg["pipeline_parallel_size"] != 1:
        with pytest.raises(ValueError, match="pipeline"):
            parse_args(args)
        return
    ns, ea = parse_args(args)
    assert ea.model == cfg["model_name"]
    assert ea.served_model_name == cfg["served_model_name"]
    assert ns.port == cfg["port"] and ns.host == cfg["host"]
    assert ea.max_num_seqs == cfg["max_num_seqs"]
    assert ea.tensor_parallel_size == cfg["tensor_parallel_size"]
    assert ea.max_model_len == cfg["max_model_len"]
    assert ea.block_size == cfg["block_size"]
    assert ea.gpu_memory_utilization == cfg["gpu_memory_utilization"]
    assert ea.quantization == cfg["quantization"]
    assert ea.dtype == cfg["dtype"]
    assert ea.resolved_dtype() in ("bfloat16", "float16")
    # which 16-bit type a request is served in, given the type the checkpoint is stored in
    from mi355_decode.config import EngineArgs as EA, ModelConfig
    rule = {("auto", "bfloat16"): "bfloat16", ("auto", "float16"): "float16", ("auto", "float32"): "float16",
            ("auto", None): "bfloat16", ("float32", "float16"): "float16", ("float32", "bfloat16"): "bfloat16",
            ("float32", "float32"): "bfloat16", ("float32", None): "bfloat16", ("half", "bfloat16"): "float16",
            ("bfloat16", "float16"): "bfloat16", ("float16", None): "float16"}
    for (asked, stored), served in rule.items():
        assert EA(model="x", dtype=asked).resolved_dtype(stored) == served, (asked, stored)
    assert ModelConfig.from_pretrained("facebook/opt-125m").torch_dtype == "float16"
    assert ModelConfig.from_pretrained("llama-3-8b").torch_dtype == "bfloat16"
    assert ModelConfig.from_dict({**{k: v for k, v in __import__("mi355_decode.config", fromlist=["KNOWN_MODELS"])
                                     .KNOWN_MODELS["qwen2-7b"].items() if k != "torch_dtype"},
                                  "torch_dtype": "torch.float32"}).torch_dtype == "float32"
    # the templates always force these; they are accepted and non-binding
    assert ea.extra["device"] == "cpu" and ea.extra["enforce-eager"] == "advisory"
    assert ea.enforce_eager is False
    # the harness repeats its default flags after the hard-coded ones: no error, same meaning
    assert args.count("--enforce-eager") >= 1


def test_golden_args_are_what_the_survey_recorded():
    a = LAUNCH["cases"]["cfg1_opt125m_cpu"]["deployment"]
    assert a["image"] == "vllm-cpu-env:latest" and a["resources"]["requests"] == {"cpu": "4000m", "memory": "16Gi"}
    assert a["args"][-4:] == ["--disable-log-stats", "--disable-log-requests", "--enforce-eager", "--disable-custom-all-reduce"]
    c3 = LAUNCH["cases"]["cfg3_llama3_8b_tp4"]["deployment"]["resources"]["requests"]
    assert c3 == {"cpu": "16000m", "memory": "128Gi", "amd.com/gpu": "4"}
    assert LAUNCH["sanitize_k8s_name"]["meta-llama/Meta-Llama-3-8B"] == "meta-llama-meta-llama-3-8b"
    assert LAUNCH["sanitize_k8s_name"]["3B_model.v2"] == "v3b-model-v2"


def test_cli_duplicates_unknown_flags_and_pp():
    ns, ea = parse_args(["--model", "a", "--model", "b", "--port", "1", "--port", "9000", "--some-new-flag", "7",
                         "--another-switch", "--swap-space", "0", "--cpu-offload-gb", "0", "--load-format", "auto"])
    assert ea.model == "b" and ns.port == 9000
    assert ea.extra["some-new-flag"] == "7" and ea.extra["another-switch"] is True and ea.extra["swap-space"] == "0"
    with pytest.raises(ValueError, match="pipeline"):
        parse_args(["--model", "m", "--pipeline-parallel-size", "2"])
    with pytest.raises(ValueError, match="quantization"):
        parse_args(["--model", "m", "--quantization", "awq"])
    _, ea = parse_args(["--model", "m", "--quantization", "fp8", "--dtype", "float32"])
    assert ea.quantization == "fp8" and ea.resolved_dtype() == "bfloat16"


def test_helm_path_args_parse():
    """Args of the live (Helm) path, restated from vllm_manager.py:399-416,451-456."""
    extra = {"disable-log-stats": True, "disable-log-requests": True, "enforce-eager": True,
             "disable-custom-all-reduce": True}
    gpu = HR.helm_args("/data/local_models/Meta-Llama-3-8B", "llama-3-8b", "0.0.0.0", 8000, None, extra, "amd.com/gpu", 1)
    assert gpu == ["--model", "/data/local_models/Meta-Llama-3-8B", "--served-model-name", "llama-3-8b", "--host",
                   "0.0.0.0", "--port", "8000", "--disable-log-stats", "--disable-log-requests", "--enforce-eager",
                   "--disable-custom-all-reduce"]
    ns, ea = parse_args(gpu)
    assert ea.model.endswith("Meta-Llama-3-8B") and ea.served_model_name == "llama-3-8b" and ns.port == 8000
    cpu = HR.helm_args("facebook/opt-125m", None, "0.0.0.0", 8000, "fp8", extra, "cpu", 0)
    assert cpu[2:4] == ["--served-model-name", "facebook/opt-125m"] and cpu[-3:] == ["--device", "cpu", "--enforce-eager", "--disable-custom-all-reduce"][-3:]
    _, ea = parse_args(cpu)
    assert ea.quantization == "fp8"


def test_env_the_harness_injects_is_sanitised():
    for env_list in (LAUNCH["cases"]["cfg2_llama3_8b_tp1"]["deployment"]["env"], HR.helm_env("amd.com/gpu", 1),
                     HR.helm_env("cpu", 0)):
        env = {e["name"]: e["value"] for e in env_list}
        assert "CUDA_VISIBLE_DEVICES" in env            # the trap: "" or "0" would hide the allocated GPUs
        env["ROCR_VISIBLE_DEVICES"] = "0,1,2,3"
        removed = sanitize_env(env)
        assert "CUDA_VISIBLE_DEVICES" not in env and "CUDA_VISIBLE_DEVICES" in removed
        assert env["ROCR_VISIBLE_DEVICES"] == "0,1,2,3"
    keep = {"CUDA_VISIBLE_DEVICES": "2", "MI355_KEEP_CUDA_VISIBLE_DEVICES": "1"}
    assert sanitize_env(keep) == {} and keep["CUDA_VISIBLE_DEVICES"] == "2"


def test_release_name_pk_and_template_known_answers():
    # benchmark-results/README.md:111
    ka = RESULTS["pk_known_answer"]
    assert HR.result_pk(ka["timestamp"], ka["benchmark_name"], ka["run_id"]) == ka["pk"]
    assert HR.result_pk("t", "b", None) == "t-b-None"
    # benchmark-eval/main.py:57
    assert HR.process_template("{model_name}|{model_name_lower}|{inference_engine_url}", "A-b", "http://x") == "A-b|a-b|http://x"
    assert HR.evaluation_model_name("Llama_3.1_8B") == "Llama-3-1-8B"
    # vllm_manager.py:345-358
    assert HR.release_name("meta-llama/Meta_Llama-3-8B", "amd.com/gpu", 4, "s", "0123456789abcdef") == \
        "vllm-meta-llama-meta-llama-3-8b-01234567-gpu-4"
    import hashlib
    h = hashlib.md5(b"m/x-cpu-0-test-model-cpu").hexdigest()[:8]
    assert HR.release_name("m/x", "cpu", 0, "test-model-cpu") == f"vllm-m-x-{h}-cpu-0"


@pytest.mark.parametrize("case", sorted(RESULTS["cases"]))
def test_ingest_validation_matches_reference(case):
    rec = RESULTS["cases"][case]
    assert HR.payload_accepted(rec["payload"]) == rec["accepted"]
    if rec["accepted"]:
        doc = HR.stored_document(rec["payload"])
        assert doc["pk"].startswith(rec["payload"]["timestamp"] + "-" + rec["payload"]["benchmark_name"])
        if case == "data_json_string":
            assert doc["data"] == {"a": 1}
        if case == "data_plain_string":
            assert doc["data"] == "not json"


def test_bench_client_payload_is_ingestible():
    p = evaluation_payload("mi355-decode-bench", {"decode_tok_s": 6421.5, "ttft_ms_p50": 44.8}, "llama-3-8b")
    assert HR.payload_accepted(p) and isinstance(p["run_id"], str) and p["run_id"]
    assert set(p) == set(RESULTS["cases"]["full"]["payload"])
    assert result_pk(p) == HR.stored_document(p)["pk"]
    json.dumps(p)                                         # wire format


def test_engine_plugin_processor_matches_the_reference_vllm_processor():
    """Mi355Processor fills every key the reference's VLLMProcessor fills with the same values
    (golden: the reference's own processor run on sample chart values), and differs only where
    it must: the image and the GPU resource key."""
    import copy
    import json
    import os
    from mi355_decode.harness_plugin import GPU_RESOURCE, Mi355Processor
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "harness_plugin.json")) as f:
        g = json.load(f)
    proc = Mi355Processor(g["eval_url"], gpus=1)
    for name, case in g["cases"].items():
        out = proc.process_yaml_data(copy.deepcopy(g["base"]), name, "run-1", "exp-7", "model-3", "2")
        assert len(out["global"].pop("timestamp")) == 15
        ref = case["values"]
        assert out["global"] == ref["global"]
        assert proc.get_application_name(name) == case["application"]
        mine, theirs = out["vllm"], ref["vllm"]
        for key in ("vllm", "fullnameOverride", "serviceAccount"):
            assert mine[key] == theirs[key], key
        assert mine["image"]["repository"].endswith("mi355-decode")
        for kind in ("limits", "requests"):
            assert "nvidia.com/gpu" not in mine["resources"][kind] and mine["resources"][kind][GPU_RESOURCE] == 1
        assert mine["resources"]["limits"]["memory"] == "64Gi"            # everything else is left alone
    nov = proc.process_yaml_data({"other": 1}, "a/b", "r", "e", "m", "1")
    nov["global"].pop("timestamp")
    assert nov == g["no_vllm_section"]
    tp = Mi355Processor(g["eval_url"], gpus=4).process_yaml_data(copy.deepcopy(g["base"]), "a/b", "r", "e", "m", "1")
    assert tp["vllm"]["vllm"]["tensor_parallel_size"] == 4 and tp["vllm"]["resources"]["limits"][GPU_RESOURCE] == 4


def test_deploy_values_files_start_this_server():
    """deploy/values-*.yaml (Helm custom values for the harness's chart) name an image, AMD GPUs matching the
    tensor-parallel degree, and `args` this server's CLI takes without a leftover; the Dockerfile copies
    directories that exist and starts the server module."""
    import glob
    import os
    import yaml
    from mi355_decode.entrypoints.cli_args import parse_args
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "deploy", "values-*.yaml")))
    assert len(files) >= 4
    for f in files:
        v = yaml.safe_load(open(f))
        assert set(v) >= {"image", "args", "resources"} and v["image"]["repository"] and v["image"]["tag"]
        ns, ea = parse_args([str(a) for a in v["args"]])
        assert ea.model.startswith("/") and ea.dtype == "bfloat16" and ns.port == 8000
        gpus = v["resources"]["limits"]["amd.com/gpu"]
        assert gpus == v["resources"]["requests"]["amd.com/gpu"] == ea.tensor_parallel_size
        assert ea.max_num_seqs in (16, 32, 64, 128) and ea.max_model_len >= 256
        assert ("fp8" in f) == (ea.quantization == "fp8")
    docker = open(os.path.join(root, "deploy", "Dockerfile")).read()
    for line in docker.splitlines():
        if line.startswith("COPY "):
            assert os.path.isdir(os.path.join(root, line.split()[1])), line
    assert "mi355_decode.entrypoints.openai.api_server" in docker and "HSA_ENABLE_IPC_MODE_LEGACY=0" in docker
