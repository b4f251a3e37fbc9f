#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.  TEST INFRASTRUCTURE.

Run in the build container only:   python oracle/gen_golden.py
(needs ``transformers`` + ``torch`` CPU and, for the harness part, read access
to /root/reference; neither exists on the GPU box, which only sees the
fixtures this script wrote).

Three groups of fixtures:
  ops_hf.npz            per-operator inputs/outputs computed by HF transformers
                        functions (LlamaRMSNorm, LlamaRotaryEmbedding +
                        apply_rotary_pos_emb, eager_attention_forward, the MLP
                        activation, torch.argmax)
  ops_opt_hf.npz        the operators OPT adds: torch.nn.LayerNorm (+ residual add), ReLU, the
                        embedding + OPTLearnedPositionalEmbedding sum
  tiny_<model>.npz      greedy generations + logits of seeded tiny checkpoints
                        (oracle/tiny_models.py) run by HF in fp32 and bf16
  sampling_kept_hf.npz  which tokens HF's TopKLogitsWarper / TopPLogitsWarper keep for the rows of
                        oracle/sampling_cases.py (bit masks)
  harness_launch.json   container args / env / resources / labels produced by
                        the reference's own benchmark-vllm/models.py and
                        vllm_templates.py for the BASELINE configs and its YAML
                        fixtures; harness_results.json: EvaluationPayload
                        accept/reject cases from benchmark-results/models.py
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

from oracle.ops_ref import round_to, to_bits  # noqa: E402
from oracle.tiny_models import TINY, make_prompts, make_weights  # noqa: E402


def f32(t):
    return t.detach().to("cpu").float().numpy()


# --------------------------------------------------------------------------- #
def gen_ops():
    import torch
    from transformers import LlamaConfig
    from transformers.models.llama import modeling_llama as ML

    rng = np.random.Generator(np.random.PCG64(2024))
    out = {}

    def grid(shape, scale=1.0):
        return round_to((rng.standard_normal(shape) * scale).astype(np.float32), "bfloat16")

    # ---- RMSNorm ----------------------------------------------------------
    for tag, (T, H, eps) in {"a": (5, 256, 1e-5), "b": (3, 4096, 1e-5), "c": (2, 448, 1e-6)}.items():
        x, w = grid((T, H), 2.0), round_to(1.0 + 0.1 * rng.standard_normal(H).astype(np.float32), "bfloat16")
        res = grid((T, H))
        out[f"rms_{tag}_x"], out[f"rms_{tag}_w"], out[f"rms_{tag}_res"] = x, w, res
        out[f"rms_{tag}_eps"] = np.float64(eps)
        for dt, tdt in (("bfloat16", torch.bfloat16), ("float32", torch.float32)):
            m = ML.LlamaRMSNorm(H, eps=eps)
            m.weight.data = torch.from_numpy(w).to(tdt)
            xt, rt = torch.from_numpy(x).to(tdt), torch.from_numpy(res).to(tdt)
            out[f"rms_{tag}_y_{dt}"] = f32(m(xt))
            z = rt + xt                                  # decoder-layer residual add
            out[f"rms_{tag}_z_{dt}"] = f32(z)
            out[f"rms_{tag}_yz_{dt}"] = f32(m(z))

    # ---- RoPE -------------------------------------------------------------
    for tag, (d, theta, nh, kvh) in {"d64": (64, 5e5, 4, 2), "d128": (128, 5e5, 8, 2),
                                     "q128": (128, 1e6, 7, 1)}.items():
        cfg = LlamaConfig(hidden_size=d * nh, num_attention_heads=nh, num_key_value_heads=kvh,
                          head_dim=d, rope_theta=theta, max_position_embeddings=4096)
        rot = ML.LlamaRotaryEmbedding(config=cfg)
        pos = np.array([0, 1, 2, 15, 16, 17, 127, 255, 300, 2303, 4095], dtype=np.int64)
        T = len(pos)
        q, k = grid((T, nh, d)), grid((T, kvh, d))
        out[f"rope_{tag}_pos"], out[f"rope_{tag}_q"], out[f"rope_{tag}_k"] = pos, q, k
        out[f"rope_{tag}_theta"] = np.float64(theta)
        for dt, tdt in (("bfloat16", torch.bfloat16), ("float32", torch.float32)):
            qt = torch.from_numpy(q).to(tdt).transpose(0, 1)[None]      # [1, nh, T, d]
            kt = torch.from_numpy(k).to(tdt).transpose(0, 1)[None]
            cos, sin = rot(qt, torch.from_numpy(pos)[None])
            qo, ko = ML.apply_rotary_pos_emb(qt, kt, cos, sin)
            out[f"rope_{tag}_qo_{dt}"] = f32(qo[0].transpose(0, 1))
            out[f"rope_{tag}_ko_{dt}"] = f32(ko[0].transpose(0, 1))
            out[f"rope_{tag}_cos_{dt}"] = f32(cos[0])
            out[f"rope_{tag}_sin_{dt}"] = f32(sin[0])

    # ---- attention (fp32 HF eager path) -------------------------------------
    def hf_attn(q, k, v, causal):
        # q [Lq, nh, d], k/v [Lk, kvh, d]; queries are the LAST Lq positions
        Lq, nh, d = q.shape
        Lk, kvh, _ = k.shape
        mod = SimpleNamespace(num_key_value_groups=nh // kvh, training=False)
        qt = torch.from_numpy(q).transpose(0, 1)[None]
        kt = torch.from_numpy(k).transpose(0, 1)[None]
        vt = torch.from_numpy(v).transpose(0, 1)[None]
        mask = None
        if causal:
            qpos = torch.arange(Lk - Lq, Lk)[:, None]
            kpos = torch.arange(Lk)[None, :]
            mask = torch.where(kpos <= qpos, 0.0, float("-inf"))[None, None]
        o, _ = ML.eager_attention_forward(mod, qt, kt, vt, mask, scaling=d ** -0.5)
        return f32(o[0])                                   # [Lq, nh, d]

    for tag, (nh, kvh, d) in {"g2d64": (4, 2, 64), "g4d128": (8, 2, 128), "g7d64": (7, 1, 64),
                              "g8d128": (8, 1, 128)}.items():
        ctxs = [1, 15, 16, 17, 33, 128, 255] if tag == "g4d128" else [1, 17, 33, 100]
        out[f"attn_{tag}_ctx"] = np.array(ctxs, dtype=np.int32)
        for L in ctxs:
            q, k, v = grid((1, nh, d)), grid((L, kvh, d)), grid((L, kvh, d))
            out[f"attn_{tag}_dec{L}_q"], out[f"attn_{tag}_dec{L}_k"], out[f"attn_{tag}_dec{L}_v"] = q, k, v
            out[f"attn_{tag}_dec{L}_o"] = hf_attn(q, k, v, causal=False)
        for L, Lq in [(1, 1), (5, 5), (16, 16), (37, 37), (50, 13)]:
            q, k, v = grid((Lq, nh, d)), grid((L, kvh, d)), grid((L, kvh, d))
            key = f"attn_{tag}_pre{L}_{Lq}"
            out[key + "_q"], out[key + "_k"], out[key + "_v"] = q, k, v
            out[key + "_o"] = hf_attn(q, k, v, causal=True)

    # ---- SwiGLU gate --------------------------------------------------------
    x = grid((6, 2 * 512), 3.0)
    out["silu_x"] = x
    for dt, tdt in (("bfloat16", torch.bfloat16), ("float32", torch.float32)):
        xt = torch.from_numpy(x).to(tdt)
        out[f"silu_y_{dt}"] = f32(torch.nn.functional.silu(xt[:, :512]) * xt[:, 512:])

    # ---- argmax ---------------------------------------------------------------
    lg = grid((6, 1000), 4.0)
    lg[1, 7] = lg[1, 500] = lg[1].max() + 1.0            # tie -> lowest index
    lg[2, :] = 0.0                                       # all equal -> 0
    lg[3, 999] = 1e30
    lg[4, 321] = np.nan                                  # NaN wins
    out["argmax_logits"] = lg
    out["argmax_idx"] = torch.argmax(torch.from_numpy(lg), dim=-1).numpy().astype(np.int64)

    # inputs drawn on the bf16 grid are stored as raw bf16 words (key + "__bf16")
    packed = {}
    for key, arr in out.items():
        if arr.dtype == np.float32 and arr.size > 64 and not np.isnan(arr).any() \
                and np.array_equal(round_to(arr, "bfloat16"), arr):
            packed[key + "__bf16"] = to_bits(arr, "bfloat16")
        else:
            packed[key] = arr
    np.savez_compressed(os.path.join(GOLD, "ops_hf.npz"), **packed)
    print("ops_hf.npz:", len(out), "arrays")


# --------------------------------------------------------------------------- #
def gen_ops_opt():
    """Operators OPT adds: torch.nn.LayerNorm (what HF modeling_opt.py builds its norms from), the
    residual add before it, ReLU, and the embedding + learned-position sum -> ops_opt_hf.npz."""
    import torch
    from transformers import OPTConfig
    from transformers.models.opt import modeling_opt as MO

    rng = np.random.Generator(np.random.PCG64(2025))
    out = {}

    def grid(shape, scale=1.0):
        return round_to((rng.standard_normal(shape) * scale).astype(np.float32), "bfloat16")

    for tag, (T, H) in {"a": (5, 384), "b": (3, 768), "c": (2, 4096)}.items():
        x, res = grid((T, H), 2.0), grid((T, H))
        x[0] += 3.0                                    # a row with a clear non-zero mean
        x = round_to(x, "bfloat16")
        w = round_to(1.0 + 0.1 * rng.standard_normal(H).astype(np.float32), "bfloat16")
        b = round_to(0.1 * rng.standard_normal(H).astype(np.float32), "bfloat16")
        out[f"ln_{tag}_x"], out[f"ln_{tag}_res"], out[f"ln_{tag}_w"], out[f"ln_{tag}_b"] = x, res, w, b
        for dt, tdt in (("bfloat16", torch.bfloat16), ("float32", torch.float32)):
            m = torch.nn.LayerNorm(H)
            m.weight.data, m.bias.data = torch.from_numpy(w), torch.from_numpy(b)
            m = m.to(tdt)
            xt, rt = torch.from_numpy(x).to(tdt), torch.from_numpy(res).to(tdt)
            out[f"ln_{tag}_y_{dt}"] = f32(m(xt))
            z = rt + xt
            out[f"ln_{tag}_z_{dt}"] = f32(z)
            out[f"ln_{tag}_yz_{dt}"] = f32(m(z))
    x = grid((4, 512), 3.0)
    x[0, 0], x[0, 1] = 0.0, -0.0
    out["relu_x"], out["relu_y"] = x, f32(torch.relu(torch.from_numpy(x).to(torch.bfloat16)))
    # embedding + learned positions through the HF module (offset 2)
    V, H, P = 64, 128, 32
    emb = MO.OPTLearnedPositionalEmbedding(P, H)
    pw, ew = grid((P + 2, H), 0.5), grid((V, H))
    emb.weight.data = torch.from_numpy(pw)
    ids = rng.integers(0, V, size=(1, 9))
    pos = np.arange(9, dtype=np.int64)[None]
    out["pos_table"], out["pos_embed"], out["pos_ids"], out["pos_positions"] = pw, ew, ids[0], pos[0]
    for dt, tdt in (("bfloat16", torch.bfloat16), ("float32", torch.float32)):
        e = torch.from_numpy(ew).to(tdt)[torch.from_numpy(ids)]
        pe = emb.to(tdt)(torch.ones(1, 9, dtype=torch.long), position_ids=torch.from_numpy(pos))
        out[f"pos_sum_{dt}"] = f32((e + pe)[0])
    np.savez_compressed(os.path.join(GOLD, "ops_opt_hf.npz"), **out)
    print("ops_opt_hf.npz:", len(out), "arrays")


# --------------------------------------------------------------------------- #
def gen_tiny(only=None):
    import torch
    import transformers

    G = 24
    lengths = [5, 16, 17, 33]
    for name in TINY:
        if only and name not in only:
            continue
        arch, cfg, w = make_weights(name)
        cfg_cls = {"LlamaForCausalLM": transformers.LlamaConfig,
                   "Qwen2ForCausalLM": transformers.Qwen2Config,
                   "OPTForCausalLM": transformers.OPTConfig}[arch]
        model_cls = getattr(transformers, arch)
        if arch == "OPTForCausalLM":
            hf_cfg = cfg_cls(**cfg, dropout=0.0, attention_dropout=0.0, layerdrop=0.0)
            w = dict(w, **{"lm_head.weight": w["model.decoder.embed_tokens.weight"]})     # tied
        else:
            hf_cfg = cfg_cls(**{k: v for k, v in cfg.items() if k != "attention_bias"},
                             **({"attention_bias": cfg["attention_bias"]} if arch == "LlamaForCausalLM" else {}))
        hf_cfg._attn_implementation = "eager"
        prompts = make_prompts(name, len(lengths), lengths)
        rec = {"lengths": np.array(lengths, dtype=np.int32), "gen_len": np.int32(G)}
        for i, p in enumerate(prompts):
            rec[f"prompt{i}"] = p
        for dt, tdt in (("float32", torch.float32), ("bfloat16", torch.bfloat16)):
            model = model_cls(hf_cfg).eval()
            missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
            model = model.to(tdt)
            toks, first, last, margins = [], [], [], []
            for p in prompts:
                ids = torch.from_numpy(p)[None]
                with torch.no_grad():
                    o = model.generate(ids, attention_mask=torch.ones_like(ids), max_new_tokens=G,
                                       min_new_tokens=G, do_sample=False, output_logits=True,
                                       return_dict_in_generate=True, pad_token_id=0,
                                       eos_token_id=None)
                toks.append(o.sequences[0, len(p):].numpy())
                lg = torch.stack([x[0].float() for x in o.logits])            # [G, V]
                first.append(lg[0].numpy())
                last.append(lg[-1].numpy())
                top2 = torch.topk(lg, 2, dim=-1).values
                margins.append((top2[:, 0] - top2[:, 1]).min().item())
            rec[f"tokens_{dt}"] = np.stack(toks).astype(np.int64)
            rec[f"logits_first_{dt}"] = np.stack(first).astype(np.float32)
            rec[f"logits_last_{dt}"] = np.stack(last).astype(np.float32)
            rec[f"min_margin_{dt}"] = np.array(margins, dtype=np.float32)
            print(name, dt, "min top-2 margins:", [f"{m:.3f}" for m in margins])
        same = (rec["tokens_float32"] == rec["tokens_bfloat16"]).all()
        print(name, "fp32 tokens == bf16 tokens:", bool(same))
        np.savez_compressed(os.path.join(GOLD, f"tiny_{name}.npz"), **rec)


# --------------------------------------------------------------------------- #
BASELINE_CONFIGS = {
    # BASELINE.json configs[0..4] expressed as VLLMConfig payloads
    "cfg1_opt125m_cpu": dict(model_name="facebook/opt-125m"),
    "cfg2_llama3_8b_tp1": dict(model_name="/data/local_models/Meta-Llama-3-8B", dtype="bfloat16",
                               max_num_seqs=32, tensor_parallel_size=1, max_model_len=256,
                               gpu_memory_utilization=0.9, served_model_name="llama-3-8b",
                               gpu_resource_type="amd.com/gpu", gpu_resource_count=1),
    "cfg3_llama3_8b_tp4": dict(model_name="/data/local_models/Meta-Llama-3-8B", dtype="bfloat16",
                               max_num_seqs=128, tensor_parallel_size=4, max_model_len=256,
                               gpu_memory_utilization=0.9, served_model_name="llama-3-8b",
                               gpu_resource_type="amd.com/gpu", gpu_resource_count=4),
    "cfg4_llama3_70b_tp8": dict(model_name="/data/local_models/Meta-Llama-3-70B", dtype="bfloat16",
                                max_num_seqs=16, tensor_parallel_size=8, max_model_len=2304,
                                gpu_memory_utilization=0.9, served_model_name="llama-3-70b",
                                gpu_resource_type="amd.com/gpu", gpu_resource_count=8),
    "cfg5_qwen2_7b_fp8": dict(model_name="/data/local_models/Qwen2-7B", dtype="bfloat16",
                              quantization="fp8", max_num_seqs=64, tensor_parallel_size=1,
                              max_model_len=1024, gpu_memory_utilization=0.9,
                              served_model_name="qwen2-7b", gpu_resource_type="amd.com/gpu",
                              gpu_resource_count=1),
}


def gen_harness():
    import yaml

    sys.path.insert(0, os.path.join(REF, "benchmark-vllm"))
    import models as ref_models                      # benchmark-vllm/models.py
    import vllm_templates as ref_tpl                 # benchmark-vllm/vllm_templates.py

    def container(manifest):
        c = manifest["spec"]["template"]["spec"]["containers"][0]
        return {"args": c["args"], "env": c["env"], "image": c["image"],
                "imagePullPolicy": c["imagePullPolicy"], "resources": c["resources"],
                "ports": c["ports"], "readinessProbe": c["readinessProbe"],
                "livenessProbe": c["livenessProbe"]}

    launch = {"source": "benchmark-vllm/vllm_templates.py create_vllm_deployment_template:212-377, "
                        "create_vllm_statefulset_template:41-210, _get_resource_requirements:451-485, "
                        "_sanitize_k8s_name:5-39; benchmark-vllm/models.py VLLMConfig:5-72",
              "cases": {}}
    payloads = dict(BASELINE_CONFIGS)
    cfg_dir = os.path.join(REF, "benchmark-vllm", "configs")
    for fn in sorted(os.listdir(cfg_dir)):
        with open(os.path.join(cfg_dir, fn)) as f:
            doc = yaml.safe_load(f)
        if isinstance(doc, dict) and "model_name" in doc:
            payloads["ref_fixture_" + fn.replace(".yaml", "")] = doc
    for name, kw in payloads.items():
        cfg = ref_models.VLLMConfig(**kw)
        dep = ref_tpl.create_vllm_deployment_template("vllm-test", cfg, "dep-0001")
        sts = ref_tpl.create_vllm_statefulset_template("vllm-test", cfg, "dep-0001")
        launch["cases"][name] = {
            "payload": kw,
            "config": json.loads(cfg.model_dump_json() if hasattr(cfg, "model_dump_json") else cfg.json()),
            "deployment": container(dep), "statefulset": container(sts),
            "labels": dep["metadata"]["labels"],
            "resource_key": cfg.get_resource_key(),
        }
    names = ["meta-llama/Meta-Llama-3-8B", "3B_model.v2", "facebook/opt-125m", "Qwen/Qwen2-7B",
             "--weird__Name--", "", "9", "a" * 70, "UPPER.case_Model", "x" * 62 + "-y"]
    launch["sanitize_k8s_name"] = {n: ref_tpl._sanitize_k8s_name(n) for n in names}
    a = ref_models.VLLMConfig(**BASELINE_CONFIGS["cfg2_llama3_8b_tp1"])
    b = ref_models.VLLMConfig(**BASELINE_CONFIGS["cfg3_llama3_8b_tp4"])
    c = ref_models.VLLMConfig(model_name="m", gpu_resource_type="nvidia.com/mig-3g.20gb", gpu_resource_count=1)
    d = ref_models.VLLMConfig(model_name="m", gpu_resource_type="nvidia.com/mig-3g.20gb", gpu_resource_count=2)
    e = ref_models.VLLMConfig(model_name="m")
    launch["config_relations"] = {
        "cfg2_matches_cfg2": a.matches_config(ref_models.VLLMConfig(**BASELINE_CONFIGS["cfg2_llama3_8b_tp1"])),
        "cfg2_matches_cfg3": a.matches_config(b),
        "cfg2_conflicts_cfg3": a.conflicts_with_gpu_resources(b),
        "mig_conflicts_mig": c.conflicts_with_gpu_resources(d),
        "cpu_conflicts_gpu": e.conflicts_with_gpu_resources(a),
    }
    with open(os.path.join(GOLD, "harness_launch.json"), "w") as f:
        json.dump(launch, f, indent=1, sort_keys=True)
    print("harness_launch.json:", len(launch["cases"]), "cases")

    # ---- results ingest schema -------------------------------------------------
    spec = importlib.util.spec_from_file_location("ref_results_models",
                                                  os.path.join(REF, "benchmark-results", "models.py"))
    rm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rm)
    base = dict(run_id="run-123", benchmark_name="mi355-decode", data={"decode_tok_s": 1.0},
                timestamp="2024-01-01T12:00:00Z", model_id="llama-3-8b", tokenizer_id="llama-3-8b",
                source="mi355-bench-client")
    cases = {
        "full": base,
        "no_run_id": {k: v for k, v in base.items() if k != "run_id"},
        "data_json_string": {**base, "data": json.dumps({"a": 1})},
        "data_plain_string": {**base, "data": "not json"},
        "missing_benchmark_name": {k: v for k, v in base.items() if k != "benchmark_name"},
        "missing_timestamp": {k: v for k, v in base.items() if k != "timestamp"},
        "timestamp_not_str": {**base, "timestamp": 1704110400},
        "missing_model_id": {k: v for k, v in base.items() if k != "model_id"},
        "missing_tokenizer_id": {k: v for k, v in base.items() if k != "tokenizer_id"},
        "missing_source": {k: v for k, v in base.items() if k != "source"},
    }
    res = {"source": "benchmark-results/models.py EvaluationPayload:6-19 (validated by the reference model "
                     "itself); pk rule benchmark-results/results_manager.py:21-22; README known answer "
                     "benchmark-results/README.md:111", "cases": {}}
    for name, payload in cases.items():
        try:
            obj = rm.EvaluationPayload(**payload)
            dumped = obj.model_dump() if hasattr(obj, "model_dump") else obj.dict()
            res["cases"][name] = {"payload": payload, "accepted": True, "parsed": dumped}
        except Exception as exc:                                   # pydantic ValidationError
            res["cases"][name] = {"payload": payload, "accepted": False, "error": type(exc).__name__}
    res["pk_known_answer"] = {"timestamp": "2024-01-01T12:00:00Z", "benchmark_name": "mmlu",
                              "run_id": "test-run-001", "pk": "2024-01-01T12:00:00Z-mmlu-test-run-001"}
    with open(os.path.join(GOLD, "harness_results.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("harness_results.json:", {k: v["accepted"] for k, v in res["cases"].items()})


def gen_plugin():
    """Outputs of the reference's own VLLMProcessor (benchmark-mlflow/vllm_processor.py) on sample chart
    values: what a sibling processor has to reproduce for the sections both fill."""
    import copy
    sys.path.insert(0, os.path.join(REF, "benchmark-mlflow"))
    from vllm_processor import VLLMProcessor
    proc = VLLMProcessor("http://benchmark-eval:8004/evaluate")
    base = {"global": {"keep": "me"},
            "vllm": {"vllm": {"model": "x", "dtype": "bfloat16", "max_num_seqs": 32}, "fullnameOverride": "x",
                     "serviceAccount": {"create": True, "name": "x"},
                     "resources": {"limits": {"nvidia.com/gpu": 1, "memory": "64Gi"}, "requests": {"nvidia.com/gpu": 1}}}}
    cases = {}
    for name in ("meta-llama/Meta-Llama-3-8B", "Qwen/Qwen2-7B", "org/My_Model.v2"):
        out = proc.process_yaml_data(copy.deepcopy(base), name, "run-1", "exp-7", "model-3", "2")
        out["global"].pop("timestamp")
        cases[name] = {"values": out, "application": proc.get_application_name(name)}
    nov = proc.process_yaml_data({"other": 1}, "a/b", "r", "e", "m", "1")
    nov["global"].pop("timestamp")
    with open(os.path.join(GOLD, "harness_plugin.json"), "w") as f:
        json.dump({"base": base, "eval_url": "http://benchmark-eval:8004/evaluate", "cases": cases,
                   "no_vllm_section": nov}, f, indent=1, sort_keys=True)
    print("harness_plugin.json:", list(cases))


def gen_sampling_kept():
    """Which tokens HF transformers' sampling warpers keep (TemperatureLogitsWarper, then
    TopKLogitsWarper, then TopPLogitsWarper -- the published top-k / nucleus definition, the order
    `generate` applies them in) for the rows of oracle/sampling_cases.py -> sampling_kept_hf.npz: one
    packed bit mask per case.  The numpy oracle's kept set (ops_ref.sample_kept_mask) is compared
    with them here, at generation time, and in tests/test_oracle_golden.py."""
    import torch
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper

    from oracle import ops_ref as R
    from oracle.sampling_cases import CASES, kept_sets_agree, row_of
    out = {}
    for name, (V, kind, t, k, p) in CASES.items():
        row = row_of(name)
        scores = torch.from_numpy(row)[None].clone()
        scores = TemperatureLogitsWarper(float(t))(None, scores)
        if k > 0:
            scores = TopKLogitsWarper(int(k))(None, scores)
        if p < 1:
            scores = TopPLogitsWarper(float(p))(None, scores)
        kept = torch.isfinite(scores[0]).numpy()
        ours = R.sample_kept_mask(row, t, k, p)
        same = kept_sets_agree(row, kept, ours)
        print(f"sampling kept set {name}: HF keeps {int(kept.sum())} of {V}; oracle {'agrees' if same else 'DIFFERS'}")
        assert same, name
        out[name] = np.packbits(kept)
    np.savez_compressed(os.path.join(GOLD, "sampling_kept_hf.npz"), **out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["ops", "tiny", "harness"]
    if "ops" in which:
        gen_ops()
    if "ops_opt" in which:
        gen_ops_opt()
    if "tiny" in which:
        gen_tiny()
    only = [a.split("=", 1)[1] for a in which if a.startswith("tiny=")]   # e.g. tiny=qwen2_d128
    if only:
        gen_tiny(only)
    if "harness" in which:
        gen_harness()
    if "plugin" in which or not sys.argv[1:]:
        gen_plugin()
    if "sampling" in which or not sys.argv[1:]:
        gen_sampling_kept()
