"""The real-model path on CPU: a checkpoint directory as the harness serves them
(/root/reference/benchmark-mlflow/vllm_processor.py:29 `/data/local_models/<name>`) -- sharded
safetensors + index, config.json + generation_config.json, tokenizer.json + tokenizer_config.json
with a chat template -- through the weight loader, the config reader, the tokenizer and the HTTP
routes (engine thread + scheduler real, GPU runner replaced by the double of test_engine_cpu.py)."""
import json

import numpy as np
import pytest
import torch
from fastapi.testclient import TestClient

from mi355_decode import weights as W
from mi355_decode.config import ModelConfig, merge_eos
from mi355_decode.engine import LLMEngine
from mi355_decode.entrypoints.openai.api_server import build_app, render_chat
from mi355_decode.serving import EngineThread
from mi355_decode.tokenizer import load_tokenizer
from real_model_fixture import BOS, CHAT_TEMPLATE, EOS, EOT, build_model_dir
from test_engine_cpu import FakeRunner, reference


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    path = tmp_path_factory.mktemp("tiny-llama-real")
    cfg, w, tk = build_model_dir(str(path), "llama_d128", shards=3, eos_generation=[1, 7])
    return str(path), cfg, w, tk


def _same(a: W.ModelWeights, b: W.ModelWeights):
    assert torch.equal(a.embed, b.embed) and torch.equal(a.final_norm, b.final_norm) and torch.equal(a.lm_head, b.lm_head)
    assert len(a.layers) == len(b.layers) > 0
    for la, lb in zip(a.layers, b.layers):
        for f in ("qkv_w", "o_w", "gate_up_w", "down_w", "input_norm", "post_norm"):
            ta, tb = getattr(la, f), getattr(lb, f)
            assert ta.dtype == tb.dtype == torch.bfloat16 and ta.shape == tb.shape and torch.equal(ta, tb), f


@pytest.mark.parametrize("tp,shard_mlp,shard_vocab", [(1, False, False), (2, False, False), (2, True, True), (2, True, False)])
def test_safetensors_directory_loads_bit_for_bit_like_the_state_dict(model_dir, tp, shard_mlp, shard_vocab):
    path, cfg, w, _ = model_dir
    assert W.has_checkpoint(path)
    mc = ModelConfig.from_pretrained(path)
    for rank in range(tp):
        kw = dict(tp_rank=rank, tp_size=tp, shard_mlp=shard_mlp, shard_vocab=shard_vocab)
        a = W.from_safetensors(mc, path, torch.bfloat16, "cpu", **kw)
        b = W.from_state_dict(mc, {k: torch.from_numpy(v) for k, v in w.items()}, torch.bfloat16, "cpu", **kw)
        _same(a, b)
        # and the shard really is the slice its rank owns
        F, V, H = cfg["intermediate_size"], cfg["vocab_size"], cfg["hidden_size"]
        assert a.layers[0].down_w.shape == (H, F // tp if shard_mlp else F)
        assert a.lm_head.shape == (V // tp if shard_vocab else V, H)
        if shard_vocab:
            full = torch.from_numpy(w["lm_head.weight"]).to(torch.bfloat16)
            assert torch.equal(a.lm_head, full[rank * V // tp:(rank + 1) * V // tp])


def test_single_file_checkpoint_without_index(tmp_path):
    cfg, w, _ = build_model_dir(str(tmp_path), "qwen2_d64", shards=1)
    mc = ModelConfig.from_pretrained(str(tmp_path))
    a = W.from_safetensors(mc, str(tmp_path), torch.bfloat16, "cpu")
    b = W.from_state_dict(mc, {k: torch.from_numpy(v) for k, v in w.items()}, torch.bfloat16, "cpu")
    _same(a, b)
    assert a.layers[0].qkv_b is not None and torch.equal(a.layers[0].qkv_b, b.layers[0].qkv_b)   # Qwen2's q/k/v bias
    with pytest.raises(KeyError, match="not found"):
        W.from_safetensors(ModelConfig.from_dict({**cfg, "num_hidden_layers": 3}), str(tmp_path), torch.bfloat16, "cpu")


def test_opt_checkpoints_load_with_and_without_the_model_prefix(tmp_path):
    """Published OPT checkpoints come in both namings: `model.decoder.*` and `decoder.*` (then with a
    separate tied lm_head.weight, which is not read)."""
    from oracle.tiny_models import make_weights
    arch, cfg, w = make_weights("opt_d64")
    mc = ModelConfig.from_dict({**cfg, "architectures": [arch]})
    a = W.from_state_dict(mc, {k: torch.from_numpy(v) for k, v in w.items()}, torch.bfloat16, "cpu")
    bare = {k[len("model."):]: torch.from_numpy(v) for k, v in w.items()}
    bare["lm_head.weight"] = bare["decoder.embed_tokens.weight"].clone()
    b = W.from_state_dict(mc, bare, torch.bfloat16, "cpu")
    for la, lb in zip(a.layers, b.layers):
        for ta, tb in zip(la.tensors(), lb.tensors()):
            assert torch.equal(ta, tb)
    assert torch.equal(a.embed, b.embed) and torch.equal(a.pos_embed, b.pos_embed) and b.lm_head is b.embed
    with pytest.raises(KeyError, match="decoder.embed_tokens.weight"):
        W.from_state_dict(mc, {}, torch.bfloat16, "cpu")
    # ... and from a directory written without the prefix
    build_model_dir(str(tmp_path), "opt_d64", shards=2, strip_model_prefix=True)
    c = W.from_safetensors(mc, str(tmp_path), torch.bfloat16, "cpu")
    assert torch.equal(c.layers[1].fc2_w, a.layers[1].fc2_w)


def test_config_reads_every_end_of_sequence_id(model_dir, tmp_path):
    path, cfg, _, tk = model_dir
    mc = ModelConfig.from_pretrained(path)
    assert tk.token_to_id(EOS) == 1 and mc.eos_token_id == [1, 7]      # config.json names 1, generation_config.json [1, 7]
    assert merge_eos(None, None) is None and merge_eos(5, None) == 5 and merge_eos(5, [5]) == 5
    assert merge_eos([3, 4], 4, [9]) == [3, 4, 9] and merge_eos(True, 7) == 7
    build_model_dir(str(tmp_path), "llama_d64", shards=1, eos_config=17)           # no generation_config.json
    assert ModelConfig.from_pretrained(str(tmp_path)).eos_token_id == 17


def test_tokenizer_json_and_its_config(model_dir):
    path, cfg, _, tk = model_dir
    t = load_tokenizer(path, cfg["vocab_size"])
    assert not t.is_fallback and t.chat_template == CHAT_TEMPLATE
    assert (t.bos_token, t.eos_token) == (BOS, EOS)
    assert (t.bos_token_id, t.eos_token_id) == (tk.token_to_id(BOS), tk.token_to_id(EOS))
    ids = t.encode("hello world")
    assert ids[0] == t.bos_token_id and ids.count(t.bos_token_id) == 1            # the post-processor's BOS
    assert t.encode("hello world", add_special_tokens=False) == ids[1:]
    assert t.decode(ids) == "hello world"                                          # special tokens are not text
    assert t.encode(f"{BOS}hi{EOT}", add_special_tokens=False)[0] == t.bos_token_id   # spelled-out specials are recognised
    text = "naive café 日本語 tokens"
    assert t.decode(t.encode(text)) == text


def _serve(path, cfg, max_model_len=512):
    th = EngineThread(lambda: LLMEngine(FakeRunner(256, 8, max_model_len), 8, 128, max_model_len))
    tok = load_tokenizer(path, cfg["vocab_size"])
    seen = []
    submit = th.submit

    def spy(rid, prompt, params, sink, **kw):
        seen.append(list(prompt))
        return submit(rid, prompt, params, sink, **kw)

    th.submit = spy
    app = build_app(th, "tiny", tok, 1000, max_model_len, requested_dtype="float32", served_dtype="bfloat16")
    th.start()
    assert th.ready.wait(10)
    return th, TestClient(app), tok, seen


def test_chat_route_renders_the_template_and_keeps_one_bos(model_dir):
    path, cfg, _, _ = model_dir
    th, c, tok, seen = _serve(path, cfg)
    msgs = [{"role": "system", "content": "be brief"}, {"role": "user", "content": "what is the capital of France?"}]
    r = c.post("/v1/chat/completions", json={"model": "tiny", "messages": msgs, "max_tokens": 4, "temperature": 0})
    assert r.status_code == 200, r.text
    want_text = f"{BOS}<|system|>\nbe brief{EOT}\n<|user|>\nwhat is the capital of France?{EOT}\n<|assistant|>\n"
    assert render_chat(msgs, tok) == want_text
    prompt, = seen
    assert prompt == tok.encode(want_text, add_special_tokens=False)
    assert prompt.count(tok.bos_token_id) == 1 and prompt[0] == tok.bos_token_id   # not two: template + post-processor
    assert prompt.count(tok.token_to_id(EOT)) == 2
    body = r.json()
    assert body["object"] == "chat.completion" and body["usage"]["prompt_tokens"] == len(prompt)
    assert body["choices"][0]["message"]["role"] == "assistant"
    # the text route still gets the post-processor's BOS, once
    r = c.post("/v1/completions", json={"prompt": "hello world", "max_tokens": 2, "temperature": 0})
    assert r.status_code == 200 and seen[-1][0] == tok.bos_token_id and seen[-1].count(tok.bos_token_id) == 1
    th.shutdown()


def test_chat_template_is_sandboxed_and_its_errors_are_400(model_dir):
    path, cfg, _, _ = model_dir
    th, c, tok, _ = _serve(path, cfg)
    msgs = [{"role": "user", "content": "hi"}]
    # a template reaching for Python internals: refused by the sandbox, answered 400
    tok.chat_template = "{{ ''.__class__.__mro__[1].__subclasses__() }}"
    r = c.post("/v1/chat/completions", json={"messages": msgs, "max_tokens": 2})
    assert r.status_code == 400 and "chat template" in r.json()["error"]["message"]
    # the helpers published templates call
    tok.chat_template = ("{% if messages[0]['role'] != 'user' %}{{ raise_exception('first message must be from the user') }}"
                         "{% endif %}{{ strftime_now('%Y') }}|{{ messages[0]['content'] }}")
    assert render_chat(list(msgs), tok).endswith("|hi") and len(render_chat(list(msgs), tok)) == 7
    r = c.post("/v1/chat/completions", json={"messages": [{"role": "system", "content": "x"}], "max_tokens": 2})
    assert r.status_code == 400 and "first message must be from the user" in r.json()["error"]["message"]
    tok.chat_template = "{% for %}"                      # a syntax error in the template
    assert c.post("/v1/chat/completions", json={"messages": msgs, "max_tokens": 2}).status_code == 400
    # a template that mutates what it is handed
    tok.chat_template = "{% set _ = messages.append({'role': 'x'}) %}ok"
    assert c.post("/v1/chat/completions", json={"messages": msgs, "max_tokens": 2}).status_code == 400
    th.shutdown()


def test_models_and_metrics_say_what_precision_is_served(model_dir):
    path, cfg, _, _ = model_dir
    th, c, _, _ = _serve(path, cfg)
    card = c.get("/v1/models").json()["data"][0]
    assert card["dtype"] == "bfloat16" and card["requested_dtype"] == "float32" and card["dtype_downgraded"] is True
    m = c.get("/metrics").text
    assert 'mi355_served_dtype{dtype="bfloat16",requested="float32"} 1' in m and "mi355_dtype_downgraded 1" in m
    th.shutdown()


def test_stream_holds_back_a_forming_stop_string_and_hides_the_stop_token(model_dir):
    path, cfg, _, _ = model_dir
    th, c, tok, _ = _serve(path, cfg)
    prompt = [5, 9, 200, 31]
    toks = reference(prompt, 40)
    pieces = [tok.decode(toks[:k + 1])[len(tok.decode(toks[:k])):] for k in range(len(toks))]
    full = "".join(pieces)
    assert full == tok.decode(toks)
    # a stop string that straddles two tokens' text
    k = next(i for i in range(3, len(pieces) - 1) if len(pieces[i]) >= 1 and len(pieces[i + 1]) >= 1
             and full.count(pieces[i][-1:] + pieces[i + 1][:1]) >= 1
             and full.find(pieces[i][-1:] + pieces[i + 1][:1]) == len("".join(pieces[:i + 1])) - 1)
    stop = pieces[k][-1:] + pieces[k + 1][:1]
    cut = full.find(stop)
    with c.stream("POST", "/v1/completions", json={"prompt": prompt, "max_tokens": 40, "temperature": 0, "stop": stop,
                                                   "stream": True}) as r:
        chunks = [json.loads(ln[6:]) for ln in r.iter_lines() if ln and ln != "data: [DONE]"]
    text = "".join(ch["choices"][0]["text"] for ch in chunks)
    assert text == full[:cut] and chunks[-1]["choices"][0]["finish_reason"] == "stop"   # nothing of the stop string was sent
    r = c.post("/v1/completions", json={"prompt": prompt, "max_tokens": 40, "temperature": 0, "stop": [stop]})
    assert r.json()["choices"][0]["text"] == full[:cut]
    # a stop TOKEN ends the sequence, is counted, and is not shown
    r = c.post("/v1/completions", json={"prompt": prompt, "max_tokens": 40, "temperature": 0, "stop_token_ids": [toks[6]]})
    ch = r.json()["choices"][0]
    n = toks.index(toks[6]) + 1
    assert ch["finish_reason"] == "stop" and ch["text"] == tok.decode(toks[:n - 1]) and r.json()["usage"]["completion_tokens"] == n
    th.shutdown()
