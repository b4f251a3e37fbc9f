"""HTTP contract of the server (routes the harness calls), on CPU: the real
FastAPI app and the real engine thread + scheduler, with the GPU model runner
replaced by the test double of test_engine_cpu.py."""
import json
import threading
import time

import pytest
from fastapi.testclient import TestClient

from mi355_decode.engine import LLMEngine
from mi355_decode.entrypoints.openai.api_server import build_app, normalise_prompts
from mi355_decode.serving import EngineThread
from mi355_decode.tokenizer import ByteTokenizer, IncrementalDetokenizer
from test_engine_cpu import FakeRunner, V, reference


def make(delay=0.0, fail=False):
    gate = threading.Event()

    def factory():
        if delay:
            gate.wait(delay)
        if fail:
            raise RuntimeError("cannot load weights")
        return LLMEngine(FakeRunner(256, 8, 512), 8, 128, 512)

    th = EngineThread(factory)
    tok = ByteTokenizer(V)
    app = build_app(th, "test-model", tok, V, 512)
    th.start()
    return th, TestClient(app), gate, tok


def test_health_and_models_503_until_ready_then_200():
    th, c, gate, _ = make(delay=30)
    assert c.get("/health").status_code == 503              # socket is up, engine still loading
    assert c.get("/v1/models").status_code == 503
    gate.set()
    assert th.ready.wait(10)
    assert c.get("/health").status_code == 200 and c.get("/health").content == b""
    r = c.get("/v1/models")
    assert r.status_code == 200 and r.json()["object"] == "list" and r.json()["data"][0]["id"] == "test-model"
    th.shutdown()


def test_fatal_engine_error_turns_health_500():
    th, c, _, _ = make(fail=True)
    th.thread.join(10)
    assert c.get("/health").status_code == 500
    assert c.post("/v1/completions", json={"prompt": [1, 2], "max_tokens": 2}).status_code == 500


def test_completion_token_ids_and_usage():
    th, c, _, tok = make()
    th.ready.wait(10)
    prompt = [5, 6, 7, 8]
    r = c.post("/v1/completions", json={"model": "test-model", "prompt": prompt, "max_tokens": 6, "temperature": 0})
    assert r.status_code == 200
    body = r.json()
    assert body["object"] == "text_completion" and body["model"] == "test-model" and body["id"].startswith("cmpl-")
    ch = body["choices"][0]
    assert ch["index"] == 0 and ch["finish_reason"] == "length" and ch["logprobs"] is None
    assert ch["text"] == tok.decode(reference(prompt, 6))
    assert body["usage"] == {"prompt_tokens": 4, "completion_tokens": 6, "total_tokens": 10}
    th.shutdown()


def test_batch_prompts_text_and_echo():
    th, c, _, tok = make()
    th.ready.wait(10)
    r = c.post("/v1/completions", json={"prompt": ["hello", "wor"], "max_tokens": 3, "temperature": 0, "echo": True})
    assert r.status_code == 200
    chs = sorted(r.json()["choices"], key=lambda x: x["index"])
    for ch, text in zip(chs, ["hello", "wor"]):
        assert ch["text"] == text + tok.decode(reference(tok.encode(text), 3))
    r = c.post("/v1/completions", json={"prompt": [[1, 2, 3], [4, 5]], "max_tokens": 2, "temperature": 0})
    assert len(r.json()["choices"]) == 2 and r.json()["usage"]["prompt_tokens"] == 5
    th.shutdown()


def test_streaming_sse():
    th, c, _, tok = make()
    th.ready.wait(10)
    prompt = [9, 8, 7]
    with c.stream("POST", "/v1/completions", json={"prompt": prompt, "max_tokens": 5, "temperature": 0, "stream": True}) as r:
        assert r.status_code == 200 and r.headers["content-type"].startswith("text/event-stream")
        lines = [ln for ln in r.iter_lines() if ln]
    assert lines[-1] == "data: [DONE]"
    chunks = [json.loads(ln[6:]) for ln in lines[:-1]]
    assert all(ch["object"] == "text_completion" for ch in chunks)
    assert "".join(ch["choices"][0]["text"] for ch in chunks) == tok.decode(reference(prompt, 5))
    assert chunks[-1]["choices"][0]["finish_reason"] == "length"
    assert all(ch["choices"][0]["finish_reason"] is None for ch in chunks[:-1])
    th.shutdown()


def test_stop_string_cuts_text_and_aborts():
    th, c, _, tok = make()
    th.ready.wait(10)
    prompt = [3, 1, 4, 1, 5]
    full = tok.decode(reference(prompt, 40))
    stop = full[5:7]
    r = c.post("/v1/completions", json={"prompt": prompt, "max_tokens": 40, "temperature": 0, "stop": stop})
    ch = r.json()["choices"][0]
    assert ch["finish_reason"] == "stop" and ch["text"] == full[: full.find(stop)]
    deadline = time.time() + 5
    while th.engine.has_unfinished() and time.time() < deadline:
        time.sleep(0.01)
    assert not th.engine.has_unfinished()                   # the aborted sequence released its pages
    th.shutdown()


def test_errors_are_openai_style_json():
    th, c, _, _ = make()
    th.ready.wait(10)
    for body, code in [({"max_tokens": 3}, 400), ({"prompt": [1], "model": "other"}, 404),
                       ({"prompt": [1], "n": 0}, 400), ({"prompt": [1], "n": 2, "best_of": 4}, 400),
                       ({"prompt": [[1], [2]], "n": 200}, 400), ({"prompt": [V + 5]}, 400), ({"prompt": []}, 400),
                       ({"prompt": list(range(600))}, 400), ({"prompt": [1], "max_tokens": 0}, 400),
                       ({"prompt": [1], "temperature": -1}, 400), ({"prompt": [1], "top_p": 0}, 400),
                       ({"prompt": [1], "logprobs": 50}, 400)]:
        r = c.post("/v1/completions", json=body)
        assert r.status_code == code, (body, r.text)
        err = r.json()["error"]
        assert set(err) >= {"message", "type", "code"}
    assert c.post("/v1/completions", content=b"{not json").status_code == 400
    th.shutdown()


def test_sampling_parameters_are_accepted():
    th, c, _, _ = make()
    th.ready.wait(10)
    r = c.post("/v1/completions", json={"prompt": [1, 2], "max_tokens": 3, "temperature": 0.8, "top_k": 20,
                                        "top_p": 0.9, "seed": 11})
    assert r.status_code == 200 and r.json()["usage"]["completion_tokens"] == 3
    s = th.engine.runner.last_sampling
    assert s is not None and s["seeds"].tolist() == [11] and s["top_k"].tolist() == [20]
    th.shutdown()


def test_n_completions_per_prompt():
    """n > 1: n sequences per prompt, choices numbered prompt-major, the prompt counted once, copy j
    drawing from seed + j; streamed chunks carry the choice index."""
    th, c, _, tok = make()
    th.ready.wait(10)
    r = c.post("/v1/completions", json={"prompt": [[5, 6, 7], [8, 9]], "max_tokens": 4, "temperature": 0, "n": 3})
    assert r.status_code == 200, r.text
    chs = sorted(r.json()["choices"], key=lambda x: x["index"])
    assert [ch["index"] for ch in chs] == list(range(6))
    for ch in chs:
        assert ch["text"] == tok.decode(reference([[5, 6, 7], [8, 9]][ch["index"] // 3], 4))
    assert r.json()["usage"] == {"prompt_tokens": 5, "completion_tokens": 24, "total_tokens": 29}
    seeds = []
    orig = th.engine.add_request

    def spy(rid, prompt, params, **kw):
        seeds.append(params.seed)
        return orig(rid, prompt, params, **kw)

    th.engine.add_request = spy
    r = c.post("/v1/completions", json={"prompt": [1, 2], "max_tokens": 2, "temperature": 0.7, "seed": 40, "n": 3,
                                        "best_of": 3})
    assert r.status_code == 200 and len(r.json()["choices"]) == 3 and sorted(seeds) == [40, 41, 42]
    with c.stream("POST", "/v1/completions", json={"prompt": [9, 8, 7], "max_tokens": 3, "temperature": 0, "n": 2,
                                                   "stream": True}) as r:
        chunks = [json.loads(ln[6:]) for ln in r.iter_lines() if ln.startswith("data: {")]
    for i in (0, 1):
        mine = [ch["choices"][0] for ch in chunks if ch["choices"][0]["index"] == i]
        assert "".join(m["text"] for m in mine) == tok.decode(reference([9, 8, 7], 3)) and mine[-1]["finish_reason"] == "length"
    th.shutdown()


def test_any_top_k_is_accepted():
    """top_k is not capped: the sampler keeps the set the published definition names, whatever its size."""
    th, c, _, _ = make()
    th.ready.wait(10)
    r = c.post("/v1/completions", json={"prompt": [1, 2], "max_tokens": 3, "temperature": 0.8, "top_k": 5000, "top_p": 0.95})
    assert r.status_code == 200 and r.json()["usage"]["completion_tokens"] == 3
    th.shutdown()


def test_chat_completions_share_the_generation_path():
    th, c, _, tok = make()
    th.ready.wait(10)
    msgs = [{"role": "system", "content": "be brief"}, {"role": "user", "content": [{"type": "text", "text": "hi"}]}]
    prompt = "<|system|>\nbe brief\n\n<|user|>\nhi\n\n<|assistant|>\n"
    r = c.post("/v1/chat/completions", json={"model": "test-model", "messages": msgs, "max_tokens": 5, "temperature": 0})
    assert r.status_code == 200, r.text
    body = r.json()
    assert body["object"] == "chat.completion" and body["id"].startswith("chatcmpl-")
    ch = body["choices"][0]
    assert ch["message"] == {"role": "assistant", "content": tok.decode(reference(tok.encode(prompt), 5))}
    assert ch["finish_reason"] == "length" and body["usage"]["prompt_tokens"] == len(tok.encode(prompt))
    # the same text through the plain route
    r2 = c.post("/v1/completions", json={"prompt": prompt, "max_tokens": 5, "temperature": 0})
    assert r2.json()["choices"][0]["text"] == ch["message"]["content"]
    # streaming: a role first, content deltas, a finish reason, [DONE]
    with c.stream("POST", "/v1/chat/completions", json={"messages": msgs, "max_tokens": 4, "temperature": 0, "stream": True}) as r:
        lines = [ln for ln in r.iter_lines() if ln.startswith("data: ")]
    assert lines[-1] == "data: [DONE]"
    chunks = [json.loads(ln[6:]) for ln in lines[:-1]]
    assert all(ch_["object"] == "chat.completion.chunk" for ch_ in chunks)
    assert chunks[0]["choices"][0]["delta"].get("role") == "assistant"
    assert "".join(ch_["choices"][0]["delta"].get("content", "") for ch_ in chunks) == tok.decode(reference(tok.encode(prompt), 4))
    assert chunks[-1]["choices"][0]["finish_reason"] == "length"
    assert c.post("/v1/chat/completions", json={"messages": []}).status_code == 400
    assert c.post("/v1/chat/completions", json={"messages": [{"content": "x"}]}).status_code == 400
    th.shutdown()


def test_many_concurrent_requests_and_health_stays_fast():
    th, c, _, tok = make()
    th.ready.wait(10)
    results = {}

    def one(i):
        p = [i + 1, i + 2, i + 3]
        r = c.post("/v1/completions", json={"prompt": p, "max_tokens": 8, "temperature": 0})
        results[i] = (r.status_code, r.json()["choices"][0]["text"], tok.decode(reference(p, 8)))

    ts = [threading.Thread(target=one, args=(i,)) for i in range(24)]
    [t.start() for t in ts]
    t0 = time.time()
    assert c.get("/health").status_code == 200 and time.time() - t0 < 5.0     # kubelet probe timeout is 5 s
    [t.join(60) for t in ts]
    assert len(results) == 24 and all(s == 200 and a == b for s, a, b in results.values())
    m = c.get("/metrics").text
    assert "mi355_generation_tokens_total" in m and "mi355_kv_blocks_free" in m
    th.shutdown()


def test_prompt_normalisation_and_detokenizer():
    tok = ByteTokenizer(1000)
    assert normalise_prompts("ab", tok, 1000) == [tok.encode("ab")]
    assert normalise_prompts(["a", "b"], tok, 1000) == [tok.encode("a"), tok.encode("b")]
    assert normalise_prompts([1, 2, 3], tok, 1000) == [[1, 2, 3]]
    assert normalise_prompts([[1], [2, 3]], tok, 1000) == [[1], [2, 3]]
    for bad in ([], [[]], [1, "a"], 5, [True]):
        with pytest.raises(ValueError):
            normalise_prompts(bad, tok, 1000)
    d = IncrementalDetokenizer(tok)
    ids = tok.encode("héllo")                               # multi-byte char split over two tokens
    out = "".join(d.push([i]) for i in ids)
    assert out == "héllo"
    # the windowed decode emits exactly the text of the whole sequence, whatever the chunking
    import random
    rnd = random.Random(0)
    for _ in range(100):
        text = "".join(rnd.choice(["a", "é", "漢", "😀", " ", "z", "\n", "ß"]) for _ in range(rnd.randint(1, 60)))
        ids = tok.encode(text)
        d, out, i = IncrementalDetokenizer(tok), "", 0
        while i < len(ids):
            n = rnd.randint(1, 3)
            out += d.push(ids[i:i + n], final=i + n >= len(ids))
            i += n
        assert out == text
    for _ in range(100):                                      # arbitrary bytes: the final text is the full decode
        ids = [rnd.randrange(3, 259) for _ in range(rnd.randint(1, 40))]
        d = IncrementalDetokenizer(tok)
        out = "".join(d.push([t], final=k == len(ids) - 1) for k, t in enumerate(ids))
        assert out == tok.decode(ids)


def test_echo_with_logprobs_scores_the_prompt():
    """`echo` + `logprobs` (log-likelihood scoring as evaluation harnesses do it): the logprobs object
    starts with one entry per prompt token -- null for the first, then the token's log-probability given
    the tokens before it, its own entry always present in top_logprobs -- followed by the generated ones."""
    th, c, _, tok = make()
    th.ready.wait(10)
    prompt = [65, 66, 67, 68]
    r = c.post("/v1/completions", json={"prompt": prompt, "max_tokens": 1, "temperature": 0, "logprobs": 1, "echo": True})
    assert r.status_code == 200, r.text
    ch = r.json()["choices"][0]
    lp = ch["logprobs"]
    n = len(prompt) + 1
    assert len(lp["tokens"]) == len(lp["token_logprobs"]) == len(lp["top_logprobs"]) == len(lp["text_offset"]) == n
    assert lp["token_logprobs"][0] is None and lp["top_logprobs"][0] is None
    assert all(v is not None and v <= 0 for v in lp["token_logprobs"][1:])
    for tok, top, v in zip(lp["tokens"][1:len(prompt)], lp["top_logprobs"][1:len(prompt)], lp["token_logprobs"][1:]):
        assert top[tok] == v and 1 <= len(top) <= 2
    assert "".join(lp["tokens"]) == ch["text"] and lp["text_offset"][0] == 0
    assert lp["text_offset"] == [sum(len(t) for t in lp["tokens"][:j]) for j in range(n)]
    # logprobs: 0 keeps the token scores and drops the alternatives
    r0 = c.post("/v1/completions", json={"prompt": prompt, "max_tokens": 1, "temperature": 0, "logprobs": 0, "echo": True})
    lp0 = r0.json()["choices"][0]["logprobs"]
    assert lp0["top_logprobs"] == [None] * n and lp0["token_logprobs"][1:] == lp["token_logprobs"][1:]
    th.shutdown()


def test_logprobs_object_of_a_completion():
    """`logprobs: k`: OpenAI text-completion logprobs object, one entry per generated token,
    top_logprobs of width k; streamed chunks carry the step's entries."""
    th, c, _, tok = make()
    th.ready.wait(10)
    r = c.post("/v1/completions", json={"prompt": [5, 6, 7], "max_tokens": 5, "temperature": 0, "logprobs": 2})
    assert r.status_code == 200, r.text
    lp = r.json()["choices"][0]["logprobs"]
    assert set(lp) == {"tokens", "token_logprobs", "top_logprobs", "text_offset"}
    assert len(lp["tokens"]) == len(lp["token_logprobs"]) == len(lp["top_logprobs"]) == len(lp["text_offset"]) == 5
    assert all(v <= 0 for v in lp["token_logprobs"]) and all(len(t) <= 2 for t in lp["top_logprobs"])
    assert lp["text_offset"] == sorted(lp["text_offset"]) and lp["text_offset"][0] == 0
    r0 = c.post("/v1/completions", json={"prompt": [5, 6, 7], "max_tokens": 3, "temperature": 0, "logprobs": 0})
    assert r0.json()["choices"][0]["logprobs"]["top_logprobs"] == [None, None, None]
    with c.stream("POST", "/v1/completions", json={"prompt": [5, 6, 7], "max_tokens": 4, "temperature": 0, "logprobs": 1,
                                                   "stream": True}) as resp:
        chunks = [json.loads(line[6:]) for line in resp.iter_lines() if line.startswith("data: {")]
    got = [t for ch in chunks if ch["choices"][0]["logprobs"] for t in ch["choices"][0]["logprobs"]["token_logprobs"]]
    assert len(got) == 4
    th.shutdown()
