"""BASELINE.json configs[0] for this engine: the plumbing pass of the harness on
one GPU.  The server process is started with the container args the reference's
template generated for a harness config (golden fixture), only --model/--port
swapped for a local tiny model directory; then the steps of the harness are
replayed: poll /health (kubelet) and /v1/models (benchmark-eval), send
/v1/completions requests (the evaluation job), and post the results to a stub of
benchmark-results that validates them like the reference's EvaluationPayload."""
import http.server
import json
import os
import socket
import subprocess
import sys
import threading
import time

import pytest

pytestmark = pytest.mark.gpu

from conftest import GOLDEN, PKG, ROOT  # noqa: E402
from oracle import harness_ref as HR  # noqa: E402
from oracle.tiny_models import TINY  # noqa: E402


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class ResultsStub(http.server.BaseHTTPRequestHandler):
    stored = {}

    def do_POST(self):
        body = json.loads(self.rfile.read(int(self.headers["Content-Length"])))
        ok = self.path in ("/raw_input", "/standardized_output") and HR.payload_accepted(body)
        if ok:
            doc = HR.stored_document(body)
            ResultsStub.stored[(self.path, doc["pk"])] = doc
        self.send_response(200 if ok else 422)
        self.end_headers()
        self.wfile.write(json.dumps({"status": "success" if ok else "error"}).encode())

    def log_message(self, *a):
        pass


@pytest.mark.parametrize("case,tp", [("cfg2_llama3_8b_tp1", 1), ("cfg3_llama3_8b_tp4", 2), ("cfg1_opt125m_cpu", 1)],
                         ids=["tp1", "tp2_one_gpu", "opt125m_as_launched"])
def test_harness_plumbing_pass(tmp_path, case, tp):
    """tp2_one_gpu: the tensor-parallel server as the harness launches it (--tensor-parallel-size from
    the TP config's args, cut to 2): rank 0 serves HTTP and starts the worker rank itself, requests
    reach the worker over the control channel, both step in lock-step exchanging partial sums -- the
    two ranks share this box's one GPU and talk over gloo, which RCCL would not allow.
    opt125m_as_launched: the args and env the harness generates when a request names no model
    (facebook/opt-125m, --device cpu, --dtype float32, 2 sequences of 512 tokens), unchanged but for
    host and port: the real OPT-125m geometry on seeded weights (nothing can be downloaded), served
    by this engine's GPU path in float16, the type OPT-125m is published in."""
    import torch
    assert torch.cuda.is_available()
    from mi355_decode.harness import BenchClient, evaluation_payload, result_pk

    launch = json.load(open(os.path.join(GOLDEN, "harness_launch.json")))
    args = list(launch["cases"][case]["deployment"]["args"])
    port = free_port()
    args[args.index("--port") + 1] = str(port)
    args[args.index("--host") + 1] = "127.0.0.1"
    if case != "cfg1_opt125m_cpu":
        args[args.index("--tensor-parallel-size") + 1] = str(tp)
        args[args.index("--max-num-seqs") + 1] = "8"
        model_dir = tmp_path / "tiny-llama"
        model_dir.mkdir()
        arch, cfg, _ = TINY["llama_d128"]
        (model_dir / "config.json").write_text(json.dumps({**cfg, "architectures": [arch]}))
        args[args.index("--model") + 1] = str(model_dir)
        args[args.index("--max-model-len") + 1] = "256"
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + ROOT)
    for e in launch["cases"][case]["deployment"]["env"]:     # the harness's env, CUDA_VISIBLE_DEVICES="" included
        env[e["name"]] = e["value"]
    if tp > 1:
        env["MI355_DIST_BACKEND"] = "gloo"
        env["MI355_ONESHOT_ALLREDUCE"] = "1"     # the production form of the step: one-shot collectives inside HIP graphs
    log = open(tmp_path / "server.log", "w")
    proc = subprocess.Popen([sys.executable, "-m", "mi355_decode.entrypoints.openai.api_server"] + args,
                            env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
    stub = http.server.ThreadingHTTPServer(("127.0.0.1", 0), ResultsStub)
    threading.Thread(target=stub.serve_forever, daemon=True).start()
    try:
        client = BenchClient(f"http://127.0.0.1:{port}")
        t0 = time.time()
        saw_503 = False
        while True:                                   # kubelet-style probe loop
            assert proc.poll() is None, open(tmp_path / "server.log").read()[-3000:]
            try:
                code = client.health()
                saw_503 |= code == 503
                if code == 200:
                    break
            except Exception:
                pass
            assert time.time() - t0 < 240, "server did not become healthy"
            time.sleep(0.2)
        served = client.wait_ready(tries=30, delay=0.5)
        assert served == args[args.index("--served-model-name") + 1]
        # what precision is served: the harness's default config asks for float32
        # (/root/reference/benchmark-vllm/models.py:14), the kernels compute in 16 bits, and the model card says so
        card = client.http.get(f"http://127.0.0.1:{port}/v1/models").json()["data"][0]
        asked = args[args.index("--dtype") + 1]
        # float32 cannot be had; a checkpoint stored in 16 bits is then served in its own type (OPT-125m: float16,
        # every stored weight exact), an unlabelled one (the tiny model's config.json names no type) in bfloat16
        served_in = "float16" if asked in ("half", "float16") or case == "cfg1_opt125m_cpu" else "bfloat16"
        assert card["requested_dtype"] == asked and card["dtype"] == served_in
        assert card["dtype_downgraded"] == (asked == "float32")
        if case == "cfg1_opt125m_cpu":
            assert asked == "float32" and card["dtype_downgraded"] is True
            assert "mi355_dtype_downgraded 1" in client.http.get(f"http://127.0.0.1:{port}/metrics").text

        # the evaluation job: 4 token-id prompts, greedy, streaming
        outs = []
        for i in range(4):
            prompt = [(7 * i + j) % 500 + 1 for j in range(16 + i)]
            outs.append(client.complete_stream(served, prompt, max_tokens=24, ignore_eos=True))
        assert all(o["finish_reason"] == "length" and o["chunks"] >= 1 for o in outs)
        again = client.complete_stream(served, [(j % 500) + 1 for j in range(16)], max_tokens=24, ignore_eos=True)
        assert again["text"] == outs[0]["text"]       # greedy decoding is reproducible
        r = client.http.post(f"http://127.0.0.1:{port}/v1/completions",
                             json={"model": served, "prompt": "hello world", "max_tokens": 8, "temperature": 0})
        assert r.status_code == 200 and r.json()["usage"]["completion_tokens"] == 8
        assert client.health() == 200

        data = {"ttft_ms_p50": sorted(o["ttft_s"] for o in outs)[2] * 1e3, "requests": len(outs),
                "decode_tok_s": sum(24 for _ in outs) / sum(o["total_s"] for o in outs)}
        payload = evaluation_payload("mi355-decode-plumbing", data, served)
        codes = client.post_results(f"http://127.0.0.1:{stub.server_address[1]}", payload)
        assert codes == [200, 200]
        assert ("/raw_input", result_pk(payload)) in ResultsStub.stored
        assert ResultsStub.stored[("/standardized_output", result_pk(payload))]["data"]["requests"] == 4
        if tp > 1:                                    # what the per-turn control broadcast to the workers costs rank 0
            m = dict(l.rsplit(" ", 1) for l in client.http.get(f"http://127.0.0.1:{port}/metrics").text.splitlines() if l)
            n, secs = int(m["mi355_control_broadcasts_total"]), float(m["mi355_control_broadcast_seconds_total"])
            assert n >= 24 and secs > 0
            print(f"control broadcast (gloo, 2 ranks on this host): {secs / n * 1e6:.1f} us per engine-loop turn over {n} turns")
        log.flush()
        text = open(tmp_path / "server.log").read()
        for r in range(tp):                           # every rank really came up, with the sharding its size implies
            assert f"tensor-parallel rank {r} of {tp}" in text, text[-2000:]
        assert ("MLP sharded, lm_head sharded" in text) == (tp > 1)
        if tp > 1:
            assert text.count("HIP graphs on") == tp and "one-shot all-reduce for messages up to" in text, text[-2000:]
    finally:
        proc.terminate()
        try:
            proc.wait(30)
        except subprocess.TimeoutExpired:
            pass
        try:                                          # whatever is left of the server's own process group (TP workers)
            os.killpg(proc.pid, 9)
        except ProcessLookupError:
            pass
        stub.shutdown()
        log.close()


def _oracle_generation(cfg, w, prompt_ids, gen):
    """Greedy tokens of the numpy oracle for one prompt, every step's logits, and every step's top-2
    margin relative to the logit scale."""
    import numpy as np
    from oracle import model_ref as M
    model = M.RefModel(M.RefConfig(**cfg), w, "bfloat16", num_blocks=16)
    toks, logits = M.greedy_generate(model, [np.asarray(prompt_ids, dtype=np.int64)], gen)
    scale = float(np.abs(logits[0]).max())
    margins = [float(np.sort(lg[0])[-1] - np.sort(lg[0])[-2]) / scale for lg in logits]
    return [int(t) for t in toks[0]], [lg[0] for lg in logits], margins, scale


def test_real_model_directory_end_to_end(tmp_path):
    """The only thing the harness ever serves is a local checkpoint directory
    (/root/reference/benchmark-mlflow/vllm_processor.py:29; polled at /root/reference/benchmark-eval/main.py:105).
    One is built here from nothing -- sharded safetensors + index, tokenizer.json trained in memory,
    tokenizer_config.json with a chat template, config.json + generation_config.json with a LIST of
    end-of-sequence ids -- and served: text goes in, text comes out and equals tokenizer.decode(oracle
    tokens); generation stops on an end-of-sequence id from either file; a stop string that straddles
    two tokens cuts the text; the chat route renders the template (one BOS)."""
    import torch
    assert torch.cuda.is_available()
    from mi355_decode.harness import BenchClient
    from mi355_decode.tokenizer import load_tokenizer
    from real_model_fixture import BOS, CORPUS, EOT, build_model_dir
    from test_model_gpu import LOGIT_TOL

    model_dir = str(tmp_path / "tiny-llama-real")
    cfg, w, _ = build_model_dir(model_dir, "llama_d128", shards=3)
    tok = load_tokenizer(model_dir, cfg["vocab_size"])
    GEN = 8
    # prompts whose oracle generation has a clear margin at every step (2^-5 of the logit scale, the bound
    # on the worst single logit of a bf16 step), so that the served text can be held to the oracle's exactly
    texts, gens = [], []
    for text in CORPUS:
        toks, _, margins, _ = _oracle_generation(cfg, w, tok.encode(text), GEN)
        if min(margins) >= 2.0 ** -5 and 1 not in toks:
            texts.append(text)
            gens.append(toks)
    assert len(texts) >= 3, "too few prompts with clear oracle margins"
    # end-of-sequence ids: config.json names a token the SECOND prompt's generation reaches, generation_config.json
    # one the FIRST prompt's reaches (and not the other way round, nor before)
    y = next(t for i, t in enumerate(gens[0][2:]) if t not in gens[0][:i + 2] and t not in gens[1])
    z = next(t for i, t in enumerate(gens[1][2:]) if t not in gens[1][:i + 2] and t not in gens[0] and t != y)
    full = json.load(open(os.path.join(model_dir, "config.json")))
    full["eos_token_id"] = z
    json.dump(full, open(os.path.join(model_dir, "config.json"), "w"))
    json.dump({"eos_token_id": [y, 1]}, open(os.path.join(model_dir, "generation_config.json"), "w"))

    port = free_port()
    args = ["--model", model_dir, "--served-model-name", "tiny-real", "--host", "127.0.0.1", "--port", str(port),
            "--dtype", "bfloat16", "--max-num-seqs", "8", "--max-model-len", "256", "--tensor-parallel-size", "1",
            "--load-format", "auto"]
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + ROOT)
    log = open(tmp_path / "server.log", "w")
    proc = subprocess.Popen([sys.executable, "-m", "mi355_decode.entrypoints.openai.api_server"] + args,
                            env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
    url = f"http://127.0.0.1:{port}"
    try:
        client = BenchClient(url)
        t0 = time.time()
        while True:
            assert proc.poll() is None, open(tmp_path / "server.log").read()[-3000:]
            try:
                if client.health() == 200:
                    break
            except Exception:
                pass
            assert time.time() - t0 < 240, "server did not become healthy"
            time.sleep(0.2)
        assert client.wait_ready(tries=30, delay=0.5) == "tiny-real"
        card = client.http.get(f"{url}/v1/models").json()["data"][0]
        assert card["dtype"] == "bfloat16" and card["dtype_downgraded"] is False
        post = lambda route, **body: client.http.post(f"{url}{route}", json={"model": "tiny-real", **body})   # noqa: E731

        # text in, text out == decode(oracle tokens); the weights came from the safetensors shards, not a seed
        for text, toks in zip(texts, gens):
            r = post("/v1/completions", prompt=text, max_tokens=GEN, temperature=0, ignore_eos=True)
            assert r.status_code == 200, r.text
            assert r.json()["choices"][0]["text"] == tok.decode(toks), text
            assert r.json()["usage"]["prompt_tokens"] == len(tok.encode(text))
        text_log = open(tmp_path / "server.log").read()
        assert "synthetic" not in text_log and "byte-wise" not in text_log, text_log[-2000:]

        # a longer generation, by token id (return_tokens_as_token_ids): the text is the decoding of the very tokens
        # the engine produced, and those leave the oracle's only at a near-tie of the oracle's logits
        LONG = 24
        want, logits, _, scale = _oracle_generation(cfg, w, tok.encode(texts[0]), LONG)
        r = post("/v1/completions", prompt=texts[0], max_tokens=LONG, temperature=0, ignore_eos=True, logprobs=0,
                 return_tokens_as_token_ids=True)
        ch = r.json()["choices"][0]
        got = [int(t.split(":")[1]) for t in ch["logprobs"]["tokens"]]
        assert len(got) == LONG and ch["text"] == tok.decode(got)
        for t in range(LONG):
            if got[t] != want[t]:
                gap = logits[t][want[t]] - logits[t][got[t]]
                assert 0 <= gap <= 2 * LOGIT_TOL["bfloat16"] * scale, f"step {t}: {got[t]} vs oracle {want[t]}, margin {gap}"
                break

        # end-of-sequence ids from either file end the generation (the id itself counted, not shown)
        for text, toks, eos in ((texts[0], gens[0], y), (texts[1], gens[1], z)):
            r = post("/v1/completions", prompt=text, max_tokens=GEN, temperature=0)
            ch, n = r.json()["choices"][0], toks.index(eos) + 1
            assert ch["finish_reason"] == "stop" and r.json()["usage"]["completion_tokens"] == n
            assert ch["text"] == tok.decode(toks[:n - 1])

        # a stop string across a token boundary (two tokens with plain ASCII text), plain and streamed
        found = None
        for text, toks in zip(texts, gens):
            pieces = [tok.decode(toks[:k + 1])[len(tok.decode(toks[:k])):] for k in range(GEN)]
            whole = "".join(pieces)
            for i in range(1, GEN - 1):
                a, b = pieces[i], pieces[i + 1]
                if a and b and (a + b).isascii() and whole.find(a[-1:] + b[:1]) == len("".join(pieces[:i + 1])) - 1:
                    found = (text, whole, a[-1:] + b[:1])
                    break
            if found:
                break
        assert found, "no pair of adjacent ASCII pieces in the oracle generations"
        stop_prompt, whole, stop = found
        r = post("/v1/completions", prompt=stop_prompt, max_tokens=GEN, temperature=0, ignore_eos=True, stop=stop)
        assert r.json()["choices"][0]["finish_reason"] == "stop"
        assert r.json()["choices"][0]["text"] == whole[:whole.find(stop)]
        with client.http.post(f"{url}/v1/completions", stream=True,
                              json={"model": "tiny-real", "prompt": stop_prompt, "max_tokens": GEN, "temperature": 0,
                                    "ignore_eos": True, "stop": [stop], "stream": True}) as r:
            lines = [ln.decode() for ln in r.iter_lines() if ln]
        chunks = [json.loads(ln[6:]) for ln in lines if ln != "data: [DONE]"]
        assert "".join(c["choices"][0]["text"] for c in chunks) == whole[:whole.find(stop)]

        # chat: the directory's template, tokenised without a second BOS
        msgs = [{"role": "user", "content": texts[0]}]
        rendered = f"{BOS}<|user|>\n{texts[0]}{EOT}\n<|assistant|>\n"
        ids = tok.encode(rendered, add_special_tokens=False)
        assert ids.count(tok.bos_token_id) == 1
        want, _, margins, _ = _oracle_generation(cfg, w, ids, 4)
        r = post("/v1/chat/completions", messages=msgs, max_tokens=4, temperature=0, ignore_eos=True)
        assert r.status_code == 200, r.text
        assert r.json()["usage"]["prompt_tokens"] == len(ids)
        if min(margins) >= 2.0 ** -5:
            assert r.json()["choices"][0]["message"]["content"] == tok.decode(want)
    finally:
        proc.terminate()
        try:
            proc.wait(30)
        except subprocess.TimeoutExpired:
            pass
        try:
            os.killpg(proc.pid, 9)
        except ProcessLookupError:
            pass
        log.close()
