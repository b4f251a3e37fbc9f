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
    by this engine's GPU path in bf16."""
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
        log.flush()
        text = open(tmp_path / "server.log").read()
        for r in range(tp):                           # every rank really came up, with the sharding its size implies
            assert f"tensor-parallel rank {r} of {tp}" in text, text[-2000:]
        assert ("MLP sharded, lm_head sharded" in text) == (tp > 1)
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
