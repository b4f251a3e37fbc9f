#!/usr/bin/env python3
"""Host-side cost of the served path, without a GPU: the real FastAPI app + engine thread +
scheduler over the CPU test double of the model runner (a step takes no time), driven by
mi355_decode/bench_client.py.  tok/s here = what the HTTP/SSE fan-out alone can sustain."""
import json, os, socket, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vllm-backend-public_amd"), os.path.join(ROOT, "tests"), ROOT]
import uvicorn
from mi355_decode.engine import LLMEngine
from mi355_decode.entrypoints.openai.api_server import build_app
from mi355_decode.serving import EngineThread
from mi355_decode.tokenizer import ByteTokenizer
from test_engine_cpu import FakeRunner, V

B, G, P = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 128, 128
th = EngineThread(lambda: LLMEngine(FakeRunner(B * 20, B, 512), B, 8192, 512))
app = build_app(th, "m", ByteTokenizer(V), V, 512)
th.start()
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
srv = uvicorn.Server(uvicorn.Config(app, host="127.0.0.1", port=port, log_level="warning", access_log=False))
threading.Thread(target=srv.run, daemon=True).start()
while not srv.started:
    time.sleep(0.01)
env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "vllm-backend-public_amd"))
p = subprocess.run([sys.executable, "-m", "mi355_decode.bench_client", "--url", f"http://127.0.0.1:{port}", "--model", "m",
                    "--batch", str(B), "--prompt-len", str(P), "--gen-len", str(G), "--vocab", str(V), "--runs", "3"],
                   env=env, capture_output=True, text=True)
print(p.stdout[-600:], p.stderr[-600:])
srv.should_exit = True
th.shutdown()
