"""Client-side glue to the harness's drive and ingest contracts.

``evaluation_payload`` builds the body that benchmark-results ingests unchanged
(benchmark-results/models.py:6-19, routes /raw_input and /standardized_output,
benchmark-results/main.py:55-83); ``BenchClient`` drives the server the way the
harness's evaluation jobs do (POST /v1/completions, streaming, to measure time
to first token and decode rate) after waiting on /health and /v1/models like
benchmark-eval does (benchmark-eval/main.py:102-110)."""
from __future__ import annotations

import json
import time
import uuid
from datetime import datetime, timezone
from typing import Any, Dict, List, Optional


def evaluation_payload(benchmark_name: str, data: Any, model_id: str, tokenizer_id: Optional[str] = None,
                       source: str = "mi355-bench-client", run_id: Optional[str] = None,
                       timestamp: Optional[str] = None) -> Dict[str, Any]:
    """Always carries a non-null run_id: the reference's response model declares it a
    string (benchmark-results/models.py:39-43) and would fail on None."""
    return {
        "run_id": run_id or uuid.uuid4().hex,
        "benchmark_name": benchmark_name,
        "data": data,
        "timestamp": timestamp or datetime.now(timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ"),
        "model_id": model_id,
        "tokenizer_id": tokenizer_id or model_id,
        "source": source,
    }


def result_pk(payload: Dict[str, Any]) -> str:
    """Primary key benchmark-results will store the payload under."""
    return f"{payload['timestamp']}-{payload['benchmark_name']}-{payload['run_id']}"


class BenchClient:
    """Minimal blocking HTTP client (requests) for the three engine routes."""

    def __init__(self, base_url: str, timeout: float = 600.0):
        import requests
        self.base, self.timeout, self.http = base_url.rstrip("/"), timeout, requests.Session()

    def wait_ready(self, tries: int = 60, delay: float = 1.0) -> str:
        """benchmark-eval's loop: GET /v1/models until 200 (connection errors retried
        here, unlike the reference, which drops the evaluation on them)."""
        last = None
        for _ in range(tries):
            try:
                r = self.http.get(f"{self.base}/v1/models", timeout=5)
                if r.status_code == 200:
                    return r.json()["data"][0]["id"]
                last = r.status_code
            except Exception as exc:  # noqa: BLE001
                last = exc
            time.sleep(delay)
        raise TimeoutError(f"engine not ready: {last}")

    def health(self) -> int:
        return self.http.get(f"{self.base}/health", timeout=5).status_code

    def complete_stream(self, model: str, prompt, max_tokens: int, **kw) -> Dict[str, Any]:
        """One streaming completion; returns text, finish_reason, ttft_s, total_s, chunks."""
        body = {"model": model, "prompt": prompt, "max_tokens": max_tokens, "temperature": 0, "stream": True, **kw}
        t0 = time.perf_counter()
        first = None
        text, reason, n = "", None, 0
        with self.http.post(f"{self.base}/v1/completions", json=body, stream=True, timeout=self.timeout) as r:
            r.raise_for_status()
            for line in r.iter_lines():
                if not line or not line.startswith(b"data: "):
                    continue
                data = line[6:]
                if data == b"[DONE]":
                    break
                obj = json.loads(data)
                if "error" in obj:
                    raise RuntimeError(obj["error"])
                ch = obj["choices"][0]
                if first is None:
                    first = time.perf_counter()
                text += ch["text"]
                n += 1
                reason = ch["finish_reason"] or reason
        t1 = time.perf_counter()
        return {"text": text, "finish_reason": reason, "ttft_s": (first or t1) - t0, "total_s": t1 - t0, "chunks": n}

    def post_results(self, results_url: str, payload: Dict[str, Any]) -> List[int]:
        out = []
        for route in ("/raw_input", "/standardized_output"):
            out.append(self.http.post(results_url.rstrip("/") + route, json=payload, timeout=30).status_code)
        return out
