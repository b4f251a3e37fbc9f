"""Load generator for the served path: what the harness's evaluation jobs do to the engine
(POST /v1/completions, benchmark-deployer/docs/deployment/post_run.md:46), with the timing the
north-star metric needs.  B concurrent streaming token-id requests (`ignore_eos`, fixed
`max_tokens`), per request the time of its first and last token:

    TTFT            = t_first_token - t_submit                      (p50 over the B requests)
    decode tok/s    = B * (G - 1) / (t_last_token - t_first_token)  (first token of the earliest
                      request to last token of the latest: whole-batch rate; requests that arrive a
                      few ms apart are prefilled in several steps, and that stagger is inside it)
    steady tok/s    = tokens received while ALL B streams are decoding (after the latest first
                      token, up to the earliest last token) / that window: the served decode rate
                      without the ramp at either end

One untimed warm-up run, then `runs` measured runs; the medians are reported and, when a results
URL is given, posted as an EvaluationPayload to /raw_input and /standardized_output
(benchmark-results/models.py:6-19, main.py:55-83).  Runs as its own process so that parsing 4 000
server-sent events per run does not compete with the server for the interpreter lock:

    python -m mi355_decode.bench_client --url http://127.0.0.1:8000 --model m --batch 32 \
        --prompt-len 128 --gen-len 128 --vocab 128256 --runs 5
"""
from __future__ import annotations

import argparse
import asyncio
import json
import random
import statistics
import time
from typing import List

from .harness import evaluation_payload


def make_prompts(vocab: int, batch: int, plen: int, seed: int) -> List[List[int]]:
    rng = random.Random(seed)
    lo, hi = min(1000, vocab // 4), max(vocab - 1000, vocab // 2)
    return [[rng.randrange(lo, hi) for _ in range(plen)] for _ in range(batch)]


async def _one(session, url: str, model: str, prompt: List[int], gen: int, t_start: float):
    body = {"model": model, "prompt": prompt, "max_tokens": gen, "temperature": 0, "stream": True,
            "ignore_eos": True}
    t_submit = time.perf_counter()
    first = last = None
    chunks = 0
    stamps = []
    async with session.post(url + "/v1/completions", json=body) as r:
        if r.status != 200:
            raise RuntimeError(f"HTTP {r.status}: {await r.text()}")
        async for raw in r.content:
            line = raw.strip()
            if not line.startswith(b"data: "):
                continue
            data = line[6:]
            if data == b"[DONE]":
                break
            now = time.perf_counter()
            obj = json.loads(data)
            if "error" in obj:
                raise RuntimeError(obj["error"])
            chunks += 1
            if first is None:
                first = now
            last = now
            stamps.append(now - t_start)
    return {"submit": t_submit - t_start, "first": first - t_start, "last": last - t_start, "chunks": chunks,
            "stamps": stamps}


async def one_run(url: str, model: str, prompts: List[List[int]], gen: int):
    import aiohttp
    timeout = aiohttp.ClientTimeout(total=600)
    conn = aiohttp.TCPConnector(limit=0)
    async with aiohttp.ClientSession(timeout=timeout, connector=conn) as session:
        t0 = time.perf_counter()
        res = await asyncio.gather(*[_one(session, url, model, p, gen, t0) for p in prompts])
    B = len(prompts)
    ttft = [r["first"] - r["submit"] for r in res]
    t_first, t_last = min(r["first"] for r in res), max(r["last"] for r in res)
    t_a, t_b = max(r["first"] for r in res), min(r["last"] for r in res)
    inside = sum(1 for r in res for t in r["stamps"] if t_a < t <= t_b)
    return {"steady_tok_s": inside / (t_b - t_a) if t_b > t_a else 0.0,
            "ttft_ms_p50": statistics.median(ttft) * 1e3, "ttft_ms_max": max(ttft) * 1e3,
            "decode_tok_s": B * (gen - 1) / (t_last - t_first) if t_last > t_first else 0.0,
            "wall_s": t_last, "chunks_min": min(r["chunks"] for r in res),
            # absolute stamps (time.perf_counter, the host's monotonic clock: comparable with the server's)
            "submit_abs": sorted(t0 + r["submit"] for r in res), "first_abs": sorted(t0 + r["first"] for r in res)}


async def post_results(results_url: str, payload: dict) -> List[int]:
    import aiohttp
    out = []
    async with aiohttp.ClientSession() as session:
        for route in ("/raw_input", "/standardized_output"):
            async with session.post(results_url.rstrip("/") + route, json=payload) as r:
                out.append(r.status)
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--url", required=True)
    ap.add_argument("--model", required=True)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--gen-len", type=int, default=128)
    ap.add_argument("--vocab", type=int, required=True)
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--results-url", default=None)
    ap.add_argument("--benchmark-name", default="decode_throughput")
    a = ap.parse_args(argv)
    runs = []
    for i in range(a.warmup + a.runs):
        prompts = make_prompts(a.vocab, a.batch, a.prompt_len, 4321 + i)
        r = asyncio.run(one_run(a.url, a.model, prompts, a.gen_len))
        if i >= a.warmup:
            runs.append(r)
    out = {"batch": a.batch, "prompt_len": a.prompt_len, "gen_len": a.gen_len, "runs": len(runs), "warmup": a.warmup,
           "decode_tok_s_p50": round(statistics.median(r["decode_tok_s"] for r in runs), 1),
           "ttft_ms_p50": round(statistics.median(r["ttft_ms_p50"] for r in runs), 2),
           "ttft_ms_max_p50": round(statistics.median(r["ttft_ms_max"] for r in runs), 2),
           "steady_decode_tok_s_p50": round(statistics.median(r["steady_tok_s"] for r in runs), 1),
           "decode_tok_s_runs": [round(r["decode_tok_s"], 1) for r in runs],
           "chunks_per_request_min": min(r["chunks_min"] for r in runs),
           "stamps": [{"submit_first": r["submit_abs"][0], "submit_p50": statistics.median(r["submit_abs"]),
                       "submit_last": r["submit_abs"][-1], "first_token_first": r["first_abs"][0],
                       "first_token_p50": statistics.median(r["first_abs"]), "first_token_last": r["first_abs"][-1]}
                      for r in runs]}
    if a.results_url:
        payload = evaluation_payload(a.benchmark_name, out, model_id=a.model, source="mi355-bench-client")
        out["results_post_status"] = asyncio.run(post_results(a.results_url, payload))
        out["results_pk"] = f"{payload['timestamp']}-{payload['benchmark_name']}-{payload['run_id']}"
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
