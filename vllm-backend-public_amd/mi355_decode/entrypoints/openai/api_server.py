"""OpenAI-compatible completion server: the process the harness starts in place
of the stock engine image.

Routes the reference calls:
  GET  /health          kubelet readiness/liveness probes (benchmark-vllm/vllm_templates.py:182-201,
                        352-369) and the deployer's wait loop (benchmark-deployer/deployer_manager.py:1597-1608)
  GET  /v1/models       benchmark-eval polls it until 200 before launching the evaluation job
                        (benchmark-eval/main.py:102-110)
  POST /v1/completions  the evaluation jobs' endpoint (benchmark-deployer/docs/deployment/post_run.md:46;
                        env VLLM_MODEL_ENDPOINT, deployer_manager.py:2195-2199)

The socket is bound before the weights are loaded: until the engine is ready
/health and /v1/models answer 503 (a refused connection would make benchmark-eval
drop the evaluation, benchmark-eval/main.py:105,131).  A fatal engine error turns
/health into 500 and the process exits non-zero, so Kubernetes shows
CrashLoopBackOff and the harness counts a failure (vllm_manager.py:921-936).

    python -m mi355_decode.entrypoints.openai.api_server --model /data/local_models/X --port 8000 ...
"""
from __future__ import annotations

import asyncio
import dataclasses
import json
import logging
import os
import sys
import time
import uuid
from typing import List, Optional

from fastapi import FastAPI, Request
from fastapi.responses import JSONResponse, Response, StreamingResponse

from ..cli_args import parse_args, sanitize_env

MAX_N = 64           # completions per prompt
MAX_CHOICES = 256    # completions per request (prompts x n)

log = logging.getLogger("mi355_decode")


def _error(status: int, message: str, etype: str = "invalid_request_error"):
    return JSONResponse(status_code=status,
                        content={"error": {"message": message, "type": etype, "param": None, "code": status}})


def normalise_prompts(prompt, tokenizer, vocab_size: int) -> List[List[int]]:
    """prompt: str | [str] | [int] | [[int]]  ->  list of token-id prompts."""
    if isinstance(prompt, str):
        return [tokenizer.encode(prompt)]
    if isinstance(prompt, list) and prompt:
        if all(isinstance(p, str) for p in prompt):
            return [tokenizer.encode(p) for p in prompt]
        # (exact types: a bool is not a token id; one pass in C instead of two isinstance calls per token --
        # a burst of 32 requests is parsed on one thread, and every microsecond of it is time to first token)
        if set(map(type, prompt)) == {int}:
            return [list(prompt)]
        if all(type(p) is list and p and set(map(type, p)) == {int} for p in prompt):
            return [list(p) for p in prompt]
    raise ValueError("prompt must be a string, a list of strings, a list of token ids or a list of token-id lists")


def render_chat(messages, tokenizer) -> str:
    """messages [{role, content}] -> one prompt string ending where the assistant's answer begins.
    The model directory's own template is used when the tokenizer carries one (`chat_template`, a
    Jinja string from tokenizer_config.json); otherwise a plain transcript: `<|role|>` line,
    content, blank line."""
    for m in messages:
        if not isinstance(m, dict) or not isinstance(m.get("role"), str):
            raise ValueError("every message needs a role")
        c = m.get("content")
        if isinstance(c, list):                           # content parts: the text ones, concatenated
            m["content"] = "".join(p.get("text", "") for p in c if isinstance(p, dict) and p.get("type") == "text")
        elif not isinstance(c, str):
            raise ValueError("message content must be a string or a list of text parts")
    template = getattr(tokenizer, "chat_template", None)
    if template:
        # the template comes with the model directory, i.e. from whoever published the model: it is
        # rendered in a sandbox (no attribute walks into Python internals, no mutation of what it is
        # handed), with the two helpers published templates call
        import datetime
        import jinja2
        import jinja2.sandbox

        def raise_exception(message):
            raise jinja2.TemplateError(message)

        env = jinja2.sandbox.ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
        env.globals["raise_exception"] = raise_exception
        env.globals["strftime_now"] = lambda fmt: datetime.datetime.now().strftime(fmt)
        try:
            return env.from_string(template).render(messages=messages, add_generation_prompt=True,
                                                    bos_token=getattr(tokenizer, "bos_token", "") or "",
                                                    eos_token=getattr(tokenizer, "eos_token", "") or "")
        except jinja2.TemplateError as exc:               # incl. SecurityError: the request cannot be rendered
            raise ValueError(f"chat template: {exc}") from None
    return "".join(f"<|{m['role']}|>\n{m['content']}\n\n" for m in messages) + "<|assistant|>\n"


def build_app(engine_thread, served_model_name: str, tokenizer, vocab_size: int, max_model_len: int,
              requested_dtype: Optional[str] = None, served_dtype: Optional[str] = None):
    """requested_dtype / served_dtype: what --dtype asked for and the arithmetic type the kernels
    run in.  The harness's default config asks for float32 (benchmark-vllm/models.py:14); this
    engine computes in 16 bits, and /v1/models and /metrics say so instead of leaving it to a log line."""
    from ...sampling_params import SamplingParams
    from ...tokenizer import IncrementalDetokenizer

    app = FastAPI(title="mi355-decode", version="0.1.0")
    created = int(time.time())
    dtype_downgraded = (requested_dtype or "").lower() in ("float32", "float")

    def not_ready():
        if engine_thread.failed is not None:
            return _error(500, f"engine failed: {engine_thread.failed!r}", "internal_error")
        if not engine_thread.ready.is_set():
            return _error(503, "engine is loading", "service_unavailable")
        return None

    @app.get("/health")
    async def health():
        bad = not_ready()
        return bad if bad is not None else Response(status_code=200)

    @app.get("/ping")
    async def ping():
        return await health()

    @app.get("/version")
    async def version():
        from ... import __version__
        return {"version": __version__, "engine": "mi355-decode"}

    @app.get("/v1/models")
    async def models():
        bad = not_ready()
        if bad is not None:
            return bad
        card = {"id": served_model_name, "object": "model", "created": created, "owned_by": "mi355-decode",
                "root": served_model_name, "parent": None, "max_model_len": max_model_len}
        if served_dtype is not None:
            card.update(dtype=served_dtype, requested_dtype=requested_dtype, dtype_downgraded=dtype_downgraded)
        return {"object": "list", "data": [card]}

    @app.get("/metrics")
    async def metrics():
        eng = engine_thread.engine
        lines = [f"mi355_engine_steps_total {engine_thread.steps}",
                 f"mi355_generation_tokens_total {engine_thread.tokens_out}",
                 f"mi355_uptime_seconds {time.time() - engine_thread.started:.1f}"]
        if engine_thread.exchanges:                              # tensor parallel: rank 0's per-turn control broadcast to the workers
            lines += [f"mi355_control_broadcasts_total {engine_thread.exchanges}",
                      f"mi355_control_broadcast_seconds_total {engine_thread.exchange_s:.6f}"]
        if served_dtype is not None:
            lines += [f'mi355_served_dtype{{dtype="{served_dtype}",requested="{requested_dtype}"}} 1',
                      f"mi355_dtype_downgraded {int(dtype_downgraded)}"]
        if eng is not None:
            lines += [f"mi355_num_requests_running {len(eng.sched.running)}",
                      f"mi355_num_requests_waiting {len(eng.sched.waiting)}",
                      f"mi355_kv_blocks_free {eng.alloc.num_free}",
                      f"mi355_kv_blocks_total {eng.alloc.num_blocks}"]
            r = getattr(eng, "runner", None)
            if r is not None and hasattr(r, "capture_failures"):   # degraded modes: eager launches instead of graph replay, untimed GEMM plans
                lines += [f"mi355_hip_graphs_enabled {int(bool(r.use_graphs))}",
                          f"mi355_graph_capture_failures_total {r.capture_failures}",
                          f"mi355_gemm_tune_failures_total {r.tune_failures}"]
                if getattr(r, "tp_size", 1) > 1:               # which all-reduce serves the decode-sized messages
                    lines += [f"mi355_oneshot_allreduce_enabled {int(getattr(r, 'oneshot', None) is not None)}",
                              f"mi355_oneshot_allreduce_fallbacks_total {getattr(r, 'oneshot_fallbacks', 0)}",
                              f"mi355_allreduce_norm_fused_total {getattr(r, 'fused_allreduce_norms', 0)}"]
                for bucket, mode in sorted(getattr(r, "_modes", {}).items()):
                    for proj, tiled in mode.items():
                        lines.append(f'mi355_gemm_form{{bucket="{bucket}",projection="{proj}"}} {int(bool(tiled))}')
        return Response("\n".join(lines) + "\n", media_type="text/plain")

    # The two generation routes are plain Starlette routes (request in, response out): the framework's
    # per-request dependency / body-model machinery has nothing to do here and costs a burst of 32
    # requests several milliseconds of time to first token on the one thread that parses them.
    async def completions(request: Request):
        return await serve(request, chat=False)

    async def chat_completions(request: Request):
        """Chat form of the same generation path: the messages are rendered to one prompt (the model
        directory's chat template when it has one, a plain role-tagged transcript otherwise) and the
        answer comes back as an assistant message (`chat.completion` / `chat.completion.chunk`)."""
        return await serve(request, chat=True)

    app.add_route("/v1/completions", completions, methods=["POST"])
    app.add_route("/v1/chat/completions", chat_completions, methods=["POST"])

    async def serve(request: Request, chat: bool):
        bad = not_ready()
        if bad is not None:
            return bad
        try:
            body = await request.json()
        except Exception:
            return _error(400, "request body is not valid JSON")
        if chat:
            if not isinstance(body, dict) or not isinstance(body.get("messages"), list) or not body["messages"]:
                return _error(400, "missing field: messages")
            try:
                # a rendered template spells its special tokens out (Llama-3's begins with bos_token):
                # tokenised without the tokenizer's own post-processor, or the sequence would start
                # with two BOS ids
                rendered = render_chat(body["messages"], tokenizer)
                body = dict(body, prompt=tokenizer.encode(rendered, add_special_tokens=False))
            except (ValueError, TypeError, KeyError) as exc:
                return _error(400, f"cannot render the messages: {exc}")
            if not body["prompt"]:
                return _error(400, "the messages render to an empty prompt")
            if body.get("max_tokens") is None:
                body["max_tokens"] = body.get("max_completion_tokens") or max(
                    1, max_model_len - len(body["prompt"]) - 1)
            body.pop("echo", None)
            if body.get("logprobs") is True:              # chat form: a boolean + top_logprobs
                body["logprobs"] = int(body.get("top_logprobs") or 0)
            elif body.get("logprobs") is False:
                body["logprobs"] = None
        if not isinstance(body, dict) or "prompt" not in body:
            return _error(400, "missing field: prompt")
        model = body.get("model") or served_model_name
        if model != served_model_name:
            return _error(404, f"The model `{model}` does not exist.", "NotFoundError")
        try:
            base_prompts = normalise_prompts(body["prompt"], tokenizer, vocab_size)
            # n completions per prompt: n independent sequences (choice index = prompt index * n + j, as the
            # engine this replaces numbers them); with a seed, copy j draws from seed + j
            n = body.get("n", 1)
            n = 1 if n is None else n
            if isinstance(n, bool) or not isinstance(n, int) or n < 1 or n > MAX_N:
                raise ValueError(f"n must be an integer in [1, {MAX_N}]")
            if body.get("best_of") not in (None, n):
                raise ValueError("best_of other than n is not supported")
            if n * len(base_prompts) > MAX_CHOICES:
                raise ValueError(f"at most {MAX_CHOICES} completions per request (prompts x n)")
            prompts = [p for p in base_prompts for _ in range(n)]
            want_lp = body.get("logprobs")
            want_lp = None if want_lp is None or want_lp is False else int(want_lp)
            # echo + logprobs: the prompt is scored too (how evaluation harnesses compute log-likelihoods)
            prompt_lp = want_lp if want_lp is not None and body.get("echo") else None
            stop = body.get("stop") or []
            params = SamplingParams(
                max_tokens=int(body.get("max_tokens", 16) if body.get("max_tokens") is not None else 16),
                min_tokens=int(body.get("min_tokens", 0) or 0),
                temperature=float(body.get("temperature", 1.0) if body.get("temperature") is not None else 1.0),
                top_p=float(body.get("top_p", 1.0) if body.get("top_p") is not None else 1.0),
                top_k=int(body.get("top_k", -1) if body.get("top_k") is not None else -1),
                seed=body.get("seed"), stop=[stop] if isinstance(stop, str) else list(stop),
                stop_token_ids=list(body.get("stop_token_ids") or []),
                ignore_eos=bool(body.get("ignore_eos", False)), logprobs=want_lp,
                prompt_logprobs=prompt_lp)
            if n > 1 and params.seed is not None:
                copies = [dataclasses.replace(params, seed=int(params.seed) + j) for j in range(n)]
            else:
                copies = [params] * n
            for p in prompts:
                if len(p) + 1 > max_model_len:
                    raise ValueError(f"This model's maximum context length is {max_model_len} tokens; the prompt has "
                                     f"{len(p)} tokens.")
                if min(p) < 0 or max(p) >= vocab_size:
                    raise ValueError("token id outside the vocabulary")
        except (ValueError, TypeError) as exc:
            return _error(400, str(exc))
        if not params.greedy and not getattr(engine_thread.engine, "supports_random_sampling", False):
            return _error(400, "only greedy decoding (temperature=0) is available in this build")

        stream = bool(body.get("stream", False))
        echo = bool(body.get("echo", False))
        cmpl_id = ("chatcmpl-" if chat else "cmpl-") + uuid.uuid4().hex
        loop = asyncio.get_running_loop()
        q: asyncio.Queue = asyncio.Queue()
        rids = [f"{cmpl_id}-{i}" for i in range(len(prompts))]
        for i, (rid, p) in enumerate(zip(rids, prompts)):
            # the sink runs on this request's loop; a step's outputs for one loop arrive in one wake-up
            engine_thread.submit(rid, p, copies[i % n], lambda o, i=i: q.put_nowait((i, o)), loop=loop)

        detoks = [IncrementalDetokenizer(tokenizer) for _ in prompts]
        texts = [tokenizer.decode(p) if echo else "" for p in prompts]
        n_out = [0] * len(prompts)
        reasons: List[Optional[str]] = [None] * len(prompts)
        # OpenAI text-completion logprobs object per choice, filled token by token
        lp_objs = [{"tokens": [], "token_logprobs": [], "top_logprobs": [], "text_offset": []} if want_lp is not None
                   else None for _ in prompts]

        # return_tokens_as_token_ids: the logprobs object names tokens as "token_id:<id>" instead of by their
        # text (pieces of a multi-byte character, or two tokens with the same text, are otherwise ambiguous)
        as_ids = bool(body.get("return_tokens_as_token_ids", False))
        piece_of = (lambda t: f"token_id:{t}") if as_ids else (lambda t: tokenizer.decode([t]))

        def add_logprobs(i: int, o, offset: int) -> Optional[dict]:
            """Appends the step's tokens to choice i's logprobs object; returns the same fields for this step alone."""
            if lp_objs[i] is None or not o.logprobs:
                return None
            step = {"tokens": [], "token_logprobs": [], "top_logprobs": [], "text_offset": []}
            if getattr(o, "prompt_logprobs", None):           # first output of an echo + logprobs request
                at = 0
                for tok, ent in zip(prompts[i], o.prompt_logprobs):
                    piece = piece_of(tok)
                    step["tokens"].append(piece)
                    step["token_logprobs"].append(None if ent is None else ent[0])
                    if ent is None or not want_lp:
                        step["top_logprobs"].append(None)
                    else:                                     # the prompt token itself is always listed
                        top = {piece_of(t): v for t, v in ent[1]}
                        top.setdefault(piece, ent[0])
                        step["top_logprobs"].append(top)
                    step["text_offset"].append(at)
                    at += len(tokenizer.decode([tok]))
            for tok, (lp, top) in zip(o.new_token_ids, o.logprobs):
                step["tokens"].append(piece_of(tok))
                step["token_logprobs"].append(lp)
                step["top_logprobs"].append({piece_of(t): v for t, v in top} if want_lp else None)
                step["text_offset"].append(offset)
            for k2, v in step.items():
                lp_objs[i][k2].extend(v)
            return step

        def apply_stop(i: int) -> bool:
            """Cut text at the first stop string; True if one was hit."""
            for s in params.stop:
                at = texts[i].find(s, len(tokenizer.decode(prompts[i])) if echo else 0)
                if at >= 0:
                    texts[i] = texts[i][:at]
                    return True
            return False

        # characters a stream holds back while a stop string may still be forming across tokens
        hold = max((len(s) for s in params.stop), default=1) - 1

        async def events():
            done = 0
            sent = [len(t) if not echo else 0 for t in texts]
            while done < len(prompts):
                i, o = await q.get()
                if isinstance(o, Exception):
                    raise o
                if reasons[i] is not None:
                    continue
                n_out[i] = o.num_output_tokens
                step_lp = add_logprobs(i, o, len(texts[i]))
                # a sequence the engine ended on a token (end-of-sequence id, stop_token_ids) does not
                # show that token in its text; it still counts as a completion token
                shown = o.new_token_ids[:-1] if o.finished and o.finish_reason == "stop" else o.new_token_ids
                texts[i] += detoks[i].push(shown, final=o.finished)
                hit = bool(params.stop) and apply_stop(i)
                if hit:
                    reasons[i] = "stop"
                    engine_thread.abort(rids[i])
                elif o.finished:
                    reasons[i] = o.finish_reason
                if reasons[i] is not None:
                    done += 1
                upto = len(texts[i]) if reasons[i] is not None else max(len(texts[i]) - hold, sent[i])
                delta = texts[i][sent[i]:upto]
                sent[i] = upto
                yield i, delta, reasons[i], step_lp

        # a prompt counts once, however many completions it gets
        usage = lambda: {"prompt_tokens": sum(len(p) for p in base_prompts), "completion_tokens": sum(n_out),  # noqa: E731
                         "total_tokens": sum(len(p) for p in base_prompts) + sum(n_out)}

        started = [False] * len(prompts)
        if stream:
            async def sse():
                try:
                    async for i, delta, reason, step_lp in events():
                        # one event per generated token, as the engine this replaces streams them: the
                        # text may be empty while the detokenizer holds back an unfinished UTF-8 sequence
                        if chat:
                            d = {"content": delta}
                            if not started[i]:
                                started[i] = True
                                d["role"] = "assistant"
                            choice = {"index": i, "delta": d, "logprobs": step_lp, "finish_reason": reason}
                        else:
                            choice = {"index": i, "text": delta, "logprobs": step_lp, "finish_reason": reason}
                        chunk = {"id": cmpl_id, "object": "chat.completion.chunk" if chat else "text_completion",
                                 "created": int(time.time()), "model": served_model_name, "choices": [choice]}
                        yield f"data: {json.dumps(chunk)}\n\n"
                    yield "data: [DONE]\n\n"
                except asyncio.CancelledError:
                    for rid in rids:
                        engine_thread.abort(rid)
                    raise
                except Exception as exc:
                    yield f"data: {json.dumps({'error': {'message': str(exc), 'type': 'internal_error', 'code': 500}})}\n\n"
            return StreamingResponse(sse(), media_type="text/event-stream")

        try:
            async for _ in events():
                if await request.is_disconnected():
                    for rid in rids:
                        engine_thread.abort(rid)
                    return _error(499, "client disconnected")
        except ValueError as exc:
            return _error(400, str(exc))
        except Exception as exc:
            return _error(500, f"engine error: {exc!r}", "internal_error")
        if chat:
            choices = [{"index": i, "message": {"role": "assistant", "content": texts[i]}, "logprobs": lp_objs[i],
                        "finish_reason": reasons[i]} for i in range(len(prompts))]
        else:
            choices = [{"index": i, "text": texts[i], "logprobs": lp_objs[i], "finish_reason": reasons[i]}
                       for i in range(len(prompts))]
        return JSONResponse({"id": cmpl_id, "object": "chat.completion" if chat else "text_completion",
                             "created": int(time.time()), "model": served_model_name, "choices": choices, "usage": usage()})

    return app


def spawn_tp_workers(tp: int, argv: List[str]):
    """One process per GPU: the server process becomes rank 0 and starts ranks
    1..tp-1 as children (before anything touches the GPU)."""
    import socket
    import subprocess
    if tp <= 1 or "RANK" in os.environ:
        return []
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(tp), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(1, tp):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, "-m", "mi355_decode.entrypoints.openai.api_server"] + argv, env=env))
    os.environ.update(base, RANK="0", LOCAL_RANK="0")
    return procs


def main(argv: Optional[List[str]] = None) -> int:
    logging.basicConfig(level=os.environ.get("VLLM_LOGGING_LEVEL", "INFO").upper() if
                        os.environ.get("VLLM_LOGGING_LEVEL", "INFO").upper() in ("DEBUG", "INFO", "WARNING", "ERROR")
                        else "INFO", format="%(asctime)s %(levelname)s %(name)s: %(message)s")
    argv = list(sys.argv[1:] if argv is None else argv)
    # the engine thread and the HTTP loop share one interpreter: hand the lock over quickly, or a
    # decode step waits for the SSE writer's time slice
    sys.setswitchinterval(float(os.environ.get("MI355_GIL_SWITCH_S", "0.0005")))
    removed = sanitize_env()
    if removed:
        log.info("ignoring harness-set %s (GPU visibility comes from the device plugin)", removed)
    ns, args = parse_args(argv)
    workers = spawn_tp_workers(args.tensor_parallel_size, argv)
    rank = int(os.environ.get("RANK", "0"))

    from ...config import ModelConfig
    from ...llm import build_engine
    from ...serving import EngineThread
    from ...tokenizer import load_tokenizer

    cfg = ModelConfig.from_pretrained(args.model)
    served = args.served_model_name or args.model
    max_model_len = args.max_model_len or min(cfg.max_position_embeddings, 8192)
    tokenizer = load_tokenizer(args.model, cfg.vocab_size)
    # generation also ends on the tokenizer's own end-of-sequence token (tokenizer_config.json)
    from ...config import merge_eos
    cfg.eos_token_id = merge_eos(cfg.eos_token_id, getattr(tokenizer, "eos_token_id", None))
    if getattr(tokenizer, "is_fallback", False):
        log.warning("no tokenizer.json under %r: text prompts are mapped byte-wise (token-id prompts are exact)", args.model)

    control = None

    def make_engine():
        nonlocal control
        eng = build_engine(args, model_config=cfg)
        if args.tensor_parallel_size > 1:
            import torch.distributed as dist
            import datetime
            control = dist.new_group(backend="gloo", timeout=datetime.timedelta(
                seconds=int(os.environ.get("MI355_DIST_TIMEOUT_S", "300"))))
            thread.control_group = control
        return eng

    thread = EngineThread(make_engine, rank=rank)
    if rank != 0:                      # TP worker: no HTTP, just follow rank 0
        thread._run()
        return 1 if thread.failed else 0

    import uvicorn
    app = build_app(thread, served, tokenizer, cfg.vocab_size, max_model_len,
                    requested_dtype=(args.dtype or "auto"), served_dtype=args.resolved_dtype(cfg.torch_dtype))
    thread.start()
    config = uvicorn.Config(app, host=ns.host, port=ns.port, log_level="warning", access_log=False)
    server = uvicorn.Server(config)

    lost = []

    async def serve():
        task = asyncio.create_task(server.serve())
        while not task.done():
            await asyncio.sleep(0.5)
            if thread.failed is not None:            # fatal engine error: exit non-zero
                server.should_exit = True
            # a tensor-parallel worker that died leaves rank 0 blocked in its next collective with
            # /health still answering: exit non-zero instead, so the pod is restarted as a whole
            dead = [p.pid for p in workers if p.poll() is not None]
            if dead and not lost:
                lost.extend(dead)
                log.error("tensor-parallel worker process(es) %s exited; shutting the server down", dead)
                thread.failed = thread.failed or RuntimeError(f"tensor-parallel worker {dead} exited")
                server.should_exit = True
        await task

    asyncio.run(serve())
    if not lost:
        thread.shutdown()
    for p in workers:
        if p.poll() is None:
            p.terminate()
    return 1 if thread.failed is not None else 0


if __name__ == "__main__":
    sys.exit(main())
