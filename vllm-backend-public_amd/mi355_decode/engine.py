"""Continuous-batching engine: page allocator, scheduler and the step loop.

Mirrors the role of the engine the harness launches (LLMEngine of the image at
benchmark-vllm/vllm_manager.py:371-375): requests are admitted up to
``max_num_seqs`` (--max-num-seqs, vllm_templates.py:53), their KV pages come from
a pool of 16-token blocks (--block-size, vllm_templates.py:54), prompts are
prefilled (in chunks of at most ``max_num_batched_tokens``) and then every
running sequence advances one token per decode step.  When the pool runs dry the
youngest running sequence is preempted and later recomputed.

Decode steps are launched asynchronously and the engine keeps one step ahead of
the host: the step after the one in flight takes its input tokens on the device,
so the GPU never waits for the host to read tokens, run the scheduler and stage
inputs.  A sequence that stops on a token (EOS, stop id) is noticed one step
late; the extra token computed for it is dropped.

Under tensor parallelism every rank runs this same deterministic state machine
on the same request stream, so all ranks issue identical steps and identical
collectives without exchanging per-step metadata.
"""
from __future__ import annotations

import time
from collections import deque
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from ._lib import BLOCK_SIZE
from .sampling_params import SamplingParams


class NoFreeBlocks(Exception):
    pass


class BlockAllocator:
    """Free list of KV pages (block ids 0 .. num_blocks-1)."""

    def __init__(self, num_blocks: int):
        self.num_blocks = num_blocks
        self._free = list(range(num_blocks - 1, -1, -1))

    @property
    def num_free(self) -> int:
        return len(self._free)

    def allocate(self, n: int) -> List[int]:
        if n > len(self._free):
            raise NoFreeBlocks(f"need {n} blocks, {len(self._free)} free")
        out = [self._free.pop() for _ in range(n)]
        return out

    def free(self, blocks: List[int]) -> None:
        self._free.extend(reversed(blocks))


@dataclass(eq=False)          # identity semantics: sequences live in scheduler lists
class Sequence:
    request_id: str
    prompt_ids: List[int]
    params: SamplingParams
    arrival_time: float = field(default_factory=time.perf_counter)
    output_ids: List[int] = field(default_factory=list)
    blocks: List[int] = field(default_factory=list)
    num_computed: int = 0            # tokens whose K/V are in the cache
    first_token_time: Optional[float] = None
    finish_time: Optional[float] = None
    finish_reason: Optional[str] = None
    num_preemptions: int = 0
    seed: int = 0                    # sampling seed (request's, or drawn by the engine)
    inflight: int = 0                # decode steps launched for this sequence and not yet harvested
    # requests with params.prompt_logprobs: entry j scores prompt token j (None for j = 0), filled
    # chunk by chunk as the prompt is prefilled
    prompt_logprobs: Optional[list] = None

    @property
    def all_ids(self) -> List[int]:
        return self.prompt_ids + self.output_ids

    @property
    def total_len(self) -> int:
        return len(self.prompt_ids) + len(self.output_ids)

    @property
    def finished(self) -> bool:
        return self.finish_reason is not None


@dataclass
class RequestOutput:
    request_id: str
    new_token_ids: List[int]
    finished: bool
    finish_reason: Optional[str]
    num_prompt_tokens: int
    num_output_tokens: int
    arrival_time: float
    first_token_time: Optional[float]
    finish_time: Optional[float]
    # one entry per new token when the request asked for logprobs: (logprob of the token,
    # [(token id, logprob)] of the most likely tokens, most likely first)
    logprobs: Optional[list] = None
    # on the output that carries the first generated token of a request with prompt_logprobs: one entry
    # per prompt token, None for the first, then (logprob, [(token id, logprob)] most likely first)
    prompt_logprobs: Optional[list] = None


class Scheduler:
    def __init__(self, allocator: BlockAllocator, max_num_seqs: int, max_num_batched_tokens: int,
                 max_model_len: int):
        self.alloc = allocator
        self.max_num_seqs = max_num_seqs
        self.max_tokens = max_num_batched_tokens
        self.max_model_len = max_model_len
        self.waiting: deque = deque()
        self.running: List[Sequence] = []

    def add(self, seq: Sequence) -> None:
        self.waiting.append(seq)

    def abort(self, request_id: str) -> Optional[Sequence]:
        for seq in list(self.waiting):
            if seq.request_id == request_id:
                self.waiting.remove(seq)
                return seq
        for seq in self.running:
            if seq.request_id == request_id:
                self.release(seq)
                return seq
        return None

    def release(self, seq: Sequence) -> None:
        if seq in self.running:
            self.running.remove(seq)
        self.alloc.free(seq.blocks)
        seq.blocks = []

    @staticmethod
    def prefill_target(seq: Sequence) -> int:
        """Tokens whose K/V must be in the cache before the sequence can decode:
        the whole prompt for a new sequence (the prefill step samples its first
        token); everything but the last token for a preempted one (recompute)."""
        return seq.total_len if not seq.output_ids else seq.total_len - 1

    @staticmethod
    def _blocks_for(n_tokens: int) -> int:
        return (n_tokens + BLOCK_SIZE - 1) // BLOCK_SIZE

    def _grow(self, seq: Sequence, n_tokens: int) -> None:
        need = self._blocks_for(n_tokens) - len(seq.blocks)
        if need > 0:
            seq.blocks.extend(self.alloc.allocate(need))

    def _preempt_youngest(self, keep: Sequence) -> bool:
        """Free the most recently admitted running sequence (recompute later)."""
        for victim in reversed(self.running):
            if victim is not keep:
                self.release(victim)
                victim.num_computed = 0
                victim.num_preemptions += 1
                self.waiting.appendleft(victim)
                return True
        return False

    def schedule(self):
        """Returns ("prefill", [(seq, start, n_tokens)...]) or ("decode", [seq...])
        or None when idle.  Prefill has priority (lowest time-to-first-token)."""
        # 1. sequences with uncomputed prompt tokens: running ones first, then new
        budget = self.max_tokens
        chunks = []
        for seq in self.running:
            todo = self.prefill_target(seq) - seq.num_computed
            if todo > 0 and budget > 0:
                n = min(todo, budget)
                chunks.append((seq, seq.num_computed, n))
                budget -= n
        while self.waiting and budget > 0 and len(self.running) < self.max_num_seqs:
            seq = self.waiting[0]
            # a sequence is admitted only if all its tokens fit in the pool right now
            if self._blocks_for(seq.total_len + 1) > self.alloc.num_free:
                break
            self.waiting.popleft()
            self._grow(seq, seq.total_len + 1)
            self.running.append(seq)
            n = min(self.prefill_target(seq), budget)
            chunks.append((seq, 0, n))
            budget -= n
        if chunks:
            return "prefill", chunks
        if not self.running:
            return None
        # 2. decode: every running sequence needs a slot for the token it is about to store
        for seq in list(self.running):
            if seq not in self.running:
                continue
            while True:
                try:
                    self._grow(seq, seq.total_len)
                    break
                except NoFreeBlocks:
                    if not self._preempt_youngest(keep=seq):
                        raise
        return "decode", list(self.running)


class LLMEngine:
    """Single-GPU (or one TP rank's) engine.  `runner` is a ModelRunner."""

    supports_random_sampling = True

    def __init__(self, runner, max_num_seqs: int, max_num_batched_tokens: int, max_model_len: int,
                 eos_token_ids=None, seed: int = 0, run_ahead: bool = True):
        import random
        self._rng = random.Random(seed)          # same stream on every TP rank
        self.runner = runner
        self.device = runner.device
        self.max_model_len = max_model_len
        self.alloc = BlockAllocator(runner.num_blocks)
        self.sched = Scheduler(self.alloc, max_num_seqs, max_num_batched_tokens, max_model_len)
        self.eos = set(eos_token_ids or [])
        self.seqs: Dict[str, Sequence] = {}
        # host mirrors (numpy views of pinned memory) of the runner's packed decode inputs,
        # one per staging set
        self._hosts = [runner.host_inputs(i) for i in range(getattr(runner, "NUM_STAGES", 1))]
        self._stage = 0
        self._pending: deque = deque()        # decode steps launched and not yet harvested: (seqs, stage)
        self.run_ahead = run_ahead and len(self._hosts) > 1
        self.stats = {"prefill_steps": 0, "decode_steps": 0, "prefill_tokens": 0, "decode_tokens": 0,
                      "preemptions": 0}

    # ------------------------------------------------------------------ #
    def add_request(self, request_id: str, prompt_token_ids: List[int], params: SamplingParams,
                    arrival_time: Optional[float] = None) -> None:
        if request_id in self.seqs:
            raise ValueError(f"duplicate request id {request_id}")
        if not prompt_token_ids:
            raise ValueError("empty prompt")
        if len(prompt_token_ids) + 1 > self.max_model_len:
            raise ValueError(f"prompt of {len(prompt_token_ids)} tokens does not fit max_model_len "
                             f"{self.max_model_len}")
        if self.sched._blocks_for(min(len(prompt_token_ids) + params.max_tokens, self.max_model_len)) \
                > self.alloc.num_blocks:
            raise ValueError("request cannot fit in the KV cache even alone")
        bad = [t for t in prompt_token_ids if not 0 <= t < self.runner.V]
        if bad:
            raise ValueError(f"token id {bad[0]} outside the vocabulary (size {self.runner.V})")
        seq = Sequence(request_id, list(prompt_token_ids), params)
        seq.seed = params.seed if params.seed is not None else self._rng.getrandbits(62)
        if arrival_time is not None:
            seq.arrival_time = arrival_time
        self.seqs[request_id] = seq
        self.sched.add(seq)

    def abort_request(self, request_id: str) -> None:
        seq = self.seqs.pop(request_id, None)
        if seq is not None:
            self.sched.abort(request_id)
            seq.finish_reason = seq.finish_reason or "abort"

    def has_unfinished(self) -> bool:
        return bool(self.sched.waiting or self.sched.running or self._pending)

    # ------------------------------------------------------------------ #
    def _append_token(self, seq: Sequence, tok: int, now: float) -> None:
        seq.output_ids.append(tok)
        if seq.first_token_time is None:
            seq.first_token_time = now
        p = seq.params
        n_out = len(seq.output_ids)
        if n_out >= p.min_tokens:
            if (not p.ignore_eos and tok in self.eos) or tok in p.stop_token_ids:
                seq.finish_reason = "stop"
        if seq.finish_reason is None and n_out >= p.max_tokens:
            seq.finish_reason = "length"
        if seq.finish_reason is None and seq.total_len >= self.max_model_len:
            seq.finish_reason = "length"
        if seq.finished:
            seq.finish_time = now
            self.sched.release(seq)
            self.seqs.pop(seq.request_id, None)

    @staticmethod
    def _sampling_arrays(seqs: List[Sequence]) -> Optional[dict]:
        """Per-row sampling parameters, or None when every row is greedy."""
        if all(s.params.greedy for s in seqs):
            return None
        return {
            "temperature": np.asarray([s.params.temperature for s in seqs], dtype=np.float32),
            "top_k": np.asarray([s.params.top_k for s in seqs], dtype=np.int32),
            "top_p": np.asarray([s.params.top_p for s in seqs], dtype=np.float32),
            "seeds": np.asarray([s.seed & 0x7FFFFFFFFFFFFFFF for s in seqs], dtype=np.int64),
            "steps": np.asarray([len(s.output_ids) + s.inflight for s in seqs], dtype=np.int64),
        }

    def _output(self, seq: Sequence, new: List[int], lp=None) -> RequestOutput:
        k = seq.params.logprobs
        if lp is not None and k is not None:
            lp = [(lp[0], lp[1][:k])]
        else:
            lp = None
        return RequestOutput(seq.request_id, new, seq.finished, seq.finish_reason, len(seq.prompt_ids),
                             len(seq.output_ids), seq.arrival_time, seq.first_token_time, seq.finish_time, lp)

    @staticmethod
    def _logprobs_k(seqs: List[Sequence]) -> int:
        """Top-k width a step has to compute: the largest any of its requests asked for; -1: none did."""
        return max([s.params.logprobs for s in seqs if s.params.logprobs is not None], default=-1)

    def _prefill(self, chunks) -> List[RequestOutput]:
        MB = self.runner.max_blocks
        toks, pos, slots, cu, ctx = [], [], [], [0], []
        bt = np.zeros((len(chunks), MB), dtype=np.int32)
        for i, (seq, start, n) in enumerate(chunks):
            ids = seq.all_ids[start:start + n]
            blocks = np.asarray(seq.blocks, dtype=np.int64)
            p = np.arange(start, start + n, dtype=np.int64)
            toks.append(np.asarray(ids, dtype=np.int64))
            pos.append(p)
            slots.append(blocks[p // BLOCK_SIZE] * BLOCK_SIZE + p % BLOCK_SIZE)
            cu.append(cu[-1] + n)
            ctx.append(start + n)
            bt[i, : len(seq.blocks)] = seq.blocks
        k = self._logprobs_k([c[0] for c in chunks])
        kw = {"logprobs": k} if k >= 0 else {}
        # prompt scoring: row r of the packed batch predicts the prompt token after it
        rows, targets, owners = [], [], []
        for i, (seq, start, n) in enumerate(chunks):
            if seq.params.prompt_logprobs is None:
                continue
            if seq.prompt_logprobs is None:
                seq.prompt_logprobs = [None]
            first = max(start, len(seq.prompt_logprobs) - 1)
            last = min(start + n, len(seq.prompt_ids) - 1)
            for p_ in range(first, last):
                rows.append(cu[i] + p_ - start)
                targets.append(seq.prompt_ids[p_ + 1])
                owners.append(seq)
        if rows:
            kw["prompt_logprobs"] = (max(c[0].params.prompt_logprobs or 0 for c in chunks),
                                     np.asarray(rows, dtype=np.int64), np.asarray(targets, dtype=np.int64))
        nxt = self.runner.run_prefill(np.concatenate(toks), np.concatenate(pos), np.concatenate(slots), bt,
                                      np.asarray(cu, dtype=np.int32), np.asarray(ctx, dtype=np.int32),
                                      max(n for _, _, n in chunks),
                                      sampling=self._sampling_arrays([c[0] for c in chunks]), **kw)
        if rows:                                     # (the usual result, one (logprob, top) per scored row)
            nxt, plps = nxt
            for seq, (lp, top) in zip(owners, plps):
                seq.prompt_logprobs.append((lp, top[: seq.params.prompt_logprobs]))
        nxt, lps = nxt if k >= 0 else (nxt, None)
        now = time.perf_counter()
        outs = []
        self.stats["prefill_steps"] += 1
        for i, (seq, start, n) in enumerate(chunks):
            seq.num_computed = start + n
            self.stats["prefill_tokens"] += n
            if seq.num_computed >= seq.total_len:        # prompt complete: this step sampled a token
                self._append_token(seq, int(nxt[i]), now)
                out = self._output(seq, [seq.output_ids[-1]], lps[i] if lps else None)
                if seq.prompt_logprobs is not None and len(seq.output_ids) == 1:
                    out.prompt_logprobs = seq.prompt_logprobs
                outs.append(out)
        return outs

    # ------------------------------------------------------------------ #
    # decode: launched asynchronously, at most one step ahead of the host
    # ------------------------------------------------------------------ #
    def _launch_decode(self, seqs: List[Sequence], ahead: bool) -> None:
        """Stages and enqueues one decode step.  `ahead`: the step before it is still
        in flight, so the input tokens are taken on the device from that step's
        samples; everything else a step needs (positions, slots, pages) does not
        depend on token values and is staged from the host as usual."""
        r = self.runner
        B, MB = r.max_num_seqs, r.max_blocks
        n = len(seqs)
        st = self._stage
        self._stage = (st + 1) % len(self._hosts)
        n64, n32 = self._hosts[st]
        n64[2 * B:] = -1                      # neutral padding: store nothing,
        n32[:B] = 0                           # attend to nothing
        bt = n32[B:].reshape(B, MB)
        for i, seq in enumerate(seqs):
            p = seq.total_len - 1 + seq.inflight   # position of the newest token (not yet in the cache)
            if not ahead:
                n64[i] = seq.output_ids[-1] if seq.output_ids else seq.prompt_ids[-1]
            n64[B + i] = p
            n64[2 * B + i] = seq.blocks[p // BLOCK_SIZE] * BLOCK_SIZE + p % BLOCK_SIZE
            n32[i] = p + 1
            bt[i, : len(seq.blocks)] = seq.blocks
        k = self._logprobs_k(seqs)
        kw = {"logprobs": k} if k >= 0 else {}
        r.launch_decode(n, st, sampling=self._sampling_arrays(seqs), ids_from_device=ahead, **kw)
        for seq in seqs:
            seq.inflight += 1
        self._pending.append((seqs, st, k))

    def _try_launch_ahead(self, seqs: List[Sequence]) -> None:
        """Launch the step after the one in flight if its batch is certain to be the
        same rows: nobody waiting for admission, nobody ending by length at the step
        in flight, a page available for every new token.  A row that ends on a stop
        token at the step in flight computes one token too many, which is dropped."""
        if not self.run_ahead or self.sched.waiting or len(seqs) != len(self.sched.running):
            return
        for s in seqs:
            if (not s.blocks or len(s.output_ids) + s.inflight >= s.params.max_tokens
                    or s.total_len + s.inflight >= self.max_model_len):
                return
        try:
            for s in seqs:
                self.sched._grow(s, s.total_len + s.inflight)
        except NoFreeBlocks:
            return
        self._launch_decode(seqs, ahead=True)

    def _harvest_decode(self) -> List[RequestOutput]:
        seqs, st, k = self._pending.popleft()
        if k >= 0:
            toks, lps = self.runner.finish_decode(len(seqs), st, logprobs=k)
        else:
            toks, lps = self.runner.finish_decode(len(seqs), st), None
        now = time.perf_counter()
        self.stats["decode_steps"] += 1
        outs = []
        for i, (seq, tok) in enumerate(zip(seqs, toks)):
            seq.inflight -= 1
            if not seq.blocks:                # finished, aborted or preempted while the step was in flight
                continue
            self.stats["decode_tokens"] += 1
            seq.num_computed = seq.total_len
            self._append_token(seq, int(tok), now)
            outs.append(self._output(seq, [seq.output_ids[-1]], lps[i] if lps else None))
        return outs

    def step(self) -> List[RequestOutput]:
        if not self._pending:
            before = sum(s.num_preemptions for s in self.sched.running) + sum(s.num_preemptions for s in self.sched.waiting)
            plan = self.sched.schedule()
            if plan is None:
                return []
            after = sum(s.num_preemptions for s in self.sched.running) + sum(s.num_preemptions for s in self.sched.waiting)
            self.stats["preemptions"] += after - before
            kind, work = plan
            if kind == "prefill":
                return self._prefill(work)
            self._launch_decode(work, ahead=False)
        if len(self._pending) == 1:
            self._try_launch_ahead(self._pending[0][0])
        return self._harvest_decode()
