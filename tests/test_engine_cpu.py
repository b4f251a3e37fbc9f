"""Host logic of the engine (page allocator, scheduler, chunked prefill,
preemption + recompute, stop conditions) on CPU, with a test double standing in
for the GPU model runner.

The double keeps a "KV cache" of token ids addressed exactly like the real one
(slot = block * 16 + offset, read back through block tables and context
lengths) and emits next = f(all tokens in the context).  Any bookkeeping error
-- wrong slot, stale block table, lost token after a preemption, bad chunk
boundary -- changes the generated ids, which are checked against a plain
sequential evaluation of f."""
import numpy as np
import pytest

from mi355_decode.engine import BlockAllocator, LLMEngine, NoFreeBlocks
from mi355_decode.sampling_params import SamplingParams

V = 1000


def next_token(history):
    h = 7
    for i, t in enumerate(history):
        h = (h * 31 + t * (i + 1) + 3) % 1000003
    return h % V


def reference(prompt, n, eos=None):
    out = []
    hist = list(prompt)
    for _ in range(n):
        t = next_token(hist)
        out.append(t)
        hist.append(t)
        if eos is not None and t == eos:
            break
    return out


class FakeRunner:
    device = "cpu"

    def __init__(self, num_blocks, max_num_seqs, max_model_len):
        self.num_blocks, self.max_num_seqs = num_blocks, max_num_seqs
        self.max_blocks = (max_model_len + 15) // 16
        self.V = V
        self.cache = np.full(num_blocks * 16, -1, dtype=np.int64)
        B = max_num_seqs
        self.h64s = [np.zeros(3 * B, dtype=np.int64) for _ in range(self.NUM_STAGES)]
        self.h32s = [np.zeros(B + B * self.max_blocks, dtype=np.int32) for _ in range(self.NUM_STAGES)]
        self.results = [None] * self.NUM_STAGES
        self.dev_next = []                    # "device" copy of the last sampled tokens
        self.decode_batches, self.prefill_tokens = [], []
        self.ahead_launches = 0
        self.sampling_log = []

    NUM_STAGES = 2

    def host_inputs(self, stage=0):
        return self.h64s[stage], self.h32s[stage]

    def _context(self, bt_row, ctx):
        idx = np.arange(ctx)
        toks = self.cache[bt_row[idx // 16].astype(np.int64) * 16 + idx % 16]
        assert (toks >= 0).all(), "read a slot that was never written"
        return toks.tolist()

    def launch_decode(self, n, stage=0, sampling=None, ids_from_device=False, logprobs=-1):
        self.logprobs_k = logprobs
        """Computes at once (the double has no queue) but hands the result over only in finish_decode."""
        self.last_sampling = sampling
        self.sampling_log.append(sampling)
        B = self.max_num_seqs
        h64, h32 = self.h64s[stage], self.h32s[stage]
        ids, pos, slots = h64[:B].copy(), h64[B:2 * B], h64[2 * B:]
        if ids_from_device:
            assert len(self.dev_next) >= n, "no previous step to take tokens from"
            ids[:n] = self.dev_next[:n]
            h64[:n] = -12345                  # the host copy of the ids must not be needed
            self.ahead_launches += 1
        ctx, bt = h32[:B], h32[B:].reshape(B, self.max_blocks)
        assert (slots[n:] == -1).all() and (ctx[n:] == 0).all(), "padding rows must be neutral"
        out = []
        for i in range(n):
            assert ctx[i] == pos[i] + 1
            self.cache[slots[i]] = ids[i]
            out.append(next_token(self._context(bt[i], int(ctx[i]))))
        self.decode_batches.append(n)
        self.dev_next = list(out)
        self.results[stage] = out

    def finish_decode(self, n, stage=0, logprobs=-1):
        out, self.results[stage] = self.results[stage], None
        assert out is not None and len(out) == n
        if logprobs >= 0:                     # fake log-probabilities: -token/1000, top-k = token, token+1, ...
            return out, [(-t / 1000.0, [(t + j, -(t + j) / 1000.0) for j in range(logprobs)]) for t in out]
        return out

    def run_prefill(self, token_ids, positions, slot_mapping, block_tables, cu_seqlens, context_lens, max_q_len,
                    sampling=None, logprobs=-1, prompt_logprobs=None):
        self.last_sampling = sampling
        assert len(token_ids) == len(positions) == len(slot_mapping) == cu_seqlens[-1]
        self.cache[slot_mapping] = token_ids
        out = []
        for i in range(len(context_lens)):
            qlen = cu_seqlens[i + 1] - cu_seqlens[i]
            assert qlen <= max_q_len
            # the chunk's positions are the last qlen positions of the context
            np.testing.assert_array_equal(positions[cu_seqlens[i]:cu_seqlens[i + 1]],
                                          np.arange(context_lens[i] - qlen, context_lens[i]))
            out.append(next_token(self._context(block_tables[i], int(context_lens[i]))))
        self.prefill_tokens.append(int(cu_seqlens[-1]))
        res = out
        if logprobs >= 0:
            res = out, [(-t / 1000.0, [(t + j, -(t + j) / 1000.0) for j in range(logprobs)]) for t in out]
        if prompt_logprobs is not None:       # fake scores: -(row's own token * 1000 + target) / 1e6
            k, rows, targets = prompt_logprobs
            assert len(rows) == len(targets) and (np.diff(rows) > 0).all() and rows[-1] < len(token_ids)
            self.scored_rows = getattr(self, "scored_rows", 0) + len(rows)
            score = lambda r, t: -(int(token_ids[r]) * 1000 + int(t)) / 1e6   # noqa: E731
            return res, [(score(r, t), [(int(t) + j, score(r, t) - j) for j in range(k)]) for r, t in zip(rows, targets)]
        return res


def run(engine):
    got = {}
    steps = 0
    while engine.has_unfinished():
        for o in engine.step():
            got.setdefault(o.request_id, []).extend(o.new_token_ids)
            if o.finished:
                got[o.request_id + ":reason"] = o.finish_reason
        steps += 1
        assert steps < 10000
    return got


def make_engine(num_blocks=64, max_num_seqs=4, max_model_len=256, max_tokens=64, eos=None, run_ahead=True):
    runner = FakeRunner(num_blocks, max_num_seqs, max_model_len)
    return LLMEngine(runner, max_num_seqs, max_tokens, max_model_len, eos_token_ids=eos, run_ahead=run_ahead), runner


def prompts(n, rng, lo=1, hi=60):
    return [rng.integers(0, V, size=int(rng.integers(lo, hi))).tolist() for _ in range(n)]


def test_block_allocator():
    a = BlockAllocator(4)
    x = a.allocate(3)
    assert sorted(x) == [0, 1, 2] and a.num_free == 1
    with pytest.raises(NoFreeBlocks):
        a.allocate(2)
    a.free(x)
    assert a.num_free == 4 and sorted(a.allocate(4)) == [0, 1, 2, 3]


def test_generation_matches_sequential_reference():
    rng = np.random.default_rng(0)
    eng, runner = make_engine()
    ps = prompts(10, rng)
    for i, p in enumerate(ps):
        eng.add_request(f"r{i}", p, SamplingParams(max_tokens=20, temperature=0.0))
    got = run(eng)
    for i, p in enumerate(ps):
        assert got[f"r{i}"] == reference(p, 20)
        assert got[f"r{i}:reason"] == "length"
    assert max(runner.decode_batches) == 4                      # max_num_seqs respected
    assert eng.alloc.num_free == eng.alloc.num_blocks            # every page returned


def test_chunked_prefill_and_token_budget():
    rng = np.random.default_rng(1)
    eng, runner = make_engine(max_tokens=16, max_model_len=512, num_blocks=128)
    ps = [rng.integers(0, V, size=n).tolist() for n in (100, 3, 47, 16, 17)]
    for i, p in enumerate(ps):
        eng.add_request(f"r{i}", p, SamplingParams(max_tokens=5, temperature=0.0))
    got = run(eng)
    for i, p in enumerate(ps):
        assert got[f"r{i}"] == reference(p, 5)
    assert max(runner.prefill_tokens) <= 16                      # never above max_num_batched_tokens


def test_preemption_recomputes_and_results_are_unchanged():
    rng = np.random.default_rng(2)
    # 3 sequences of up to 40+60 tokens need up to 21 pages; give 12 so decode runs out
    eng, runner = make_engine(num_blocks=12, max_num_seqs=3, max_model_len=128)
    ps = [rng.integers(0, V, size=40).tolist() for _ in range(3)]
    for i, p in enumerate(ps):
        eng.add_request(f"r{i}", p, SamplingParams(max_tokens=60, temperature=0.0))
    got = run(eng)
    assert eng.stats["preemptions"] > 0
    for i, p in enumerate(ps):
        assert got[f"r{i}"] == reference(p, 60)
    assert eng.alloc.num_free == 12


def test_stop_conditions():
    rng = np.random.default_rng(3)
    p = rng.integers(0, V, size=9).tolist()
    full = reference(p, 30)
    eos = full[4]
    first = full.index(eos)                       # eos may also occur earlier
    eng, _ = make_engine(eos=[eos])
    eng.add_request("a", p, SamplingParams(max_tokens=30, temperature=0.0))
    eng.add_request("b", p, SamplingParams(max_tokens=30, temperature=0.0, ignore_eos=True))
    eng.add_request("c", p, SamplingParams(max_tokens=30, temperature=0.0, stop_token_ids=[full[7]], ignore_eos=True))
    eng.add_request("d", p, SamplingParams(max_tokens=30, temperature=0.0, min_tokens=first + 2))
    got = run(eng)
    assert got["a"] == full[: first + 1] and got["a:reason"] == "stop"
    assert got["b"] == full and got["b:reason"] == "length"
    assert got["c"] == full[: full.index(full[7]) + 1] and got["c:reason"] == "stop"
    assert len(got["d"]) >= first + 2


def test_max_model_len_truncates_generation():
    eng, _ = make_engine(max_model_len=32)
    p = list(range(20))
    eng.add_request("a", p, SamplingParams(max_tokens=100, temperature=0.0))
    got = run(eng)
    assert len(got["a"]) == 12 and got["a:reason"] == "length"     # 20 + 12 == max_model_len


def test_request_validation():
    eng, _ = make_engine(max_model_len=32)
    with pytest.raises(ValueError, match="empty"):
        eng.add_request("a", [], SamplingParams())
    with pytest.raises(ValueError, match="max_model_len"):
        eng.add_request("a", list(range(32)), SamplingParams())
    with pytest.raises(ValueError, match="vocabulary"):
        eng.add_request("a", [V], SamplingParams())
    eng.add_request("a", [1], SamplingParams())
    with pytest.raises(ValueError, match="duplicate"):
        eng.add_request("a", [1], SamplingParams())
    with pytest.raises(ValueError):
        SamplingParams(max_tokens=0)
    with pytest.raises(ValueError):
        SamplingParams(top_p=0.0)


def test_abort_frees_pages():
    eng, _ = make_engine()
    eng.add_request("a", list(range(40)), SamplingParams(max_tokens=50, temperature=0.0))
    eng.add_request("b", list(range(10)), SamplingParams(max_tokens=5, temperature=0.0))
    early = []
    for _ in range(2):
        early += [t for o in eng.step() if o.request_id == "b" for t in o.new_token_ids]
    eng.abort_request("a")
    got = run(eng)
    assert early + got["b"] == reference(list(range(10)), 5)
    assert eng.alloc.num_free == eng.alloc.num_blocks


def test_late_arrivals_join_running_batch():
    rng = np.random.default_rng(4)
    eng, runner = make_engine(max_num_seqs=8)
    ps = prompts(6, rng)
    got = {}
    for i, p in enumerate(ps):
        eng.add_request(f"r{i}", p, SamplingParams(max_tokens=12, temperature=0.0))
        for o in eng.step():                       # one step between arrivals
            got.setdefault(o.request_id, []).extend(o.new_token_ids)
    for k, v in run(eng).items():
        if not k.endswith(":reason"):
            got.setdefault(k, []).extend(v)
    for i, p in enumerate(ps):
        assert got[f"r{i}"] == reference(p, 12)


def test_sampling_parameters_reach_the_runner():
    eng, runner = make_engine()
    eng.add_request("g", [1, 2, 3], SamplingParams(max_tokens=3, temperature=0.0))
    eng.step()
    assert runner.last_sampling is None                       # all greedy: no sampler launch
    eng.add_request("r", [4, 5], SamplingParams(max_tokens=3, temperature=0.7, top_k=5, top_p=0.9, seed=42))
    eng.step()                                                # prefill of "r"
    s = runner.last_sampling
    assert s["temperature"].tolist() == [np.float32(0.7)] and s["top_k"].tolist() == [5] and s["seeds"].tolist() == [42]
    eng.step()                                                # decode of both, and the step after it launched ahead
    s, s_ahead = runner.sampling_log[-2:]
    assert s["temperature"].tolist() == [0.0, np.float32(0.7)] and s["steps"].tolist() == [1, 1]
    assert s_ahead["steps"].tolist() == [2, 2] and s_ahead["seeds"].tolist() == s["seeds"].tolist()
    eng.add_request("u", [9], SamplingParams(max_tokens=2, temperature=1.0))      # no seed: engine draws one
    assert 0 <= eng.seqs["u"].seed < 2 ** 62


def test_run_ahead_launches_no_step_that_length_limits_make_useless():
    """One sequence, 20 tokens: the prefill samples the first, 19 decode steps the rest.
    All but the first decode step are launched ahead of the host, and the engine does
    not launch a 20th (it knows the sequence ends by length)."""
    eng, runner = make_engine()
    p = list(range(7))
    eng.add_request("a", p, SamplingParams(max_tokens=20, temperature=0.0))
    got = run(eng)
    assert got["a"] == reference(p, 20)
    assert len(runner.decode_batches) == 19 and runner.ahead_launches == 18
    assert eng.stats["decode_steps"] == 19 and eng.stats["decode_tokens"] == 19


def test_run_ahead_drops_the_token_computed_past_a_stop_token():
    rng = np.random.default_rng(5)
    p = rng.integers(0, V, size=11).tolist()
    full = reference(p, 30)
    stop = full[6]
    cut = full.index(stop)
    eng, runner = make_engine()
    eng.add_request("a", p, SamplingParams(max_tokens=30, temperature=0.0, stop_token_ids=[stop]))
    eng.add_request("b", p[:5], SamplingParams(max_tokens=30, temperature=0.0))
    got = run(eng)
    assert got["a"] == full[: cut + 1] and got["a:reason"] == "stop"
    assert got["b"] == reference(p[:5], 30)                     # the other row of those steps is unaffected
    assert eng.alloc.num_free == eng.alloc.num_blocks
    assert eng.stats["decode_tokens"] == cut + 29               # the dropped token is not counted


def test_run_ahead_on_and_off_generate_the_same_tokens():
    rng = np.random.default_rng(6)
    ps = prompts(9, rng)
    outs = []
    for ahead in (True, False):
        eng, runner = make_engine(num_blocks=20, max_num_seqs=3, max_model_len=128, run_ahead=ahead)   # tight pool: preemptions
        for i, p in enumerate(ps):
            eng.add_request(f"r{i}", p, SamplingParams(max_tokens=25, temperature=0.0))
        outs.append(run(eng))
        assert (runner.ahead_launches > 0) == ahead
        assert eng.alloc.num_free == 20
    assert outs[0] == outs[1]
    for i, p in enumerate(ps):
        assert outs[0][f"r{i}"] == reference(p, 25)


def test_abort_while_a_step_is_in_flight():
    eng, _ = make_engine()
    eng.add_request("a", list(range(30)), SamplingParams(max_tokens=40, temperature=0.0))
    eng.add_request("b", list(range(12)), SamplingParams(max_tokens=10, temperature=0.0))
    early = []
    for _ in range(4):                                  # 2 prefills (token budget), then decode steps: one is now in flight
        early += [t for o in eng.step() if o.request_id == "b" for t in o.new_token_ids]
    assert eng._pending
    eng.abort_request("a")
    got = run(eng)
    assert early + got["b"] == reference(list(range(12)), 10)
    assert "a" not in got and eng.alloc.num_free == eng.alloc.num_blocks


def test_logprobs_follow_each_request_own_width():
    """A step computes the widest top-k any of its requests asked for; every request gets its own width
    (and none at all if it did not ask), for the prefill token and for tokens of steps launched ahead."""
    eng, runner = make_engine()
    eng.add_request("a", [1, 2, 3], SamplingParams(max_tokens=4, temperature=0.0, logprobs=3))
    eng.add_request("b", [4, 5], SamplingParams(max_tokens=4, temperature=0.0, logprobs=0))
    eng.add_request("c", [6], SamplingParams(max_tokens=4, temperature=0.0))
    seen = {"a": [], "b": [], "c": []}
    toks = {"a": [], "b": [], "c": []}
    while eng.has_unfinished():
        for o in eng.step():
            toks[o.request_id] += o.new_token_ids
            seen[o.request_id].append(o.logprobs)
    assert runner.logprobs_k == 3
    assert all(x is None for x in seen["c"])
    for lp, t in zip(seen["a"], toks["a"]):
        assert lp == [(-t / 1000.0, [(t + j, -(t + j) / 1000.0) for j in range(3)])]
    for lp, t in zip(seen["b"], toks["b"]):
        assert lp == [(-t / 1000.0, [])]
    assert len(seen["a"]) == 4 and toks["a"] == reference([1, 2, 3], 4)


def test_prompt_logprobs_cover_every_prompt_token_once_across_chunks():
    """echo + logprobs: every prompt token but the first is scored exactly once, by the row before
    it, however the prompt is cut into prefill chunks; the scores arrive with the first token."""
    eng, runner = make_engine(num_blocks=64, max_num_seqs=4, max_model_len=256, max_tokens=7)   # 7-token chunks
    rng = np.random.default_rng(5)
    ps = [rng.integers(0, V, size=n).tolist() for n in (1, 7, 8, 23)]
    for i, p in enumerate(ps):
        eng.add_request(f"r{i}", p, SamplingParams(max_tokens=3, temperature=0.0, logprobs=2 if i % 2 else None,
                                                   prompt_logprobs=2 if i != 2 else None))
    first = {}
    while eng.has_unfinished():
        for o in eng.step():
            if o.request_id not in first:
                first[o.request_id] = o
            else:
                assert o.prompt_logprobs is None          # only on the first output
    for i, p in enumerate(ps):
        o = first[f"r{i}"]
        if i == 2:
            assert o.prompt_logprobs is None
            continue
        plp = o.prompt_logprobs
        assert len(plp) == len(p) and plp[0] is None
        for j in range(1, len(p)):
            lp, top = plp[j]
            assert lp == -(p[j - 1] * 1000 + p[j]) / 1e6       # scored by the row of token j-1 against token j
            assert [t for t, _ in top] == [p[j], p[j] + 1]
    assert runner.scored_rows == sum(len(p) - 1 for i, p in enumerate(ps) if i != 2)
