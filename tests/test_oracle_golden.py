"""The numpy oracle (oracle/ops_ref.py, oracle/model_ref.py) against the HF
transformers vectors in tests/golden/ (written by oracle/gen_golden.py).
CPU only.  These tests pin the oracle; the GPU tests then pin the HIP path to
the oracle."""
import numpy as np
import pytest

import os

from conftest import GOLDEN, load_npz
from oracle import model_ref as M
from oracle import ops_ref as R
from oracle.tiny_models import TINY, make_prompts, make_weights

BF16_ULP = 2.0 ** -7          # one bf16 ulp relative to the value


def close_dtype(got, ref, dtype, max_ulp_frac=0.002):
    """fp32: tight tolerance.  bf16: identical up to single-ulp flips caused by
    summation order in the reduction (allowed on a small fraction of elements)."""
    if dtype == "float32":
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6)
        return
    diff = np.abs(got - ref)
    tol = BF16_ULP * np.maximum(np.abs(ref), 2.0 ** -126) * 1.01
    assert (diff <= tol).all(), f"max diff {diff.max()}"
    assert (diff > 0).mean() <= max_ulp_frac, f"{(diff > 0).mean():.4f} of elements differ by one ulp"


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
def test_rms_norm(ops_golden, tag, dtype):
    g = ops_golden
    x, w, res, eps = g[f"rms_{tag}_x"], g[f"rms_{tag}_w"], g[f"rms_{tag}_res"], float(g[f"rms_{tag}_eps"])
    close_dtype(R.rms_norm(x, w, eps, dtype), g[f"rms_{tag}_y_{dtype}"], dtype)
    y, z = R.fused_add_rms_norm(x, res, w, eps, dtype)
    np.testing.assert_array_equal(z, g[f"rms_{tag}_z_{dtype}"])
    close_dtype(y, g[f"rms_{tag}_yz_{dtype}"], dtype)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
def test_layer_norm(tag, dtype):
    """torch.nn.LayerNorm (the norm of HF's OPT layers), alone and after the residual add."""
    g = load_npz("ops_opt_hf.npz")
    x, res, w, b = (g[f"ln_{tag}_{k}"] for k in ("x", "res", "w", "b"))
    close_dtype(R.layer_norm(x, w, b, 1e-5, dtype), g[f"ln_{tag}_y_{dtype}"], dtype)
    y, z = R.fused_add_layer_norm(x, res, w, b, 1e-5, dtype)
    np.testing.assert_array_equal(z, g[f"ln_{tag}_z_{dtype}"])
    close_dtype(y, g[f"ln_{tag}_yz_{dtype}"], dtype)


def test_relu_and_learned_positions():
    g = load_npz("ops_opt_hf.npz")
    got = R.relu(g["relu_x"])
    np.testing.assert_array_equal(got, g["relu_y"])
    np.testing.assert_array_equal(np.signbit(got), np.signbit(g["relu_y"]))     # torch keeps -0.0
    for dtype in ("bfloat16", "float32"):
        np.testing.assert_array_equal(
            R.embed_positions(g["pos_embed"], g["pos_table"], g["pos_ids"], g["pos_positions"], 2, dtype),
            g[f"pos_sum_{dtype}"])


@pytest.mark.parametrize("tag,d", [("d64", 64), ("d128", 128), ("q128", 128)])
@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
def test_rope(ops_golden, tag, d, dtype):
    g = ops_golden
    pos, theta = g[f"rope_{tag}_pos"], float(g[f"rope_{tag}_theta"])
    table = R.rope_cos_sin_table(4096, d, theta, dtype)
    # HF duplicates the half table: cos = cat(freqs, freqs).cos()
    hf_tab = np.concatenate([g[f"rope_{tag}_cos_{dtype}"][:, : d // 2],
                             g[f"rope_{tag}_sin_{dtype}"][:, : d // 2]], axis=1)
    # the table is host-side set-up; numpy and torch libm differ in the last
    # fp32 bit of pow/sin/cos at large arguments, which can flip one bf16 ulp
    diff = np.abs(table[pos] - hf_tab)
    assert diff.max() <= (2.0 ** -8 if dtype == "bfloat16" else 5e-4)   # 1 fp32 ulp of inv_freq x pos 4095
    assert (diff > 0).mean() <= (0.02 if dtype == "bfloat16" else 1.0)
    # the rotation itself is pinned exactly, using HF's own cos/sin rows
    for name in ("q", "k"):
        got = R.rope_neox(g[f"rope_{tag}_{name}"], np.arange(len(pos)), hf_tab, dtype)
        ref = g[f"rope_{tag}_{name}o_{dtype}"]
        if dtype == "bfloat16":
            np.testing.assert_array_equal(got, ref)
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)


def _paged(k, v, block_size=16, shuffle_seed=0):
    """Scatter contiguous K/V [L, kvh, d] into a shuffled paged cache."""
    L, kvh, d = k.shape
    nblk = (L + block_size - 1) // block_size
    total = nblk + 3
    perm = np.random.Generator(np.random.PCG64(shuffle_seed)).permutation(total)[:nblk].astype(np.int32)
    kc = np.full((total, kvh, block_size, d), 7.0, np.float32)   # garbage outside the context
    vc = np.full((total, kvh, d, block_size), -7.0, np.float32)
    slots = perm[np.arange(L) // block_size].astype(np.int64) * block_size + np.arange(L) % block_size
    R.kv_write(k, v, slots, kc, vc)
    return kc, vc, perm


@pytest.mark.parametrize("tag", ["g2d64", "g4d128", "g7d64", "g8d128"])
def test_attention_decode(ops_golden, tag):
    g = ops_golden
    for L in g[f"attn_{tag}_ctx"]:
        q, k, v = (g[f"attn_{tag}_dec{L}_{n}"] for n in "qkv")
        kc, vc, bt = _paged(k, v, shuffle_seed=int(L))
        got = R.paged_attn_decode(q, kc, vc, bt[None], np.array([L], np.int32), q.shape[-1] ** -0.5, "float32")
        np.testing.assert_allclose(got, g[f"attn_{tag}_dec{L}_o"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("tag", ["g2d64", "g4d128", "g7d64", "g8d128"])
def test_attention_prefill(ops_golden, tag):
    g = ops_golden
    for L, Lq in [(1, 1), (5, 5), (16, 16), (37, 37), (50, 13)]:
        q, k, v = (g[f"attn_{tag}_pre{L}_{Lq}_{n}"] for n in "qkv")
        kc, vc, bt = _paged(k, v, shuffle_seed=L)
        got = R.paged_attn_prefill(q, kc, vc, bt[None], np.array([0, Lq], np.int32),
                                   np.array([L], np.int32), q.shape[-1] ** -0.5, "float32")
        np.testing.assert_allclose(got, g[f"attn_{tag}_pre{L}_{Lq}_o"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
def test_silu_and_mul(ops_golden, dtype):
    got = R.silu_and_mul(ops_golden["silu_x"], dtype)
    close_dtype(got, ops_golden[f"silu_y_{dtype}"], dtype, max_ulp_frac=0.0 if dtype == "bfloat16" else 1)


def test_argmax(ops_golden):
    np.testing.assert_array_equal(R.argmax_lowest(ops_golden["argmax_logits"]), ops_golden["argmax_idx"])


def test_argmax_shard_keys_reduce_to_the_golden_argmax(ops_golden):
    """The vocabulary-parallel form (one key per shard, signed max, decode) gives the committed
    argmax indices for every way of cutting the golden rows into equal shards."""
    lg = ops_golden["argmax_logits"]
    V = lg.shape[1]
    for shards in (1, 2, 4, 8):
        if V % shards:
            continue
        Vs = V // shards
        keys = np.stack([R.argmax_shard_key(lg[:, r * Vs:(r + 1) * Vs], r * Vs) for r in range(shards)])
        assert keys.dtype == np.int64
        np.testing.assert_array_equal(R.argmax_key_to_index(keys.max(axis=0)), ops_golden["argmax_idx"])


def test_round_to_bits_roundtrip():
    rng = np.random.Generator(np.random.PCG64(5))
    x = rng.standard_normal(1000).astype(np.float32) * 100
    for dt in ("bfloat16", "float16"):
        r = R.round_to(x, dt)
        np.testing.assert_array_equal(R.from_bits(R.to_bits(r, dt), dt), r)
        assert np.abs(r - x).max() <= np.abs(x).max() * 2.0 ** -8
    # ties to even on the bf16 grid: 1 + 2^-8 is halfway between 1 and 1 + 2^-7
    assert R.round_to(np.float32(1 + 2.0 ** -8), "bfloat16") == 1.0
    assert R.round_to(np.float32(1 + 3 * 2.0 ** -8), "bfloat16") == np.float32(1 + 2.0 ** -6)


# --------------------------------------------------------------------------- #
# whole-model restatement against HF greedy generations
# --------------------------------------------------------------------------- #
def _run_oracle(name, dtype, gen):
    arch, cfg, w = make_weights(name)
    g = load_npz(f"tiny_{name}.npz")
    lengths = [int(x) for x in g["lengths"]]
    prompts = make_prompts(name, len(lengths), lengths)
    for i, p in enumerate(prompts):
        np.testing.assert_array_equal(p, g[f"prompt{i}"])
    model = M.make_ref(arch, cfg, w, dtype, num_blocks=len(prompts) * 8)
    toks, logits = M.greedy_generate(model, prompts, gen)
    return g, toks, logits


@pytest.mark.parametrize("name", list(TINY))
def test_tiny_model_fp32_matches_hf(name):
    """fp32 oracle == fp32 HF: same tokens for the whole generation and logits
    to fp32 accumulation accuracy.  Pins the algorithm (GQA mapping, rotary
    pairing, norm placement, paged cache bookkeeping)."""
    g, toks, logits = _run_oracle(name, "float32", int(load_npz(f"tiny_{name}.npz")["gen_len"]))
    np.testing.assert_allclose(logits[0], g["logits_first_float32"], rtol=1e-3, atol=2e-4)
    ref = g["tokens_float32"]
    for b in range(ref.shape[0]):
        # identical unless HF's own top-2 margin is below fp32 noise
        if g["min_margin_float32"][b] > 1e-3:
            np.testing.assert_array_equal(toks[b], ref[b])
    np.testing.assert_allclose(logits[-1], g["logits_last_float32"], rtol=1e-3, atol=5e-4)


@pytest.mark.parametrize("name", list(TINY))
def test_tiny_model_bf16_close_to_hf(name):
    """bf16 oracle vs bf16 HF (CPU): first-step logits agree to a few bf16 ulps
    of the logit scale; tokens agree until the first near-tie."""
    g, toks, logits = _run_oracle(name, "bfloat16", 8)
    ref_first = g["logits_first_bfloat16"]
    scale = np.abs(ref_first).max()
    # HF's bf16 eager attention also rounds the scores and probabilities, so the
    # two bf16 runs differ by accumulated rounding noise, not by algorithm
    err = np.abs(logits[0] - ref_first)
    assert err.max() <= 8 * BF16_ULP * scale
    assert err.mean() <= BF16_ULP * scale
    ref = g["tokens_bfloat16"]
    agree = (toks[:, :1] == ref[:, :1]).mean()
    assert agree >= 0.75


def test_fp8_e4m3fn_rounding_matches_torch():
    """The oracle's e4m3fn grid (decode table and round-to-nearest-even) against PyTorch's own
    float8_e4m3fn conversion, an independent implementation available on the CPU."""
    import torch
    from oracle import ops_ref as R
    table = R.fp8_e4m3fn_decode_table()
    ref = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
    np.testing.assert_array_equal(np.nan_to_num(table, nan=777.0), np.nan_to_num(ref, nan=777.0))
    rng = np.random.Generator(np.random.PCG64(5))
    for scale in (150.0, 1.0, 0.01, 2.0 ** -8):
        v = np.clip((rng.standard_normal(100000) * scale).astype(np.float32), -448, 448)
        np.testing.assert_array_equal(R.fp8_e4m3fn_round(v), torch.from_numpy(v).to(torch.float8_e4m3fn).float().numpy())
    ties = (np.arange(0, 448 * 16 + 1) / 16.0).astype(np.float32)     # many exact halfway cases
    np.testing.assert_array_equal(R.fp8_e4m3fn_round(ties), torch.from_numpy(ties).to(torch.float8_e4m3fn).float().numpy())
    q, s = R.quantize_rows_fp8(np.array([[0.0, 0.0], [1.0, -3.5], [448.0, 2.0]], np.float32))
    np.testing.assert_array_equal(s, np.array([1.0, 3.5 / 448, 1.0], np.float32))
    assert np.abs(q).max() == 448.0 and (q[0] == 0).all()


# --------------------------------------------------------------------------- #
# random sampling, full-vocabulary form: the pieces of the restatement against numpy's own
# log / softmax (no engine output pins the drawn tokens themselves: parity unpinned; what is
# pinned is that the draw follows softmax(logits / T) and that the GPU equals this code)
# --------------------------------------------------------------------------- #
def test_ln_det_is_a_natural_logarithm():
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(5e-8, 20, 200000), 1 - 2.0 ** -np.arange(1, 25), 2.0 ** -np.arange(0, 25),
                        [np.sqrt(2), 1.41421354, 1.4142137]]).astype(np.float32)
    got = R.ln_det(x).astype(np.float64)
    want = np.log(x.astype(np.float64))
    assert np.abs(got - want).max() <= 1e-6
    assert (np.abs(got - want) <= 4e-7 * np.abs(want) + 1e-9).all()


def test_sample_uniform_at_stays_inside_the_open_interval():
    for h in (0, 1, 0xFFFFFFFF, 0x9E3779B9):
        u = R.sample_uniform_at(h, 200000)
        assert u.dtype == np.float32 and u.min() >= 2.0 ** -24 and u.max() <= 1 - 2.0 ** -24
        assert abs(float(u.mean()) - 0.5) < 0.01
        g = -R.ln_det(-R.ln_det(u))
        assert np.isfinite(g).all()
    # the vectorised hash is the scalar one
    i = 123456
    h = R._mix32(0x1234 ^ R._mix32((i + 0x632BE5AB) & 0xFFFFFFFF))
    assert R.sample_uniform_at(0x1234, i + 1)[i] == (np.float32(h >> 9) + np.float32(0.5)) * np.float32(2.0 ** -23)


def test_full_vocabulary_draw_follows_the_softmax():
    """top_k disabled and top_p = 1: tokens outside any 1024-candidate set are drawn with their
    softmax probability (a nearly flat row of 4096 logits, where the tail holds most of the mass)."""
    rng = np.random.default_rng(12)
    V, n, T = 4096, 6000, 1.3
    row = (rng.standard_normal(V) * 0.5).astype(np.float32)
    row[:8] += 4.0
    ids = R.sample_topk_topp(np.tile(row, (n, 1)), np.full(n, T, np.float32), np.full(n, -1, np.int32),
                             np.ones(n, np.float32), np.arange(n) * 31 + 5, np.arange(n) % 97)
    p = np.exp(row.astype(np.float64) / T)
    p /= p.sum()
    order = np.argsort(-row, kind="stable")
    tail = np.zeros(V, bool)
    tail[order[1024:]] = True
    assert abs(tail[ids].mean() - p[tail].sum()) < 4 * np.sqrt(p[tail].sum() / n)
    for t in range(8):
        assert abs((ids == t).mean() - p[t]) < 4 * np.sqrt(p[t] / n) + 1e-3
    # special rows: a NaN wins, -inf never does, an all-tied row is still a uniform draw
    lg = np.zeros((3, 64), np.float32)
    lg[0, 9] = np.nan
    lg[1, :] = -np.inf
    lg[1, 40] = -3.0
    out = R.sample_topk_topp(lg, np.ones(3, np.float32), np.zeros(3, np.int32), np.ones(3, np.float32), [1, 2, 3], [0, 0, 0])
    assert out[0] == 9 and out[1] == 40 and 0 <= out[2] < 64


def test_kept_sets_match_the_hf_warpers():
    """The oracle's kept set against what HF transformers' TopKLogitsWarper / TopPLogitsWarper keep
    (tests/golden/sampling_kept_hf.npz, written by oracle/gen_golden.py): flat rows whose nucleus is
    far larger than 1024 tokens, peaked rows, top_k of 1 / beyond 1024 / the whole vocabulary, tied rows
    (top-k keeps every token tied with the k-th), the Llama-3 vocabulary -- the same set up to which of
    the equal logits at the cut are taken (kept_sets_agree)."""
    from oracle.sampling_cases import CASES, kept_sets_agree, row_of
    z = np.load(os.path.join(GOLDEN, "sampling_kept_hf.npz"))
    assert sorted(z.files) == sorted(CASES)
    sizes = {}
    for name, (V, kind, t, k, p) in CASES.items():
        row = row_of(name)
        hf = np.unpackbits(z[name])[:V].astype(bool)
        ours = R.sample_kept_mask(row, t, k, p)
        assert kept_sets_agree(row, hf, ours), name
        sizes[name] = int(ours.sum())
        if 0 < k < V:                                        # top-k keeps at least k and every tie with the k-th
            assert ours.sum() >= min(k, V) or p < 1
    assert sizes["flat_nucleus_beyond_1024"] > 10000 and sizes["tied_topk100_keeps_every_tie"] > 100
    assert sizes["peaked_topk1"] == 1 and sizes["llama3_vocab_flat_topp"] > 100000


def test_integer_masses_make_the_cut_independent_of_summation_order():
    rng = np.random.default_rng(13)
    row = R.round_to((rng.standard_normal(5000) * 2).astype(np.float32), "bfloat16")
    m = R.sample_mass(row, row.max(), np.float32(0.9))
    assert m.dtype == np.uint64 and int(m.max()) == 1 << 40 and (m <= np.uint64(1 << 40)).all()
    perm = rng.permutation(5000)
    assert int(m.sum()) == int(m[perm].sum()) == sum(int(v) for v in m)
    assert R.mass_threshold(0.0, 12345) == 1 and R.mass_threshold(0.5, 10) == 5 and R.mass_threshold(0.9, 10 ** 15 + 1) == 899999976158143          # float32(0.9) = 0.89999998
    # -inf logits weigh nothing, the maximum weighs 2^40, even when it is +inf
    assert R.sample_mass(np.asarray([-np.inf, 3.0], np.float32), 3.0, 1.0).tolist() == [0, 1 << 40]
    assert R.sample_mass(np.asarray([np.inf, 3.0], np.float32), np.inf, 1.0).tolist() == [1 << 40, 0]


def test_top_p_and_top_k_cut_the_candidates():
    """The restatement's cut semantics on a row small enough to enumerate: top-k keeps k tokens,
    top-p the shortest prefix reaching the mass, and no draw ever leaves the kept set."""
    rng = np.random.default_rng(14)
    V, n = 50, 4000
    row = (rng.standard_normal(V) * 2).astype(np.float32)
    order = np.argsort(-row, kind="stable")
    p = np.exp(row[order].astype(np.float64))
    p /= p.sum()
    lg = np.tile(row, (n, 1))
    ones = np.ones(n, np.float32)
    for k, tp in ((5, 1.0), (-1, 0.7), (10, 0.5)):
        kk = k if k > 0 else V
        pk = p[:kk] / p[:kk].sum()
        keep = kk if tp >= 1 else int(np.argmax(np.cumsum(pk) >= tp * (1 - 1e-6))) + 1
        ids = R.sample_topk_topp(lg, ones, np.full(n, k, np.int32), ones * tp, np.arange(n) + 99, np.arange(n) % 13)
        assert set(ids) <= set(order[:keep].tolist())
        q = pk[:keep] / pk[:keep].sum()
        freq = np.array([(ids == t).mean() for t in order[:keep]])
        assert np.abs(freq - q).max() < 4 * np.sqrt(q.max() / n) + 0.01
