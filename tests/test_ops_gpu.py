"""Parity of the HIP operators (called through the C ABI via mi355_decode.ops)
against the numpy oracle on the same seeded inputs, and against the committed HF
vectors.  GPU only.

Bars (written next to each check):
  * rotary embedding, KV-cache store, residual add, argmax: bit-exact.
  * RMSNorm, SwiGLU gate: identical up to single-ulp flips of the 16-bit result
    on a small fraction of elements (reduction order / expf rounding).
  * attention: fp32 softmax/accumulation with probabilities rounded to the model
    dtype before the PV product: |err| <= 8e-3 * max|ref| for bf16,
    <= 1e-3 * max|ref| for fp16 (the north-star tolerance), plus one output ulp.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from conftest import GOLDEN  # noqa: E402
from oracle import ops_ref as R  # noqa: E402

DT = {"bfloat16": torch.bfloat16, "float16": torch.float16}
ULP = {"bfloat16": 2.0 ** -7, "float16": 2.0 ** -10}
ATTN_TOL = {"bfloat16": 8e-3, "float16": 1e-3}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mi355_decode import ops as _ops
    return _ops


def rng_for(*key):
    import zlib
    return np.random.Generator(np.random.PCG64(zlib.crc32(repr(key).encode())))


def grid(rng, shape, dtype, scale=1.0):
    return R.round_to((rng.standard_normal(shape) * scale).astype(np.float32), dtype)


def dev(x, dtype):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DT[dtype]).cuda()


def host(t):
    return t.float().cpu().numpy()


SUBNORMAL_STEP = {"bfloat16": 2.0 ** -133, "float16": 2.0 ** -24}


def assert_ulp_close(got, ref, dtype, max_frac, atol=0.0):
    """Equal except for rare flips: a one-ulp flip of the first rounding, scaled by
    the second factor and rounded again, moves the result by at most two ulps
    (ulps of subnormal fp16 values are absolute: 2^-24).  `atol`: for results that are a
    difference of larger terms (a norm's scaled value plus its bias), where the fp32 noise
    of the terms, not an ulp of the small result, bounds the error."""
    diff = np.abs(got - ref)
    tol = np.maximum(np.maximum(2 * ULP[dtype] * np.abs(ref) * 1.01, 2 * SUBNORMAL_STEP[dtype]), atol)
    assert (diff <= tol).all(), f"max diff {diff.max()} (worst ref {ref.flat[diff.argmax()]})"
    frac = (diff > 0).mean()
    assert frac <= max_frac, f"{frac:.5f} of elements differ by one ulp (> {max_frac})"


# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("T,H", [(1, 256), (5, 448), (32, 4096), (3, 8192), (2, 16384), (130, 1024)])
def test_rms_norm(ops, dtype, T, H):
    rng = rng_for("rms", dtype, T, H)
    x, w, res = grid(rng, (T, H), dtype, 2.0), R.round_to(1 + 0.1 * rng.standard_normal(H), dtype), grid(rng, (T, H), dtype)
    eps = 1e-5
    y = ops.rms_norm(dev(x, dtype), dev(w, dtype), eps)
    assert_ulp_close(host(y), R.rms_norm(x, w, eps, dtype), dtype, 0.005)

    xd, rd = dev(x, dtype), dev(res, dtype)
    out = ops.fused_add_rms_norm(xd, rd, dev(w, dtype), eps)
    y_ref, z_ref = R.fused_add_rms_norm(x, res, w, eps, dtype)
    np.testing.assert_array_equal(host(rd), z_ref)            # residual add: bit-exact
    assert out.data_ptr() == xd.data_ptr()                     # in place by default
    assert_ulp_close(host(out), y_ref, dtype, 0.005)


def test_rms_norm_strided_rows(ops):
    """Row strides larger than the width (a column slice of a wider buffer)."""
    dtype = "bfloat16"
    rng = rng_for("rms-strided")
    big = grid(rng, (7, 1024), dtype)
    w = R.round_to(1 + 0.1 * rng.standard_normal(512), dtype)
    bd = dev(big, dtype)
    y = ops.rms_norm(bd[:, 256:768], dev(w, dtype), 1e-6)
    assert_ulp_close(host(y), R.rms_norm(big[:, 256:768], w, 1e-6, dtype), dtype, 0.005)


def test_rms_norm_golden(ops, ops_golden):
    g = ops_golden
    for tag in "abc":
        x, w, eps = g[f"rms_{tag}_x"], g[f"rms_{tag}_w"], float(g[f"rms_{tag}_eps"])
        y = ops.rms_norm(dev(x, "bfloat16"), dev(w, "bfloat16"), eps)
        assert_ulp_close(host(y), g[f"rms_{tag}_y_bfloat16"], "bfloat16", 0.005)


# --------------------------------------------------------------------------- #
# LayerNorm family (OPT): identical to the oracle up to single-ulp flips of the one final rounding
# on a small fraction of elements (fp32 reduction order); the residual add and the embedding +
# position sum are bit-exact.
LN_ATOL = 2e-6      # normalised values of a few units x fp32 epsilon: where weight * value and bias cancel


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("T,H", [(1, 384), (5, 768), (32, 4096), (3, 8192), (2, 12288), (130, 1024)])
def test_layer_norm(ops, dtype, T, H):
    rng = rng_for("ln", dtype, T, H)
    x, res = grid(rng, (T, H), dtype, 2.0), grid(rng, (T, H), dtype)
    x[0] = R.round_to(x[0] + 5.0, dtype)                        # a row with a large mean
    w = R.round_to(1 + 0.1 * rng.standard_normal(H), dtype)
    b = R.round_to(0.1 * rng.standard_normal(H), dtype)
    eps = 1e-5
    y = ops.layer_norm(dev(x, dtype), dev(w, dtype), dev(b, dtype), eps)
    assert_ulp_close(host(y), R.layer_norm(x, w, b, eps, dtype), dtype, 0.005, atol=LN_ATOL)

    xd, rd = dev(x, dtype), dev(res, dtype)
    out = ops.fused_add_layer_norm(xd, rd, dev(w, dtype), dev(b, dtype), eps)
    y_ref, z_ref = R.fused_add_layer_norm(x, res, w, b, eps, dtype)
    np.testing.assert_array_equal(host(rd), z_ref)            # residual add: bit-exact
    assert out.data_ptr() == xd.data_ptr()                     # in place by default
    assert_ulp_close(host(out), y_ref, dtype, 0.005, atol=LN_ATOL)


def test_layer_norm_golden(ops):
    from conftest import load_npz
    g = load_npz("ops_opt_hf.npz")
    for tag in "abc":
        x, res, w, b = (g[f"ln_{tag}_{k}"] for k in ("x", "res", "w", "b"))
        y = ops.layer_norm(dev(x, "bfloat16"), dev(w, "bfloat16"), dev(b, "bfloat16"), 1e-5)
        assert_ulp_close(host(y), g[f"ln_{tag}_y_bfloat16"], "bfloat16", 0.005, atol=LN_ATOL)
        xd, rd = dev(x, "bfloat16"), dev(res, "bfloat16")
        ops.fused_add_layer_norm(xd, rd, dev(w, "bfloat16"), dev(b, "bfloat16"), 1e-5)
        np.testing.assert_array_equal(host(rd), g[f"ln_{tag}_z_bfloat16"])
        assert_ulp_close(host(xd), g[f"ln_{tag}_yz_bfloat16"], "bfloat16", 0.005, atol=LN_ATOL)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_embed_pos_layer_norm(ops, dtype):
    rng = rng_for("embed-pos", dtype)
    V, P, H, T = 320, 64, 768, 37
    emb, pos_tab = grid(rng, (V, H), dtype), grid(rng, (P + 2, H), dtype, 0.5)
    w = R.round_to(1 + 0.1 * rng.standard_normal(H), dtype)
    b = R.round_to(0.1 * rng.standard_normal(H), dtype)
    ids, pos = rng.integers(0, V, T), rng.integers(0, P, T)
    resid = torch.empty(T, H, dtype=DT[dtype], device="cuda")
    out = torch.empty_like(resid)
    ops.embed_pos_layer_norm(dev(emb, dtype), torch.from_numpy(ids).cuda(), dev(pos_tab, dtype),
                             torch.from_numpy(pos).cuda(), 2, dev(w, dtype), dev(b, dtype), 1e-5, resid, out)
    want = R.embed_positions(emb, pos_tab, ids, pos, 2, dtype)
    np.testing.assert_array_equal(host(resid), want)          # gather + add: bit-exact
    assert_ulp_close(host(out), R.layer_norm(want, w, b, 1e-5, dtype), dtype, 0.005, atol=LN_ATOL)
    # ids / positions outside their tables: row 0 / the nearest row, nothing faults
    bad_ids, bad_pos = torch.tensor([-1, V, 5], device="cuda"), torch.tensor([0, 1, 10 ** 9], device="cuda")
    r3, o3 = torch.empty(3, H, dtype=DT[dtype], device="cuda"), torch.empty(3, H, dtype=DT[dtype], device="cuda")
    ops.embed_pos_layer_norm(dev(emb, dtype), bad_ids, dev(pos_tab, dtype), bad_pos, 2, dev(w, dtype), dev(b, dtype),
                             1e-5, r3, o3)
    np.testing.assert_array_equal(host(r3), R.embed_positions(emb, pos_tab, [0, 0, 5], [0, 1, P - 1], 2, dtype))


def test_embed_pos_golden(ops):
    from conftest import load_npz
    g = load_npz("ops_opt_hf.npz")
    H = g["pos_embed"].shape[1]
    T = len(g["pos_ids"])
    resid = torch.empty(T, H, dtype=torch.bfloat16, device="cuda")
    ones, zeros = torch.ones(H, dtype=torch.bfloat16, device="cuda"), torch.zeros(H, dtype=torch.bfloat16, device="cuda")
    ops.embed_pos_layer_norm(dev(g["pos_embed"], "bfloat16"), torch.from_numpy(g["pos_ids"]).cuda(),
                             dev(g["pos_table"], "bfloat16"), torch.from_numpy(g["pos_positions"]).cuda(), 2, ones, zeros,
                             1e-5, resid, torch.empty_like(resid))
    np.testing.assert_array_equal(host(resid), g["pos_sum_bfloat16"])


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_relu(ops, dtype):
    rng = rng_for("relu", dtype)
    x = grid(rng, (33, 1024), dtype, 3.0)
    x[0, :4] = [0.0, -0.0, np.inf, -np.inf]
    x[1, 0] = np.nan
    xd = dev(x, dtype)
    y = ops.relu(xd)
    assert y.data_ptr() == xd.data_ptr()
    np.testing.assert_array_equal(R.to_bits(host(y), dtype), R.to_bits(R.relu(x), dtype))   # bit-exact, -0.0 and NaN included
    from mi355_decode._lib import Mi355Error
    with pytest.raises(Mi355Error, match="multiple of 8"):
        ops.relu(torch.zeros(7, dtype=DT[dtype], device="cuda"))


def test_rejects_bad_arguments(ops):
    from mi355_decode._lib import Mi355Error
    x = torch.zeros(4, 250, dtype=torch.bfloat16, device="cuda")      # hidden not a multiple of 8
    with pytest.raises(Mi355Error, match="multiple of 8"):
        ops.rms_norm(x, torch.ones(250, dtype=torch.bfloat16, device="cuda"), 1e-5)
    with pytest.raises(TypeError):
        ops.rms_norm(x.float(), torch.ones(250, device="cuda"), 1e-5)
    with pytest.raises(ValueError):
        ops.rms_norm(x.cpu(), torch.ones(250, dtype=torch.bfloat16), 1e-5)


# --------------------------------------------------------------------------- #
def make_cache(num_blocks, kvh, d, dtype, fill=None):
    kc = torch.zeros(num_blocks, kvh, 16, d, dtype=DT[dtype], device="cuda")
    vc = torch.zeros(num_blocks, kvh, d, 16, dtype=DT[dtype], device="cuda")
    if fill is not None:
        kc.fill_(fill), vc.fill_(fill)
    return kc, vc


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("nh,kvh,d", [(4, 2, 64), (8, 2, 128), (32, 8, 128), (7, 1, 64)])
def test_rope_kv_write(ops, dtype, nh, kvh, d):
    rng = rng_for("rope", dtype, nh, kvh, d)
    T, nblk, max_pos = 37, 12, 512
    table = R.rope_cos_sin_table(max_pos, d, 5e5, dtype)
    qkv = grid(rng, (T, (nh + 2 * kvh) * d), dtype)
    pos = rng.integers(0, max_pos, T).astype(np.int64)
    slots = rng.permutation(nblk * 16)[:T].astype(np.int64)
    slots[3] = -1                                             # padding token: rotated, not stored
    kc_ref = np.zeros((nblk, kvh, 16, d), np.float32)
    vc_ref = np.zeros((nblk, kvh, d, 16), np.float32)
    ref = R.rope_kv_write(qkv, pos, table, slots, kc_ref, vc_ref, nh, kvh, d, dtype)

    kc, vc = make_cache(nblk, kvh, d, dtype)
    qd = dev(qkv, dtype)
    ops.rope_kv_write(qd, torch.from_numpy(pos).cuda(), dev(table, dtype), torch.from_numpy(slots).cuda(),
                      kc, vc, nh, kvh, d)
    np.testing.assert_array_equal(host(qd), ref)              # bit-exact
    np.testing.assert_array_equal(host(kc), kc_ref)
    np.testing.assert_array_equal(host(vc), vc_ref)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("nh,kvh,d", [(8, 2, 128), (32, 8, 128), (7, 1, 64), (4, 4, 256)])
def test_rope_kv_write_prefill_sized(ops, dtype, nh, kvh, d):
    """>= 64 tokens: the V heads go through the page-transposing kernel.  Packed sequences whose
    16-token groups fill pages in order (whole-page 16-byte stores), a sequence that starts
    mid-page and one whose length is not a multiple of 16 (groups straddle sequences and pages:
    element-wise stores), padding slots, a slot beyond the cache, pages in shuffled order, and a
    token count that is not a multiple of 16 -- all bit-exact against the oracle; untouched
    pages stay as they were."""
    rng = rng_for("rope-prefill", dtype, nh, kvh, d)
    nblk, max_pos = 64, 1024
    pages = list(rng.permutation(nblk - 1))                     # the last page stays untouched
    slots, pos = [], []

    def seq(start, n):
        blocks = [pages.pop() for _ in range((start + n + 15) // 16)]
        for p_ in range(start, start + n):
            slots.append(blocks[p_ // 16] * 16 + p_ % 16)
            pos.append(p_)

    seq(0, 128)           # aligned: eight whole pages
    seq(0, 48)
    seq(5, 40)            # chunk that starts mid-page
    seq(0, 37)            # length not a multiple of 16: the groups after it straddle sequences
    seq(0, 64)
    slots, pos = np.asarray(slots, np.int64), np.asarray(pos, np.int64)
    slots[200] = -1                                            # padding token inside an otherwise whole page
    slots[130] = nblk * 16 + 3                                 # beyond the cache: dropped
    T = len(slots)
    assert T % 16 != 0 and T >= 64
    table = R.rope_cos_sin_table(max_pos, d, 5e5, dtype)
    qkv = grid(rng, (T, (nh + 2 * kvh) * d), dtype)
    kc_ref = np.full((nblk, kvh, 16, d), 0.5, np.float32)
    vc_ref = np.full((nblk, kvh, d, 16), 0.5, np.float32)
    ref_slots = np.where(slots >= nblk * 16, -1, slots)
    ref = R.rope_kv_write(qkv, pos, table, ref_slots, kc_ref, vc_ref, nh, kvh, d, dtype)
    kc, vc = make_cache(nblk, kvh, d, dtype, fill=0.5)
    qd = dev(qkv, dtype)
    ops.rope_kv_write(qd, torch.from_numpy(pos).cuda(), dev(table, dtype), torch.from_numpy(slots).cuda(),
                      kc, vc, nh, kvh, d)
    np.testing.assert_array_equal(host(qd), ref)
    np.testing.assert_array_equal(host(kc), kc_ref)
    np.testing.assert_array_equal(host(vc), vc_ref)
    assert (host(vc)[nblk - 1] == 0.5).all()


def test_rope_golden(ops, ops_golden):
    """HF apply_rotary_pos_emb outputs (bf16), using HF's own cos/sin rows as the table."""
    g = ops_golden
    for tag, d, nh, kvh in (("d64", 64, 4, 2), ("d128", 128, 8, 2), ("q128", 128, 7, 1)):
        q, k = g[f"rope_{tag}_q"], g[f"rope_{tag}_k"]
        T = q.shape[0]
        tab = np.concatenate([g[f"rope_{tag}_cos_bfloat16"][:, : d // 2], g[f"rope_{tag}_sin_bfloat16"][:, : d // 2]], 1)
        v = np.zeros_like(k)
        qkv = np.concatenate([q.reshape(T, -1), k.reshape(T, -1), v.reshape(T, -1)], 1)
        kc, vc = make_cache(2, kvh, d, "bfloat16")
        qd = dev(qkv, "bfloat16")
        ops.rope_kv_write(qd, torch.arange(T, device="cuda"), dev(tab, "bfloat16"),
                          torch.arange(T, device="cuda"), kc, vc, nh, kvh, d)
        out = host(qd)
        np.testing.assert_array_equal(out[:, : nh * d].reshape(T, nh, d), g[f"rope_{tag}_qo_bfloat16"])
        np.testing.assert_array_equal(out[:, nh * d:(nh + kvh) * d].reshape(T, kvh, d), g[f"rope_{tag}_ko_bfloat16"])
        # and the rotated keys landed in the page
        np.testing.assert_array_equal(host(kc).transpose(0, 2, 1, 3).reshape(-1, kvh, d)[:T], g[f"rope_{tag}_ko_bfloat16"])


# --------------------------------------------------------------------------- #
def build_paged(rng, ctxs, kvh, d, dtype, extra_blocks=5, poison=True):
    """Random K/V for each sequence scattered into shuffled pages; unused slots
    are poisoned with NaN to prove the kernels never let them through."""
    nblks = [(c + 15) // 16 for c in ctxs]
    total = sum(nblks) + extra_blocks
    perm = rng.permutation(total).astype(np.int32)
    max_blocks = max(max(nblks), 1) + 2
    bt = np.zeros((len(ctxs), max_blocks), np.int32)
    fill = np.nan if poison else 0.0
    kc = np.full((total, kvh, 16, d), fill, np.float32)
    vc = np.full((total, kvh, d, 16), fill, np.float32)
    used = 0
    for i, c in enumerate(ctxs):
        bt[i, : nblks[i]] = perm[used: used + nblks[i]]
        bt[i, nblks[i]:] = perm[-1]                       # never-read filler ids
        used += nblks[i]
        if c:
            k, v = grid(rng, (c, kvh, d), dtype), grid(rng, (c, kvh, d), dtype)
            slots = bt[i, np.arange(c) // 16].astype(np.int64) * 16 + np.arange(c) % 16
            R.kv_write(k, v, slots, kc, vc)
    return kc, vc, bt


def attn_close(got, ref, dtype):
    scale = np.abs(ref).max() if ref.size else 1.0
    err = np.abs(got - ref)
    tol = ATTN_TOL[dtype] * scale + ULP[dtype] * np.abs(ref)
    assert not np.isnan(got).any(), "NaN in attention output"
    assert (err <= tol).all(), f"max err {err.max():.4g} vs tol {tol.flat[err.argmax()]:.4g} (scale {scale:.3g})"


DECODE_CTX = [1, 15, 16, 17, 31, 32, 33, 64, 127, 128, 129, 200, 255, 256, 1000, 0]


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("nh,kvh,d", [(4, 2, 64), (8, 2, 128), (32, 8, 128), (7, 1, 64), (8, 1, 128), (4, 4, 128)])
@pytest.mark.parametrize("splits", [1, 2, 5])
def test_paged_attn_decode(ops, dtype, nh, kvh, d, splits):
    rng = rng_for("dec", dtype, nh, kvh, d)
    ctxs = DECODE_CTX
    kc, vc, bt = build_paged(rng, ctxs, kvh, d, dtype)
    q = grid(rng, (len(ctxs), nh, d), dtype)
    scale = d ** -0.5
    ref = R.paged_attn_decode(q, np.nan_to_num(kc), np.nan_to_num(vc), bt, np.array(ctxs, np.int32), scale, dtype)
    out = ops.paged_attn_decode(dev(q.reshape(len(ctxs), -1), dtype), dev(kc, dtype), dev(vc, dtype),
                                torch.from_numpy(bt).cuda(), torch.tensor(ctxs, dtype=torch.int32, device="cuda"),
                                nh, kvh, d, scale, num_splits=splits)
    attn_close(host(out).reshape(len(ctxs), nh, d), ref, dtype)


def test_paged_attn_decode_q_inside_qkv_rows(ops):
    """q passed as a strided view of the packed QKV buffer (how the engine calls it)."""
    dtype, nh, kvh, d = "bfloat16", 8, 2, 128
    rng = rng_for("dec-strided")
    ctxs = [40, 7, 129]
    kc, vc, bt = build_paged(rng, ctxs, kvh, d, dtype)
    qkv = grid(rng, (3, (nh + 2 * kvh) * d), dtype)
    ref = R.paged_attn_decode(qkv[:, : nh * d].reshape(3, nh, d), np.nan_to_num(kc), np.nan_to_num(vc), bt,
                              np.array(ctxs, np.int32), d ** -0.5, dtype)
    qd = dev(qkv, dtype)
    out = ops.paged_attn_decode(qd[:, : nh * d], dev(kc, dtype), dev(vc, dtype), torch.from_numpy(bt).cuda(),
                                torch.tensor(ctxs, dtype=torch.int32, device="cuda"), nh, kvh, d, d ** -0.5)
    attn_close(host(out).reshape(3, nh, d), ref, dtype)


def test_paged_attn_decode_golden(ops, ops_golden):
    """HF eager attention outputs (fp32) for single-query attention."""
    g = ops_golden
    for tag, nh, kvh, d in (("g2d64", 4, 2, 64), ("g4d128", 8, 2, 128), ("g7d64", 7, 1, 64), ("g8d128", 8, 1, 128)):
        for L in g[f"attn_{tag}_ctx"]:
            q, k, v = (g[f"attn_{tag}_dec{L}_{n}"] for n in "qkv")
            nblk = (int(L) + 15) // 16
            kc = np.zeros((nblk, kvh, 16, d), np.float32)
            vc = np.zeros((nblk, kvh, d, 16), np.float32)
            R.kv_write(k, v, np.arange(L, dtype=np.int64), kc, vc)
            bt = np.arange(nblk, dtype=np.int32)[None]
            out = ops.paged_attn_decode(dev(q.reshape(1, -1), "bfloat16"), dev(kc, "bfloat16"), dev(vc, "bfloat16"),
                                        torch.from_numpy(bt).cuda(), torch.tensor([L], dtype=torch.int32, device="cuda"),
                                        nh, kvh, d, d ** -0.5)
            attn_close(host(out).reshape(1, nh, d), g[f"attn_{tag}_dec{L}_o"], "bfloat16")


def test_paged_attn_decode_softmax_spike(ops):
    """Force the online-softmax rescale: one late key dominates all earlier ones."""
    dtype, nh, kvh, d = "bfloat16", 8, 2, 128
    rng = rng_for("spike")
    L = 300
    k, v = grid(rng, (L, kvh, d), dtype, 0.5), grid(rng, (L, kvh, d), dtype)
    q = grid(rng, (1, nh, d), dtype)
    for pos in (5, 170, 299):
        k2 = k.copy()
        k2[pos] = R.round_to(8.0 * q[0, ::4][:kvh], dtype)      # huge positive score for head 0 of each group
        nblk = (L + 15) // 16
        kc, vc = np.zeros((nblk, kvh, 16, d), np.float32), np.zeros((nblk, kvh, d, 16), np.float32)
        R.kv_write(k2, v, np.arange(L, dtype=np.int64), kc, vc)
        bt = np.arange(nblk, dtype=np.int32)[None]
        ref = R.paged_attn_decode(q, kc, vc, bt, np.array([L], np.int32), d ** -0.5, dtype)
        out = ops.paged_attn_decode(dev(q.reshape(1, -1), dtype), dev(kc, dtype), dev(vc, dtype),
                                    torch.from_numpy(bt).cuda(), torch.tensor([L], dtype=torch.int32, device="cuda"),
                                    nh, kvh, d, d ** -0.5)
        attn_close(host(out).reshape(1, nh, d), ref, dtype)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("nh,kvh,d", [(4, 2, 64), (8, 2, 128), (32, 8, 128), (7, 1, 64), (16, 2, 128)])
def test_paged_attn_prefill(ops, dtype, nh, kvh, d):
    rng = rng_for("pre", dtype, nh, kvh, d)
    # (context_len, q_len): full prefill and chunked continuation
    cases = [(1, 1), (5, 5), (16, 16), (17, 17), (37, 37), (128, 128), (50, 13), (200, 33), (64, 1)]
    ctxs = [c for c, _ in cases]
    qlens = [q for _, q in cases]
    kc, vc, bt = build_paged(rng, ctxs, kvh, d, dtype)
    T = sum(qlens)
    q = grid(rng, (T, nh, d), dtype)
    cu = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32)
    scale = d ** -0.5
    ref = R.paged_attn_prefill(q, np.nan_to_num(kc), np.nan_to_num(vc), bt, cu, np.array(ctxs, np.int32), scale, dtype)
    out = ops.paged_attn_prefill(dev(q.reshape(T, -1), dtype), dev(kc, dtype), dev(vc, dtype),
                                 torch.from_numpy(bt).cuda(), torch.from_numpy(cu).cuda(),
                                 torch.tensor(ctxs, dtype=torch.int32, device="cuda"), max(qlens), nh, kvh, d, scale)
    attn_close(host(out).reshape(T, nh, d), ref, dtype)


def test_paged_attn_prefill_golden(ops, ops_golden):
    g = ops_golden
    for tag, nh, kvh, d in (("g2d64", 4, 2, 64), ("g4d128", 8, 2, 128), ("g7d64", 7, 1, 64), ("g8d128", 8, 1, 128)):
        for L, Lq in [(1, 1), (5, 5), (16, 16), (37, 37), (50, 13)]:
            q, k, v = (g[f"attn_{tag}_pre{L}_{Lq}_{n}"] for n in "qkv")
            nblk = (L + 15) // 16
            kc, vc = np.zeros((nblk, kvh, 16, d), np.float32), np.zeros((nblk, kvh, d, 16), np.float32)
            R.kv_write(k, v, np.arange(L, dtype=np.int64), kc, vc)
            out = ops.paged_attn_prefill(dev(q.reshape(Lq, -1), "bfloat16"), dev(kc, "bfloat16"), dev(vc, "bfloat16"),
                                         torch.arange(nblk, dtype=torch.int32, device="cuda")[None].contiguous(),
                                         torch.tensor([0, Lq], dtype=torch.int32, device="cuda"),
                                         torch.tensor([L], dtype=torch.int32, device="cuda"), Lq, nh, kvh, d, d ** -0.5)
            attn_close(host(out).reshape(Lq, nh, d), g[f"attn_{tag}_pre{L}_{Lq}_o"], "bfloat16")


def test_prefill_then_decode_consistent(ops):
    """The last row of a prefill equals a decode step over the same cache."""
    dtype, nh, kvh, d = "bfloat16", 8, 2, 128
    rng = rng_for("consistency")
    ctxs = [70, 33]
    kc, vc, bt = build_paged(rng, ctxs, kvh, d, dtype)
    q = grid(rng, (sum(ctxs), nh, d), dtype)
    cu = np.array([0, 70, 103], np.int32)
    args = (dev(kc, dtype), dev(vc, dtype), torch.from_numpy(bt).cuda())
    cl = torch.tensor(ctxs, dtype=torch.int32, device="cuda")
    pre = host(ops.paged_attn_prefill(dev(q.reshape(103, -1), dtype), *args, torch.from_numpy(cu).cuda(), cl, 70,
                                      nh, kvh, d, d ** -0.5))
    dec = host(ops.paged_attn_decode(dev(q[[69, 102]].reshape(2, -1), dtype), *args, cl, nh, kvh, d, d ** -0.5))
    np.testing.assert_allclose(pre[[69, 102]], dec, rtol=2 ** -6, atol=2e-2)


# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("T,F", [(1, 512), (32, 14336), (7, 1000), (300, 640)])
def test_silu_and_mul(ops, dtype, T, F):
    rng = rng_for("silu", dtype, T, F)
    x = grid(rng, (T, 2 * F), dtype, 3.0)
    y = ops.silu_and_mul(dev(x, dtype))
    assert_ulp_close(host(y), R.silu_and_mul(x, dtype), dtype, 0.002)


def test_silu_golden(ops, ops_golden):
    y = ops.silu_and_mul(dev(ops_golden["silu_x"], "bfloat16"))
    assert_ulp_close(host(y), ops_golden["silu_y_bfloat16"], "bfloat16", 0.002)


# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("ldt", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("B,V", [(1, 8), (6, 1000), (32, 128256), (3, 50272), (2, 152064), (4, 1027)])
def test_argmax(ops, ldt, B, V):
    rng = rng_for("argmax", ldt, B, V)
    stride = (V + 7) // 8 * 8
    lg = (rng.standard_normal((B, V)) * 4).astype(np.float32)
    lg = R.round_to(lg, ldt)
    hi = float(np.abs(lg).max()) + 1.0
    if V > 16:
        lg[0, [5, V - 3]] = hi                           # tie -> lowest index
        lg[B - 1, :] = 0.0
        lg[B - 1, V // 2] = -0.0                         # signed zeros are equal -> index 0
    ref = R.argmax_lowest(lg)
    tdt = {"float32": torch.float32, **DT}[ldt]
    buf = torch.full((B, stride), float("-inf"), dtype=tdt, device="cuda")
    buf[:, :V] = torch.from_numpy(lg).to(tdt).cuda()
    out = ops.argmax(buf[:, :V])
    np.testing.assert_array_equal(out.cpu().numpy(), ref)  # bit-exact indices


@pytest.mark.parametrize("ldt", ["bfloat16", "float16", "float32"])
@pytest.mark.parametrize("B,V,shards", [(5, 512, 2), (32, 128256, 8), (3, 448, 4), (1, 16, 1)])
def test_argmax_over_vocabulary_shards(ops, ops_golden, ldt, B, V, shards):
    """Greedy sampling with the vocabulary split over ranks: every shard's key equals the oracle's
    bit for bit, and the signed max of the shards' keys decodes to the argmax of the whole row
    (ties across shards -> lowest global index, NaN on top, signed zeros equal)."""
    rng = rng_for("argmax-shards", ldt, B, V)
    lg = R.round_to((rng.standard_normal((B, V)) * 4).astype(np.float32), ldt)
    hi = 64.0                                            # above every drawn logit, exact in all three types
    Vs = V // shards
    if shards > 1:
        lg[0, [Vs - 1, Vs + 2, V - 1]] = hi              # the same maximum in two shards
        lg[B - 1, :] = -0.0
        lg[B - 1, V - 2] = 0.0                           # all equal: index 0 of shard 0
        if B > 2:
            lg[1, Vs + 3] = np.nan                       # NaN beats everything
            lg[1, 0] = hi
    tdt = {"float32": torch.float32, **DT}[ldt]
    full = torch.from_numpy(lg).to(tdt).cuda()
    keys = []
    for r in range(shards):
        k = ops.argmax_shard_key(full[:, r * Vs:(r + 1) * Vs].contiguous(), r * Vs)
        np.testing.assert_array_equal(k.cpu().numpy(), R.argmax_shard_key(lg[:, r * Vs:(r + 1) * Vs], r * Vs))
        keys.append(k)
    best = torch.stack(keys).max(dim=0).values           # what the all-reduce(MAX) over ranks computes
    ids = ops.argmax_key_to_index(best).cpu().numpy()
    np.testing.assert_array_equal(ids, R.argmax_key_to_index(best.cpu().numpy()))
    np.testing.assert_array_equal(ids, R.argmax_lowest(lg))
    np.testing.assert_array_equal(ids, ops.argmax(full).cpu().numpy())


@pytest.mark.parametrize("ldt", ["bfloat16", "float16", "float32"])
@pytest.mark.parametrize("B,V", [(1, 16), (3, 1000), (32, 128256), (7, 50272), (128, 152064), (2, 8200)])
def test_argmax_chunked(ops, ldt, B, V):
    """The many-workgroup argmax of a decode batch: the oracle's ids (lowest index on ties, also
    across column ranges; NaN on top; signed zeros equal), both destinations written, the key
    buffer left zero, same ids on a second launch."""
    rng = rng_for("argmax-chunked", ldt, B, V)
    lg = R.round_to((rng.standard_normal((B, V)) * 4).astype(np.float32), ldt)
    hi = 64.0
    if V > 16:
        lg[0, [V // 3, V - 3]] = hi                      # a tie across column ranges -> lowest index
        lg[B - 1, :] = 0.0
        lg[B - 1, V // 2] = -0.0
        if B > 2:
            lg[1, V - 5] = np.nan
            lg[1, 3] = hi
    tdt = {"float32": torch.float32, **DT}[ldt]
    stride = (V + 7) // 8 * 8
    buf = torch.full((B, stride), float("-inf"), dtype=tdt, device="cuda")
    buf[:, :V] = torch.from_numpy(lg).to(tdt).cuda()
    keys = torch.zeros(B + 3, dtype=torch.int64, device="cuda")
    out = torch.full((B,), -7, dtype=torch.int64, device="cuda")
    out2 = torch.full((B,), -7, dtype=torch.int64, device="cuda")
    ref = R.argmax_lowest(lg)
    for _ in range(2):
        ops.argmax_chunked(buf[:, :V], keys, out, out2)
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
        np.testing.assert_array_equal(out2.cpu().numpy(), ref)
        assert int(keys.abs().sum().item()) == 0
    np.testing.assert_array_equal(ops.argmax(buf[:, :V]).cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("T,H,V", [(1, 256, 50), (32, 4096, 1000), (5, 8192, 77), (3, 16384, 9), (130, 448, 300)])
def test_embed_rms_norm(ops, dtype, T, H, V):
    """Embedding gather + first input norm in one launch: the residual rows are the embedding rows
    bit for bit, the normed rows meet the RMSNorm bar; ids outside the table read row 0."""
    rng = rng_for("embed-norm", dtype, T, H)
    emb = grid(rng, (V, H), dtype, 2.0)
    w = R.round_to(1 + 0.1 * rng.standard_normal(H), dtype)
    ids = rng.integers(0, V, T).astype(np.int64)
    if T > 2:
        ids[1], ids[2] = -4, V + 9
    safe = np.where((ids < 0) | (ids >= V), 0, ids)
    resid = torch.full((T, H), float("nan"), dtype=DT[dtype], device="cuda")
    x = torch.full((T, H), float("nan"), dtype=DT[dtype], device="cuda")
    ops.embed_rms_norm(dev(emb, dtype), torch.from_numpy(ids).cuda(), dev(w, dtype), 1e-5, resid, x)
    np.testing.assert_array_equal(host(resid), emb[safe])
    assert_ulp_close(host(x), R.rms_norm(emb[safe], w, 1e-5, dtype), dtype, 0.005)


def test_argmax_golden_and_nan(ops, ops_golden):
    lg = ops_golden["argmax_logits"]
    out = ops.argmax(torch.from_numpy(lg).cuda())
    np.testing.assert_array_equal(out.cpu().numpy(), ops_golden["argmax_idx"])


# --------------------------------------------------------------------------- #
# decode-time linear layer (weights streamed once; fp32 accumulation, one rounding)
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,bias", [(1, 64, 256, False), (4, 320, 256, True), (32, 4096, 4096, False),
                                        (32, 6144, 4096, True), (7, 1000, 448, False), (33, 512, 1024, True),
                                        (128, 768, 512, False), (17, 100, 96, True)])
@pytest.mark.parametrize("mode", [0, 1, 2, 4])
def test_linear_decode(ops, dtype, M, N, K, bias, mode):
    rng = rng_for("linear", dtype, M, N, K)
    x = grid(rng, (M, K), dtype)
    w = grid(rng, (N, K), dtype, 0.05)
    b = grid(rng, (N,), dtype) if bias else None
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    y = ops.linear_decode(dev(x, dtype), dev(w, dtype), dev(b, dtype) if bias else None, k_slices=mode)
    got = host(y)
    # fp32 accumulation in another order than the reference, then one rounding:
    # within one ulp of the exactly rounded result (plus fp32 summation noise)
    tol = ULP[dtype] * np.abs(ref64) * 1.01 + 1e-5 * np.sqrt(K) * np.abs(x).max() * np.abs(w).max() + 2 * SUBNORMAL_STEP[dtype]
    err = np.abs(got - ref64)
    assert (err <= tol).all(), f"max err {err.max()} at ref {ref64.flat[err.argmax()]}"
    # and mostly identical to the oracle's rounding of the fp32 product
    exact = R.linear(x, w, b, dtype)
    assert (got != exact).mean() <= 0.02


def test_linear_decode_strided_out_and_in(ops):
    """x and y as column slices of wider buffers (how the engine packs activations)."""
    dtype = "bfloat16"
    rng = rng_for("linear-strided")
    xb, w = grid(rng, (8, 1024), dtype), grid(rng, (256, 512), dtype, 0.05)
    yb = torch.zeros(8, 640, dtype=torch.bfloat16, device="cuda")
    xd = dev(xb, dtype)
    ops.linear_decode(xd[:, 256:768], dev(w, dtype), out=yb[:, 128:384])
    ref = R.linear(xb[:, 256:768], w, None, dtype)
    assert (host(yb[:, 128:384]) != ref).mean() <= 0.02
    assert float(yb[:, :128].abs().max()) == 0.0 and float(yb[:, 384:].abs().max()) == 0.0


# --------------------------------------------------------------------------- #
# random sampling: bit-exact token ids (fixed-order fp32 algorithm on both sides)
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("ldt", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("B,V", [(8, 1000), (32, 128256), (5, 696), (16, 50272)])
def test_sample_topk_topp(ops, ldt, B, V):
    rng = rng_for("sample", ldt, B, V)
    lg = R.round_to((rng.standard_normal((B, V)) * 3).astype(np.float32), ldt)
    lg[0, :] = 0.0                                          # all tied: index order decides
    lg[0, 5] = -0.0
    if B > 2:
        lg[1, : V // 2] = lg[1, 0]                          # a huge tie that straddles the top-k threshold
        lg[2, 7] = np.nan                                   # a NaN sorts on top: the row yields its index, like argmax
    temperature = rng.choice([0.0, 0.3, 0.7, 1.0, 1.5], B).astype(np.float32)
    temperature[:3] = [1.0, 0.8, 1.2][: min(3, B)]
    top_k = rng.choice([-1, 1, 2, 5, 40, 1024, 5000], B).astype(np.int32)
    top_p = rng.choice([1.0, 0.95, 0.5, 0.1], B).astype(np.float32)
    seeds = rng.integers(0, 2 ** 62, B).astype(np.int64)
    tdt = {"float32": torch.float32, **DT}[ldt]
    lgd = torch.from_numpy(lg).to(tdt).cuda()
    for step in (0, 1, 17):
        steps = np.full(B, step, np.int64) + np.arange(B)
        ref = R.sample_topk_topp(lg, temperature, top_k, top_p, seeds, steps)
        out = ops.sample(lgd, torch.from_numpy(temperature).cuda(), torch.from_numpy(top_k).cuda(),
                         torch.from_numpy(top_p).cuda(), torch.from_numpy(seeds).cuda(), torch.from_numpy(steps).cuda())
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
        if B > 2:
            assert out[2].item() == 7


def test_sample_cut_corner_rows(ops):
    """Rows that stress the candidate-list construction: mostly NaN, denormals around zero, a
    vocabulary smaller than the per-thread top-2 table, one huge outlier, a sorted row."""
    rng = rng_for("sample-corners")
    V, B = 1504, 8
    lg = (rng.standard_normal((B, V)) * 2).astype(np.float32)
    lg[0, 10:] = np.nan
    lg[1] = (rng.integers(-400, 400, V) * np.float32(1e-42)).astype(np.float32)   # denormals of both signs
    lg[2] = np.sort(lg[2])[::-1]
    lg[3] = np.sort(lg[3])
    lg[4, 77] = 1e30
    lg[5, :] = -np.inf
    lg[5, 3] = -1e30
    lg[6, ::2] = lg[6, 1]                                                          # half the row tied
    temperature = np.full(B, 1.0, np.float32)
    seeds = np.arange(B, dtype=np.int64) * 1000003
    lgd = torch.from_numpy(lg).cuda()
    for k, p in ((1024, 1.0), (-1, 0.9), (7, 0.5), (1000, 0.999)):
        top_k, top_p = np.full(B, k, np.int32), np.full(B, p, np.float32)
        for step in range(4):
            steps = np.full(B, step, np.int64)
            ref = R.sample_topk_topp(lg, temperature, top_k, top_p, seeds, steps)
            out = ops.sample(lgd, torch.from_numpy(temperature).cuda(), torch.from_numpy(top_k).cuda(),
                             torch.from_numpy(top_p).cuda(), torch.from_numpy(seeds).cuda(), torch.from_numpy(steps).cuda())
            np.testing.assert_array_equal(out.cpu().numpy(), ref)
            assert out[0].item() == 10 and out[5].item() == 3


def test_sample_distribution(ops):
    """Frequencies over many (seed, step) pairs follow softmax(logits / T) over the top-k set."""
    rng = rng_for("sample-dist")
    V, k, T, n = 4096, 8, 0.8, 4096
    row = R.round_to((rng.standard_normal(V) * 2).astype(np.float32), "bfloat16")
    lg = torch.from_numpy(np.tile(row, (n, 1))).to(torch.bfloat16).cuda()
    out = ops.sample(lg, torch.full((n,), T, device="cuda"), torch.full((n,), k, dtype=torch.int32, device="cuda"),
                     torch.ones(n, device="cuda"), torch.arange(n, device="cuda") * 7919 + 3,
                     torch.arange(n, device="cuda")).cpu().numpy()
    top = np.flatnonzero(R.sample_kept_mask(row, T, k, 1.0))      # the k largest and every token tied with the k-th
    assert k <= len(top) <= k + 4
    p = np.exp((row[top] - row[top].max()) / T)
    p /= p.sum()
    assert set(out) <= set(top)
    freq = np.array([(out == t).mean() for t in top])
    assert np.abs(freq - p).max() < 4 * np.sqrt(p.max() / n) + 0.01


@pytest.mark.parametrize("ldt", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("B,V", [(7, 1000), (32, 128256), (3, 72)])
def test_sample_full_vocabulary(ops, ldt, B, V):
    """top_k disabled and top_p >= 1: the Gumbel-max draw over the whole row, bit-exact."""
    rng = rng_for("sample-full", ldt, B, V)
    lg = R.round_to((rng.standard_normal((B, V)) * 2).astype(np.float32), ldt)
    lg[0, :] = 0.0                                          # all tied: a uniform draw
    lg[0, 5] = -0.0
    lg[1, :] = -np.inf                                      # one finite logit among -inf
    lg[1, V - 1] = -2.5
    lg[2, 7] = np.nan                                       # a NaN wins, lowest index first
    lg[2, 40] = np.nan
    if B > 3:
        lg[3, 11] = np.inf
        lg[4, :] = -np.inf                                  # nothing finite: index 0
    temperature = rng.choice([0.05, 0.7, 1.0, 1.5, 30.0], B).astype(np.float32)
    top_k = rng.choice([-1, 0], B).astype(np.int32)
    top_p = rng.choice([1.0, 1.5], B).astype(np.float32)
    seeds = rng.integers(-2 ** 62, 2 ** 62, B).astype(np.int64)
    tdt = {"float32": torch.float32, **DT}[ldt]
    lgd = torch.from_numpy(lg).to(tdt).cuda()
    for step in (0, 3, 2 ** 40):
        steps = np.full(B, step, np.int64) + np.arange(B)
        ref = R.sample_topk_topp(lg, temperature, top_k, top_p, seeds, steps)
        out = ops.sample(lgd, torch.from_numpy(temperature).cuda(), torch.from_numpy(top_k).cuda(),
                         torch.from_numpy(top_p).cuda(), torch.from_numpy(seeds).cuda(), torch.from_numpy(steps).cuda())
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
        assert out[1].item() == V - 1 and out[2].item() == 7
        if B > 3:
            assert out[3].item() == 11 and out[4].item() == 0


def test_sample_full_vocabulary_distribution(ops):
    """With no cut, tokens beyond the 1024 most likely ones are drawn with their softmax mass."""
    rng = rng_for("sample-full-dist")
    V, T, n = 8192, 1.3, 8192
    row = R.round_to((rng.standard_normal(V) * 0.5).astype(np.float32), "bfloat16")
    row[:8] += 4.0
    row = R.round_to(row, "bfloat16")
    lg = torch.from_numpy(np.tile(row, (n, 1))).to(torch.bfloat16).cuda()
    out = ops.sample(lg, torch.full((n,), T, device="cuda"), torch.full((n,), -1, dtype=torch.int32, device="cuda"),
                     torch.ones(n, device="cuda"), torch.arange(n, device="cuda") * 7919 + 3,
                     torch.arange(n, device="cuda") % 101).cpu().numpy()
    p = np.exp(row.astype(np.float64) / T)
    p /= p.sum()
    tail = np.zeros(V, bool)
    tail[np.argsort(-row, kind="stable")[1024:]] = True
    assert p[tail].sum() > 0.3
    assert abs(tail[out].mean() - p[tail].sum()) < 4 * np.sqrt(p[tail].sum() / n)
    for t in range(8):
        assert abs((out == t).mean() - p[t]) < 4 * np.sqrt(p[t] / n) + 1e-3


def test_sample_kept_set_is_the_hf_warpers_set(ops):
    """The rows of oracle/sampling_cases.py: every draw equals the oracle's (whose kept set is pinned to HF
    transformers' TopKLogitsWarper / TopPLogitsWarper in tests/test_oracle_golden.py), and no draw leaves the
    set HF keeps (tests/golden/sampling_kept_hf.npz) except among the equal logits at the cut, where HF's order
    is unspecified.  Covers the list path, the exact path (top_k beyond 1024, a nucleus of thousands) and the
    Llama-3 vocabulary."""
    from oracle.sampling_cases import CASES, row_of
    z = np.load(os.path.join(GOLDEN, "sampling_kept_hf.npz"))
    n = 48
    for name, (V, kind, t, k, p) in CASES.items():
        row = row_of(name)
        hf = np.unpackbits(z[name])[:V].astype(bool)
        lg = torch.from_numpy(np.tile(row, (n, 1))).to(torch.bfloat16).cuda()
        temperature, top_k, top_p = np.full(n, t, np.float32), np.full(n, k, np.int32), np.full(n, p, np.float32)
        seeds, steps = (np.arange(n, dtype=np.int64) * 104729 + 11), np.arange(n, dtype=np.int64) % 7
        out = ops.sample(lg, torch.from_numpy(temperature).cuda(), torch.from_numpy(top_k).cuda(),
                         torch.from_numpy(top_p).cuda(), torch.from_numpy(seeds).cuda(),
                         torch.from_numpy(steps).cuda()).cpu().numpy()
        np.testing.assert_array_equal(out, R.sample_topk_topp(np.tile(row, (n, 1)), temperature, top_k, top_p, seeds, steps),
                                      err_msg=name)
        cut = row[hf].min()
        assert (hf[out] | (row[out] == cut)).all(), name
        assert (row[out] >= cut).all(), name


def test_sample_whole_vocabulary_nucleus_distribution(ops):
    """top_p over the WHOLE vocabulary: on a flat row the 0.9 nucleus holds thousands of tokens; the draws never
    leave it, tokens beyond the 1024 most likely ones are drawn with their renormalised mass, and so are the
    most likely ones."""
    rng = rng_for("sample-nucleus-dist")
    V, T, p, n = 8192, 1.3, 0.9, 8192
    row = (rng.standard_normal(V) * 0.5).astype(np.float32)
    row[:8] += 3.0
    row = R.round_to(row, "bfloat16")
    kept = R.sample_kept_mask(row, T, -1, p)
    assert kept.sum() > 3000
    lg = torch.from_numpy(np.tile(row, (n, 1))).to(torch.bfloat16).cuda()
    out = ops.sample(lg, torch.full((n,), T, device="cuda"), torch.full((n,), -1, dtype=torch.int32, device="cuda"),
                     torch.full((n,), p, device="cuda"), torch.arange(n, device="cuda") * 7919 + 3,
                     torch.arange(n, device="cuda") % 101).cpu().numpy()
    assert kept[out].all()
    q = np.where(kept, np.exp(row.astype(np.float64) / T), 0.0)
    q /= q.sum()
    tail = np.zeros(V, bool)
    tail[np.argsort(-row, kind="stable")[1024:]] = True
    tail &= kept
    assert q[tail].sum() > 0.3
    assert abs(tail[out].mean() - q[tail].sum()) < 4 * np.sqrt(q[tail].sum() / n)
    for t in range(8):
        assert abs((out == t).mean() - q[t]) < 4 * np.sqrt(q[t] / n) + 1e-3
    # the same row with top_k = 3000 (beyond the list: the exact path) and no top-p
    kept_k = R.sample_kept_mask(row, T, 3000, 1.0)
    out = ops.sample(lg, torch.full((n,), T, device="cuda"), torch.full((n,), 3000, dtype=torch.int32, device="cuda"),
                     torch.ones(n, device="cuda"), torch.arange(n, device="cuda") * 7919 + 3,
                     torch.arange(n, device="cuda") % 101).cpu().numpy()
    assert kept_k.sum() >= 3000 and kept_k[out].all()
    qk = np.where(kept_k, np.exp(row.astype(np.float64) / T), 0.0)
    qk /= qk.sum()
    for t in range(8):
        assert abs((out == t).mean() - qk[t]) < 4 * np.sqrt(qk[t] / n) + 1e-3


@pytest.mark.parametrize("ldt", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("B,V,k", [(3, 128256, 20), (8, 1000, 5), (4, 50272, 64), (5, 40, 64), (6, 696, 1)])
def test_topk_logits(ops, ldt, B, V, k):
    """The top_logprobs selection: ids and values bit-exact against the oracle (logit descending, id ascending),
    with a row of ties, a NaN, signed zeros, k beyond the vocabulary, and as log-probabilities."""
    rng = rng_for("topk-logits", ldt, B, V, k)
    lg = R.round_to((rng.standard_normal((B, V)) * 2).astype(np.float32), ldt)
    lg[0, :] = 0.5                                           # all tied: the lowest ids, in order
    if B > 2:
        lg[1, V // 3] = np.nan                               # a NaN sorts on top
        lg[2, : V // 2] = 0.0                                # signed zeros are one value
        lg[2, 1::2] = -0.0
        lg[2, V // 2:] = -1.0
    tdt = {"float32": torch.float32, **DT}[ldt]
    wide = torch.zeros(B, (V + 7) // 8 * 8, dtype=tdt, device="cuda")
    lgd = wide[:, :V]
    lgd.copy_(torch.from_numpy(lg).to(tdt))
    ids, vals = ops.topk_logits(lgd, k)
    rid, rval = R.topk_logits(lg, k)
    np.testing.assert_array_equal(ids.cpu().numpy(), rid)
    def same(a, b):                                          # bit for bit, NaN payloads aside
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(np.signbit(a) | np.isnan(a), np.signbit(b) | np.isnan(b))
    same(vals.cpu().numpy(), rval)
    lse = torch.from_numpy(rng.standard_normal(B).astype(np.float32)).cuda()
    ids2, lp = ops.topk_logits(lgd, k, lse=lse)
    rid2, rlp = R.topk_logits(lg, k, lse=lse.cpu().numpy())
    np.testing.assert_array_equal(ids2.cpu().numpy(), rid2)
    same(lp.cpu().numpy(), rlp)
    with pytest.raises(Exception):
        ops.topk_logits(lgd, 65)


@pytest.mark.parametrize("ldt", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("B,V", [(3, 128256), (32, 50272), (300, 16384), (40, 4100)])
def test_sample_rows_beyond_the_candidate_list(ops, ldt, B, V):
    """Rows whose kept set cannot be found among the 4096 largest logits: a nucleus over a flat row, a deep top-k
    with and without top-p, a tie over half the row that top-p cuts inside (the first tokens of the tie by index
    stay, across the column ranges of the follow-up launches), an all-equal row.  The launches that spread such
    rows over many workgroups, the one-workgroup form (cut_rows=False, and rows 256 .. of a batch) and the oracle
    agree to the token."""
    rng = rng_for("sample-beyond", ldt, B, V)
    lg = R.round_to((rng.standard_normal((B, V)) * 0.7).astype(np.float32), ldt)
    temperature = rng.choice([0.6, 1.0, 1.7], B).astype(np.float32)
    top_k = rng.choice([-1, 1500, 4000, V - 1], B).astype(np.int32)
    top_p = rng.choice([1.0, 0.97, 0.6, 0.2], B).astype(np.float32)
    lg[0, :] = 1.25                                          # all equal: top-p keeps a prefix of the row by index
    top_k[0], top_p[0], temperature[0] = -1, 0.5, 1.0
    lg[1, V // 4: 3 * V // 4] = lg[1].max() + 1.0            # a tie over half the row on top, cut inside by top-p ...
    top_k[1], top_p[1] = -1, 0.4
    lg[2, ::2] = np.float32(0.5)                             # ... and one below a few larger logits, straddling a deep top-k
    top_k[2], top_p[2] = min(4000, V - 1), 0.9
    if B > 20:
        lg[5, 100] = np.nan
        temperature[6] = 0.0
    seeds = rng.integers(0, 2 ** 62, B).astype(np.int64)
    tdt = {"float32": torch.float32, **DT}[ldt]
    wide = torch.zeros(B, (V + 7) // 8 * 8, dtype=tdt, device="cuda")    # rows start on 16-byte boundaries; the row may end anywhere
    lgd = wide[:, :V]
    lgd.copy_(torch.from_numpy(lg).to(tdt))
    args = [torch.from_numpy(a).cuda() for a in (temperature, top_k, top_p, seeds)]
    for step in (0, 3):
        steps = np.full(B, step, np.int64) + 5 * np.arange(B)
        ref = R.sample_topk_topp(lg, temperature, top_k, top_p, seeds, steps)
        sd = torch.from_numpy(steps).cuda()
        spread = ops.sample(lgd, *args, sd).cpu().numpy()
        single = ops.sample(lgd, *args, sd, cut_rows=False).cpu().numpy()
        np.testing.assert_array_equal(spread, ref)
        np.testing.assert_array_equal(single, ref)
    # a second call with other rows flagged than the first: nothing of the first call's scratch leaks
    top_k2 = np.roll(top_k, 1)
    ref = R.sample_topk_topp(lg, temperature, top_k2, top_p, seeds, steps)
    out = ops.sample(lgd, args[0], torch.from_numpy(top_k2).cuda(), args[2], args[3], sd).cpu().numpy()
    np.testing.assert_array_equal(out, ref)


# --------------------------------------------------------------------------- #
# decode-time linear layer on fragment-tiled weights
# --------------------------------------------------------------------------- #
def _linear_tol(ref64, x, w, K, dtype):
    return ULP[dtype] * np.abs(ref64) * 1.01 + 1e-5 * np.sqrt(K) * np.abs(x).max() * np.abs(w).max() + 2 * SUBNORMAL_STEP[dtype]


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,bias", [(1, 64, 1024, False), (5, 320, 1024, True), (32, 4096, 4096, False),
                                        (32, 6144, 4096, True), (17, 1000, 2048, True), (32, 28672, 4096, False),
                                        (16, 100, 3072, False)])
@pytest.mark.parametrize("tiles", [0, 4, 8, 104, 108])
def test_linear_tiled_store(ops, dtype, M, N, K, bias, tiles):
    rng = rng_for("tiled", dtype, M, N, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.05)
    b = grid(rng, (N,), dtype) if bias else None
    wt = ops.tile_weights(dev(w, dtype))
    y = host(ops.linear_tiled(dev(x, dtype), wt, N, dev(b, dtype) if bias else None, tiles_per_wg=tiles))
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    err = np.abs(y - ref64)
    assert (err <= _linear_tol(ref64, x, w, K, dtype)).all(), f"max err {err.max()}"
    assert (y != R.linear(x, w, b, dtype)).mean() <= 0.02


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,bias", [(16, 1280, 8192, False), (32, 8192, 1024, True), (5, 320, 1024, True),
                                        (17, 1000, 2048, False), (1, 16, 1024, False), (32, 4096, 4096, True)])
@pytest.mark.parametrize("tiles", [302, 304, 308, 402, 404])
def test_linear_tiled_k_split_across_waves(ops, dtype, M, N, K, bias, tiles):
    """The narrow-slice geometries (the shapes of a tensor-parallel rank's qkv / o projection among them):
    store epilogue against the oracle, partial-sum slabs against float64 products of their K ranges."""
    rng = rng_for("tiled-kw", dtype, M, N, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.05)
    b = grid(rng, (N,), dtype) if bias else None
    wt = ops.tile_weights(dev(w, dtype))
    y = host(ops.linear_tiled(dev(x, dtype), wt, N, dev(b, dtype) if bias else None, tiles_per_wg=tiles))
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    err = np.abs(y - ref64)
    assert (err <= _linear_tol(ref64, x, w, K, dtype)).all(), f"max err {err.max()}"
    assert (y != R.linear(x, w, b, dtype)).mean() <= 0.02
    unit = ops.tiled_slice_width(M, tiles)
    for groups in {1, min(2, K // unit), min(4, K // unit)}:
        part = ops.linear_tiled(dev(x, dtype), wt, N, k_groups=groups, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles)
        part = part.cpu().numpy()
        assert part.shape == (groups, M, N)
        units = K // unit
        start = 0
        for g in range(groups):                  # slices dealt as evenly as they go, the first groups one more
            n = units // groups + (1 if g < units % groups else 0)
            a, e = start * unit, (start + n) * unit
            start += n
            ref = x[:, a:e].astype(np.float64) @ w[:, a:e].astype(np.float64).T
            np.testing.assert_allclose(part[g], ref, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M,tiles", [(33, 302), (64, 304), (50, 402), (65, 302), (128, 302)])
@pytest.mark.parametrize("N,K,bias", [(4096, 7168, True), (1000, 1024, False)])
def test_linear_tiled_k_split_across_waves_up_to_128_rows(ops, M, tiles, N, K, bias):
    """The same geometries at 33..128 rows (the rank shapes of wide tensor-parallel batches): store,
    slabs and the SwiGLU epilogue."""
    dtype = "bfloat16"
    rng = rng_for("tiled-kw-rows", M, tiles, N, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.05)
    b = grid(rng, (N,), dtype) if bias else None
    xd, wt = dev(x, dtype), ops.tile_weights(dev(w, dtype))
    y = host(ops.linear_tiled(xd, wt, N, dev(b, dtype) if bias else None, tiles_per_wg=tiles))
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    err = np.abs(y - ref64)
    assert (err <= _linear_tol(ref64, x, w, K, dtype)).all(), f"max err {err.max()}"
    assert (y != R.linear(x, w, b, dtype)).mean() <= 0.02
    part = ops.linear_tiled(xd, wt, N, k_groups=2, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles).cpu().numpy()
    np.testing.assert_allclose(part[0] + part[1], x.astype(np.float64) @ w.astype(np.float64).T, rtol=1e-4, atol=2e-3)
    if N % 32 == 0:
        wi = ops.tile_weights(dev(w, dtype), interleave_halves=True)
        g = host(ops.linear_tiled(xd, wi, N, epilogue=ops.EPI_SILU_MUL, tiles_per_wg=tiles))
        np.testing.assert_array_equal(g, host(ops.silu_and_mul(ops.linear_tiled(xd, wt, N, tiles_per_wg=tiles))))
    from mi355_decode._lib import Mi355Error
    with pytest.raises(Mi355Error, match="waves per tile need an x slice"):
        ops.linear_tiled(xd, wt, N, tiles_per_wg=308)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,F,K", [(16, 3584, 8192), (7, 512, 1024), (32, 1536, 1024), (1, 64, 2048)])
@pytest.mark.parametrize("tiles", [302, 304, 402, 404])
def test_linear_tiled_k_split_across_waves_swiglu(ops, dtype, M, F, K, tiles):
    rng = rng_for("tiled-kw-silu", dtype, M, F, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (2 * F, K), dtype, 0.05)
    wt = ops.tile_weights(dev(w, dtype), interleave_halves=True)
    y = host(ops.linear_tiled(dev(x, dtype), wt, 2 * F, epilogue=ops.EPI_SILU_MUL, tiles_per_wg=tiles))
    ref = R.silu_and_mul(R.linear(x, w, None, dtype), dtype)
    diff = np.abs(y - ref)
    assert (diff <= 4 * ULP[dtype] * np.abs(ref) + 4 * ULP[dtype] * 1e-2).all(), f"max diff {diff.max()}"
    assert (diff > 0).mean() <= 0.03
    # the gate on the same sums, unfused
    gu = ops.linear_tiled(dev(x, dtype), ops.tile_weights(dev(w, dtype)), 2 * F, tiles_per_wg=tiles)
    np.testing.assert_array_equal(y, host(ops.silu_and_mul(gu)))
    from mi355_decode._lib import Mi355Error
    with pytest.raises(Mi355Error, match="at most 4 waves per tile"):
        ops.linear_tiled(dev(x, dtype), wt, 2 * F, epilogue=ops.EPI_SILU_MUL, tiles_per_wg=308)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,groups", [(32, 4096, 4096, 4), (32, 4096, 14336, 7), (9, 512, 2048, 2), (32, 256, 1024, 1)])
def test_linear_tiled_partial_slabs(ops, dtype, M, N, K, groups):
    rng = rng_for("tiled-partial", dtype, M, N, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.05)
    wt = ops.tile_weights(dev(w, dtype))
    part = ops.linear_tiled(dev(x, dtype), wt, N, k_groups=groups, epilogue=ops.EPI_PARTIAL).cpu().numpy()
    part2 = ops.linear_tiled(dev(x, dtype), wt, N, k_groups=groups, epilogue=ops.EPI_PARTIAL, tiles_per_wg=104).cpu().numpy()
    np.testing.assert_array_equal(part, part2)             # geometry does not change the sums
    assert part.shape == (groups, M, N)
    kg = K // groups
    for g in range(groups):                      # every slab is the product over its own K range
        ref = x[:, g * kg:(g + 1) * kg].astype(np.float64) @ w[:, g * kg:(g + 1) * kg].astype(np.float64).T
        np.testing.assert_allclose(part[g], ref, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,F,K", [(32, 14336, 4096), (7, 512, 1024), (32, 1536, 1024), (1, 64, 2048)])
def test_linear_tiled_swiglu_epilogue(ops, dtype, M, F, K):
    """gate/up projection with the SwiGLU gate fused == silu_and_mul(linear(x, [gate; up]))."""
    rng = rng_for("tiled-silu", dtype, M, F, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (2 * F, K), dtype, 0.05)
    wt = ops.tile_weights(dev(w, dtype), interleave_halves=True)
    y = host(ops.linear_tiled(dev(x, dtype), wt, 2 * F, epilogue=ops.EPI_SILU_MUL))
    ref = R.silu_and_mul(R.linear(x, w, None, dtype), dtype)
    # the GEMM sums may round differently on a few elements (other accumulation order);
    # each such flip moves the gated product by at most a few ulps
    diff = np.abs(y - ref)
    assert (diff <= 4 * ULP[dtype] * np.abs(ref) + 4 * ULP[dtype] * 1e-2).all(), f"max diff {diff.max()}"
    assert (diff > 0).mean() <= 0.03
    # and against the unfused HIP path on the same machine
    gu = ops.linear_tiled(dev(x, dtype), ops.tile_weights(dev(w, dtype)), 2 * F)
    np.testing.assert_array_equal(y, host(ops.silu_and_mul(gu)))


def _slab_sum(part, bias, dtype):
    """round(slab_0 + slab_1 + ... + bias) in fp32, slab order: what the consumers of partial sums form."""
    acc = np.zeros(part.shape[1:], np.float32)
    for s in range(part.shape[0]):
        acc = acc + part[s]
    if bias is not None:
        acc = acc + bias.astype(np.float32)
    return R.round_to(acc, dtype)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("T,H,S,bias", [(32, 4096, 4, False), (5, 448, 1, True), (3, 8192, 7, False),
                                        (2, 16384, 2, True), (1, 256, 8, False)])
def test_fused_add_rms_norm_partial(ops, dtype, T, H, S, bias):
    rng = rng_for("rms-partial", dtype, T, H, S)
    part = (rng.standard_normal((S, T, H)) * 0.7).astype(np.float32)
    b = grid(rng, (H,), dtype) if bias else None
    w, res = R.round_to(1 + 0.1 * rng.standard_normal(H), dtype), grid(rng, (T, H), dtype)
    x = _slab_sum(part, b, dtype)
    y_ref, z_ref = R.fused_add_rms_norm(x, res, w, 1e-5, dtype)
    rd = dev(res, dtype)
    out = torch.empty(T, H, dtype=DT[dtype], device="cuda")
    ops.fused_add_rms_norm_partial(torch.from_numpy(part).cuda(), S, rd, dev(w, dtype), 1e-5, out,
                                   bias=dev(b, dtype) if bias else None)
    np.testing.assert_array_equal(host(rd), z_ref)            # slab sum + residual add: bit-exact
    assert_ulp_close(host(out), y_ref, dtype, 0.005)
    # and identical to the two-kernel path on the same machine
    rd2 = dev(res, dtype)
    out2 = ops.fused_add_rms_norm(dev(x, dtype), rd2, dev(w, dtype), 1e-5)
    np.testing.assert_array_equal(host(out), host(out2))


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("nh,kvh,d,S,bias", [(4, 2, 64, 2, False), (32, 8, 128, 2, False), (28, 4, 128, 4, True),
                                             (7, 1, 64, 1, True)])
def test_rope_kv_write_partial(ops, dtype, nh, kvh, d, S, bias):
    rng = rng_for("rope-partial", dtype, nh, kvh, d, S)
    T, nblk, max_pos = 29, 10, 512
    width = (nh + 2 * kvh) * d
    table = R.rope_cos_sin_table(max_pos, d, 5e5, dtype)
    part = (rng.standard_normal((S, T, width)) * 0.6).astype(np.float32)
    b = grid(rng, (width,), dtype) if bias else None
    qkv = _slab_sum(part, b, dtype)
    pos = rng.integers(0, max_pos, T).astype(np.int64)
    slots = rng.permutation(nblk * 16)[:T].astype(np.int64)
    slots[5] = -1
    kc_ref = np.zeros((nblk, kvh, 16, d), np.float32)
    vc_ref = np.zeros((nblk, kvh, d, 16), np.float32)
    ref = R.rope_kv_write(qkv, pos, table, slots, kc_ref, vc_ref, nh, kvh, d, dtype)

    kc, vc = make_cache(nblk, kvh, d, dtype)
    qd = torch.full((T, width), 7.0, dtype=DT[dtype], device="cuda")     # every element must be overwritten
    ops.rope_kv_write(qd, torch.from_numpy(pos).cuda(), dev(table, dtype), torch.from_numpy(slots).cuda(),
                      kc, vc, nh, kvh, d, partial=torch.from_numpy(part).cuda(), num_slabs=S,
                      bias=dev(b, dtype) if bias else None)
    np.testing.assert_array_equal(host(qd), ref)              # bit-exact, v rows included
    np.testing.assert_array_equal(host(kc), kc_ref)
    np.testing.assert_array_equal(host(vc), vc_ref)


def test_tiled_decode_chain_matches_plain_chain(ops):
    """o_proj in partial-sum form + fused slab-sum/add/norm == linear -> fused_add_rms_norm."""
    dtype = "bfloat16"
    rng = rng_for("tiled-chain")
    M, N, K, G = 32, 4096, 4096, 4
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.02)
    res, nw = grid(rng, (M, N), dtype), R.round_to(1 + 0.1 * rng.standard_normal(N), dtype)
    wt = ops.tile_weights(dev(w, dtype))
    part = ops.linear_tiled(dev(x, dtype), wt, N, k_groups=G, epilogue=ops.EPI_PARTIAL, tiles_per_wg=104)
    rd = dev(res, dtype)
    out = torch.empty(M, N, dtype=DT[dtype], device="cuda")
    ops.fused_add_rms_norm_partial(part, G, rd, dev(nw, dtype), 1e-5, out)
    lin = R.linear(x, w, None, dtype)
    y_ref, z_ref = R.fused_add_rms_norm(lin, res, nw, 1e-5, dtype)
    assert (host(rd) != z_ref).mean() <= 0.02                  # different summation order: rare 1-ulp flips
    # a flip is one ulp of the GEMM output, which may be larger than the sum it enters
    assert (np.abs(host(rd) - z_ref) <= 2 * ULP[dtype] * (np.abs(lin) + np.abs(z_ref))).all()
    d = np.abs(host(out) - y_ref)
    assert (d <= 4 * ULP[dtype] * np.abs(y_ref) + 1e-3).all()


# --------------------------------------------------------------------------- #
# GEMM launches with the K-split reduction / residual add / RMSNorm fused in
# --------------------------------------------------------------------------- #
def _groups(N, tiles):
    return -(-((N + 15) // 16) // (tiles % 100))


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,tiles,groups,bias", [(32, 4096, 4096, 104, 4, False), (32, 4096, 14336, 104, 4, False),
                                                     (9, 520, 2048, 104, 2, True), (32, 1024, 1024, 8, 1, True),
                                                     (17, 6144, 4096, 104, 2, True), (1, 64, 512, 104, 1, False)])
def test_linear_tiled_reduce_in_launch(ops, dtype, M, N, K, tiles, groups, bias):
    """Epilogues 3 and 4: the slabs are added by the workgroup that finishes last, in slab
    order -- bit-identical to adding the slabs of the partial-sum epilogue afterwards."""
    rng = rng_for("tiled-reduce", dtype, M, N, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.05)
    b = grid(rng, (N,), dtype) if bias else None
    res = grid(rng, (M, N), dtype)
    wt, xd = ops.tile_weights(dev(w, dtype)), dev(x, dtype)
    bd = dev(b, dtype) if bias else None
    part = ops.linear_tiled(xd, wt, N, k_groups=groups, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles).cpu().numpy()
    y_ref = _slab_sum(part, b, dtype)
    counters = torch.zeros(_groups(N, tiles), dtype=torch.int32, device="cuda")
    for _ in range(2):                                   # the counters re-arm themselves
        y = ops.linear_tiled_fused(xd, wt, N, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_REDUCE_STORE,
                                   bias=bd, counters=counters)
        np.testing.assert_array_equal(host(y), y_ref)
    assert int(counters.abs().sum()) == 0
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    assert (np.abs(y_ref - ref64) <= _linear_tol(ref64, x, w, K, dtype)).all()

    rd = dev(res, dtype)
    sumsq = torch.full((_groups(N, tiles), 32), -1.0, dtype=torch.float32, device="cuda")
    ops.linear_tiled_fused(xd, wt, N, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_REDUCE_RESID, bias=bd,
                           residual=rd, sumsq_out=sumsq, counters=counters)
    z_ref = R.round_to(y_ref + res, dtype)
    np.testing.assert_array_equal(host(rd), z_ref)
    ss = sumsq.cpu().numpy()[:, :M].astype(np.float64).sum(0)
    np.testing.assert_allclose(ss, (z_ref.astype(np.float64) ** 2).sum(1), rtol=1e-5)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,H,N2,tiles,epi", [(32, 4096, 6144, 104, "partial"), (32, 4096, 28672, 8, "silu"),
                                              (5, 1024, 1000, 8, "store"), (20, 1024, 1536, 104, "partial")])
def test_linear_tiled_norm_prologue(ops, dtype, M, H, N2, tiles, epi):
    """A GEMM that leaves residual + sums of squares behind, then a GEMM that normalises the
    residual while staging it == linear(rms_norm(residual)) of the oracle."""
    rng = rng_for("tiled-norm", dtype, M, H, N2)
    K1 = 1024
    a, w1 = grid(rng, (M, K1), dtype), grid(rng, (H, K1), dtype, 0.05)
    res0 = grid(rng, (M, H), dtype)
    nw, w2 = R.round_to(1 + 0.1 * rng.standard_normal(H), dtype), grid(rng, (N2, H), dtype, 0.03)
    b2 = grid(rng, (N2,), dtype) if epi == "store" else None
    g1 = _groups(H, 104)
    counters = torch.zeros(g1, dtype=torch.int32, device="cuda")
    sumsq = torch.zeros(g1, 32, dtype=torch.float32, device="cuda")
    rd = dev(res0, dtype)
    ops.linear_tiled_fused(dev(a, dtype), ops.tile_weights(dev(w1, dtype)), H, tiles_per_wg=104, k_groups=2,
                           epilogue=ops.EPI_REDUCE_RESID, residual=rd, sumsq_out=sumsq, counters=counters)
    res = host(rd)                                       # checked bit-exactly by the test above
    eps = 1e-5
    xn = R.rms_norm(res, nw, eps, dtype)
    kw = dict(tiles_per_wg=tiles, norm_weight=dev(nw, dtype), norm_sumsq=sumsq, norm_parts=g1, eps=eps)
    if epi == "silu":
        wt2 = ops.tile_weights(dev(w2, dtype), interleave_halves=True)
        y = host(ops.linear_tiled_fused(rd, wt2, N2, epilogue=ops.EPI_SILU_MUL, **kw))
        ref = R.silu_and_mul(R.linear(xn, w2, None, dtype), dtype)
        diff = np.abs(y - ref)
        assert (diff <= 6 * ULP[dtype] * np.abs(ref) + 6 * ULP[dtype] * 1e-2).all(), f"max diff {diff.max()}"
        assert (diff > 0).mean() <= 0.05
        return
    wt2 = ops.tile_weights(dev(w2, dtype))
    if epi == "partial":
        part = ops.linear_tiled_fused(rd, wt2, N2, k_groups=2, epilogue=ops.EPI_PARTIAL, **kw).cpu().numpy()
        y = _slab_sum(part, None, dtype)
    else:
        y = host(ops.linear_tiled_fused(rd, wt2, N2, epilogue=ops.EPI_STORE, bias=dev(b2, dtype), **kw))
    ref64 = xn.astype(np.float64) @ w2.astype(np.float64).T + (b2.astype(np.float64) if b2 is not None else 0.0)
    # a few normalised inputs may round the other way (another summation order of the squares)
    tol = _linear_tol(ref64, xn, w2, H, dtype) + 4 * ULP[dtype] * np.abs(xn).max() * np.abs(w2).max()
    err = np.abs(y - ref64)
    assert (err <= tol).all(), f"max err {err.max()}"
    assert (y != R.linear(xn, w2, b2, dtype)).mean() <= 0.05


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,tiles,groups", [(64, 4096, 4096, 104, 4), (33, 1000, 1024, 8, 1), (64, 6144, 4096, 8, 1),
                                                (128, 4096, 14336, 8, 7), (100, 1536, 4096, 8, 8), (65, 320, 512, 8, 1),
                                                (128, 28672, 4096, 8, 1), (48, 4096, 1024, 104, 2),
                                                (64, 4096, 4096, 108, 4), (40, 1000, 1024, 108, 1), (64, 4736, 3584, 108, 1),
                                                (64, 4096, 4096, 208, 4), (40, 1000, 1024, 216, 1), (64, 4736, 3584, 208, 1),
                                                (128, 4096, 14336, 216, 7), (100, 1544, 4096, 216, 8), (65, 336, 512, 216, 1),
                                                (64, 4736, 3584, 210, 1), (50, 1000, 1024, 212, 2), (33, 4096, 4096, 214, 4),
                                                (128, 4096, 14336, 210, 7), (100, 1544, 4096, 212, 8), (65, 336, 512, 214, 1)])
def test_linear_tiled_up_to_128_rows(ops, dtype, M, N, K, tiles, groups):
    """33..128 rows (more m-tiles per wave, narrower x slices): store, partial slabs and the
    in-launch reduction against the fp64 product; the reduction bit-identical to the slab sum."""
    rng = rng_for("tiled-rows", dtype, M, N, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.05)
    wt, xd = ops.tile_weights(dev(w, dtype)), dev(x, dtype)
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T
    tol = _linear_tol(ref64, x, w, K, dtype)
    if groups == 1:
        y = host(ops.linear_tiled(xd, wt, N, tiles_per_wg=tiles))
        assert (np.abs(y - ref64) <= tol).all()
        assert (y != R.linear(x, w, None, dtype)).mean() <= 0.02
        y0 = host(ops.linear_tiled(xd, wt, N))                       # geometry chosen by the wrapper
        assert (np.abs(y0 - ref64) <= tol).all()
    part = ops.linear_tiled(xd, wt, N, k_groups=groups, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles).cpu().numpy()
    kg = K // groups
    for g in range(groups):
        ref = x[:, g * kg:(g + 1) * kg].astype(np.float64) @ w[:, g * kg:(g + 1) * kg].astype(np.float64).T
        np.testing.assert_allclose(part[g], ref, rtol=1e-4, atol=2e-3)
    y_ref = _slab_sum(part, None, dtype)
    assert (np.abs(y_ref - ref64) <= tol).all()
    if N % 8 == 0 and tiles < 200:                                   # the two-tiles-per-wave forms have no in-launch reduction
        counters = torch.zeros(_groups(N, tiles), dtype=torch.int32, device="cuda")
        for _ in range(2):
            y = ops.linear_tiled_fused(xd, wt, N, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_REDUCE_STORE,
                                       counters=counters)
            np.testing.assert_array_equal(host(y), y_ref)


@pytest.mark.parametrize("M,F,K", [(64, 14336, 4096), (128, 14336, 4096), (40, 512, 1024), (128, 64, 512)])
def test_linear_tiled_swiglu_epilogue_up_to_128_rows(ops, M, F, K):
    dtype = "bfloat16"
    rng = rng_for("tiled-silu-rows", M, F, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (2 * F, K), dtype, 0.05)
    xd = dev(x, dtype)
    wi = ops.tile_weights(dev(w, dtype), interleave_halves=True)
    y = host(ops.linear_tiled(xd, wi, 2 * F, epilogue=ops.EPI_SILU_MUL))
    gu = ops.linear_tiled(xd, ops.tile_weights(dev(w, dtype)), 2 * F)
    np.testing.assert_array_equal(y, host(ops.silu_and_mul(gu)))
    # two tiles per wave: the wave holds the gate and the up tile itself; same sums, same bits
    for tiles in ([208, 210, 212, 214, 216] if M <= 64 else [210, 212, 214, 216]):
        if K % ops.tiled_slice_width(M, tiles) == 0:
            y2 = host(ops.linear_tiled(xd, wi, 2 * F, epilogue=ops.EPI_SILU_MUL, tiles_per_wg=tiles))
            np.testing.assert_array_equal(y2, y)
    ref = R.silu_and_mul(R.linear(x, w, None, dtype), dtype)
    diff = np.abs(y - ref)
    assert (diff <= 4 * ULP[dtype] * np.abs(ref) + 4 * ULP[dtype] * 1e-2).all()


def test_linear_tiled_rejects_unbuilt_geometries(ops):
    from mi355_decode._lib import Mi355Error
    x = torch.zeros(128, 1024, dtype=torch.bfloat16, device="cuda")
    wt = ops.tile_weights(torch.zeros(64, 1024, dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(Mi355Error, match="no kernel for M 128"):
        ops.linear_tiled(x, wt, 64, tiles_per_wg=4)
    with pytest.raises(Mi355Error, match="multiple of the x slice"):
        ops.linear_tiled(x[:, :640].contiguous(), ops.tile_weights(torch.zeros(64, 640, dtype=torch.bfloat16, device="cuda")),
                         64, tiles_per_wg=8)
    with pytest.raises(Mi355Error, match="outside"):
        ops.linear_tiled(torch.zeros(257, 1024, dtype=torch.bfloat16, device="cuda"), wt, 64)
    with pytest.raises(Mi355Error, match="up to 128 rows"):
        ops.linear_tiled_fused(torch.zeros(129, 1024, dtype=torch.bfloat16, device="cuda"), wt, 64, tiles_per_wg=8, k_groups=2,
                               epilogue=ops.EPI_REDUCE_STORE, counters=torch.zeros(8, dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,tiles,groups", [(128, 1536, 4096, 8, 4), (192, 1536, 4096, 8, 4), (256, 1536, 4096, 8, 8),
                                                (129, 320, 512, 8, 1), (256, 4096, 1024, 302, 1), (192, 4096, 3584, 8, 7),
                                                (256, 7168, 4096, 216, 4), (200, 1544, 4096, 212, 8), (256, 336, 512, 214, 1),
                                                (256, 4096, 512, 302, 2), (130, 1000, 1024, 210, 2), (256, 16032, 4096, 8, 1)])
def test_linear_tiled_129_to_256_rows(ops, dtype, M, N, K, tiles, groups):
    """129 .. 256 rows (the decode step of a tensor-parallel rank that serves 256 sequences): two row
    blocks of one launch.  Store, partial slabs and both slab consumers against the fp64 product, and
    every row block bit-identical to the same rows run as a launch of their own."""
    rng = rng_for("tiled-rowblocks", dtype, M, N, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.05)
    wt, xd = ops.tile_weights(dev(w, dtype)), dev(x, dtype)
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T
    tol = _linear_tol(ref64, x, w, K, dtype)
    if groups == 1:
        yd = ops.linear_tiled(xd, wt, N, tiles_per_wg=tiles)
        y = host(yd)
        assert (np.abs(y - ref64) <= tol).all()
        assert (y != R.linear(x, w, None, dtype)).mean() <= 0.02
        for a in range(0, M, 128):                                   # a row block computes what a launch of its rows computes
            if M - a <= 64:                                          # (fewer rows take another geometry: another slice width)
                continue
            np.testing.assert_array_equal(host(ops.linear_tiled(xd[a:a + 128].contiguous(), wt, N, tiles_per_wg=tiles)),
                                          y[a:a + 128])
        y0 = host(ops.linear_tiled(xd, wt, N))                       # geometry chosen by the wrapper
        assert (np.abs(y0 - ref64) <= tol).all()
    partd = ops.linear_tiled(xd, wt, N, k_groups=groups, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles)
    part = partd.cpu().numpy()
    assert part.shape == (groups, M, N)
    unit = ops.tiled_slice_width(M, tiles)
    units = K // unit
    edges = [unit * (g * (units // groups) + min(g, units % groups)) for g in range(groups + 1)]
    for g in range(groups):
        ref = x[:, edges[g]:edges[g + 1]].astype(np.float64) @ w[:, edges[g]:edges[g + 1]].astype(np.float64).T
        np.testing.assert_allclose(part[g], ref, rtol=1e-4, atol=2e-3)
    y_ref = _slab_sum(part, None, dtype)
    assert (np.abs(y_ref - ref64) <= tol).all()
    if N % 8 == 0:
        ys = ops.sum_slabs(partd, groups, torch.empty(M, N, dtype=DT[dtype], device="cuda"))
        np.testing.assert_array_equal(host(ys), y_ref)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,tiles,groups", [(128, 1536, 4096, 1008, 8), (128, 4096, 1024, 1108, 4), (128, 7168, 4096, 1214, 4),
                                                (128, 4096, 3584, 1008, 4), (256, 768, 4096, 1208, 8), (256, 4096, 512, 1302, 1),
                                                (256, 4096, 512, 2104, 1), (128, 4096, 1024, 2302, 1), (200, 1544, 4096, 1104, 2),
                                                (100, 1000, 1024, 1210, 2), (65, 320, 1024, 2008, 1), (256, 4096, 1792, 1402, 1),
                                                (192, 1536, 4096, 2108, 4), (128, 336, 1024, 1304, 1)])
def test_linear_tiled_row_blocks_of_64_and_32(ops, dtype, M, N, K, tiles, groups):
    """tiles_per_wg + 1000 / + 2000: the rows of a 65 .. 256-row launch in blocks of 64 / 32, each block by the
    geometry of that row class.  Store, partial slabs and the slab sum against the fp64 product and the oracle,
    and every (full enough) block bit-identical to the same rows run as a launch of their own with the base code."""
    rng = rng_for("tiled-rowblocks-64-32", dtype, M, N, K, tiles)
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.05)
    wt, xd = ops.tile_weights(dev(w, dtype)), dev(x, dtype)
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T
    tol = _linear_tol(ref64, x, w, K, dtype)
    rb = ops.row_block_rows(tiles)
    assert rb == (64 if tiles < 2000 else 32) and ops.tiled_slice_width(M, tiles) == ops.tiled_slice_width(rb, tiles % 1000)
    if groups == 1:
        y = host(ops.linear_tiled(xd, wt, N, tiles_per_wg=tiles))
        assert (np.abs(y - ref64) <= tol).all()
        assert (y != R.linear(x, w, None, dtype)).mean() <= 0.02
        for a in range(0, M, rb):
            if M - a <= rb // 2:                                     # (fewer rows take another geometry: another slice width)
                continue
            np.testing.assert_array_equal(host(ops.linear_tiled(xd[a:a + rb].contiguous(), wt, N, tiles_per_wg=tiles % 1000)),
                                          y[a:a + rb])
    partd = ops.linear_tiled(xd, wt, N, k_groups=groups, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles)
    part = partd.cpu().numpy()
    assert part.shape == (groups, M, N)
    unit = ops.tiled_slice_width(M, tiles)
    units = K // unit
    edges = [unit * (g * (units // groups) + min(g, units % groups)) for g in range(groups + 1)]
    for g in range(groups):
        ref = x[:, edges[g]:edges[g + 1]].astype(np.float64) @ w[:, edges[g]:edges[g + 1]].astype(np.float64).T
        np.testing.assert_allclose(part[g], ref, rtol=1e-4, atol=2e-3)
    y_ref = _slab_sum(part, None, dtype)
    assert (np.abs(y_ref - ref64) <= tol).all()
    if N % 8 == 0:
        ys = ops.sum_slabs(partd, groups, torch.empty(M, N, dtype=DT[dtype], device="cuda"))
        np.testing.assert_array_equal(host(ys), y_ref)


@pytest.mark.parametrize("M,F,K,tiles", [(128, 3584, 4096, 1302), (256, 1792, 4096, 1216), (128, 3584, 4096, 2008),
                                         (200, 512, 1024, 1104)])
def test_linear_tiled_row_blocks_swiglu_epilogue(ops, M, F, K, tiles):
    """The SwiGLU epilogue in row blocks of 64 / 32: bit-identical to the unfused pair of the same form."""
    dtype = "bfloat16"
    rng = rng_for("tiled-rowblocks-swiglu", M, F, K, tiles)
    x, w = grid(rng, (M, K), dtype), grid(rng, (2 * F, K), dtype, 0.05)
    xd = dev(x, dtype)
    wt_i = ops.tile_weights(dev(w, dtype), interleave_halves=True)
    act = host(ops.linear_tiled(xd, wt_i, 2 * F, tiles_per_wg=tiles, epilogue=ops.EPI_SILU_MUL))
    gu = ops.linear_tiled(xd, ops.tile_weights(dev(w, dtype)), 2 * F, tiles_per_wg=tiles)
    np.testing.assert_array_equal(act, host(ops.silu_and_mul(gu)))
    ref = R.silu_and_mul(R.linear(x, w, None, dtype), dtype)
    assert (act != ref).mean() <= 0.02


def test_linear_tiled_row_blocks_reject_fused_forms(ops):
    from mi355_decode._lib import Mi355Error
    wt = ops.tile_weights(torch.zeros(64, 1024, dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(Mi355Error, match="row blocks of 64 / 32"):
        ops.linear_tiled_fused(torch.zeros(128, 1024, dtype=torch.bfloat16, device="cuda"), wt, 64, tiles_per_wg=1008, k_groups=2,
                               epilogue=ops.EPI_REDUCE_STORE, counters=torch.zeros(8, dtype=torch.int32, device="cuda"))
    with pytest.raises(Mi355Error):
        ops.linear_tiled(torch.zeros(128, 1024, dtype=torch.bfloat16, device="cuda"), wt, 64, tiles_per_wg=3008)
    with pytest.raises(Mi355Error, match="no kernel"):                # two tiles per wave exist from 33 rows
        ops.linear_tiled(torch.zeros(128, 1024, dtype=torch.bfloat16, device="cuda"), wt, 64, tiles_per_wg=2216)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,S,bias", [(128, 4096, 4, False), (256, 4096, 8, True), (5, 448, 1, True), (33, 1000, 3, False),
                                        (192, 8192, 2, False), (1, 8, 7, False)])
def test_sum_slabs(ops, dtype, M, N, S, bias):
    """Slabs -> the 16-bit tensor an all-reduce takes: the fixed-order fp32 slab sum (+ bias), one rounding; bit-exact
    against the oracle's slab sum, with a strided destination and slabs beyond num_slabs ignored."""
    rng = rng_for("sum-slabs", dtype, M, N, S)
    part = (rng.standard_normal((S + 1, M, N)) * 0.6).astype(np.float32)
    b = grid(rng, (N,), dtype) if bias else None
    pd = torch.from_numpy(part).cuda()
    out = torch.full((M, N + 8), 7.0, dtype=DT[dtype], device="cuda")
    ops.sum_slabs(pd, S, out[:, :N], dev(b, dtype) if bias else None)      # the slab after the first S is not read
    ref = _slab_sum(part[:S], b, dtype)
    np.testing.assert_array_equal(host(out[:, :N]), ref)
    assert (host(out[:, N:]) == 7.0).all()
    from mi355_decode._lib import Mi355Error
    with pytest.raises((Mi355Error, ValueError)):
        ops.sum_slabs(pd, 9, out[:, :N])
    with pytest.raises((Mi355Error, ValueError)):
        ops.sum_slabs(pd.to(torch.float64), S, out[:, :N])


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,F,K,tiles,groups", [(128, 3584, 4096, 8, 4), (256, 1792, 4096, 8, 4), (32, 1536, 1024, 104, 2),
                                                (64, 512, 1024, 216, 2), (200, 3584, 4096, 302, 2), (7, 64, 2048, 4, 2),
                                                (128, 3584, 4096, 212, 8)])
def test_silu_and_mul_slabs(ops, dtype, M, F, K, tiles, groups):
    """SwiGLU from the slabs of a K-split gate_up projection: bit-identical to the oracle's SwiGLU of the rounded
    fixed-order slab sums (de-interleaved), and to the GEMM's own SwiGLU epilogue when there is one slab."""
    rng = rng_for("silu-slabs", dtype, M, F, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (2 * F, K), dtype, 0.05)
    xd = dev(x, dtype)
    wi = ops.tile_weights(dev(w, dtype), interleave_halves=True)
    partd = ops.linear_tiled(xd, wi, 2 * F, k_groups=groups, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles)
    act = host(ops.silu_and_mul_slabs(partd, groups, torch.empty(M, F, dtype=DT[dtype], device="cuda")))
    # slab columns: 16-column tiles alternating gate / up
    summed = _slab_sum(partd.cpu().numpy(), None, dtype).reshape(M, F // 16, 2, 16)
    gate_up = np.concatenate([summed[:, :, 0].reshape(M, F), summed[:, :, 1].reshape(M, F)], axis=1)
    np.testing.assert_array_equal(act, host(ops.silu_and_mul(dev(gate_up, dtype))))     # the SwiGLU kernel's bits
    assert_ulp_close(act, R.silu_and_mul(gate_up, dtype), dtype, 0.005)                 # ... and the oracle's bar for SwiGLU
    # against the exact product: SwiGLU of the correctly rounded gate / up values, up to the GEMM's ulp flips
    ref = R.silu_and_mul(R.linear(x, w, None, dtype), dtype)
    diff = np.abs(act - ref)
    assert (diff <= 4 * ULP[dtype] * np.abs(ref) + 4 * ULP[dtype] * 1e-2).all()
    one = ops.linear_tiled(xd, wi, 2 * F, k_groups=1, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles)
    np.testing.assert_array_equal(host(ops.silu_and_mul_slabs(one, 1, torch.empty(M, F, dtype=DT[dtype], device="cuda"))),
                                  host(ops.linear_tiled(xd, wi, 2 * F, epilogue=ops.EPI_SILU_MUL, tiles_per_wg=tiles)))


# --------------------------------------------------------------------------- #
# decode attention with the rotary embedding + KV store fused in
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("nh,kvh,d", [(4, 2, 64), (32, 8, 128), (28, 4, 128), (7, 1, 64), (8, 1, 128)])
@pytest.mark.parametrize("splits,slabs", [(1, 0), (1, 2), (3, 4), (2, 0)])
def test_paged_attn_decode_fused(ops, dtype, nh, kvh, d, splits, slabs):
    """One launch == rope_kv_write followed by paged_attn_decode: the attention output, the
    K pages and the V pages are bit-identical, from 16-bit rows (slabs == 0) and from fp32
    partial-sum slabs + bias; and the output matches the oracle."""
    _check_decode_fused(ops, dtype, nh, kvh, d, splits, slabs, [1, 16, 17, 33, 128, 129, 200, 255, 513, 0, 64])


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("nh,kvh,d,splits,slabs", [(32, 8, 128, 1, 0), (32, 8, 128, 1, 4), (32, 8, 128, 2, 0), (16, 8, 64, 1, 2)])
def test_paged_attn_decode_more_workgroups_than_cus(ops, dtype, nh, kvh, d, splits, slabs):
    """More than 256 workgroups (sequences x kv heads x splits: a wide batch): the launch takes the form with ONE
    register stage of K/V per wave, two workgroups to a CU.  Same checks as above, on 44 sequences, and the plain
    kernel against the oracle on its own."""
    ctxs = [1, 16, 17, 33, 128, 129, 200, 255, 513, 0, 64, 300, 31, 1000, 2, 100, 257, 640, 48, 15, 96, 191] * 2
    assert len(ctxs) * kvh * splits > 256
    _check_decode_fused(ops, dtype, nh, kvh, d, splits, slabs, ctxs)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("nh,kvh,d,splits,slabs", [(32, 8, 128, 1, 0), (32, 8, 128, 1, 4), (32, 8, 128, 2, 1), (16, 8, 64, 1, 2),
                                                   (40, 4, 128, 1, 0)])
def test_paged_attn_decode_one_wave_per_group(ops, dtype, nh, kvh, d, splits, slabs):
    """From 1024 (sequence, kv head, split) groups on, a launch gives every group ONE wave that walks all of its
    32-token steps (no merge across waves).  Same checks as above on 132 or 264 sequences; the last case has 10
    query heads per kv head, more than the one-wave prologue has threads for, and must take the four-wave form."""
    ctxs = [1, 16, 17, 33, 128, 129, 200, 255, 513, 0, 64, 300, 31, 1000, 2, 100, 257, 640, 48, 15, 96, 191] * (6 if kvh == 8 else 12)
    assert len(ctxs) * kvh * splits >= 1024
    _check_decode_fused(ops, dtype, nh, kvh, d, splits, slabs, ctxs)


def _check_decode_fused(ops, dtype, nh, kvh, d, splits, slabs, ctxs):
    rng = rng_for("dec-fused", dtype, nh, kvh, d, splits, slabs, len(ctxs))
    n = len(ctxs)
    kc, vc, bt = build_paged(rng, ctxs, kvh, d, dtype, poison=False)
    width = (nh + 2 * kvh) * d
    max_pos = 1024
    table = R.rope_cos_sin_table(max_pos, d, 5e5, dtype)
    pos = np.array([max(c - 1, 0) for c in ctxs], np.int64)
    slot = np.array([bt[i, p // 16] * 16 + p % 16 if c else -1 for i, (c, p) in enumerate(zip(ctxs, pos))], np.int64)
    if slabs:
        part = (rng.standard_normal((slabs, n, width)) * 0.6).astype(np.float32)
        bias = grid(rng, (width,), dtype)
        qkv = _slab_sum(part, bias, dtype)
    else:
        qkv = grid(rng, (n, width), dtype)
    dv = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    ctx_d, bt_d, pos_d, slot_d, tab_d = dv(np.array(ctxs, np.int32)), dv(bt), dv(pos), dv(slot), dev(table, dtype)
    scale = d ** -0.5

    # two kernels
    kc1, vc1, q1 = dev(kc, dtype), dev(vc, dtype), dev(qkv, dtype)
    ops.rope_kv_write(q1, pos_d, tab_d, slot_d, kc1, vc1, nh, kvh, d)
    out1 = ops.paged_attn_decode(q1[:, : nh * d], kc1, vc1, bt_d, ctx_d, nh, kvh, d, scale, num_splits=splits)
    # one kernel
    kc2, vc2 = dev(kc, dtype), dev(vc, dtype)
    if slabs:
        out2 = ops.paged_attn_decode_fused(None, tab_d, slot_d, kc2, vc2, bt_d, ctx_d, nh, kvh, d, scale,
                                           num_splits=splits, partial=dv(part), num_slabs=slabs, bias=dev(bias, dtype))
    else:
        q2 = dev(qkv, dtype)
        out2 = ops.paged_attn_decode_fused(q2, tab_d, slot_d, kc2, vc2, bt_d, ctx_d, nh, kvh, d, scale,
                                           num_splits=splits)
        np.testing.assert_array_equal(host(q2), qkv)             # the packed rows are left as they were
    np.testing.assert_array_equal(host(kc2), host(kc1))
    np.testing.assert_array_equal(host(vc2), host(vc1))
    np.testing.assert_array_equal(host(out2), host(out1))

    # and against the oracle
    kc_ref, vc_ref = kc.copy(), vc.copy()
    rot = R.rope_kv_write(qkv, pos, table, slot, kc_ref, vc_ref, nh, kvh, d, dtype)
    ref = R.paged_attn_decode(rot[:, : nh * d].reshape(n, nh, d), kc_ref, vc_ref, bt, np.array(ctxs, np.int32), scale, dtype)
    attn_close(host(out2).reshape(n, nh, d), ref, dtype)
    np.testing.assert_array_equal(host(kc2), kc_ref)
    np.testing.assert_array_equal(host(vc2), vc_ref)


def _k_ranges(K, unit, groups):
    """K range of every group: units of `unit` dealt out as evenly as they go, larger groups first."""
    units = K // unit
    base, rem = divmod(units, groups)
    out, start = [], 0
    for g in range(groups):
        n = (base + (1 if g < rem else 0)) * unit
        out.append((start, start + n))
        start += n
    return out


@pytest.mark.parametrize("M,N,K,tiles,groups,unit", [(32, 3584, 3584, 104, 3, 512), (64, 3584, 18944, 104, 4, 256),
                                                     (16, 512, 7168, 8, 3, 1024), (128, 1024, 18944, 8, 5, 256),
                                                     (20, 256, 2560, 104, 5, 512), (100, 512, 768, 8, 2, 256)])
def test_linear_tiled_uneven_k_split(ops, M, N, K, tiles, groups, unit):
    """K / slice is not a multiple of the group count (Qwen2-7B: 3584 = 7 x 512, 18944 = 37 x 512)."""
    dtype = "bfloat16"
    rng = rng_for("tiled-uneven", M, N, K, groups)
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.05)
    wt, xd = ops.tile_weights(dev(w, dtype)), dev(x, dtype)
    part = ops.linear_tiled(xd, wt, N, k_groups=groups, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles).cpu().numpy()
    for g, (a, b) in enumerate(_k_ranges(K, unit, groups)):
        ref = x[:, a:b].astype(np.float64) @ w[:, a:b].astype(np.float64).T
        np.testing.assert_allclose(part[g], ref, rtol=1e-4, atol=2e-3)
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T
    y = _slab_sum(part, None, dtype)
    assert (np.abs(y - ref64) <= _linear_tol(ref64, x, w, K, dtype)).all()
    counters = torch.zeros(_groups(N, tiles), dtype=torch.int32, device="cuda")
    y2 = ops.linear_tiled_fused(xd, wt, N, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_REDUCE_STORE,
                                counters=counters)
    np.testing.assert_array_equal(host(y2), y)


# --------------------------------------------------------------------------- #
# fp8 (e4m3fn) weight-only decode GEMM
# --------------------------------------------------------------------------- #
def _untile_fp8(qbytes, N, K):
    """[K/128][tiles][2][lane][16] bytes -> [tiles*16, K] codes (1 KB blocks of 16 rows x 64 k ordered
    k group, tile, pair in group; lane (i, h): bytes 0-7 are row i's k = 64p + 8h + j, bytes 8-15 its
    k = 64p + 32 + 8h + j)."""
    tiles = (N + 15) // 16
    b = qbytes.reshape(K // 128, tiles, 2, 4, 16, 2, 8)      # k group, tile, pair, h, i, half, j
    return b.transpose(1, 4, 0, 2, 5, 3, 6).reshape(tiles * 16, K)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("N,K,inter", [(64, 512, False), (1000, 1024, False), (4096, 4096, False), (1536, 1024, True)])
def test_tile_weights_fp8(ops, dtype, N, K, inter):
    """Quantise + tile: codes and scales bit-exact against the oracle's e4m3fn rounding."""
    rng = rng_for("fp8-tile", dtype, N, K)
    w = grid(rng, (N, K), dtype, 0.05)
    w[3] = 0.0                                                  # an all-zero row: scale 1, codes 0
    w[5, 7] = R.round_to(np.float32(3.0), dtype)                # an outlier sets its row's scale
    q, scale = ops.tile_weights_fp8(dev(w, dtype), interleave_halves=inter)
    codes = _untile_fp8(q.cpu().numpy(), N, K)
    vals = R.fp8_e4m3fn_decode_table()[codes]
    q_ref, s_ref = R.quantize_rows_fp8(w)
    if inter:                                                   # tiles alternate gate / up rows
        order = np.concatenate([np.concatenate([np.arange(16 * t, 16 * t + 16), N // 2 + np.arange(16 * t, 16 * t + 16)])
                                for t in range(N // 32)])
    else:
        order = np.arange(N)
    np.testing.assert_array_equal(scale.cpu().numpy()[: len(order)], s_ref[order])
    np.testing.assert_array_equal(vals[: len(order)], q_ref[order])
    assert (vals[len(order):] == 0).all()                       # padding rows


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("N,K,inter", [(64, 512, False), (1000, 1024, False), (4096, 4096, False), (1536, 1024, True),
                                       (40, 128, False)])
def test_untile_weights_is_the_inverse_of_tile_weights(ops, dtype, N, K, inter):
    """Tiled -> row-major gives back the same bits (rows the tiling padded are dropped), and a
    destination larger than the matrix is not written beyond it."""
    rng = rng_for("untile", dtype, N, K)
    w = grid(rng, (N, K), dtype, 0.05)
    wd = dev(w, dtype)
    t = ops.tile_weights(wd, interleave_halves=inter)
    guard = torch.full((N + 3, K), 7.0, dtype=wd.dtype, device="cuda")
    back = ops.untile_weights(t, N, K, wd.dtype, interleave_halves=inter, out=guard[:N])
    assert torch.equal(back.view(torch.int16), wd.view(torch.int16))
    assert (guard[N:] == 7.0).all()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("N,K,inter", [(64, 512, False), (1000, 1024, False), (1536, 1024, True), (4096, 4096, False)])
def test_untile_weights_fp8_decodes_to_the_quantised_matrix(ops, dtype, N, K, inter):
    """fp8 tiles -> row-major 16-bit: round(fp8 * row scale) bit for bit, i.e. the oracle's quantised
    model, which is also what the engine's row-major copies hold under --quantization fp8."""
    rng = rng_for("untile-fp8", dtype, N, K)
    w = grid(rng, (N, K), dtype, 0.05)
    w[3] = 0.0
    w[5, 7] = R.round_to(np.float32(3.0), dtype)
    wd = dev(w, dtype)
    qs = ops.tile_weights_fp8(wd, interleave_halves=inter)
    back = host(ops.untile_weights(qs, N, K, wd.dtype, interleave_halves=inter))
    q_ref, s_ref = R.quantize_rows_fp8(w)
    want = R.round_to((q_ref * s_ref[:, None]).astype(np.float32), dtype)
    np.testing.assert_array_equal(back, want)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,tiles,groups,bias", [(32, 4096, 4096, 104, 4, False), (32, 28672, 4096, 8, 1, False),
                                                     (7, 1000, 1024, 104, 1, True), (64, 3584, 18944, 104, 4, False),
                                                     (128, 4608, 3584, 8, 7, True), (1, 64, 512, 104, 1, False),
                                                     (48, 6144, 4096, 8, 1, True), (64, 3584, 18944, 108, 8, False),
                                                     (48, 6144, 4096, 108, 1, True), (64, 3584, 18944, 208, 8, False),
                                                     (48, 6144, 4096, 216, 1, True), (128, 4616, 3584, 216, 7, True),
                                                     (64, 3584, 18944, 210, 8, False), (48, 6144, 4096, 212, 1, True),
                                                     (128, 4616, 3584, 214, 7, True)])
def test_linear_tiled_w8(ops, dtype, M, N, K, tiles, groups, bias):
    rng = rng_for("w8", dtype, M, N, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.05)
    b = grid(rng, (N,), dtype) if bias else None
    q, scale = ops.tile_weights_fp8(dev(w, dtype))
    q_ref, s_ref = R.quantize_rows_fp8(w)
    ref = R.linear_w8(x, q_ref, s_ref, b, dtype)
    ref64 = (x.astype(np.float64) @ q_ref.astype(np.float64).T) * s_ref[None, :].astype(np.float64)
    if bias:
        ref64 = ref64 + b.astype(np.float64)
    tol = _linear_tol(ref64, x, q_ref * s_ref[:, None], K, dtype)
    xd = dev(x, dtype)
    if groups == 1:
        y = host(ops.linear_tiled_w8(xd, q, scale, N, bias=dev(b, dtype) if bias else None, tiles_per_wg=tiles))
        assert (np.abs(y - ref64) <= tol).all(), f"max err {np.abs(y - ref64).max()}"
        assert (y != ref).mean() <= 0.02
    part = ops.linear_tiled_w8(xd, q, scale, N, k_groups=groups, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles).cpu().numpy()
    y = _slab_sum(part, b, dtype)
    assert (np.abs(y - ref64) <= tol).all(), f"max err {np.abs(y - ref64).max()}"
    assert (y != ref).mean() <= 0.02
    # the quantisation error itself, against the unquantised product, is what fp8 costs: a few percent of the scale
    full = x.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    assert np.abs(y - full).max() <= 0.05 * np.abs(full).max() + 0.05


@pytest.mark.parametrize("M,F,K", [(32, 14336, 4096), (64, 18944, 3584), (5, 512, 1024)])
def test_linear_tiled_w8_swiglu(ops, M, F, K):
    dtype = "bfloat16"
    rng = rng_for("w8-silu", M, F, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (2 * F, K), dtype, 0.05)
    xd = dev(x, dtype)
    qi, si = ops.tile_weights_fp8(dev(w, dtype), interleave_halves=True)
    y = host(ops.linear_tiled_w8(xd, qi, si, 2 * F, epilogue=ops.EPI_SILU_MUL))
    q, s = ops.tile_weights_fp8(dev(w, dtype))
    gu = ops.linear_tiled_w8(xd, q, s, 2 * F)
    np.testing.assert_array_equal(y, host(ops.silu_and_mul(gu)))          # == the unfused HIP path
    if M > 32:                                                            # two tiles per wave, 5 .. 8 waves: same sums, same bits
        for tiles in (210, 212, 214, 216):
            y2 = host(ops.linear_tiled_w8(xd, qi, si, 2 * F, epilogue=ops.EPI_SILU_MUL, tiles_per_wg=tiles))
            np.testing.assert_array_equal(y2, y)
    q_ref, s_ref = R.quantize_rows_fp8(w)
    ref = R.silu_and_mul(R.linear_w8(x, q_ref, s_ref, None, dtype), dtype)
    diff = np.abs(y - ref)
    assert (diff <= 4 * ULP[dtype] * np.abs(ref) + 4 * ULP[dtype] * 1e-2).all()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_linear_tiled_w8_decodes_every_fp8_code_exactly(ops, dtype):
    """Weights whose quantised codes run over the whole e4m3fn table (subnormals and both zeros
    included), read back through one-hot inputs: the in-register fp8 -> 16-bit conversion is exact."""
    table = R.fp8_e4m3fn_decode_table()
    codes = np.array([c for c in range(256) if np.isfinite(table[c])])
    K, N = 512, 64
    w = np.zeros((N, K), np.float32)
    for n in range(N):
        w[n] = table[np.roll(np.resize(codes, K), 7 * n)]
        w[n, (3 * n) % K] = 448.0                                  # row scale exactly 1
    q, scale = ops.tile_weights_fp8(dev(w, dtype))
    np.testing.assert_array_equal(scale.cpu().numpy()[:N], np.ones(N, np.float32))
    for k0 in range(0, K, 32):
        x = np.zeros((32, K), np.float32)
        x[np.arange(32), k0 + np.arange(32)] = 1.0
        y = host(ops.linear_tiled_w8(dev(x, dtype), q, scale, N, tiles_per_wg=104))
        np.testing.assert_array_equal(y, w[:, k0:k0 + 32].T)


# --------------------------------------------------------------------------- #
# W8A8: fp8 activations x fp8 weights on the fp8 matrix instruction
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,K,pad", [(1, 8, 0), (5, 1000, 24), (64, 3584, 0), (128, 18944, 0), (33, 4096, 64)])
def test_quant_rows_fp8(ops, dtype, M, K, pad):
    """Activation rows -> e4m3fn codes + one scale per row: the codes decode to the oracle's values
    bit for bit, the scales are the oracle's; an all-zero row gets scale 1; strided inputs."""
    rng = rng_for("quant-rows", dtype, M, K)
    x = grid(rng, (M, K + pad), dtype, 3.0)
    if M > 2:
        x[1] = 0.0
        x[2, : K // 2] *= 1e-3                                       # subnormal codes after scaling
        x = R.round_to(x, dtype)
    xd = dev(x, dtype)[:, :K]
    xq, sc = ops.quant_rows_fp8(xd)
    q_ref, s_ref = R.quantize_rows_fp8(x[:, :K])
    np.testing.assert_array_equal(sc.cpu().numpy(), s_ref)
    got = R.fp8_e4m3fn_decode_table()[xq.cpu().numpy()]
    np.testing.assert_array_equal(got, q_ref)
    np.testing.assert_array_equal(np.signbit(got), np.signbit(q_ref))
    if M > 2:
        assert sc[1].item() == 1.0 and not xq[1].any()


@pytest.mark.parametrize("M,tiles", [(16, 208), (32, 216), (64, 208), (64, 210), (64, 216), (128, 212), (128, 216)])
def test_linear_tiled_w8a8_operand_order_is_exact_on_integers(ops, M, tiles):
    """Integer-valued fp8 operands (every product and sum exact in fp32), asymmetric, both scales 1:
    the fp32 slabs equal the integer product exactly -- pins the lane / k order of both operands of the
    fp8 instruction, the k-step pairing of the weight blocks and the row / column order of the result."""
    rng = rng_for("w8a8-int", M, tiles)
    N, K = 208, 2048
    x = rng.integers(-8, 9, (M, K)).astype(np.float32)
    w = rng.integers(-8, 9, (N, K)).astype(np.float32)
    x[:, 5] = 448.0                                                  # row scales exactly 1
    w[np.arange(N), (7 * np.arange(N)) % K] = 448.0
    xq, sx = ops.quant_rows_fp8(dev(x, "bfloat16"))
    q, sw = ops.tile_weights_fp8(dev(w, "bfloat16"))
    assert (sx.cpu().numpy() == 1).all() and (sw.cpu().numpy()[:N] == 1).all()
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    assert np.abs(ref).max() < 2 ** 24
    for groups in (1, 2):
        part = ops.linear_tiled_w8a8(xq, sx, q, sw, N, k_groups=groups, epilogue=ops.EPI_PARTIAL,
                                     tiles_per_wg=tiles).cpu().numpy()
        np.testing.assert_array_equal(part.astype(np.float64).sum(axis=0), ref)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K,tiles,groups,bias", [(64, 3584, 18944, 208, 8, False), (64, 4608, 3584, 210, 7, True),
                                                     (48, 6144, 4096, 212, 1, True), (128, 4616, 3584, 214, 7, True),
                                                     (32, 4096, 4096, 216, 4, False), (7, 1000, 1024, 208, 1, True),
                                                     (100, 1544, 4096, 216, 8, False), (1, 64, 1024, 216, 1, False)])
def test_linear_tiled_w8a8(ops, dtype, M, N, K, tiles, groups, bias):
    rng = rng_for("w8a8", dtype, M, N, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (N, K), dtype, 0.05)
    b = grid(rng, (N,), dtype) if bias else None
    q, sw = ops.tile_weights_fp8(dev(w, dtype))
    xq, sx = ops.quant_rows_fp8(dev(x, dtype))
    q_ref, sw_ref = R.quantize_rows_fp8(w)
    xq_ref, sx_ref = R.quantize_rows_fp8(x)
    ref = R.linear_w8a8(xq_ref, sx_ref, q_ref, sw_ref, b, dtype)
    ref64 = (xq_ref.astype(np.float64) @ q_ref.astype(np.float64).T) * sw_ref[None, :].astype(np.float64) * \
        sx_ref[:, None].astype(np.float64)
    if bias:
        ref64 = ref64 + b.astype(np.float64)
    tol = _linear_tol(ref64, xq_ref * sx_ref[:, None], q_ref * sw_ref[:, None], K, dtype)
    bd = dev(b, dtype) if bias else None
    # rounding flips against the oracle: it sums in float64, the device in fp32 over K products of magnitude up
    # to 448^2 -- ~1e-5 relative on the sum, a visible fraction of an fp16 ulp at K = 18944
    flips = 0.02 if dtype == "bfloat16" else 0.10
    if groups == 1:
        y = host(ops.linear_tiled_w8a8(xq, sx, q, sw, N, dtype=DT[dtype], bias=bd, tiles_per_wg=tiles))
        assert (np.abs(y - ref64) <= tol).all(), f"max err {np.abs(y - ref64).max()}"
        assert (y != ref).mean() <= flips
    part = ops.linear_tiled_w8a8(xq, sx, q, sw, N, k_groups=groups, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles).cpu().numpy()
    y = _slab_sum(part, b, dtype)
    assert (np.abs(y - ref64) <= tol).all(), f"max err {np.abs(y - ref64).max()}"
    assert (y != ref).mean() <= flips
    # what 8-bit activations cost on top of 8-bit weights, against the unquantised product
    full = x.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    assert np.abs(y - full).max() <= 0.08 * np.abs(full).max() + 0.08


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,F,K", [(64, 18944, 3584), (128, 2048, 1024), (20, 512, 1024)])
def test_linear_tiled_w8a8_swiglu(ops, dtype, M, F, K):
    """The SwiGLU epilogue on the interleaved matrix == the store epilogue + the SwiGLU kernel, bit for bit,
    for every workgroup shape; and close to the oracle's composition."""
    rng = rng_for("w8a8-silu", dtype, M, F, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (2 * F, K), dtype, 0.05)
    xq, sx = ops.quant_rows_fp8(dev(x, dtype))
    q, s = ops.tile_weights_fp8(dev(w, dtype))
    qi, si = ops.tile_weights_fp8(dev(w, dtype), interleave_halves=True)
    gu = ops.linear_tiled_w8a8(xq, sx, q, s, 2 * F, dtype=DT[dtype])
    want = host(ops.silu_and_mul(gu))
    for tiles in ([208, 216] if M <= 32 else [208, 210, 212, 214, 216] if M <= 64 else [210, 212, 214, 216]):
        y = host(ops.linear_tiled_w8a8(xq, sx, qi, si, 2 * F, dtype=DT[dtype], epilogue=ops.EPI_SILU_MUL, tiles_per_wg=tiles))
        np.testing.assert_array_equal(y, want)
    q_ref, s_ref = R.quantize_rows_fp8(w)
    xq_ref, sx_ref = R.quantize_rows_fp8(x)
    # against the oracle (float64 sums): a flipped rounding of gate and of up, each amplified by the gate's
    # slope, plus the roundings of silu and of the product
    gu_ref = R.linear_w8a8(xq_ref, sx_ref, q_ref, s_ref, None, dtype)
    ref = R.silu_and_mul(gu_ref, dtype)
    diff = np.abs(want - ref)
    # absolute part: the fp32 accumulation error of gate / up (the _linear_tol term), times the other factor
    acc_err = 1e-5 * np.sqrt(K) * np.abs(x).max() * np.abs(w).max()
    atol = acc_err * (2 * np.abs(gu_ref).max() + 1) + 6 * ULP[dtype] * 1e-2
    assert (diff <= 6 * ULP[dtype] * np.abs(ref) + atol).all(), (diff - 6 * ULP[dtype] * np.abs(ref)).max() / atol
    assert (want != ref).mean() <= (0.05 if dtype == "bfloat16" else 0.15)


def test_linear_tiled_w8a8_rejects_what_it_does_not_build(ops):
    from mi355_decode._lib import Mi355Error
    x = torch.zeros(128, 1024, dtype=torch.bfloat16, device="cuda")
    xq, sx = ops.quant_rows_fp8(x)
    q, s = ops.tile_weights_fp8(torch.zeros(64, 1024, dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(Mi355Error, match="tiles_per_wg must be"):
        ops.linear_tiled_w8a8(xq, sx, q, s, 64, tiles_per_wg=8)
    with pytest.raises(Mi355Error, match="no kernel for M 128"):
        ops.linear_tiled_w8a8(xq, sx, q, s, 64, tiles_per_wg=208)
    with pytest.raises(Mi355Error, match="partial-sum epilogue"):
        ops.linear_tiled_w8a8(xq, sx, q, s, 64, tiles_per_wg=216, k_groups=2)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,F,K", [(128, 14336, 4096), (64, 2048, 1024)])
def test_linear_tiled_w8_swiglu_fused_equals_unfused_at_more_rows(ops, dtype, M, F, K):
    """fp16 is where a folded conversion shows: the scaled accumulator and silu(g) must be rounded
    from their fp32 values in the fused epilogue exactly as in the store epilogue + SwiGLU kernel."""
    rng = rng_for("w8-silu-rows", dtype, M, F, K)
    x, w = grid(rng, (M, K), dtype), grid(rng, (2 * F, K), dtype, 0.05)
    xd = dev(x, dtype)
    q, s = ops.tile_weights_fp8(dev(w, dtype))
    qi, si = ops.tile_weights_fp8(dev(w, dtype), interleave_halves=True)
    y = host(ops.linear_tiled_w8(xd, qi, si, 2 * F, epilogue=ops.EPI_SILU_MUL))
    np.testing.assert_array_equal(y, host(ops.silu_and_mul(ops.linear_tiled_w8(xd, q, s, 2 * F))))


@pytest.mark.parametrize("ldt", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("B,V", [(1, 8), (7, 1000), (32, 128256), (3, 152064)])
def test_token_logprobs(ops, ldt, B, V):
    rng = rng_for("logprob", ldt, B, V)
    logits = (rng.standard_normal((B, V)) * 3).astype(np.float32)
    logits[0, V // 2] = 30.0                                   # a dominant token: logprob ~ 0
    if ldt != "float32":
        logits = R.round_to(logits, ldt)
    ids = rng.integers(0, V, B).astype(np.int64)
    ids[0] = V // 2
    t = torch.from_numpy(logits).to({"float32": torch.float32, **DT}[ldt]).cuda()
    lp, lse = ops.token_logprobs(t, torch.from_numpy(ids).cuda())
    lp_ref, lse_ref = R.token_logprobs(logits, ids)
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref, rtol=1e-5, atol=5e-6)
    lp2, _ = ops.token_logprobs(t, torch.from_numpy(ids).cuda())
    np.testing.assert_array_equal(lp.cpu().numpy(), lp2.cpu().numpy())      # fixed-order sums: same bits every launch
