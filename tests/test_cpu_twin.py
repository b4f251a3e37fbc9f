"""The CPU twin of the operator ABI (libmi355_decode_cpu.so, include/mi355_decode_cpu.h): loads and
exports what its header declares, mirrors the GPU entry points' argument lists, matches the numpy
oracle to the bars the GPU kernels are held to (tests/test_ops_gpu.py), and runs clean under
AddressSanitizer + UBSan.  No GPU."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import PKG, ROOT
from oracle import ops_ref as R

LIB = os.path.join(PKG, "mi355_decode", "libmi355_decode_cpu.so")
HEADER = os.path.join(ROOT, "include", "mi355_decode_cpu.h")
GPU_HEADER = os.path.join(ROOT, "include", "mi355_decode.h")
DT = {"bfloat16": 0, "float16": 1}
ULP = {"bfloat16": 2.0 ** -7, "float16": 2.0 ** -10}
P, I32, I64, F = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float


def _decls(path, prefix):
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"\b(%s[a-z0-9_]+)\s*\(([^;]*?)\)\s*;" % prefix, text, flags=re.S):
        args = [re.sub(r"\s+", " ", a.strip()) for a in m.group(2).split(",")]
        out[m.group(1)] = [] if args == ["void"] else args
    return out


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), os.path.relpath(LIB, os.path.join(PKG, "csrc"))], check=True)
    lib = ctypes.CDLL(LIB)
    sigs = {
        "mi355_cpu_rms_norm": [P, P, P, I64, I32, I64, I64, F, I32],
        "mi355_cpu_fused_add_rms_norm": [P, P, P, P, I64, I32, I64, I64, I64, F, I32],
        "mi355_cpu_layer_norm": [P, P, P, P, I64, I32, I64, I64, F, I32],
        "mi355_cpu_fused_add_layer_norm": [P, P, P, P, P, I64, I32, I64, I64, I64, F, I32],
        "mi355_cpu_relu": [P, P, I64, I32],
        "mi355_cpu_rope_kv_write": [P, I64, P, P, I32, P, P, P, I32, I64, I32, I32, I32, I32],
        "mi355_cpu_paged_attn_decode": [P, P, I64, P, P, I32, P, I32, P, I32, I32, I32, I32, F, I32],
        "mi355_cpu_paged_attn_prefill": [P, I64, P, I64, P, P, I32, P, I32, P, P, I32, I32, I32, I32, I32, F, I32],
        "mi355_cpu_silu_and_mul": [P, P, I64, I64, I32, I32],
        "mi355_cpu_argmax": [P, P, I64, I32, I32, I32],
        "mi355_cpu_linear": [P, I64, P, I64, P, P, I32, I32, I32, I32],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = I32, args
    lib.mi355_cpu_last_error.restype = ctypes.c_char_p
    lib._sigs = sigs
    return lib


def test_exports_and_argument_lists_mirror_the_gpu_abi(lib):
    cpu, gpu = _decls(HEADER, "mi355_cpu_"), _decls(GPU_HEADER, "mi355_")
    assert set(cpu) == set(lib._sigs) | {"mi355_cpu_last_error"}
    for name in cpu:
        assert hasattr(lib, name), f"{name} declared but not exported"
    twin_of = {n: n.replace("mi355_cpu_", "mi355_") for n in cpu if n not in ("mi355_cpu_linear", "mi355_cpu_last_error")}
    twin_of["mi355_cpu_linear"] = "mi355_linear_decode"
    drop = {"mi355_paged_attn_decode": {"int32_t num_splits", "void* workspace", "int64_t workspace_bytes"},
            "mi355_linear_decode": {"int32_t k_slices"}}
    for c, g in twin_of.items():
        want = [a for a in gpu[g] if a != "void* stream" and a not in drop.get(g, ())]
        assert cpu[c] == want, f"{c} does not take the arguments of {g}"


def bits(x, dtype):
    a = np.ascontiguousarray(R.to_bits(x, dtype))
    assert a.ctypes.data % 16 == 0 or a.size == 0
    return a


def aligned(n, dt=np.uint16):
    raw = np.zeros(n * np.dtype(dt).itemsize + 16, np.uint8)
    off = (-raw.ctypes.data) % 16
    return raw[off:off + n * np.dtype(dt).itemsize].view(dt)


def put(x, dtype):
    b = R.to_bits(np.asarray(x, np.float32), dtype).ravel()
    a = aligned(b.size)
    a[:] = b
    return a


def ok(lib, rc):
    assert rc == 0, lib.mi355_cpu_last_error().decode()


def ulp_close(got, ref, dtype, max_frac):
    diff = np.abs(got - ref)
    assert (diff <= 2 * ULP[dtype] * np.abs(ref) * 1.01 + 2.0 ** -24).all(), diff.max()
    assert (diff > 0).mean() <= max_frac


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_conversions_are_exact(lib, dtype):
    """x -> 16 bits -> x through the library (a SwiGLU with gate = +large passes `up` through a
    rounding; simpler: linear with an identity weight) equals the oracle's rounding, on every
    finite 16-bit pattern and on random floats including ties, subnormals and overflow."""
    rng = np.random.Generator(np.random.PCG64(3))
    pats = np.arange(65536, dtype=np.uint16)
    vals = R.from_bits(pats, dtype)
    vals = vals[np.isfinite(vals)]
    extra = np.concatenate([rng.standard_normal(5000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 5000),
                            np.float32([65504, 65519.9, 65520, 70000, -65520, 2.0 ** -25, 2.0 ** -25 * 1.0001, 2.0 ** -24 * 1.5,
                                        2.0 ** -24 * 2.5, 6.1e-5, 0.0, -0.0])])
    # y = round(1 * x + 0) through mi355_cpu_linear with K = 8 (x in column 0, a one in the weight):
    # the input must already be on the grid, so test from_f via a bias instead: y = round(0 + bias)
    for chunk in (vals, R.round_to(extra, dtype)):
        n = (len(chunk) + 7) // 8 * 8
        b = np.zeros(n, np.float32)
        b[:len(chunk)] = chunk
        x, w, y, bias = put(np.zeros(8), dtype), put(np.zeros((n, 8)), dtype), aligned(n), put(b, dtype)
        ok(lib, lib.mi355_cpu_linear(y.ctypes.data, n, x.ctypes.data, 8, w.ctypes.data, bias.ctypes.data, 1, n, 8, DT[dtype]))
        np.testing.assert_array_equal(y, R.to_bits(np.float32(0.0) + b, dtype))   # to_f then from_f: the identity on the grid (0 + -0 = +0)
    # from_f on off-grid values: the rounded sum x1 + x2 of two grid values whose sum is off the grid
    a = R.round_to(extra, dtype)
    c = R.round_to(extra[::-1] * np.float32(0.37), dtype)
    n = len(a) // 8 * 8
    a, c = a[:n], c[:n]
    with np.errstate(over="ignore", invalid="ignore"):
        want = R.round_to(a + c, dtype)
    xa, res, w, out = put(a.reshape(1, n), dtype), put(c.reshape(1, n), dtype), put(np.ones(n), dtype), aligned(n)
    ok(lib, lib.mi355_cpu_fused_add_rms_norm(out.ctypes.data, xa.ctypes.data, res.ctypes.data, w.ctypes.data, 1, n, n, n, n,
                                             1e-5, DT[dtype]))
    keep = np.isfinite(want)
    np.testing.assert_array_equal(res[keep], R.to_bits(want, dtype)[keep])


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_norms_rope_silu_against_the_oracle(lib, dtype):
    rng = np.random.Generator(np.random.PCG64(5))
    g = lambda *s, sc=1.0: R.round_to(rng.standard_normal(s).astype(np.float32) * sc, dtype)   # noqa: E731
    T, H = 7, 448
    x, w, res = g(T, H, sc=2.0), R.round_to(1 + 0.1 * rng.standard_normal(H), dtype), g(T, H)
    xa, wa, ra, out = put(x, dtype), put(w, dtype), put(res, dtype), aligned(T * H)
    ok(lib, lib.mi355_cpu_rms_norm(out.ctypes.data, xa.ctypes.data, wa.ctypes.data, T, H, H, H, 1e-5, DT[dtype]))
    ulp_close(R.from_bits(out, dtype).reshape(T, H), R.rms_norm(x, w, 1e-5, dtype), dtype, 0.005)
    ok(lib, lib.mi355_cpu_fused_add_rms_norm(out.ctypes.data, xa.ctypes.data, ra.ctypes.data, wa.ctypes.data, T, H, H, H, H,
                                             1e-5, DT[dtype]))
    y_ref, z_ref = R.fused_add_rms_norm(x, res, w, 1e-5, dtype)
    np.testing.assert_array_equal(R.from_bits(ra, dtype).reshape(T, H), z_ref)            # residual add: bit-exact
    ulp_close(R.from_bits(out, dtype).reshape(T, H), y_ref, dtype, 0.005)

    nh, kvh, d, T, nblk, max_pos = 8, 2, 128, 21, 6, 300
    table = R.rope_cos_sin_table(max_pos, d, 5e5, dtype)
    qkv = g(T, (nh + 2 * kvh) * d)
    pos = rng.integers(0, max_pos, T).astype(np.int64)
    slots = rng.permutation(nblk * 16)[:T].astype(np.int64)
    slots[3] = -1
    kc_ref, vc_ref = np.zeros((nblk, kvh, 16, d), np.float32), np.zeros((nblk, kvh, d, 16), np.float32)
    ref = R.rope_kv_write(qkv, pos, table, slots, kc_ref, vc_ref, nh, kvh, d, dtype)
    qa, ta, kc, vc = put(qkv, dtype), put(table, dtype), put(kc_ref * 0, dtype), put(vc_ref * 0, dtype)
    ok(lib, lib.mi355_cpu_rope_kv_write(qa.ctypes.data, qkv.shape[1], pos.ctypes.data, ta.ctypes.data, max_pos, slots.ctypes.data,
                                        kc.ctypes.data, vc.ctypes.data, nblk, T, nh, kvh, d, DT[dtype]))
    np.testing.assert_array_equal(R.from_bits(qa, dtype).reshape(qkv.shape), ref)         # bit-exact
    np.testing.assert_array_equal(R.from_bits(kc, dtype).reshape(kc_ref.shape), kc_ref)
    np.testing.assert_array_equal(R.from_bits(vc, dtype).reshape(vc_ref.shape), vc_ref)

    T, Fi = 5, 1000
    x = g(T, 2 * Fi, sc=3.0)
    xa, out = put(x, dtype), aligned(T * Fi)
    ok(lib, lib.mi355_cpu_silu_and_mul(out.ctypes.data, xa.ctypes.data, 2 * Fi, T, Fi, DT[dtype]))
    ulp_close(R.from_bits(out, dtype).reshape(T, Fi), R.silu_and_mul(x, dtype), dtype, 0.01)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_layer_norm_and_relu_against_the_oracle(lib, dtype):
    rng = np.random.Generator(np.random.PCG64(6))
    g = lambda *s, sc=1.0: R.round_to(rng.standard_normal(s).astype(np.float32) * sc, dtype)   # noqa: E731
    T, H = 5, 768
    x, res = g(T, H, sc=2.0), g(T, H)
    x[0] = R.round_to(x[0] + 5.0, dtype)
    w, b = R.round_to(1 + 0.1 * rng.standard_normal(H), dtype), R.round_to(0.1 * rng.standard_normal(H), dtype)
    xa, wa, ba, ra, out = put(x, dtype), put(w, dtype), put(b, dtype), put(res, dtype), aligned(T * H)
    ok(lib, lib.mi355_cpu_layer_norm(out.ctypes.data, xa.ctypes.data, wa.ctypes.data, ba.ctypes.data, T, H, H, H, 1e-5,
                                     DT[dtype]))
    ulp_close(R.from_bits(out, dtype).reshape(T, H), R.layer_norm(x, w, b, 1e-5, dtype), dtype, 0.005)
    ok(lib, lib.mi355_cpu_fused_add_layer_norm(out.ctypes.data, xa.ctypes.data, ra.ctypes.data, wa.ctypes.data,
                                               ba.ctypes.data, T, H, H, H, H, 1e-5, DT[dtype]))
    y_ref, z_ref = R.fused_add_layer_norm(x, res, w, b, 1e-5, dtype)
    np.testing.assert_array_equal(R.from_bits(ra, dtype).reshape(T, H), z_ref)            # residual add: bit-exact
    ulp_close(R.from_bits(out, dtype).reshape(T, H), y_ref, dtype, 0.005)
    v = g(64, sc=3.0)
    v[:5] = [0.0, -0.0, np.inf, -np.inf, np.nan]
    va, out = put(v, dtype), aligned(64)
    ok(lib, lib.mi355_cpu_relu(out.ctypes.data, va.ctypes.data, 64, DT[dtype]))
    np.testing.assert_array_equal(out, R.to_bits(R.relu(v), dtype))                       # bit-exact, -0.0 and NaN kept
    assert lib.mi355_cpu_relu(out.ctypes.data, va.ctypes.data, 63, DT[dtype]) == -2
    assert b"multiple of 8" in lib.mi355_cpu_last_error()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_attention_against_the_oracle(lib, dtype):
    rng = np.random.Generator(np.random.PCG64(7))
    g = lambda *s: R.round_to(rng.standard_normal(s).astype(np.float32), dtype)   # noqa: E731
    nh, kvh, d = 8, 2, 64
    ctxs = [1, 15, 16, 17, 33, 100, 0]
    nblks = [(c + 15) // 16 for c in ctxs]
    total = sum(nblks) + 3
    perm = rng.permutation(total).astype(np.int32)
    MB = max(nblks) + 1
    bt = np.zeros((len(ctxs), MB), np.int32)
    kc, vc = np.zeros((total, kvh, 16, d), np.float32), np.zeros((total, kvh, d, 16), np.float32)
    used = 0
    for i, c in enumerate(ctxs):
        bt[i, :nblks[i]] = perm[used:used + nblks[i]]
        used += nblks[i]
        if c:
            sl = bt[i, np.arange(c) // 16].astype(np.int64) * 16 + np.arange(c) % 16
            R.kv_write(g(c, kvh, d), g(c, kvh, d), sl, kc, vc)
    q = g(len(ctxs), nh, d)
    scale = d ** -0.5
    ref = R.paged_attn_decode(q, kc, vc, bt, np.array(ctxs, np.int32), scale, dtype)
    qa, ka, va, out = put(q, dtype), put(kc, dtype), put(vc, dtype), aligned(q.size)
    cl = np.array(ctxs, np.int32)
    ok(lib, lib.mi355_cpu_paged_attn_decode(out.ctypes.data, qa.ctypes.data, nh * d, ka.ctypes.data, va.ctypes.data, total,
                                            bt.ctypes.data, MB, cl.ctypes.data, len(ctxs), nh, kvh, d, scale, DT[dtype]))
    got = R.from_bits(out, dtype).reshape(q.shape)
    tol = {"bfloat16": 8e-3, "float16": 1e-3}[dtype] * np.abs(ref).max() + ULP[dtype] * np.abs(ref)
    assert (np.abs(got - ref) <= tol).all()
    # prefill: two chunks, the second continuing a sequence already in the cache
    lens, cctx = [17, 5], [17, 100]
    cu = np.array([0, 17, 22], np.int32)
    qp = g(22, nh, d)
    pbt = np.stack([bt[4], bt[5]])
    ref = R.paged_attn_prefill(qp, kc, vc, pbt, cu, np.array(cctx, np.int32), scale, dtype)
    qa, out = put(qp, dtype), aligned(qp.size)
    cc = np.array(cctx, np.int32)
    ok(lib, lib.mi355_cpu_paged_attn_prefill(out.ctypes.data, nh * d, qa.ctypes.data, nh * d, ka.ctypes.data, va.ctypes.data, total,
                                             np.ascontiguousarray(pbt).ctypes.data, MB, cu.ctypes.data, cc.ctypes.data, 2, 17, nh,
                                             kvh, d, scale, DT[dtype]))
    got = R.from_bits(out, dtype).reshape(qp.shape)
    tol = {"bfloat16": 8e-3, "float16": 1e-3}[dtype] * np.abs(ref).max() + ULP[dtype] * np.abs(ref)
    assert (np.abs(got - ref) <= tol).all()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_linear_and_argmax_against_the_oracle(lib, dtype):
    rng = np.random.Generator(np.random.PCG64(9))
    M, N, K = 7, 320, 264
    x = R.round_to(rng.standard_normal((M, K)).astype(np.float32), dtype)
    w = R.round_to(rng.standard_normal((N, K)).astype(np.float32) * 0.05, dtype)
    b = R.round_to(rng.standard_normal(N).astype(np.float32), dtype)
    xa, wa, ba, y = put(x, dtype), put(w, dtype), put(b, dtype), aligned(M * N)
    ok(lib, lib.mi355_cpu_linear(y.ctypes.data, N, xa.ctypes.data, K, wa.ctypes.data, ba.ctypes.data, M, N, K, DT[dtype]))
    got = R.from_bits(y, dtype).reshape(M, N)
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T + b
    assert (np.abs(got - ref64) <= ULP[dtype] * np.abs(ref64) * 1.01 + 1e-5 * np.sqrt(K) * np.abs(x).max() * np.abs(w).max()
            + 2.0 ** -23).all()
    assert (got != R.linear(x, w, b, dtype)).mean() <= 0.02

    B, V = 5, 1003
    lg = R.round_to((rng.standard_normal((B, V)) * 4).astype(np.float32), dtype)
    lg[0, [5, V - 3]] = 64.0
    lg[B - 1, :] = 0.0
    lg[B - 1, V // 2] = -0.0
    lg[1, 700] = np.nan
    lg[1, 3] = 64.0
    stride = (V + 7) // 8 * 8
    buf = np.full((B, stride), -np.inf, np.float32)
    buf[:, :V] = lg
    la, ids = put(buf, dtype), np.zeros(B, np.int64)
    ok(lib, lib.mi355_cpu_argmax(ids.ctypes.data, la.ctypes.data, stride, B, V, DT[dtype]))
    np.testing.assert_array_equal(ids, R.argmax_lowest(lg))


def test_rejected_arguments_say_why(lib):
    a = aligned(64)
    rc = lib.mi355_cpu_rms_norm(a.ctypes.data, a.ctypes.data, a.ctypes.data, 1, 60, 64, 64, 1e-5, 0)
    assert rc == -2 and b"multiple of 8" in lib.mi355_cpu_last_error()
    rc = lib.mi355_cpu_paged_attn_decode(a.ctypes.data, a.ctypes.data, 4096, a.ctypes.data, a.ctypes.data, 8, a.ctypes.data, 4,
                                         a.ctypes.data, 2, 34, 2, 128, 0.1, 0)
    assert rc == -2 and b"16 query heads" in lib.mi355_cpu_last_error()
    rc = lib.mi355_cpu_linear(None, 8, a.ctypes.data, 8, a.ctypes.data, None, 1, 8, 8, 0)
    assert rc == -1 and b"null pointer" in lib.mi355_cpu_last_error()


def test_runs_clean_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The twin's source + a driver that calls every entry point on exactly-sized heap buffers,
    compiled with -fsanitize=address,undefined: any access past a buffer end or undefined
    arithmetic aborts the driver."""
    exe = tmp_path / "cpu_twin_asan"
    src = [os.path.join(ROOT, "tests", "cpu_twin", "driver.cpp"), os.path.join(PKG, "csrc", "cpu", "mi355_cpu.cpp")]
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fopenmp", "-mavx2", "-mfma", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"), *src, "-o",
                    str(exe)], check=True)
    env = dict(os.environ, OMP_NUM_THREADS="2", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    env.pop("LD_PRELOAD", None)
    p = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "sanitizer driver clean" in p.stdout
