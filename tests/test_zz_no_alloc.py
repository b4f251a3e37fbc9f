"""The decode-attention entry point that allocates nothing (mi355_paged_attn_decode_static: the
split-KV partials live in a device array built into the library) against the entry point it is a
variant of (mi355_paged_attn_decode with a caller workspace): same kernels, so the outputs must be
bit-identical -- and both are held to the oracle's attention bar.  Two calls in a row on the shared
array, the second with a smaller input; sizes beyond the array are refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import ops_ref as R  # noqa: E402
from test_ops_gpu import attn_close, build_paged, dev, grid, host, rng_for  # noqa: E402


def _case(ops, tag, ctxs, nh, kvh, d, splits, dtype="bfloat16"):
    rng = rng_for("static", tag)
    kc, vc, bt = build_paged(rng, ctxs, kvh, d, dtype)
    q = grid(rng, (len(ctxs), nh, d), dtype)
    scale = d ** -0.5
    ref = R.paged_attn_decode(q, np.nan_to_num(kc), np.nan_to_num(vc), bt, np.array(ctxs, np.int32), scale, dtype)
    qd, kd, vd = dev(q.reshape(len(ctxs), -1), dtype), dev(kc, dtype), dev(vc, dtype)
    btd, cd = torch.from_numpy(bt).cuda(), torch.tensor(ctxs, dtype=torch.int32, device="cuda")
    want = ops.paged_attn_decode(qd, kd, vd, btd, cd, nh, kvh, d, scale, num_splits=splits)
    out = torch.full_like(want, float("nan"))
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    ops.paged_attn_decode_static(qd, kd, vd, btd, cd, nh, kvh, d, scale, splits, out)
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before, "the wrapper allocated a tensor"
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host(out), host(want))               # same kernels: bit-identical
    attn_close(host(out).reshape(len(ctxs), nh, d), ref, dtype)        # and the oracle's bar


def test_static_scratch_entry_point_matches_the_workspace_one():
    from mi355_decode import ops
    # first call: 16 sequences, GQA 8, contexts to 2300 tokens, 9 splits (the 70B TP8 rank shape)
    _case(ops, "big", [2300, 2049, 17, 1, 0, 640, 129, 2304, 33, 1000, 16, 255, 1999, 64, 777, 2176], 8, 1, 128, 9)
    # second call, smaller in every dimension, on the array the first call left behind
    _case(ops, "small", [40, 7, 300], 4, 2, 64, 3)
    # one split: no partials at all
    _case(ops, "one", [129, 5], 8, 2, 128, 1)
    # fp16, larger again
    _case(ops, "f16", [500, 1, 31, 96, 2048], 32, 8, 128, 5, dtype="float16")


def test_static_scratch_entry_point_refuses_what_does_not_fit():
    from mi355_decode import ops
    from mi355_decode._lib import Mi355Error
    B, nh, kvh, d, splits = 128, 32, 8, 128, 16       # 128 * 32 * 16 * 130 * 4 B = 34 MB > 16 MiB
    assert ops.decode_workspace_bytes(B, nh, d, splits) > ops.STATIC_SCRATCH_BYTES
    q = torch.zeros(B, nh * d, dtype=torch.bfloat16, device="cuda")
    kc = torch.zeros(4, kvh, 16, d, dtype=torch.bfloat16, device="cuda")
    vc = torch.zeros(4, kvh, d, 16, dtype=torch.bfloat16, device="cuda")
    bt = torch.zeros(B, 4, dtype=torch.int32, device="cuda")
    ctx = torch.zeros(B, dtype=torch.int32, device="cuda")
    out = torch.empty_like(q)
    with pytest.raises(Mi355Error, match="built-in scratch holds"):
        ops.paged_attn_decode_static(q, kc, vc, bt, ctx, nh, kvh, d, d ** -0.5, splits, out)
