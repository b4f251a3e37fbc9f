"""The model geometries and rank shapes of the benchmark configurations, at real width
(two layers deep): one decode step of one rank against the numpy oracle run on the same
rank's weights, KV pages and step inputs (same 16-bit rounding points), on the
weight-streaming tiled GEMM path and on library GEMMs.  This pins the kernels at the shapes
the tiny-model oracle tests cannot reach (K = 3584 / 8192 / 28672, GQA groups 7 and 8,
2304-token contexts with split-KV attention, 128 and 256 rows), and a second assertion keeps the two
GEMM paths within the same bound of each other.

configs (BASELINE.json): [1] Llama-3-8B TP1 batch 32; [2] Llama-3-8B TP4 batch 128;
[3] Llama-3-70B TP8 batch 16 with 2048+256-token contexts; [4] Qwen2-7B batch 64, in bf16
and with fp8 (e4m3fn, per-row scales) weights -- the oracle then runs on the quantised model.
A tensor-parallel rank runs with its collectives replaced by no-ops, so the step computes that
rank's share only; the oracle sees a model with that rank's heads, MLP slice and vocabulary
slice."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import model_ref as M  # noqa: E402
from oracle import ops_ref as R  # noqa: E402

CASES = [
    # name, model, tp, rank, batch, context, max_model_len, MLP and lm_head sharded over the ranks, fp8 weights
    ("cfg1", "llama-3-8b", 1, 0, 32, 200, 256, False, False),
    ("cfg2", "llama-3-8b", 4, 2, 128, 200, 256, False, False),
    ("cfg2_mlp_sharded", "llama-3-8b", 4, 2, 128, 200, 256, True, False),
    # the 8-GPU point of the scaling run (bench.py --gpus 8: batch 32 per GPU): 256 rows on a TP8 rank's slices
    ("cfg2_tp8_256rows", "llama-3-8b", 8, 3, 256, 200, 256, True, False),
    # one GPU at 136 sequences (the harness's deployment default allows 256 at a time): 129+ GEMM rows, and
    # 1088 (sequence, kv head) groups -- the decode attention's one-wave-per-group form
    ("one_gpu_136rows", "llama-3-8b", 1, 0, 136, 200, 256, False, False),
    ("cfg3", "llama-3-70b", 8, 5, 16, 2100, 2304, False, False),
    ("cfg3_mlp_sharded", "llama-3-70b", 8, 5, 16, 2100, 2304, True, False),
    ("cfg4_bf16", "qwen2-7b", 1, 0, 64, 700, 1024, False, False),
    ("cfg4_fp8", "qwen2-7b", 1, 0, 64, 700, 1024, False, True),
]

# the whole-model bars of tests/test_model_gpu.py (bf16): max <= 2^-5 S, mean <= 2^-8 S, S = max |logit|
MAX_TOL, MEAN_TOL = 2.0 ** -5, 2.0 ** -8


def _np(t):
    return t.detach().float().cpu().numpy()


def _quantised(w):
    """What an fp8 copy of w decodes to, by the oracle's e4m3fn rounding, 4096 rows at a time."""
    out = np.empty_like(w)
    for a in range(0, w.shape[0], 4096):
        q, s = R.quantize_rows_fp8(w[a:a + 4096])
        out[a:a + 4096] = q * s[:, None]
    return out


def _rank_oracle(cfg, ws, nh, nkv, token_ids, max_len, num_blocks, fp8):
    """RefModel of one rank's share: its heads, its MLP and vocabulary slices, the embedding rows
    of this step's tokens only (row b = token b)."""
    d = cfg.head_dim
    F = int(ws.layers[0].down_w.shape[1])
    fq = _quantised if fp8 else (lambda a: a)
    w = {"model.embed_tokens.weight": _np(ws.embed[torch.as_tensor(token_ids, device=ws.embed.device)]),
         "model.norm.weight": _np(ws.final_norm), "lm_head.weight": _np(ws.lm_head)}
    for i, lw in enumerate(ws.layers):
        p = f"model.layers.{i}."
        qkv = fq(_np(lw.qkv_w))
        w[p + "self_attn.q_proj.weight"] = qkv[:nh * d]
        w[p + "self_attn.k_proj.weight"] = qkv[nh * d:(nh + nkv) * d]
        w[p + "self_attn.v_proj.weight"] = qkv[(nh + nkv) * d:]
        if lw.qkv_b is not None:
            b = _np(lw.qkv_b)
            w[p + "self_attn.q_proj.bias"], w[p + "self_attn.k_proj.bias"], w[p + "self_attn.v_proj.bias"] = \
                b[:nh * d], b[nh * d:(nh + nkv) * d], b[(nh + nkv) * d:]
        w[p + "self_attn.o_proj.weight"] = fq(_np(lw.o_w))
        gu = fq(_np(lw.gate_up_w))
        w[p + "mlp.gate_proj.weight"], w[p + "mlp.up_proj.weight"] = gu[:F], gu[F:]
        w[p + "mlp.down_proj.weight"] = fq(_np(lw.down_w))
        w[p + "input_layernorm.weight"] = _np(lw.input_norm)
        w[p + "post_attention_layernorm.weight"] = _np(lw.post_norm)
    rc = M.RefConfig(vocab_size=int(ws.lm_head.shape[0]), hidden_size=cfg.hidden_size, intermediate_size=F,
                     num_hidden_layers=len(ws.layers), num_attention_heads=nh, num_key_value_heads=nkv,
                     head_dim=d, rope_theta=cfg.rope_theta, rope_scaling=cfg.rope_scaling,
                     rms_norm_eps=cfg.rms_norm_eps, max_position_embeddings=max_len,
                     attention_bias=cfg.attention_bias, tie_word_embeddings=False)
    return M.RefModel(rc, w, "bfloat16", num_blocks=num_blocks)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_rank_decode_step_matches_oracle(case):
    from mi355_decode import weights as W
    from mi355_decode.config import ModelConfig
    from mi355_decode.model_runner import ModelRunner
    _, model, tp, rank, batch, ctx, max_len, shard_mlp, fp8 = case
    cfg = ModelConfig.from_pretrained(model)
    blocks_per_seq = (max_len + 15) // 16
    num_blocks = batch * blocks_per_seq
    rng = np.random.Generator(np.random.PCG64(9))
    ids = rng.integers(0, cfg.vocab_size, batch)
    pos = np.asarray([ctx - 1 - (b % 7) for b in range(batch)], dtype=np.int64)     # ragged contexts
    tables = np.arange(num_blocks, dtype=np.int32).reshape(batch, blocks_per_seq)
    slots = tables[np.arange(batch), pos // 16].astype(np.int64) * 16 + pos % 16
    out, ref = {}, None
    for tiled in (True, False):
        ws = W.synthetic(cfg, torch.bfloat16, "cuda", seed=5, tp_rank=rank, tp_size=tp, num_layers=2, std=0.03,
                         shard_mlp=shard_mlp, shard_vocab=shard_mlp)
        if ref is None:            # before the runner: with fp8 it replaces the row-major matrices by their dequantised copies
            ref = _rank_oracle(cfg, ws, cfg.num_attention_heads // tp, cfg.num_key_value_heads // tp, ids, max_len,
                               num_blocks, fp8)
        r = ModelRunner(cfg, ws, torch.bfloat16, "cuda", num_blocks=num_blocks, max_num_seqs=batch,
                        max_model_len=max_len, tp_rank=rank, tp_size=tp, tp_group=None, tiled_linears=tiled,
                        weight_fp8=fp8)
        r._allreduce = lambda t: None        # this rank's share only
        r._allreduce_max = lambda t: None    # ... and the best token of its vocabulary slice
        assert r._use_tiled(r.bucket_for(batch)) == tiled and r.mlp_sharded == shard_mlp == r.vocab_sharded
        if fp8 and tiled:
            assert isinstance(r.tiled[0]["qkv"], tuple), "fp8 weight stream not active"
        g = torch.Generator(device="cuda").manual_seed(11)
        for kc, vc in zip(r.k_cache, r.v_cache):
            kc.normal_(0, 1.0, generator=g)
            vc.normal_(0, 1.0, generator=g)
        if tiled:                  # the oracle starts from the same pages
            for i in range(r.L):
                ref.k_cache[i][:] = _np(r.k_cache[i])
                ref.v_cache[i][:] = _np(r.v_cache[i])
        n64, n32 = r.host_inputs()
        B = r.max_num_seqs
        n64[:B], n64[B:2 * B], n64[2 * B:] = ids, pos, slots
        n32[:B] = pos + 1
        n32[B:].reshape(B, r.max_blocks)[:, :blocks_per_seq] = tables
        toks = r.run_decode(B)
        assert r.use_graphs and r.bucket_for(B) in r.graphs, "the decode step did not run as a HIP graph"
        out[tiled] = (_np(r.logits[:B]), np.asarray(toks))
        del r, ws, g, kc, vc
        gc.collect()
        torch.cuda.empty_cache()
    want = ref.forward(np.arange(batch), pos, slots, tables, (pos + 1).astype(np.int32))
    want_tok = R.argmax_lowest(want) + (rank * want.shape[1] if shard_mlp else 0)
    scale = np.abs(want).max()
    (lt, tt), (ll, tl) = out[True], out[False]
    assert lt.shape == want.shape == ll.shape
    for name, lg, tk in (("tiled", lt, tt), ("library", ll, tl)):
        assert np.isfinite(lg).all()
        err = np.abs(lg - want)
        same = tk == want_tok
        # a token may differ from the oracle's only at a near-tie of the oracle's logits
        rows = np.flatnonzero(~same)
        local = tk[rows] - (rank * want.shape[1] if shard_mlp else 0)
        gaps = want[rows, want_tok[rows] - (rank * want.shape[1] if shard_mlp else 0)] - want[rows, local]
        print(f"{case[0]} {name} vs oracle: logits scale {scale:.3f}, max diff {err.max():.2e} "
              f"({err.max() / scale:.2e} rel), mean {err.mean() / scale:.2e} rel; {int(same.sum())}/{len(tk)} "
              f"greedy tokens equal, worst margin of the others {gaps.max() if len(gaps) else 0:.2e}")
        assert err.max() <= MAX_TOL * scale and err.mean() <= MEAN_TOL * scale
        # the sampled token is exactly the lowest-index argmax of the step's own logits ...
        np.testing.assert_array_equal(tk - (rank * want.shape[1] if shard_mlp else 0), R.argmax_lowest(lg))
        # ... so it can leave the oracle's only where the oracle's margin is inside twice the logit error
        # (random weights give flat logits: near-ties are common, 2 of 16 rows were seen)
        assert (gaps >= 0).all() and (gaps <= 2 * err.max() + 1e-6).all()
    err = np.abs(lt - ll)
    assert err.max() <= MAX_TOL * scale and err.mean() <= MEAN_TOL * scale
    # the two GEMM forms may pick different tokens only where the margin between the two picks, in either
    # form's logits, is inside twice the difference between the forms
    rows = np.flatnonzero(tt != tl)
    off = rank * want.shape[1] if shard_mlp else 0
    assert (np.abs(lt[rows, tt[rows] - off] - lt[rows, tl[rows] - off]) <= 2 * err.max() + 1e-6).all()
    print(f"{case[0]}: {int((tt == tl).sum())}/{len(tt)} greedy tokens equal between the tiled and the library GEMM path")
