"""The model geometries and rank shapes of the benchmark configurations, at real width
(two layers deep): one decode step of one rank on the weight-streaming tiled GEMM path
against the same step on library GEMMs.  The two paths share the attention, rotary and
sampling kernels and differ in every linear layer, the SwiGLU gate and the norm inputs, so
agreement to the logit bound pins the tiled path at the shapes the tiny-model oracle tests
cannot reach (K = 3584 / 8192 / 28672, GQA groups 7 and 8, 2304-token contexts, 128 rows).

configs (BASELINE.json): [1] Llama-3-8B TP1 batch 32; [2] Llama-3-8B TP4 batch 128;
[3] Llama-3-70B TP8 batch 16 with 2048+256-token contexts; [4] Qwen2-7B batch 64 (served in
bf16: fp8 weights are not built)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

CASES = [
    # name, model, tp, rank, batch, context, max_model_len, MLP and lm_head sharded over the ranks
    ("cfg1", "llama-3-8b", 1, 0, 32, 200, 256, False),
    ("cfg2", "llama-3-8b", 4, 2, 128, 200, 256, False),
    ("cfg2_mlp_sharded", "llama-3-8b", 4, 2, 128, 200, 256, True),
    ("cfg3", "llama-3-70b", 8, 5, 16, 2100, 2304, False),
    ("cfg3_mlp_sharded", "llama-3-70b", 8, 5, 16, 2100, 2304, True),
    ("cfg4", "qwen2-7b", 1, 0, 64, 700, 1024, False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_rank_decode_step_tiled_vs_library(case, monkeypatch):
    from mi355_decode import weights as W
    from mi355_decode.config import ModelConfig
    from mi355_decode.model_runner import ModelRunner
    _, model, tp, rank, batch, ctx, max_len, shard_mlp = case
    cfg = ModelConfig.from_pretrained(model)
    blocks_per_seq = (max_len + 15) // 16
    out = {}
    for tiled in (True, False):
        ws = W.synthetic(cfg, torch.bfloat16, "cuda", seed=5, tp_rank=rank, tp_size=tp, num_layers=2, std=0.03,
                         shard_mlp=shard_mlp, shard_vocab=shard_mlp)
        r = ModelRunner(cfg, ws, torch.bfloat16, "cuda", num_blocks=batch * blocks_per_seq, max_num_seqs=batch,
                        max_model_len=max_len, tp_rank=rank, tp_size=tp, tp_group=None, tiled_linears=tiled)
        monkeypatch.setattr(r, "_allreduce", lambda t: None)      # this rank's share only
        monkeypatch.setattr(r, "_allreduce_max", lambda t: None)  # ... and the best token of its vocabulary slice
        assert r._use_tiled(r.bucket_for(batch)) == tiled and r.mlp_sharded == shard_mlp == r.vocab_sharded
        g = torch.Generator(device="cuda").manual_seed(11)
        for kc, vc in zip(r.k_cache, r.v_cache):
            kc.normal_(0, 1.0, generator=g)
            vc.normal_(0, 1.0, generator=g)
        n64, n32 = r.host_inputs()
        B = r.max_num_seqs
        rng = np.random.Generator(np.random.PCG64(9))
        n64[:B] = rng.integers(0, cfg.vocab_size, B)
        bt = n32[B:].reshape(B, r.max_blocks)
        for b in range(B):
            p = ctx - 1 - (b % 7)                                   # ragged contexts
            bt[b, :blocks_per_seq] = np.arange(b * blocks_per_seq, (b + 1) * blocks_per_seq)
            n64[B + b], n64[2 * B + b], n32[b] = p, bt[b, p // 16] * 16 + p % 16, p + 1
        toks = r.run_decode(B)
        out[tiled] = (r.logits[:B].float().cpu().numpy(), np.asarray(toks))
        del r, ws
        torch.cuda.empty_cache()
    (lt, tt), (ll, tl) = out[True], out[False]
    if shard_mlp:                                              # tokens of this rank's slice carry their global ids
        V = cfg.vocab_size // tp
        assert lt.shape[1] == V and (tt >= rank * V).all() and (tt < (rank + 1) * V).all()
    assert np.isfinite(lt).all() and np.isfinite(ll).all()
    scale = np.abs(ll).max()
    err = np.abs(lt - ll)
    print(f"{case[0]}: logits scale {scale:.3f}, max diff {err.max():.2e} ({err.max() / scale:.2e} rel), "
          f"mean {err.mean():.2e}; {int((tt == tl).sum())}/{len(tt)} greedy tokens equal")
    assert err.max() <= 2.0 ** -5 * scale and err.mean() <= 2.0 ** -8 * scale
    assert (tt == tl).mean() >= 0.9
