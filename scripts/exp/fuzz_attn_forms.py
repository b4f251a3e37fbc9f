"""Random shapes through the decode attention (plain entry point) under whichever form MI355_ATTN_FORM forces:
each result against an fp32 softmax over the gathered pages; with the bounds-checking build
(MI355_DECODE_LIB=.../libmi355_decode_bounds.so) every global access is checked against the allocations as well."""
import random, sys, torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops, bounds

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
torch.manual_seed(0)
checked = bounds.enabled()
ran = bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    d = random.choice([64, 128])
    nkv = random.choice([1, 2, 4, 8])
    group = random.choice([1, 2, 4, 5] if d == 128 else [1, 2, 4, 7, 13])
    nh = nkv * group
    B = random.choice([1, 3, 32, 130, 257, 300]) if it % 2 else random.randint(1, 320)
    splits = random.choice([1, 1, 2, 4])
    max_ctx = random.choice([40, 200, 600])
    bps = (max_ctx + 15) // 16
    ctxs = [random.choice([0, 1, 15, 16, 17, 31, 32, 33, random.randint(1, max_ctx), max_ctx]) for _ in range(B)]
    nb = B * bps + 3
    dt = random.choice([torch.bfloat16, torch.float16])
    kc = torch.randn(nb, nkv, 16, d, dtype=dt, device="cuda")
    vc = torch.randn(nb, nkv, d, 16, dtype=dt, device="cuda")
    perm = torch.randperm(nb, device="cuda")[: B * bps].to(torch.int32).view(B, bps)
    ctx = torch.tensor(ctxs, dtype=torch.int32, device="cuda")
    q = torch.randn(B, nh * d, dtype=dt, device="cuda")
    scale = d ** -0.5
    if checked:
        bounds.sync_allocations()
    out = ops.paged_attn_decode(q, kc, vc, perm, ctx, nh, nkv, d, scale, num_splits=splits)
    torch.cuda.synchronize()
    if checked:
        rep = bounds.report()
        if rep is not None and rep.get("refused", 0) > 0:
            print("BOUNDS", it, B, nh, nkv, d, splits, rep, flush=True)
            bad += 1
    # reference: gather the pages, fp32 softmax
    K = kc[perm.long()].permute(0, 2, 1, 3, 4).reshape(B, nkv, bps * 16, d).float()          # [B, kv, T, d]
    V = vc[perm.long()].permute(0, 2, 1, 4, 3).reshape(B, nkv, bps * 16, d).float()
    qq = q.view(B, nkv, group, d).float()
    s = torch.einsum("bkgd,bktd->bkgt", qq, K) * scale
    t = torch.arange(bps * 16, device="cuda")[None, None, None, :]
    s = s.masked_fill(t >= ctx.view(B, 1, 1, 1), float("-inf"))
    p = torch.softmax(s, dim=-1).nan_to_num(0.0)                                             # context 0: all masked -> zeros
    ref = torch.einsum("bkgt,bktd->bkgd", p.to(dt).float(), V).reshape(B, nh * d)
    err = (out.float() - ref).abs().max().item()
    ran += 1
    if not (err <= 0.03):
        print("MISMATCH", it, dict(B=B, nh=nh, nkv=nkv, d=d, splits=splits, max_ctx=max_ctx, dtype=str(dt)), err, flush=True)
        bad += 1
print(f"ran {ran} shapes, bad {bad}, bounds-checked {checked}")
sys.exit(1 if bad else 0)
