"""Random shapes through every candidate form of the decode GEMM at 33..256 rows (the row-block forms among them):
each result against torch.mm; with the bounds-checking build (MI355_DECODE_LIB=.../libmi355_decode_bounds.so) every
global access is checked against the allocations as well."""
import random, sys, torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops, bounds
from mi355_decode._lib import Mi355Error

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
torch.manual_seed(0)
checked = bounds.enabled()
ran = bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    M = random.choice([33, 48, 64, 65, 96, 100, 128, 129, 160, 192, 200, 255, 256])
    N = random.choice([8, 24, 320, 336, 1000, 1536, 2048, 4096, 4104]) if it % 3 else 16 * random.randint(1, 300)
    K = 128 * random.choice([4, 8, 14, 16, 28, 32])
    swiglu = N % 32 == 0 and random.random() < 0.3
    x = torch.randn(M, K, dtype=torch.bfloat16, device="cuda")
    w = torch.empty(N, K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.05)
    wt = ops.tile_weights(w, interleave_halves=swiglu)
    ref = torch.mm(x.float(), w.float().t())
    if swiglu:
        g, u = ref[:, : N // 2].bfloat16().float(), ref[:, N // 2:].bfloat16().float()
        ref = (torch.nn.functional.silu(g).bfloat16().float() * u)
    tol = 0.02 * ref.abs().max().item() + 0.05
    for tiles, groups in ops.linear_tiled_candidates(N, K, True, M, swiglu=swiglu):
        if swiglu and groups > 1:
            continue
        out = torch.empty(M, N // 2 if swiglu else N, dtype=torch.bfloat16, device="cuda")
        part = torch.empty(groups, M, N, dtype=torch.float32, device="cuda") if groups > 1 else None
        if checked:
            bounds.sync_allocations()
        try:
            if groups > 1:
                ops.linear_tiled(x, wt, N, partial=part, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_PARTIAL)
                got = part.sum(0)
            else:
                ops.linear_tiled(x, wt, N, out=out, tiles_per_wg=tiles, epilogue=ops.EPI_SILU_MUL if swiglu else ops.EPI_STORE)
                got = out.float()
        except Mi355Error:
            continue
        torch.cuda.synchronize()
        ran += 1
        rep = bounds.report() if checked else None
        err = (got - ref).abs().max().item()
        if (rep is not None and rep.get("refused", 0)) or not err <= tol:   # (a base outside the synced ranges is skipped, not refused)
            bad += 1
            print("BAD", dict(M=M, N=N, K=K, tiles=tiles, groups=groups, swiglu=swiglu, err=err, tol=tol, report=rep), flush=True)
print(f"{ran} launches, {bad} bad, bounds checks {'on' if checked else 'off'}")
