"""EXPERIMENT: what the library's best algorithm (PyTorch TunableOp over every hipBLASLt / rocBLAS solution) takes
for the projections of a Llama-3-8B tensor-parallel rank at 128 and 256 rows, and of one GPU at 256 rows --
the ceiling an LDS-staged decode GEMM built from library kernels could reach.  Times are graph replays over
rotating weight copies (the weights come from HBM, as in the step)."""
import json, sys, time
import torch
import torch.cuda.tunable as tn
dev = torch.device("cuda:0")
SHAPES = [  # (label, M, N, K)
    ("tp4 qkv", 128, 1536, 4096), ("tp4 o", 128, 4096, 1024), ("tp4 gate_up", 128, 7168, 4096),
    ("tp4 down", 128, 4096, 3584), ("tp4 lm_head", 128, 32064, 4096),
    ("tp8 qkv", 256, 768, 4096), ("tp8 o", 256, 4096, 512), ("tp8 gate_up", 256, 3584, 4096),
    ("tp8 down", 256, 4096, 1792), ("tp8 lm_head", 256, 16032, 4096),
    ("tp1 gate_up", 256, 28672, 4096), ("tp1 down", 256, 4096, 14336),
]
def timed(M, N, K, copies):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
    ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(copies)]
    for w, y in zip(ws, ys):
        torch.mm(x, w.t(), out=y)          # tuning (when enabled) happens here, outside capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for w, y in zip(ws, ys):
            torch.mm(x, w.t(), out=y)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * copies)
out = {}
for label, M, N, K in SHAPES:
    copies = max(2, min(16, int(600e6 // (N * K * 2))))
    out[label] = {"M": M, "N": N, "K": K, "default_us": round(timed(M, N, K, copies), 2)}
    print(label, out[label], flush=True)
tn.enable(True); tn.tuning_enable(True); tn.set_max_tuning_duration(60); tn.set_max_tuning_iterations(30)
tn.set_filename("/tmp/tunable_rank_shapes.csv")
for label, M, N, K in SHAPES:
    t0 = time.time()
    copies = max(2, min(16, int(600e6 // (N * K * 2))))
    out[label]["tuned_us"] = round(timed(M, N, K, copies), 2)
    out[label]["tuning_s"] = round(time.time() - t0, 1)
    print(label, out[label], flush=True)
res = {k: v for k, v in tn.get_results()} if False else [list(r) for r in tn.get_results()]
json.dump({"shapes": out, "picked": res}, open(sys.argv[1], "w"), indent=1)
