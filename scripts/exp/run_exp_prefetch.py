"""Times [prefetch(W_a); gemm(W_b)] pairs in a HIP graph with a == b (the GEMM finds its weights
in the memory-side cache) and a != b (same work, no reuse), plus the GEMM and the prefetch alone."""
import ctypes
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

lib = ctypes.CDLL("scripts/exp/exp_prefetch.so")
lib.exp_prefetch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda"
M = 32
sink = torch.zeros(4, dtype=torch.int32, device=dev)


def timed(body, n):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1000 / (5 * n)


for name, N, K, tiles, groups in (("o", 4096, 4096, 104, 4), ("qkv", 6144, 4096, 104, 2), ("down", 4096, 14336, 104, 4),
                                   ("gate_up", 28672, 4096, 8, 1)):
    copies = max(4, int(1200e6 // (N * K * 2)) + 1)
    wts = [ops.tile_weights(torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)) for _ in range(copies)]
    x = torch.randn(M, K, dtype=torch.bfloat16, device=dev)
    part = torch.empty(groups, M, N, dtype=torch.float32, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731

    def gemm(w):
        if groups > 1:
            ops.linear_tiled(x, w, N, partial=part, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_PARTIAL)
        else:
            ops.linear_tiled(x, w, N, out=out, tiles_per_wg=tiles)

    def pre(w, blocks, frac=1.0):
        lib.exp_prefetch(w.data_ptr(), int(w.numel() * 2 * frac) // 16 * 16, blocks, sink.data_ptr(), st())

    t_g = timed(lambda: [gemm(w) for w in wts], copies)
    line = [f"gemm alone {t_g:6.1f}us"]
    for blocks in (256, 1024):
        t_p = timed(lambda: [pre(w, blocks) for w in wts], copies)
        t_same = timed(lambda: [(pre(w, blocks), gemm(w)) for w in wts], copies)
        t_diff = timed(lambda: [(pre(wts[(i + copies // 2) % copies], blocks), gemm(w)) for i, w in enumerate(wts)], copies)
        line.append(f"[{blocks} WGs] prefetch alone {t_p:6.1f}us  pre+gemm same {t_same:6.1f}us  other {t_diff:6.1f}us")
    print(f"{name:8s} {N}x{K} ({N*K*2/1e6:.0f} MB): " + "  ".join(line), flush=True)
