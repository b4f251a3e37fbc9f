"""Does a norm launch that also pulls the head of the next GEMM's weight stream towards the
memory-side cache shorten the (norm; GEMM) pair?  The slab-summing add + norm kernel keeps HBM
nearly idle for ~3.5 us; extra workgroups of the same launch read the first `mb` MB the GEMM
will ask for (one stretch per K group).  Pairs on distinct weight copies inside one HIP graph.
Needs a build of the library whose rms_norm_slabs_kernel takes (num_rows, base, region stride,
regions, vectors per region), lets the workgroups beyond num_rows xor-read those vectors, and
exports mi355_exp_norm_prefetch(base, region_stride_bytes, regions, bytes_per_region, wgs) to arm
the next norm launch; that patch is not in the tree (result: slower in every setting, see
profiles/r02_exp_norm_prefetch.jsonl).  MI355_DECODE_LIB=<that build> python3 scripts/exp/run_norm_prefetch.py"""
import ctypes
import json
import os
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

lib = ctypes.CDLL(os.environ["MI355_DECODE_LIB"])
lib.mi355_exp_norm_prefetch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32]
lib.mi355_exp_norm_prefetch.restype = None
dev = "cuda"
M, H = 32, 4096


def timed(body, n):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        body()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            g.replay()
        e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) * 1000 / (5 * n))
    return best


resid = torch.randn(M, H, dtype=torch.bfloat16, device=dev)
nw = torch.ones(H, dtype=torch.bfloat16, device=dev)
slabs = torch.randn(8, M, H, dtype=torch.float32, device=dev)
xn = torch.empty(M, H, dtype=torch.bfloat16, device=dev)

for name, N, K, tiles, groups, nslab in (("qkv", 6144, 4096, 8, 4, 4), ("gate_up", 28672, 4096, 8, 1, 8)):
    copies = max(4, int(1200e6 // (N * K * 2)) + 1)
    wts = [ops.tile_weights(torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)) for _ in range(copies)]
    part = torch.empty(groups, M, N, dtype=torch.float32, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    wbytes = N * K * 2

    def gemm(w):
        if groups > 1:
            ops.linear_tiled(xn, w, N, partial=part, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_PARTIAL)
        else:
            ops.linear_tiled(xn, w, N, out=out, tiles_per_wg=tiles)

    def pairs(mb, wgs):
        def body():
            for w in wts:
                if mb:
                    per = (int(mb * 2 ** 20) // groups) // 4096 * 4096
                    lib.mi355_exp_norm_prefetch(w.data_ptr(), wbytes // groups, groups, per, wgs)
                ops.fused_add_rms_norm_partial(slabs, nslab, resid, nw, 1e-5, out=xn)
                gemm(w)
        return body

    rec = {"gemm": name, "N": N, "K": K, "pairs_per_graph": copies, "serial_us": round(timed(pairs(0, 0), copies), 2)}
    for wgs in (96, 224):
        for mb in (2, 4, 8, 16, 32):
            rec[f"prefetch_{mb}MB_{wgs}wg_us"] = round(timed(pairs(mb, wgs), copies), 2)
    rec["serial_again_us"] = round(timed(pairs(0, 0), copies), 2)
    print(json.dumps(rec), flush=True)
