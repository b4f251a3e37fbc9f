"""EXPERIMENT: a launch that only reads (scripts/exp/exp_prefetch.hip) over ONE buffer again and again -- served
by the memory-side cache once it fits -- against the same launch over rotating buffers (from HBM): does data
that a previous kernel left in that cache arrive faster than data from HBM?
Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o scripts/exp/exp_prefetch.so scripts/exp/exp_prefetch.hip"""
import ctypes
import json
import torch

lib = ctypes.CDLL("scripts/exp/exp_prefetch.so")
lib.exp_prefetch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
pool = torch.empty(3 * 2 ** 30, dtype=torch.uint8, device="cuda").random_(0, 255)


def timed(nbytes, blocks, hot):
    n = 24
    ptrs = [pool.data_ptr() + (0 if hot else i * nbytes) for i in range(n)]

    def body():
        for p in ptrs:
            lib.exp_prefetch(p, nbytes, blocks, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        body()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) * 1000 / n)
    return best


for mb in (8.4, 16.8, 33.5, 67.1, 117.4, 134.2):
    nbytes = int(mb * 1e6) // 65536 * 65536
    rec = {"MB": mb}
    for blocks in (256, 512):
        rec[f"cold_us_{blocks}wg"] = round(timed(nbytes, blocks, False), 2)
        rec[f"hot_us_{blocks}wg"] = round(timed(nbytes, blocks, True), 2)
    print(json.dumps(rec), flush=True)
