"""EXPERIMENT: where the time of the fused decode-attention kernel goes, from per-workgroup time stamps
(library built with -DMI355_ATTN_TRACE into scripts/exp/libmi355_decode_trace.so; run with
MI355_DECODE_LIB pointing at it).  The kernel runs as in a decode step: after the K-split qkv GEMM
whose slabs it sums, 32 layers deep, replayed from a HIP graph; the stamps of the last layer's launch
are read back."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops  # noqa: E402
from mi355_decode._lib import lib  # noqa: E402

dev, dt = "cuda", torch.bfloat16
B, nh, nkv, d, L = 32, 32, 8, 128, 32
H = 4096
ctx_len = int(sys.argv[1]) if len(sys.argv) > 1 else 158
width = (nh + 2 * nkv) * d
bps = 16
nb = B * bps
kc = [torch.randn(nb, nkv, 16, d, dtype=dt, device=dev) for _ in range(L)]
vc = [torch.randn(nb, nkv, d, 16, dtype=dt, device=dev) for _ in range(L)]
bt = torch.arange(nb, dtype=torch.int32, device=dev).view(B, bps)
ctx = torch.full((B,), ctx_len, dtype=torch.int32, device=dev)
pos = (ctx - 1).to(torch.int64)
slots = (bt[torch.arange(B), (pos // 16)].to(torch.int64) * 16 + pos % 16)
cs = torch.randn(4096, d, dtype=dt, device=dev)
out = torch.empty(B, nh * d, dtype=dt, device=dev)
x = torch.randn(B, H, dtype=dt, device=dev)
wq = [ops.tile_weights(torch.randn(width, H, dtype=dt, device=dev) * 0.02) for _ in range(L)]
part = torch.empty(2 * B * width, dtype=torch.float32, device=dev)


def layer(i):
    ops.linear_tiled(x, wq[i], width, partial=part, tiles_per_wg=104, k_groups=2, epilogue=ops.EPI_PARTIAL)
    ops.paged_attn_decode_fused(None, cs, slots, kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088, out=out, partial=part,
                                num_slabs=2)


for i in range(L):
    layer(i)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    for i in range(L):
        layer(i)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    g.replay()
e.record()
e.synchronize()
print(f"ctx {ctx_len}: qkv GEMM + attention pair {s.elapsed_time(e) * 1000 / (10 * L):.2f} us per layer (trace build)")
n = B * nkv
buf = (ctypes.c_ulonglong * (n * 16))()
lib.mi355_attn_trace_read.restype = ctypes.c_int32
assert lib.mi355_attn_trace_read(buf, n * 16) == 0
t = np.array(buf, dtype=np.uint64).reshape(n, 8, 2).astype(np.int64)
wall, clk = t[:, :7, 0], t[:, :7, 1]
w7 = t[:, 7, 0]
first = wall[:, 0].min()
names = ["entry", "ctx known", "prologue done (q rotated)", "first K/V stages arrived", "loop done", "merge barrier", "end"]
print("wall clock (100 MHz): per phase, median / max over the 256 workgroups, us since the first workgroup's entry")
for k, nm in enumerate(names):
    v = (wall[:, k] - first) / 100.0
    print(f"  {nm:32s} median {np.median(v):6.2f}  min {v.min():6.2f}  max {v.max():6.2f}")
v = (w7 - first) / 100.0
print(f"  {'(thread 0: slab data arrived)':32s} median {np.median(v):6.2f}  min {v.min():6.2f}  max {v.max():6.2f}")
dur = (wall[:, 6] - wall[:, 0]) / 100.0
print(f"workgroup lifetime: median {np.median(dur):.2f} us, max {dur.max():.2f}; launch span (first entry -> last end) "
      f"{(wall[:, 6].max() - first) / 100.0:.2f} us")
cyc = (clk[:, 1:7] - clk[:, 0:6])
print("shader-clock ticks per phase (median):", [int(np.median(cyc[:, k])) for k in range(6)])
