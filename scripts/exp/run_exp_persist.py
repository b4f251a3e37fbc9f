"""Persistent-grid GEMM phases of 32 decoder layers: barriers vs barriers + early weight ring vs no barriers."""
import ctypes
import numpy as np
import torch

lib = ctypes.CDLL("scripts/exp/exp_persist.so")
lib.exp_persist.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                            ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                            ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = "cuda"
H, F, QW = 4096, 14336, 6144
shapes = [("qkv", QW, H, 4), ("o", H, H, 4), ("gate_up", 2 * F, H, 1), ("down", H, F, 7)]
dt = np.dtype([("tiles", "i4"), ("kgroups", "i4"), ("slices", "i4"), ("ksteps", "i4"), ("w_off", "i8")])
ph = np.zeros(len(shapes), dt)
off = 0
for i, (name, N, K, kg) in enumerate(shapes):
    ph[i] = (N // 16, kg, K // 1024 // kg, K // 32, off)
    off += N * K
layer_elems, pool = off, 3
weights = torch.empty(pool * layer_elems, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
x = torch.randn(32, F, dtype=torch.bfloat16, device=dev)
out = torch.zeros(256 * 512 * 8, dtype=torch.float32, device=dev)
bar = torch.zeros(4, dtype=torch.int32, device=dev)
abort = torch.zeros(1, dtype=torch.int32, device=dev)
phd = torch.from_numpy(ph.view(np.uint8)).to(dev)
st = torch.cuda.current_stream().cuda_stream
L = 32
layer_bytes = layer_elems * 2
for mode, label in ((2, "no barriers"), (0, "barrier after every phase"), (1, "barrier, next ring issued before the wait")):
    def run():
        bar.zero_()
        assert lib.exp_persist(phd.data_ptr(), len(shapes), L, weights.data_ptr(), layer_elems, pool, x.data_ptr(), F,
                               out.data_ptr(), bar.data_ptr(), abort.data_ptr(), mode, 256, st) == 0
    run(); torch.cuda.synchronize()
    assert int(abort.item()) == 0, "a barrier timed out"
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        run()
    e.record(); e.synchronize()
    us = s.elapsed_time(e) * 1000 / 3 / L
    print(f"{label:45s}: {us:6.1f} us/layer  {layer_bytes / us / 1e3:6.0f} GB/s  (abort={int(abort.item())})", flush=True)
