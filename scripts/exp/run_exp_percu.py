"""Per-CU streaming rate vs number of active workgroups (contiguous 1-KB-per-instruction loads)."""
import ctypes, torch
lib = ctypes.CDLL("scripts/exp/exp_stream.so")
lib.exp_stream.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
dev = "cuda"
K = 4096
big = torch.empty(600 * 1024 * 1024, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)   # 1.2 GB pool
x = torch.randn(32, K, dtype=torch.bfloat16, device=dev)
out = torch.zeros(1 << 22, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for v, u in [(5, 8), (1, 4)]:
    for wgs in [32, 64, 128, 192, 256, 384, 512, 1024, 2048]:
        N = wgs * 4 * 16
        per = N * K                      # elements per launch
        copies = max(1, min(16, big.numel() // per))
        def run():
            for c in range(copies):
                w = big[c * per:(c + 1) * per]
                assert lib.exp_stream(v, u, out.data_ptr(), x.data_ptr(), w.data_ptr(), N, K, K, st) == 0
        run(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): run()
        e.record(); e.synchronize()
        us = s.elapsed_time(e) * 1000 / (3 * copies)
        gbs = per * 2 / us / 1e3
        print(f"v{v}u{u} wgs={wgs:5d} bytes={per*2/1e6:7.1f}MB {us:7.1f}us {gbs:7.0f}GB/s  per-WG {gbs/wgs:6.1f}GB/s", flush=True)
