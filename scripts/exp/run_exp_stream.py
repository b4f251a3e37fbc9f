import ctypes, sys, torch
lib = ctypes.CDLL("scripts/exp/exp_stream.so")
lib.exp_stream.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
dev = "cuda"
for name, N, K in [("gate_up", 28672, 4096), ("lm_head", 128256, 4096), ("down", 4096, 14336)]:
    copies = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02) for _ in range(copies)]
    xb = torch.randn(32, K + 256, dtype=torch.bfloat16, device=dev)
    out = torch.zeros(N * 4 + 1024, dtype=torch.float32, device=dev)
    line = []
    for v, u, xs in [(1, 4, K), (17, 4, K), (18, 4, K), (19, 4, K)]:
        st = torch.cuda.current_stream().cuda_stream
        def run():
            for w in ws:
                assert lib.exp_stream(v, u, out.data_ptr(), xb.data_ptr(), w.data_ptr(), N, K, xs, st) == 0
        run(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): run()
        e.record(); e.synchronize()
        us = s.elapsed_time(e) * 1000 / (5 * copies)
        line.append(f"v{v}u{u}xs+{xs-K}:{us:6.1f}us/{N*K*2/us/1e3:5.0f}GB/s")
    print(name, " ".join(line), flush=True)
