import ctypes
import torch
lib = ctypes.CDLL("scripts/exp/exp_barrier.so")
lib.exp_barrier.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                            ctypes.c_int, ctypes.c_void_p]
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
for wgs in (256, 512):
    for mode in (0, 1):
        bar = torch.zeros(1024, dtype=torch.int32, device=dev)
        abort = torch.zeros(1, dtype=torch.int32, device=dev)
        hist = torch.zeros(16, dtype=torch.int32, device=dev)
        cyc = torch.zeros(1, dtype=torch.int64, device=dev)
        iters = 2000
        assert lib.exp_barrier(bar.data_ptr(), iters, mode, abort.data_ptr(), hist.data_ptr(), cyc.data_ptr(), wgs, st) == 0
        torch.cuda.synchronize()
        print(f"wgs={wgs} mode={mode}: {cyc.item() / 100.0 / iters:6.2f} us per barrier  abort={abort.item()}  "
              f"workgroups per XCD {hist.tolist()[:8]}", flush=True)
