import ctypes, torch
lib = ctypes.CDLL("scripts/exp/exp_cvt.so")
lib.exp_cvt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(256 * 512, dtype=torch.int32, device="cuda")
inp = torch.randint(0, 2**31 - 1, (1024,), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
iters = 100000
for mode, label in ((0, "v_cvt_pk_f32_fp8 + v_cvt_pk_bf16_f32"), (1, "integer bit placement")):
    lib.exp_cvt(out.data_ptr(), inp.data_ptr(), 1000, mode, st); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); lib.exp_cvt(out.data_ptr(), inp.data_ptr(), iters, mode, st); e.record(); e.synchronize()
    ns = s.elapsed_time(e) * 1e6 / iters
    print(f"{label:40s}: {ns:6.1f} ns per 8-element operand per wave-slot (2 waves per SIMD) = {ns * 2.4 / 2:6.1f} cycles per wave")
