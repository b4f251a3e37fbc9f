"""Do hipBLASLt's other algorithms (PyTorch TunableOp) beat its default choice on the prefill GEMM
shapes (4096 packed tokens of Llama-3-8B)?  Prints us per GEMM with the default and after tuning."""
import json, os, sys, time
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shapes = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (28672, 4096), "down": (4096, 14336)}
dev = "cuda"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / n


ws = {k: torch.empty(n, kk, dtype=torch.bfloat16, device=dev).normal_(0, 0.02) for k, (n, kk) in shapes.items()}
xs = {k: torch.randn(M, kk, dtype=torch.bfloat16, device=dev) for k, (n, kk) in shapes.items()}
outs = {k: torch.empty(M, n, dtype=torch.bfloat16, device=dev) for k, (n, kk) in shapes.items()}
res = {"M": M, "default_us": {}, "tuned_us": {}, "tflops_default": {}, "tflops_tuned": {}}
for k in shapes:
    res["default_us"][k] = round(timed(lambda: torch.mm(xs[k], ws[k].t(), out=outs[k])), 1)
import torch.cuda.tunable as tn
tn.enable(True)
tn.tuning_enable(True)
tn.set_max_tuning_duration(30)
tn.set_max_tuning_iterations(20)
tn.set_filename(os.environ.get("TUNABLE_FILE", "gpurun_out/tunableop_results.csv"))
t0 = time.time()
for k in shapes:
    torch.mm(xs[k], ws[k].t(), out=outs[k])
    torch.cuda.synchronize()
    print(k, "tuned after", round(time.time() - t0, 1), "s", flush=True)
tn.tuning_enable(False)
for k, (n, kk) in shapes.items():
    res["tuned_us"][k] = round(timed(lambda: torch.mm(xs[k], ws[k].t(), out=outs[k])), 1)
    fl = 2.0 * M * n * kk
    res["tflops_default"][k] = round(fl / res["default_us"][k] / 1e6, 1)
    res["tflops_tuned"][k] = round(fl / res["tuned_us"][k] / 1e6, 1)
res["layer_us_default"] = round(sum(res["default_us"].values()), 1)
res["layer_us_tuned"] = round(sum(res["tuned_us"].values()), 1)
print(json.dumps(res), flush=True)
