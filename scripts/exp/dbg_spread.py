import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from mi355_decode import ops
from oracle import ops_ref as R
from test_ops_gpu import rng_for
ldt, B, V = "float32", 32, 128256
rng = rng_for("sample", ldt, B, V)
lg = R.round_to((rng.standard_normal((B, V)) * 3).astype(np.float32), ldt)
lg[0, :] = 0.0; lg[0, 5] = -0.0
lg[1, : V // 2] = lg[1, 0]; lg[2, 7] = np.nan
temperature = rng.choice([0.0, 0.3, 0.7, 1.0, 1.5], B).astype(np.float32)
temperature[:3] = [1.0, 0.8, 1.2]
top_k = rng.choice([-1, 1, 2, 5, 40, 1024, 5000], B).astype(np.int32)
top_p = rng.choice([1.0, 0.95, 0.5, 0.1], B).astype(np.float32)
seeds = rng.integers(0, 2 ** 62, B).astype(np.int64)
lgd = torch.from_numpy(lg).cuda()
for step in (0, 1, 17):
    steps = np.full(B, step, np.int64) + np.arange(B)
    ref = R.sample_topk_topp(lg, temperature, top_k, top_p, seeds, steps)
    out = ops.sample(lgd, torch.from_numpy(temperature).cuda(), torch.from_numpy(top_k).cuda(),
                     torch.from_numpy(top_p).cuda(), torch.from_numpy(seeds).cuda(), torch.from_numpy(steps).cuda()).cpu().numpy()
    bad = np.nonzero(out != ref)[0]
    print("step", step, "bad rows", bad, [(temperature[b], top_k[b], top_p[b], out[b], ref[b]) for b in bad])
    for b in bad:
        m = R.sample_kept_mask(lg[b], temperature[b], top_k[b], top_p[b]) if hasattr(R, "sample_kept_mask") else None
        if m is not None:
            print(" kept size", m.sum(), "got in kept", m[out[b]], "lg got", lg[b, out[b]], "lg ref", lg[b, ref[b]], "min kept", lg[b][m].min())
print("repeat")
from collections import Counter
c = Counter()
for rep in range(30):
    for step in (0, 17):
        steps = np.full(B, step, np.int64) + np.arange(B)
        ref = R.sample_topk_topp(lg, temperature, top_k, top_p, seeds, steps) if rep == 0 else refs[step]
        if rep == 0:
            refs = globals().setdefault("refs", {}); refs[step] = ref
        out = ops.sample(lgd, torch.from_numpy(temperature).cuda(), torch.from_numpy(top_k).cuda(),
                         torch.from_numpy(top_p).cuda(), torch.from_numpy(seeds).cuda(), torch.from_numpy(steps).cuda()).cpu().numpy()
        for b in np.nonzero(out != ref)[0]:
            c[(step, int(b), int(out[b]))] += 1
print(c)
print("exact rows:", [(b, top_k[b], top_p[b], temperature[b]) for b in range(B) if top_k[b] > 1024])
