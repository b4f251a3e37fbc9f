"""The bounds-checking build of the kernel library (every global-memory access of every kernel
compared with the allocation it was computed from) over the real-width rank steps of every
benchmark configuration and an engine generation: no access leaves its allocation, and a
deliberately wrong call is caught and named instead of faulting.  Runs in a process of its own
because a process loads one build of the library."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUNDS_LIB = os.path.join(ROOT, "vllm-backend-public_amd", "mi355_decode", "libmi355_decode_bounds.so")


def test_shipped_library_has_no_checks():
    from mi355_decode import bounds
    from mi355_decode._lib import Mi355Error
    assert not bounds.enabled()
    with pytest.raises(Mi355Error, match="without -DMI355_BOUNDS"):
        bounds.clear()


def test_no_access_leaves_its_allocation():
    assert os.path.exists(BOUNDS_LIB), f"{BOUNDS_LIB} missing: make -C vllm-backend-public_amd/csrc bounds"
    env = dict(os.environ, MI355_DECODE_LIB=BOUNDS_LIB)
    p = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tests", "bounds_driver.py")], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    print(json.dumps(res, indent=1))
    assert res["negative_control"]["refused"] > 0
    assert len(res["rank_steps"]) == 18 and len(res["engine"]) == 3 and len(res["sampler"]) == 12
