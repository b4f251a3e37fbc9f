"""Shared pytest configuration: markers, import paths, golden-vector loading, the order of the
GPU files and what a native abort leaves behind."""
import faulthandler
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vllm-backend-public_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# GPU files in the order they run: operator parity against the oracle first, then the whole
# model, tensor parallelism and the served path; the real-width configuration steps (the
# largest allocations and the widest kernels) after them, the allocation-free entry point last.
# With `-x` one failing file then cannot hide the parity results of the files before it.
GPU_ORDER = ("test_ops_gpu", "test_model_gpu", "test_tp_gpu", "test_plumbing_gpu", "test_configs_gpu",
             "test_allreduce_gpu", "test_rccl_gpu", "test_bounds_gpu")
_fault_log = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")
    # A native abort (HIP / HSA runtime, glibc) prints its reason to fd 2 and raises SIGABRT.
    # pytest.ini keeps fd 2 uncaptured (--capture=sys) and pytest's own all-threads traceback
    # dump off, so that reason stays the last thing in the log; the Python frames of every
    # thread go to a side file instead.
    global _fault_log
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        _fault_log = open(os.path.join(ROOT, "gpurun_out", "pytest_fault.log"), "w")
        faulthandler.enable(file=_fault_log, all_threads=True)
    except OSError:
        faulthandler.enable(all_threads=False)


# Operator forms that are built and parity-tested but that the serving path does not run (measured
# slower than what it runs): their tests go after every file above, so that with `-x` a failure
# in one of them cannot hide the parity results of the operators that serve.
OFF_PATH_TESTS = ("reduce_in_launch", "norm_prologue", "w8a8", "quant_rows_fp8")


def pytest_collection_modifyitems(config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name.startswith("test_zz"):
            return (2, 0)
        if name in GPU_ORDER:
            if any(frag in item.name for frag in OFF_PATH_TESTS):
                return (1, len(GPU_ORDER))
            return (1, GPU_ORDER.index(name))
        return (0, 0)
    items.sort(key=key)                       # stable: the order inside a file is kept


_last_file = [None]


def pytest_runtest_logstart(nodeid, location):
    """One line on fd 2 per GPU test file (per case for the real-width steps): a native abort
    then names where it happened even though pytest never gets to report the test."""
    fname = nodeid.split("::")[0]
    if "_gpu" not in fname and "test_zz" not in fname:
        return
    if fname != _last_file[0] or "test_configs_gpu" in fname:
        _last_file[0] = fname
        try:
            os.write(2, f"\n[gpu tests] {nodeid}\n".encode())
        except OSError:
            pass


def load_npz(name):
    """Load a golden .npz; arrays stored as raw bf16 words (key suffix __bf16)
    are decoded to float32."""
    z = np.load(os.path.join(GOLDEN, name))
    out = {}
    for k in z.files:
        if k.endswith("__bf16"):
            out[k[:-6]] = (z[k].astype(np.uint32) << 16).view(np.float32)
        else:
            out[k] = z[k]
    return out


@pytest.fixture(scope="session")
def ops_golden():
    return load_npz("ops_hf.npz")
