"""The C-ABI library loads without a GPU and exports exactly the entry points
that include/mi355_decode.h declares (no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mi355_decode.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "mi355_paged_attn_decode" in syms and "mi355_rms_norm" in syms and len(syms) >= 10


def test_library_exports_every_declared_symbol():
    from mi355_decode import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), f"{name} declared in the header but not exported"
    assert sorted(_lib.EXPORTS) == declared_symbols(), "host binding and header disagree"


def test_abi_version_and_target():
    from mi355_decode import _lib
    assert _lib.lib.mi355_abi_version() == _lib.ABI_VERSION == 1
    assert _lib.lib.mi355_target_arch() == b"gfx950"
    assert _lib.BLOCK_SIZE == 16


def test_argument_validation_needs_no_gpu():
    """Host-side checks reject bad calls before any launch."""
    from mi355_decode import _lib
    rc = _lib.lib.mi355_rms_norm(None, None, None, 4, 256, 256, 256, 1e-5, 0, None)
    assert rc == -1 and b"null pointer" in _lib.lib.mi355_last_error()
    buf = (ctypes.c_char * 64)()
    p = ctypes.addressof(buf)
    p += (-p) % 16
    rc = _lib.lib.mi355_rms_norm(p, p, p, 4, 250, 256, 256, 1e-5, 0, None)
    assert rc == -2 and b"multiple of 8" in _lib.lib.mi355_last_error()
    rc = _lib.lib.mi355_paged_attn_decode(p, p, 4096, p, p, 8, p, 4, p, 2, 34, 2, 128, 0.1, 1, None, 0, 0, None)
    assert rc == -2 and b"16 query heads" in _lib.lib.mi355_last_error()
    rc = _lib.lib.mi355_paged_attn_decode(p, p, 4096, p, p, 8, p, 4, p, 2, 8, 2, 96, 0.1, 1, None, 0, 0, None)
    assert rc == -2 and b"head_dim" in _lib.lib.mi355_last_error()
    rc = _lib.lib.mi355_paged_attn_decode(p, p, 4096, p, p, 8, p, 4, p, 2, 8, 2, 128, 0.1, 4, None, 0, 0, None)
    assert rc == -1 and b"workspace" in _lib.lib.mi355_last_error()
    assert _lib.lib.mi355_paged_attn_decode_workspace(32, 32, 128, 4) == 32 * 32 * 4 * 130 * 4


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    import sys
    monkeypatch.setenv("MI355_DECODE_LIB", str(tmp_path / "nope.so"))
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.startswith("mi355_decode")}
    try:
        try:
            importlib.import_module("mi355_decode")
            raised = False
        except ImportError as e:
            raised = "no fallback" in str(e)
        assert raised
    finally:
        for k in list(sys.modules):
            if k.startswith("mi355_decode"):
                del sys.modules[k]
        sys.modules.update(saved)
