"""Driver of the bounds-checking build of the kernel library (libmi355_decode_bounds.so,
`make -C csrc bounds`, loaded with MI355_DECODE_LIB=<path>).

In that build every global-memory access of every kernel is compared with the device
allocation its base pointer lies in; the list of allocations comes from here, read off
PyTorch's caching allocator.  Typical use, in a process started with the bounds library:

    run the workload once            (allocations settle, graphs are captured)
    bounds.sync_allocations()
    run it again
    torch.cuda.synchronize(); assert bounds.report() is None

A debugging aid: the serving engine never needs it, and the library as shipped answers
`enabled() == False`.
"""
from __future__ import annotations

import ctypes

import torch

from ._lib import check, lib

MAX_RANGES = 4096


def enabled() -> bool:
    return bool(lib.mi355_bounds_enabled())


def live_allocations(granularity: str = "allocation") -> list[tuple[int, int]]:
    """[lo, hi) ranges of the caching allocator's memory, sorted.
    "allocation": every live block at the size that was asked for (an access into the allocator's
                  rounding is refused too) -- for workloads whose buffers all exist beforehand;
    "block":      every live block at its rounded size;
    "segment":    everything the allocator holds, free blocks included -- for workloads that
                  allocate temporaries while they run (those would otherwise be refused); an
                  overflow into a neighbouring tensor then passes, an access outside PyTorch's
                  memory does not."""
    if granularity not in ("allocation", "block", "segment"):
        raise ValueError(f"unknown granularity {granularity!r}")
    out = []
    for seg in torch.cuda.memory_snapshot():
        if granularity == "segment":
            out.append((seg["address"], seg["address"] + seg["total_size"]))
            continue
        addr = seg["address"]
        for b in seg["blocks"]:
            size = b["size"]
            if b["state"].startswith("active"):
                lo = b.get("address", addr)
                want = b.get("requested_size", size) if granularity == "allocation" else size
                out.append((lo, lo + (want if want > 0 else size)))
            addr += size
    out.sort()
    return out


def sync_allocations(extra: list[tuple[int, int]] | tuple = (), granularity: str = "allocation") -> int:
    """Hand the library the allocator's ranges (plus `extra` ones, e.g. memory that did not come
    from PyTorch).  More than MAX_RANGES of them: neighbours are merged, closest first, until
    they fit (coarser, never wrong).  Returns the number of ranges set."""
    torch.cuda.synchronize()
    ranges = sorted(list(live_allocations(granularity)) + [tuple(r) for r in extra])
    merged: list[list[int]] = []
    for lo, hi in ranges:                      # overlapping entries become one
        if merged and lo < merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], hi)
        else:
            merged.append([lo, hi])
    while len(merged) > MAX_RANGES:
        gaps = sorted(range(1, len(merged)), key=lambda i: merged[i][0] - merged[i - 1][1])
        drop = set(gaps[:len(merged) - MAX_RANGES])
        nxt: list[list[int]] = []
        for i, r in enumerate(merged):
            if i in drop:
                nxt[-1][1] = r[1]
            else:
                nxt.append(r)
        merged = nxt
    flat = (ctypes.c_uint64 * (2 * len(merged)))(*[v for r in merged for v in r])
    check(lib.mi355_bounds_set_ranges(flat, len(merged)))
    return len(merged)


def clear() -> None:
    """Checks off: every access passes."""
    check(lib.mi355_bounds_set_ranges(None, 0))


def report(unknown_ok: bool = False) -> dict | None:
    """None if every access since the last report was checked and passed; else the number of
    refused accesses with the first one, and the number of accesses whose base the list did not
    hold (memory allocated after sync_allocations: performed unchecked).  With `unknown_ok` only
    refused accesses make a report."""
    torch.cuda.synchronize()
    out = (ctypes.c_uint64 * 6)()
    name = ctypes.create_string_buffer(256)
    check(lib.mi355_bounds_report(out, name, 256))
    if out[0] == 0 and (unknown_ok or out[5] == 0):
        return None
    return {"refused": int(out[0]), "file": name.value.decode(), "line": int(out[1]), "base": int(out[2]),
            "address": int(out[3]), "bytes": int(out[4]), "unknown_base": int(out[5])}
