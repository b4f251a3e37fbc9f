"""Host side of the one-shot all-reduce (csrc/allreduce_oneshot.hip): buffer set-up, the exchange
of the IPC handles over an existing torch.distributed group, and the per-call wrapper.

    ar = OneShotAllReduce(group, max_bytes=1 << 20)     # every rank, same arguments
    ar.run(t)                                            # in place, on the current stream
    ar.check()                                           # raises if a peer never arrived

Opt-in for the serving engine (MI355_ONESHOT_ALLREDUCE=1): it has run with two ranks on one GPU
(tests/test_tp_gpu.py), not yet across xGMI links."""
from __future__ import annotations

import ctypes

import torch

from ._lib import check, lib
from .ops import _DT, _stream


class OneShotAllReduce:
    def __init__(self, group=None, max_bytes: int = 1 << 20, rank: int | None = None, world: int | None = None,
                 exchange=None):
        """`exchange(handle_bytes) -> [handle_bytes of rank 0, 1, ...]` overrides the gather over
        `group` (tests hand the handles over a pipe)."""
        import torch.distributed as dist
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.max_bytes = int(max_bytes)
        n = int(lib.mi355_allreduce_oneshot_handle_bytes())
        handle = ctypes.create_string_buffer(n)
        comm = ctypes.c_void_p()
        check(lib.mi355_allreduce_oneshot_init(ctypes.byref(comm), self.rank, self.world, self.max_bytes, handle))
        self._comm = comm
        mine = handle.raw
        if exchange is not None:
            handles = exchange(mine)
        elif self.world == 1:
            handles = [mine]
        else:
            handles = [None] * self.world
            dist.all_gather_object(handles, mine, group=group)
        blob = b"".join(handles)
        if len(blob) != n * self.world:
            raise RuntimeError("IPC handle exchange returned the wrong number of bytes")
        check(lib.mi355_allreduce_oneshot_connect(self._comm, blob))

    def fits(self, t: torch.Tensor) -> bool:
        return (t.is_cuda and t.dtype in (torch.bfloat16, torch.float16) and t.is_contiguous()
                and t.numel() % 8 == 0 and t.numel() * 2 <= self.max_bytes)

    def run(self, t: torch.Tensor) -> torch.Tensor:
        if not self.fits(t):
            raise ValueError("one-shot all-reduce takes a contiguous bf16/fp16 GPU tensor of a multiple of 8 elements "
                             f"and at most {self.max_bytes} bytes")
        check(lib.mi355_allreduce_oneshot_run(self._comm, t.data_ptr(), t.numel(), _DT[t.dtype], _stream()))
        return t

    def check(self) -> None:
        check(lib.mi355_allreduce_oneshot_status(self._comm))

    def close(self) -> None:
        if self._comm is not None:
            lib.mi355_allreduce_oneshot_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
