"""Host side of the one-shot all-reduce (csrc/allreduce_oneshot.hip): buffer set-up, the exchange
of the IPC handles over an existing torch.distributed group, a self-test that every rank must pass
before the engine relies on it, and the per-call wrappers.

    ar = OneShotAllReduce(group, max_bytes=1 << 21)     # every rank, same arguments
    ar.run(t)                                            # in place, on the current stream
    ar.run_slabs(slabs, n, out)                          # out = sum over ranks of round(sum of n fp32 slabs)
    ar.check()                                           # raises if a peer never arrived

The engine uses it for the decode-sized messages of a tensor-parallel step whenever RCCL is the
process group's backend (MI355_ONESHOT_ALLREDUCE=0 keeps RCCL for everything, =1 forces it under
any backend) -- the harness switches the stock engine's custom all-reduce off
(benchmark-vllm/vllm_templates.py:62,157-160), so this collective is the engine's own.  It has run
with several ranks on ONE GPU; across xGMI links it is unmeasured, which is why `connect_or_none`
maps the peers' buffers, runs a known-answer all-reduce through them on every rank, lets the ranks
agree on the outcome over the process group, and hands back None -- RCCL stays in charge -- if any
rank could not map a buffer or saw a wrong sum."""
from __future__ import annotations

import ctypes
import os
import logging

import torch

from ._lib import Mi355Error, check, lib
from .ops import _DT, _stream

log = logging.getLogger("mi355_decode")


class OneShotAllReduce:
    def __init__(self, group=None, max_bytes: int = 1 << 20, rank: int | None = None, world: int | None = None,
                 exchange=None):
        """`exchange(handle_bytes) -> [handle_bytes of rank 0, 1, ...]` overrides the gather over
        `group` (tests hand the handles over a pipe)."""
        import torch.distributed as dist
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.max_bytes = int(max_bytes)
        self._comm = None
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
        # the error word of the step in flight lands here (pinned), next to the step's sampled tokens
        self.host_word = torch.zeros(2, dtype=torch.int32).pin_memory()

    def fits(self, t: torch.Tensor) -> bool:
        return (t.is_cuda and t.dtype in (torch.bfloat16, torch.float16) and t.is_contiguous()
                and t.numel() % 8 == 0 and t.numel() * 2 <= self.max_bytes)

    def run(self, t: torch.Tensor) -> torch.Tensor:
        if not self.fits(t):
            raise ValueError("one-shot all-reduce takes a contiguous bf16/fp16 GPU tensor of a multiple of 8 elements "
                             f"and at most {self.max_bytes} bytes")
        check(lib.mi355_allreduce_oneshot_run(self._comm, t.data_ptr(), t.numel(), _DT[t.dtype], _stream()))
        return t

    def run_slabs(self, slabs: torch.Tensor, num_slabs: int, out: torch.Tensor) -> torch.Tensor:
        """out = sum over the ranks of round(slabs[0] + ... + slabs[num_slabs - 1]): the slab sum of a K-split
        projection formed inside the all-reduce launch (slabs: fp32, [num_slabs, out.numel()] contiguous)."""
        if not self.fits(out):
            raise ValueError("one-shot all-reduce takes a contiguous bf16/fp16 GPU tensor of a multiple of 8 elements "
                             f"and at most {self.max_bytes} bytes")
        if (slabs.dtype != torch.float32 or not slabs.is_cuda or not slabs.is_contiguous() or num_slabs < 1
                or slabs.numel() < num_slabs * out.numel()):
            raise ValueError("slabs must be contiguous float32 holding [num_slabs, out.numel()]")
        check(lib.mi355_allreduce_oneshot_run_slabs(self._comm, out.data_ptr(), slabs.data_ptr(), num_slabs, out.numel(),
                                                    _DT[out.dtype], _stream()))
        return out

    MAX_NORM_ROWS, MAX_NORM_HIDDEN = 256, 8192

    def fits_add_norm(self, rows: int, hidden: int, dtype) -> bool:
        return (dtype in (torch.bfloat16, torch.float16) and 0 < rows <= self.MAX_NORM_ROWS and hidden % 8 == 0
                and hidden <= self.MAX_NORM_HIDDEN and rows * hidden * 2 <= self.max_bytes)

    TWO_SHOT_MIN_BYTES = 512 << 10

    def two_shot(self, rows: int, hidden: int) -> bool:
        """Which form run_add_norm takes for a [rows, hidden] message -- a function of the communicator and the
        shape only, so every rank answers alike.  The two-hop form moves 2 (world - 1) / world message sizes over
        a rank's links instead of world - 1 and waits for one flag more: from four ranks and 512 KiB on the bytes
        outweigh the hop (arithmetic on the links' rate; no multi-GPU box has timed either form).
        MI355_ALLREDUCE_TWO_SHOT=0 / 1 forces a form."""
        forced = os.environ.get("MI355_ALLREDUCE_TWO_SHOT", "")
        if forced in ("0", "1"):
            return forced == "1"
        return self.world >= 4 and rows >= self.world and rows * hidden * 2 >= self.TWO_SHOT_MIN_BYTES

    def run_add_norm(self, residual: torch.Tensor, weight: torch.Tensor, eps: float, out: torch.Tensor,
                     data: torch.Tensor | None = None, slabs: torch.Tensor | None = None, num_slabs: int = 0,
                     two_shot: bool | None = None) -> torch.Tensor:
        """The all-reduce of a projection's [rows, hidden] result and the residual add + RMSNorm behind it in one
        launch: residual += sum over the ranks (of `data`, or of the rounded slab sum), out = norm(residual) * weight;
        bit-identical to run / run_slabs followed by ops.fused_add_rms_norm.  The reduced projection is not stored.
        `two_shot`: the two-hop form (same results); None: self.two_shot(rows, hidden).  Every rank must pass the same."""
        rows, hidden = residual.shape
        if two_shot is None:
            two_shot = self.two_shot(rows, hidden)
        if not self.fits_add_norm(rows, hidden, residual.dtype):
            raise ValueError(f"one-shot all-reduce + norm takes up to {self.MAX_NORM_ROWS} rows of up to "
                             f"{self.MAX_NORM_HIDDEN} 16-bit elements within {self.max_bytes} bytes")
        if (data is None) == (slabs is None):
            raise ValueError("exactly one of data / slabs")
        for t, n in ((residual, "residual"), (out, "out")):
            if not t.is_cuda or t.dtype != residual.dtype or tuple(t.shape) != (rows, hidden) or t.stride(1) != 1:
                raise ValueError(f"{n} must be a [rows, hidden] GPU tensor of the message's dtype with unit column stride")
        if weight.dtype != residual.dtype or weight.numel() != hidden or not weight.is_contiguous() or not weight.is_cuda:
            raise ValueError("weight must be contiguous [hidden] in the message's dtype")
        if data is not None and (not data.is_cuda or data.dtype != residual.dtype or not data.is_contiguous()
                                 or data.numel() != rows * hidden):
            raise ValueError("data must be a contiguous [rows, hidden] GPU tensor of the message's dtype")
        if slabs is not None and (slabs.dtype != torch.float32 or not slabs.is_cuda or not slabs.is_contiguous()
                                  or num_slabs < 1 or slabs.numel() < num_slabs * rows * hidden):
            raise ValueError("slabs must be contiguous float32 holding [num_slabs, rows * hidden]")
        fn = lib.mi355_allreduce_twoshot_run_add_norm if two_shot else lib.mi355_allreduce_oneshot_run_add_norm
        check(fn(
            self._comm, data.data_ptr() if data is not None else 0, slabs.data_ptr() if slabs is not None else 0,
            num_slabs, residual.data_ptr(), residual.stride(0), weight.data_ptr(), out.data_ptr(), out.stride(0),
            rows, hidden, float(eps), _DT[residual.dtype], _stream()))
        return out

    MAX_KEY_ROWS = 256

    def run_max_keys(self, keys: torch.Tensor, out_ids: torch.Tensor, out_ids2: torch.Tensor | None = None) -> torch.Tensor:
        """Token ids of the max over the ranks of one (logit, index) key per row (ops.argmax_shard_key): the MAX
        all-reduce of the keys and ops.argmax_key_to_index in one launch; `out_ids2`: a second destination."""
        n = keys.numel()
        for t, name in ((keys, "keys"), (out_ids, "out_ids"), (out_ids2, "out_ids2")):
            if t is not None and (not t.is_cuda or t.dtype != torch.int64 or not t.is_contiguous() or t.numel() < n):
                raise ValueError(f"{name} must be a contiguous int64 GPU tensor of one element per row")
        if not 0 < n <= self.MAX_KEY_ROWS:
            raise ValueError(f"one-shot key reduction takes 1 .. {self.MAX_KEY_ROWS} rows")
        check(lib.mi355_allreduce_oneshot_run_max_keys(self._comm, keys.data_ptr(), out_ids.data_ptr(),
                                                       out_ids2.data_ptr() if out_ids2 is not None else 0, n, _stream()))
        return out_ids

    def check(self) -> None:
        """Synchronises with the queued work; raises if a wait of any call so far timed out."""
        check(lib.mi355_allreduce_oneshot_status(self._comm))

    def queue_status(self, slot: int = 0) -> None:
        """Queues a copy of the error word into host_word[slot] on the current stream (no synchronisation)."""
        check(lib.mi355_allreduce_oneshot_status_async(self._comm, self.host_word.data_ptr() + 4 * slot, _stream()))

    def raise_if_failed(self, slot: int = 0) -> None:
        """After the stream work behind queue_status(slot) has completed: raises if a peer failed to arrive."""
        word = int(self.host_word[slot])
        if word != 0:
            raise Mi355Error(f"one-shot all-reduce: rank {self.rank} waited 10 s for rank {word - 1}, which never delivered "
                             "its chunk; this rank's activations were not reduced -- stopping instead of serving them")

    def self_test(self, device) -> bool:
        """A known-answer all-reduce through the mapped buffers, at two sizes and on both slot parities,
        plain and from slabs, then through the add + norm and the key back ends.  True if this rank saw the exact
        results and no wait timed out."""
        try:
            ok = True
            for count in (8 * self.world, min(self.max_bytes // 2, 1 << 19)):
                count -= count % 8
                for _ in range(2):                               # both parities
                    t = torch.full((count,), float(self.rank + 1), dtype=torch.bfloat16, device=device)
                    t[::3] = 0.5 * (self.rank + 1)
                    self.run(t)
                    want = self.world * (self.world + 1) / 2.0
                    ok &= bool((t[1::3] == want).all().item()) and bool((t[::3] == 0.5 * want).all().item())
                    s = torch.full((2, count), 0.25 * (self.rank + 1), dtype=torch.float32, device=device)
                    out = torch.zeros(count, dtype=torch.bfloat16, device=device)
                    self.run_slabs(s, 2, out)
                    ok &= bool((out == 0.5 * want).all().item())
            # the add + norm back end: residual 0, weight 1, every rank's rows constant r + 1 -> the summed row is the
            # constant W (W + 1) / 2, its norm round(c * rsqrt(c * c + eps)); and the key back end: rank r offers the
            # key (r + 1) << 32 | ~(100 * r + row) for every row, the last rank's wins
            rows, hidden = 4, 1024
            if self.fits_add_norm(rows, hidden, torch.bfloat16):
                for _ in range(2):
                    data = torch.full((rows, hidden), float(self.rank + 1), dtype=torch.bfloat16, device=device)
                    resid = torch.zeros(rows, hidden, dtype=torch.bfloat16, device=device)
                    x = torch.zeros(rows, hidden, dtype=torch.bfloat16, device=device)
                    c = self.world * (self.world + 1) / 2.0
                    want_x = torch.tensor(c, dtype=torch.float32).mul(torch.rsqrt(torch.tensor(c * c + 1e-5))).to(torch.bfloat16)
                    for two in (False, True):                    # one hop, then the two-hop form
                        resid.zero_()
                        x.zero_()
                        self.run_add_norm(resid, torch.ones(hidden, dtype=torch.bfloat16, device=device), 1e-5, x, data=data,
                                          two_shot=two)
                        ok &= bool((resid == c).all().item()) and bool((x.float() - want_x.float()).abs().max().item() <= 2 ** -7)
                    keys = torch.tensor([((self.rank + 1) << 32) | (0xFFFFFFFF - (100 * self.rank + i)) for i in range(rows)],
                                        dtype=torch.int64, device=device)
                    ids = torch.full((rows,), -1, dtype=torch.int64, device=device)
                    self.run_max_keys(keys, ids)
                    ok &= ids.tolist() == [100 * (self.world - 1) + i for i in range(rows)]
            self.check()
            return ok
        except Mi355Error as exc:
            log.warning("one-shot all-reduce self-test failed on rank %d: %s", self.rank, exc)
            return False

    def close(self) -> None:
        if self._comm is not None:
            lib.mi355_allreduce_oneshot_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def connect_or_none(group, max_bytes: int, device):
    """The communicator for `group` if EVERY rank could map every peer's buffer and passed the
    self-test; else None on every rank (the caller keeps the process group's all-reduce).  The
    ranks agree on the outcome with a MIN all-reduce over the group, so either all use the one-shot
    kernel or none does."""
    import torch.distributed as dist
    ar, ok = None, 0
    try:
        ar = OneShotAllReduce(group, max_bytes=max_bytes)
        connected = 1
    except (Mi355Error, RuntimeError) as exc:
        log.warning("one-shot all-reduce: set-up failed on rank %d (%s)", dist.get_rank(group), exc)
        connected = 0
    # everyone must have mapped everything before anyone launches into a peer's buffer
    flag = torch.tensor([connected], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 1:
        ok = int(ar.self_test(device))
        flag.fill_(ok)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = int(flag.item())
    if ok != 1:
        if ar is not None:
            ar.close()
        return None
    return ar
