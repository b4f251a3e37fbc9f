"""Thread-side engine loop for the HTTP server.

The LLMEngine runs on its own thread so that the ASGI loop (and with it
``GET /health``, which kubelet probes with a 5 s timeout,
benchmark-vllm/vllm_templates.py:182-201) never waits on the GPU.  Requests enter
through a thread-safe inbox; every generated token is handed back to the
request's asyncio queue with ``loop.call_soon_threadsafe``.

Under tensor parallelism rank 0 owns the HTTP socket; each loop iteration it
broadcasts how many bytes of requests/aborts it has collected (gloo, one 8-byte
tensor) and the pickled messages only when there are any, then all ranks advance
their identical engines by one step.
"""
from __future__ import annotations

import logging
import queue
import threading
import time
from dataclasses import dataclass
from typing import Callable, List, Optional

from .engine import LLMEngine, RequestOutput
from .sampling_params import SamplingParams

log = logging.getLogger("mi355_decode")


@dataclass
class _Add:
    request_id: str
    prompt_ids: List[int]
    params: SamplingParams
    arrival: float


@dataclass
class _Abort:
    request_id: str


class EngineThread:
    def __init__(self, make_engine: Callable[[], LLMEngine], control_group=None, rank: int = 0):
        self._make_engine = make_engine
        self.engine: Optional[LLMEngine] = None
        self.inbox: "queue.Queue" = queue.Queue()
        self.sinks = {}                      # request_id -> callable(RequestOutput | Exception)
        self.ready = threading.Event()
        self.failed: Optional[BaseException] = None
        self.stop_flag = False
        self.control_group, self.rank = control_group, rank
        self.thread = threading.Thread(target=self._run, name="mi355-engine", daemon=True)
        self.sink_loops = {}                 # request_id -> asyncio loop its sink must run on (or absent)
        import os
        self.admission_window = float(os.environ.get("MI355_ADMISSION_WINDOW_MS", "2.0")) * 1e-3
        self.admission_cap = float(os.environ.get("MI355_ADMISSION_CAP_MS", "20.0")) * 1e-3
        self.bursts = []                     # (requests, arrival spread ms, wait after the last arrival ms) of recent admissions
        self.burst_gaps = []                 # (requests, longest and second longest pause between two arrivals ms) of the same admissions
        self.steps = 0
        self.exchanges = 0                   # control broadcasts of a tensor-parallel server (one per loop turn) ...
        self.exchange_s = 0.0                # ... and the time rank 0 spent in them
        self.tokens_out = 0
        self.started = time.time()

    def start(self):
        self.thread.start()

    # ---- called from the server thread ---------------------------------- #
    def submit(self, request_id: str, prompt_ids: List[int], params: SamplingParams, sink, loop=None) -> None:
        """`sink(RequestOutput | Exception)` is called from the engine thread, or -- with `loop` -- on
        that asyncio loop's thread: all outputs of one engine step that go to one loop are handed
        over with a single thread-safe wake-up."""
        if loop is not None:
            self.sink_loops[request_id] = loop
        self.sinks[request_id] = sink
        self.inbox.put(_Add(request_id, prompt_ids, params, time.perf_counter()))

    def abort(self, request_id: str) -> None:
        self.sinks.pop(request_id, None)
        self.sink_loops.pop(request_id, None)
        self.inbox.put(_Abort(request_id))

    def shutdown(self):
        self.stop_flag = True
        self.inbox.put(None)

    # ---- engine thread ----------------------------------------------------- #
    def _collect(self, block: bool) -> list:
        msgs = []
        try:
            if block:
                m = self.inbox.get(timeout=0.05)
                msgs.append(m)
            while True:
                msgs.append(self.inbox.get_nowait())
        except queue.Empty:
            pass
        # Admission window: requests of one client burst arrive a fraction of a millisecond apart
        # (now and then more than a millisecond: the HTTP thread shares the interpreter with this
        # one -- with a 1 ms window two of five 32-request bursts were split, which costs the late
        # half a whole prefill step of latency; 2 ms costs every burst 1 ms.  Every 32-request burst
        # holds two pauses of 1.05-1.5 ms between arrivals -- the server accepting the burst's other
        # connections before it reads them -- so a shorter window after the first few requests does
        # not work either: 0.5 ms split 14 of 16 bursts);
        # prefilling each as it lands would stagger the batch over several prefill steps.  While
        # new requests keep arriving within `admission_window` of each other (and for at most
        # `admission_cap` in all) the engine waits for the rest of the burst before it schedules.
        # (The engine this replaces has the same knob: its scheduler delay factor.)
        if self.admission_window > 0 and any(isinstance(m, _Add) for m in msgs):
            t_end = time.perf_counter() + self.admission_cap
            while time.perf_counter() < t_end:
                try:
                    m = self.inbox.get(timeout=self.admission_window)
                except queue.Empty:
                    break
                msgs.append(m)
                if m is None:
                    break
        adds = [m.arrival for m in msgs if isinstance(m, _Add)]
        if adds:          # how long the burst took to arrive and how long the engine then waited for more of it
            now = time.perf_counter()
            # + absolute stamps (time.perf_counter: one clock for every process of the host): first and last
            # arrival, hand-over to the scheduler; _run adds the time the burst's first tokens existed
            self.bursts.append([len(adds), (max(adds) - min(adds)) * 1e3, (now - max(adds)) * 1e3,
                                min(adds), max(adds), now, None])
            del self.bursts[:-64]
            srt = sorted(adds)                # the longest pause between two arrivals of the burst: what the window must outlast
            gaps = sorted((b - a for a, b in zip(srt, srt[1:])), reverse=True) + [0.0, 0.0]
            self.burst_gaps.append((len(adds), gaps[0] * 1e3, gaps[1] * 1e3))
            del self.burst_gaps[:-64]
        return msgs

    def _exchange(self, msgs: list) -> list:
        """Rank 0 -> all ranks: this iteration's control messages.  Every iteration broadcasts one
        8-byte length; a payload (the pickled messages) follows only when rank 0 collected any, so
        a decode step with an empty inbox costs one small gloo broadcast and no serialisation."""
        if self.control_group is None:
            return msgs
        import pickle

        import torch
        import torch.distributed as dist
        payload = pickle.dumps(msgs) if self.rank == 0 and msgs else b""
        n = torch.tensor([len(payload)], dtype=torch.int64)
        dist.broadcast(n, src=0, group=self.control_group)
        size = int(n.item())
        if size == 0:
            return []
        buf = (torch.frombuffer(bytearray(payload), dtype=torch.uint8) if self.rank == 0
               else torch.empty(size, dtype=torch.uint8))
        dist.broadcast(buf, src=0, group=self.control_group)
        return msgs if self.rank == 0 else pickle.loads(buf.numpy().tobytes())

    def _deliver(self, batch) -> None:
        """batch: (request_id, sink, output) triples of one engine step."""
        groups = {}
        for rid, sink, o in batch:
            loop = self.sink_loops.get(rid)
            if getattr(o, "finished", True):              # last output (or an exception): forget the loop
                self.sink_loops.pop(rid, None)
            if loop is None:
                sink(o)
            else:
                groups.setdefault(loop, []).append((sink, o))
        for loop, items in groups.items():
            def run(items=items):
                for sink, o in items:
                    sink(o)
            try:
                loop.call_soon_threadsafe(run)
            except RuntimeError:                          # the client's loop is gone: nobody is listening
                pass

    def _run(self):
        try:
            self.engine = self._make_engine()
            # Everything allocated so far (weights' Python objects, kernels' bindings, the HTTP app)
            # lives as long as the server: moved out of the collector's reach, so that a full
            # collection -- tens of milliseconds over a few million objects, in whichever thread
            # happens to allocate -- cannot land in the middle of a burst of requests or a decode step.
            import gc
            gc.collect()
            gc.freeze()
            self.ready.set()
            eng = self.engine
            while True:
                msgs = self._collect(block=not eng.has_unfinished()) if self.rank == 0 else []
                if self.stop_flag:
                    msgs.append(None)
                if self.control_group is not None:
                    t_x = time.perf_counter()
                    msgs = self._exchange(msgs)
                    self.exchanges += 1
                    self.exchange_s += time.perf_counter() - t_x
                if any(m is None for m in msgs):
                    break
                for m in msgs:
                    if isinstance(m, _Add):
                        try:
                            eng.add_request(m.request_id, m.prompt_ids, m.params, arrival_time=m.arrival)
                        except Exception as exc:         # bad request: report to its sink only
                            sink = self.sinks.pop(m.request_id, None)
                            if sink:
                                self._deliver([(m.request_id, sink, exc)])
                    elif isinstance(m, _Abort):
                        eng.abort_request(m.request_id)
                if not eng.has_unfinished():
                    continue
                t_step = time.perf_counter()
                outs: List[RequestOutput] = eng.step()
                self.steps += 1
                if self.bursts and self.bursts[-1][6] is None and outs and outs[0].first_token_time is not None \
                        and outs[0].num_output_tokens == 1:
                    self.bursts[-1][6] = outs[0].first_token_time      # the step that prefilled the last admitted burst
                    self.bursts[-1].append(t_step)
                batch = []
                for o in outs:
                    self.tokens_out += len(o.new_token_ids)
                    sink = self.sinks.get(o.request_id)
                    if sink:
                        batch.append((o.request_id, sink, o))
                        if o.finished:
                            self.sinks.pop(o.request_id, None)
                self._deliver(batch)
        except BaseException as exc:                     # fatal: surface on /health and to waiters
            log.exception("engine thread died")
            self.failed = exc
            self.ready.set()
            try:
                self._deliver([(rid, sink, exc) for rid, sink in list(self.sinks.items())])
            except Exception:
                pass
            self.sinks.clear()
