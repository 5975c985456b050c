"""HIP-graph capture of a whole render / training step.

One step of the hot path is ~45 kernel launches with fixed shapes; launched eagerly from Python the host
costs more than the GPU work.  `GraphedStep` runs the step a few times eagerly on a side stream (first-use
uploads of tables, hipFuncSetAttribute, allocator warm-up), captures it once into a HIP graph through
torch.cuda.graphs (the C-ABI kernels are launched on torch's current stream, so they are captured like any
other work) and then replays it.  Inputs must live in fixed tensors that the caller updates in place.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Optional

import torch


def _capture_mode() -> str:
    """How the capture treats HIP calls of OTHER threads.  With a torch.distributed process group alive its watchdog
    thread polls the events of collectives that were enqueued eagerly (the warm-up runs' all-reduces) with
    hipEventQuery every ~100 ms; under the default "global" mode such a call from any thread while a capture is open
    fails with hipErrorStreamCaptureUnsupported, the watchdog rethrows and the process aborts (seen on the MI355X in
    one of three `bench.py --force-dp` runs, round 4).  "thread_local" confines the restriction to the capturing
    thread — the only one that launches work here."""
    dist = torch.distributed
    forced = os.environ.get("HN_CAPTURE_MODE")           # diagnosis only: "global" reproduces the abort
    if forced:
        return forced
    return "thread_local" if dist.is_available() and dist.is_initialized() else "global"


def _quiesce_collectives():
    """Before a capture: let every eagerly enqueued collective finish and give the process group's watchdog one poll
    period to retire it from its list, so that it has nothing to query while the capture is open (second line of
    defence next to _capture_mode)."""
    torch.cuda.synchronize()
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized() and os.environ.get("HN_CAPTURE_QUIESCE", "1") != "0":
        time.sleep(0.25)


class GraphedStep:
    def __init__(self, fn: Callable[[], object], warmup: int = 3, warmup_fn: Optional[Callable[[], object]] = None,
                 pool=None, mutates_params: bool = True):
        """warmup_fn: what the warm-up runs execute instead of `fn` (a step whose tail is captured separately runs
        that tail as well); warmup=0 with warmup_fn=None: capture only (`fn` consumes state a warm-up run would use
        up); pool: share the memory pool of another graph (`other.graph.pool()`); mutates_params: the captured work
        contains an optimizer launch (False for inference / forward+backward-only graphs: their replays then do not
        invalidate every packed weight stream of the process)."""
        self.mutates_params = bool(mutates_params)
        if warmup > 0 or warmup_fn is not None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    (warmup_fn or fn)()
            torch.cuda.current_stream().wait_stream(side)
        _quiesce_collectives()
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph, pool=pool, capture_error_mode=_capture_mode()):
                self.out = fn()
        except BaseException:
            # an operation that cannot be captured (a host-staged collective, a synchronising copy) invalidates the
            # capture; torch has ended it on the way out — leave the device idle for whoever falls back to eager work
            try:
                torch.cuda.synchronize()
            except RuntimeError:
                pass
            raise

    def __call__(self):
        self.graph.replay()
        # the replay may have stepped an optimizer (parameters changed behind Python's back): packed weight streams
        # of inference forwards that follow must be rebuilt
        if self.mutates_params:
            from . import machine
            machine.note_parameters_changed()
        return self.out
