"""Live timing of the C-ABI calls of a run: HIP events around every `libpuflow_hip.so` entry point, recorded on the stream the
call launches on (torch's current stream is the stream handed to the library).  bench.py uses it to find and time the
dominant kernel of a training step inside the benchmark itself (the rocprofv3 summary under profiles/ must agree)."""
from __future__ import annotations

import contextlib
from collections import defaultdict

import torch

from . import _lib


class CallProfile:
    def __init__(self):
        self.events = defaultdict(list)          # name -> [(start, end)]

    def table(self):
        """name -> (calls, total ms, average ms); call after torch.cuda.synchronize()."""
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = (len(ms), sum(ms), sum(ms) / max(len(ms), 1))
        return out


@contextlib.contextmanager
def profile_calls():
    """Wrap every bound function of the loaded library for the duration of the block.  Eager launches only (events cannot be
    recorded usefully inside a graph capture)."""
    lib = _lib.load()
    prof = CallProfile()
    saved = {}

    def wrap(name, fn):
        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a)
            e1.record()
            prof.events[name].append((e0, e1))
            return r
        return timed

    for name in _lib.SIGNATURES:
        fn = getattr(lib, name)
        saved[name] = fn
        try:
            setattr(lib, name, wrap(name, fn))
        except Exception:                          # pragma: no cover
            pass
    try:
        yield prof
    finally:
        for name, fn in saved.items():
            setattr(lib, name, fn)
