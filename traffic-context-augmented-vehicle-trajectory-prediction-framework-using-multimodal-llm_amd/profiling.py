"""In-situ kernel timing with HIP events on the launching stream (bench.py's roofline leg).

``torch.cuda.Event`` objects are hipEvents recorded on torch's current stream, which is the
stream every C-ABI launch goes to (capi.stream_ptr), so an event pair brackets exactly the
kernels launched between ``start`` and ``stop``.  Events are only read after the timed region.
"""
from collections import defaultdict

import torch


class KernelTimer:
    def __init__(self):
        self._pairs = defaultdict(list)
        self._open = {}

    def start(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._open[name] = e

    def stop(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._pairs[name].append((self._open.pop(name), e))

    def reset(self):
        self._pairs.clear()
        self._open.clear()

    def summary(self):
        """-> {name: (count, mean_ms)}; call after torch.cuda.synchronize()."""
        out = {}
        for name, pairs in self._pairs.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = (len(ms), sum(ms) / max(1, len(ms)))
        return out
