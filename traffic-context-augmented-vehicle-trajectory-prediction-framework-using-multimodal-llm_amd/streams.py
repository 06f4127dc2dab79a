"""A small shared pool of side streams.

HIP maps every stream onto one of a few hardware queues (four by default); streams that share a queue execute in
order with respect to one another, whatever events say.  The forward and the backward together want seven
logical side channels, but never more than three at a time besides the caller's stream, so they draw them from ONE
pool of three streams (three side queues + the caller's = the four hardware queues) instead of creating their own:

    slot 0: forward  - lane-polygon encoder + LTSF front      | backward - lane-polygon encoder's backward
    slot 1: forward  - cross-attention K/V projections        | backward - leaf chain of the cross-attention K/V branch
    slot 2: forward  - Q-Former prefetch of the next batch    | backward - all other leaves (weight / bias gradients)
"""
import os

import torch

_POOL = {}
N_SLOTS = int(os.environ.get("TCAVT_SIDE_STREAMS", "3"))  # A/B: 6 gives every logical channel its own stream


_ACTIVE = None  # set_active_slots(): use only the first n streams of the pool


def set_active_slots(n):
    """Fold the logical side channels onto the first `n` streams of the pool (training.Trainer, data-parallel ranks: the
    process group's RCCL stream takes one place in the five-stream budget of the pipelined step).  Streams handed out earlier
    stay valid; callers that cached one should draw again."""
    global _ACTIVE
    _ACTIVE = max(1, min(int(n), N_SLOTS))


def side_stream(device, slot):
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _POOL:
        _POOL[key] = [torch.cuda.Stream(device=dev) for _ in range(N_SLOTS)]
    return _POOL[key][slot % (_ACTIVE or N_SLOTS)]
