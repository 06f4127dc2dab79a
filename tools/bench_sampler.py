"""Token selection in isolation (tcavt_sample_logits): one-stage vs two-stage form, with / without the logits processors, sampling
and greedy, at the decode step's shape (V = 128 256).  HIP-event time per call over a loop.  Measurement only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tcavt_amd import capi, ops

capi.init(0)
dev = torch.device("cuda:0")
V = 128256
for B in (8, 32):
    g = torch.Generator().manual_seed(B)
    base = (torch.randn(B, V, generator=g) * 6.0).to(dev)
    cap = 320
    hist0 = torch.randint(0, V, (B, cap), generator=g).to(dev)
    for name, (do_sample, rep, ng) in (("sampling + processors", (1, 1.2, 3)), ("sampling, no processors", (1, 1.0, 0)),
                                       ("greedy + processors", (0, 1.2, 3)), ("greedy, no processors", (0, 1.0, 0))):
        line = f"B={B:2d} {name:26s}"
        for two in (False, True):
            wsp = ops.sample_workspace(B, dev) if two else None
            sp = capi.SampleParams(0.9, 0.9, rep, 40, ng, do_sample, -1, 0, 7)
            logits = base.clone()
            hist = hist0.clone()
            hl = torch.full((B,), 300, dtype=torch.int32, device=dev)
            step = torch.zeros(1, dtype=torch.int32, device=dev)
            cur = torch.zeros(B, dtype=torch.int64, device=dev)
            pos = torch.zeros(B, dtype=torch.int32, device=dev)
            fin = torch.zeros(B, dtype=torch.int32, device=dev)
            out = torch.zeros(B, 4096, dtype=torch.int64, device=dev)

            def call():
                hl.fill_(300)
                logits.copy_(base)  # (the penalties are applied in place: fresh scores per call)
                ops.sample_logits(logits, hist, hl, sp, step, cur, pos, fin, out, advance_pos=True, workspace=wsp)

            for _ in range(5):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 200
            e0.record()
            for _ in range(n):
                call()
            e1.record()
            torch.cuda.synchronize()
            line += f" | {'two-stage' if two else 'one-stage'} {e0.elapsed_time(e1) / n * 1e3:6.1f} us"
        print(line + "   (incl. a fill_ and a 4 MB x B copy_ launch)", flush=True)
