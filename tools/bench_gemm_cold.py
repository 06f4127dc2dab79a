"""gate|up and down GEMMs with ROTATING weight buffers (16 layers' worth, > 256 MiB Infinity Cache in total)
vs one reused buffer: separates 'weights streamed from HBM' (as inside the model) from 'weights MALL-warm'."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops
capi.init(0)
dev = torch.device("cuda:0")
M = 8192
for name, N, K in (("gateup", 16384, 2048), ("down", 2048, 8192)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(16)]
    out = torch.empty(M, N // 2 if name == "gateup" else N, dtype=torch.bfloat16, device=dev)
    kw = dict(silu_mul=True) if name == "gateup" else {}
    for tile in (255, 256, 254):
        for rotate in (False, True):
            for i in range(4):
                ops.gemm_bf16(a, ws[i % 16 if rotate else 0], out=out, tile=tile, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 32
            e0.record()
            for i in range(n):
                ops.gemm_bf16(a, ws[i % 16 if rotate else 0], out=out, tile=tile, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print(f"{name:7s} tile={tile} rotate={rotate}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)

# --- can a preceding "touch" of the next weights (bringing them into the Infinity Cache) recover the warm rate?
name, N, K = "gateup", 16384, 2048
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(16)]
out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
sink = torch.empty(N * K // 2, dtype=torch.float32, device=dev)
for tile in (256, 254):
    tot = 0.0
    n = 32
    for i in range(n + 4):
        w = ws[i % 16]
        # touch: a device-to-device copy of the weights (reads every line once)
        sink.view(torch.bfloat16)[: N * K].copy_(w.view(-1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm_bf16(a, w, out=out, tile=tile, silu_mul=True)
        e1.record()
        torch.cuda.synchronize()
        if i >= 4:
            tot += e0.elapsed_time(e1)
    print(f"gateup tile={tile} rotating weights, touched just before: {tot / n * 1e3:8.1f} us", flush=True)
