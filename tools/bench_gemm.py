"""Micro-benchmark of tcavt_gemm_bf16 on the decoder's shapes (random data, HIP-event timing)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops

capi.init(0)
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
shapes = [("qkv", 3072, 2048), ("o", 2048, 2048), ("gateup", 16384, 2048), ("down", 2048, 8192)]
for name, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    for tile in (255, 256):
        out = torch.empty(M, N // 2 if name == "gateup" else N, dtype=torch.bfloat16, device=dev)
        kw = dict(silu_mul=True) if name == "gateup" else {}
        for _ in range(3):
            ops.gemm_bf16(a, w, out=out, tile=tile, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.gemm_bf16(a, w, out=out, tile=tile, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"{name:7s} M={M} N={N} K={K} tile={tile}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
