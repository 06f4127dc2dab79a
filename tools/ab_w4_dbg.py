"""Timing-only elimination experiments on the 4-wave gate|up GEMM (codes 261-267 give wrong results by design)."""
import os, sys
os.environ["TCAVT_GEMM_TIMING_EXPERIMENTS"] = "1"  # codes 261-267 refuse to run without it
os.environ.setdefault("TCAVT_LIB", "exp")  # the -DTCAVT_EXPERIMENTS build: python -m tcavt_amd.build --experiments
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops
capi.init(0)
dev = torch.device("cuda:0")
M, N, K = 8192, 16384, 2048
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(16)]
out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
names = {256: "8-wave", 257: "w4", 261: "w4 no-DMA", 262: "w4 no-barrier", 263: "w4 no-DMA no-barrier", 264: "w4 no-fragment-loads", 265: "w4 DMA never waited for", 267: "w4 MFMA only"}
for tile, nm in names.items():
    for i in range(4):
        ops.gemm_bf16(a, ws[i], out=out, tile=tile, silu_mul=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(32):
        ops.gemm_bf16(a, ws[i % 16], out=out, tile=tile, silu_mul=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 32
    print(f"{nm:24s} {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)
