"""Headroom probe (measurement only, never on the product path): the four decoder GEMM shapes through the vendor
library (torch.nn.functional.linear -> hipBLASLt) next to tcavt_gemm_bf16, with rotating (HBM-cold) weights as inside
the model.  The library computes the plain GEMM only (no SiLU*up / RoPE / residual epilogue), so its time is a lower
bound for what a fused kernel built on it would need."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tcavt_amd import capi, ops

capi.init(0)
dev = torch.device("cuda:0")
M = 8192


def timeit(fn, n=32, warm=4):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, N, K in (("qkv", 3072, 2048), ("o", 2048, 2048), ("gateup", 16384, 2048), ("down", 2048, 8192)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(16)]
    out = torch.empty(M, N // 2 if name == "gateup" else N, dtype=torch.bfloat16, device=dev)
    kw = dict(silu_mul=True) if name == "gateup" else {}
    for rotate in (False, True):
        ms_lib = timeit(lambda i: F.linear(a, ws[i % 16 if rotate else 0]))
        ms_own = timeit(lambda i: ops.gemm_bf16(a, ws[i % 16 if rotate else 0], out=out, **kw))
        fl = 2.0 * M * N * K
        print(f"{name:7s} M={M} N={N} K={K} rotate={rotate}: hipBLASLt {ms_lib*1e3:7.1f} us {fl/ms_lib/1e9:7.1f} TF | "
              f"tcavt {ms_own*1e3:7.1f} us {fl/ms_own/1e9:7.1f} TF", flush=True)
