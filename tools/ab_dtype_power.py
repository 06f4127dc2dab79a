"""gate|up GEMM (M=8192, N=16384, K=2048): bf16 vs fp16 operands, random vs all-zero data, plain SiLU epilogue vs the fused
row scale -- how much of the time is the data-dependent clock (power) and how much the instruction stream."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops

capi.init(0)
dev = torch.device("cuda:0")
M, N, K = 8192, 16384, 2048


def timeit(fn, n=30, warm=8):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


part = torch.rand(M, K // 64, device=dev) + 0.5
for rep in range(2):
    for dt in (torch.bfloat16, torch.float16):
        for data in ("randn", "zeros", "small"):
            if data == "randn":
                x = torch.randn(M, K, device=dev).to(dt)
                ws = [(torch.randn(N, K, device=dev) * 0.02).to(dt) for _ in range(6)]
            elif data == "small":  # few significant bits: values from {-1, 0, 1} * 2^-3
                x = (torch.randint(-1, 2, (M, K), device=dev).float() * 0.125).to(dt)
                ws = [(torch.randint(-1, 2, (N, K), device=dev).float() * 0.125).to(dt) for _ in range(6)]
            else:
                x = torch.zeros(M, K, device=dev, dtype=dt)
                ws = [torch.zeros(N, K, device=dev, dtype=dt) for _ in range(6)]
            out = torch.empty(M, N // 2, dtype=dt, device=dev)
            for fused in (False, True):
                def run(i):
                    g = capi.GemmArgs()
                    w = ws[i % 6]
                    g.A, g.lda, g.W, g.ldw, g.C, g.ldc = x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N // 2
                    g.M, g.N, g.K, g.tile = M, N, K, 0
                    g.in_dtype = g.out_dtype = ops._DT[dt]
                    g.epilogue = capi.EPI_SILU_MUL | (capi.EPI_ROWSCALE if fused else 0)
                    if fused:
                        g.rowscale_part, g.rowscale_npart, g.rowscale_h, g.rowscale_eps = part.data_ptr(), K // 64, K, 1e-5
                    capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(g), capi.stream_ptr()), "gemm")
                us = timeit(run)
                print(f"rep {rep} {str(dt)[6:]:9s} {data:6s} {'rowscale' if fused else 'plain   '}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)
            del x, ws, out
