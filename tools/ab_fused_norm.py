"""A/B of the fused-RMSNorm epilogues on the decoder's shapes (one box, one process): the four projections with and
without TCAVT_EPI_ROWSCALE / TCAVT_EPI_NORM_OUT, the stand-alone RMSNorm kernel they replace, rotating (HBM-cold) weights."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes

import torch
from tcavt_amd import capi, ops

capi.init(0)
dev = torch.device("cuda:0")
dt = torch.float16 if "--bf16" not in sys.argv else torch.bfloat16
M, H, I, NQKV = 8192, 2048, 8192, 3072


def timeit(fn, n=48, warm=8):
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


def raw_gemm(a, w, out, epi, **kw):
    g = capi.GemmArgs()
    g.A, g.lda, g.W, g.ldw, g.C, g.ldc = a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0)
    g.M, g.N, g.K = a.shape[0], w.shape[0], a.shape[1]
    g.in_dtype = ops._DT[a.dtype]
    g.out_dtype = ops._DT[out.dtype]
    g.epilogue = epi
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(g), capi.stream_ptr()), "gemm")


x = (torch.randn(M, H, device=dev)).to(dt)
part = torch.rand(M, H // 64, device=dev) + 0.5
h = torch.randn(M, H, device=dev)
h16 = torch.empty(M, H, dtype=dt, device=dev)
part_out = torch.empty(M, H // 64, device=dev)
cos = torch.rand(256, 32, device=dev)
sin = torch.rand(256, 32, device=dev)
t = torch.randn(M, 64, device=dev).to(dt)
rows = []

w_gu = [(torch.randn(2 * I, H, device=dev) * 0.02).to(dt) for _ in range(12)]
act = torch.empty(M, I, dtype=dt, device=dev)
rows.append(("gate|up SiLU", timeit(lambda i: raw_gemm(x, w_gu[i % 12], act, capi.EPI_SILU_MUL)),
             timeit(lambda i: raw_gemm(x, w_gu[i % 12], act, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, rowscale_part=part,
                                       rowscale_npart=H // 64, rowscale_h=H, rowscale_eps=1e-5))))
del w_gu
w_qkv = [(torch.randn(NQKV, H, device=dev) * 0.02).to(dt) for _ in range(12)]
b_ext = (torch.randn(NQKV, 64, device=dev) * 0.02).to(dt)
qkv = torch.empty(M, NQKV, dtype=dt, device=dev)
kw = dict(A2=t, lda2=64, W2=b_ext, ldw2=64, K2=64, rope_cos=cos, rope_sin=sin, rope_L=256, rope_cols=2560)
rows.append(("q|k|v + LoRA + RoPE", timeit(lambda i: raw_gemm(x, w_qkv[i % 12], qkv, capi.EPI_ROPE, **kw)),
             timeit(lambda i: raw_gemm(x, w_qkv[i % 12], qkv, capi.EPI_ROPE | capi.EPI_ROWSCALE, rowscale_part=part,
                                       rowscale_npart=H // 64, rowscale_h=H, rowscale_eps=1e-5, **kw))))
del w_qkv
w_o = [(torch.randn(H, H, device=dev) * 0.02).to(dt) for _ in range(12)]
rows.append(("o_proj + residual", timeit(lambda i: raw_gemm(x, w_o[i % 12], h, capi.EPI_RESIDUAL, residual=h, ldr=H)),
             timeit(lambda i: raw_gemm(x, w_o[i % 12], h, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, residual=h, ldr=H,
                                       norm_h16=h16, norm_part=part_out))))
del w_o
a_d = torch.randn(M, I, device=dev).to(dt)
w_d = [(torch.randn(H, I, device=dev) * 0.02).to(dt) for _ in range(12)]
rows.append(("down_proj + residual", timeit(lambda i: raw_gemm(a_d, w_d[i % 12], h, capi.EPI_RESIDUAL, residual=h, ldr=H)),
             timeit(lambda i: raw_gemm(a_d, w_d[i % 12], h, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, residual=h, ldr=H,
                                       norm_h16=h16, norm_part=part_out))))
gamma = torch.ones(H, device=dev)
xn = torch.empty(M, H, dtype=dt, device=dev)
rms = timeit(lambda i: ops.rmsnorm(h, gamma, 1e-5, out_bf16=xn))
print(f"storage {dt}; stand-alone rmsnorm kernel: {rms:.1f} us (x2 per layer, + a launch gap each)")
tot_a = tot_b = 0.0
for name, a, b in rows:
    print(f"{name:24s} plain {a:7.1f} us   fused-norm form {b:7.1f} us   ({b - a:+.1f})")
    tot_a += a
    tot_b += b
print(f"per layer: plain GEMMs + 2 rmsnorm = {tot_a + 2 * rms:.1f} us, fused = {tot_b:.1f} us ({tot_b - tot_a - 2 * rms:+.1f} us before launch gaps)")
