"""Tile-form A/B of the four decoder projections in their fused-norm forms (fp16, rotating weights), one box, one process:
tile codes 0 (auto), 257 (4-wave 256x256), 271 (256x192), 272 (two-barrier deep-prefetch), 256 (8-wave)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops

capi.init(0)
dev = torch.device("cuda:0")
dt = torch.float16
M, H, I, NQKV = 8192, 2048, 8192, 3072


def timeit(fn, n=40, warm=10):
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


def raw(a, w, out, epi, tile, **kw):
    g = capi.GemmArgs()
    g.A, g.lda, g.W, g.ldw = a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0)
    g.C, g.ldc = (None, kw.pop("ldc")) if out is None else (out.data_ptr(), out.stride(0))
    g.M, g.N, g.K, g.tile = a.shape[0], w.shape[0], a.shape[1], tile
    g.in_dtype, g.out_dtype, g.epilogue = ops._DT[a.dtype], capi.F32 if out is None else ops._DT[out.dtype], epi
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(g), capi.stream_ptr()), "gemm")


x = (torch.randn(M, H, device=dev) * 0.05).to(dt)  # (small: the in-place 16-bit stream accumulates over the timing loop)
part = torch.rand(M, H // 64, device=dev) + 0.5
h = torch.randn(M, H, device=dev)
h16 = torch.empty(M, H, dtype=dt, device=dev)
pout = torch.empty(M, H // 64, device=dev)
rs = dict(rowscale_part=part, rowscale_npart=H // 64, rowscale_h=H, rowscale_eps=1e-5)
res = {}
w = [(torch.randn(2 * I, H, device=dev) * 0.02).to(dt) for _ in range(10)]
act = torch.empty(M, I, dtype=dt, device=dev)
for t in (0, 257, 272, 256):
    res[("gateup", t)] = timeit(lambda i: raw(x, w[i % 10], act, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, t, **rs))
del w
w = [(torch.randn(NQKV, H, device=dev) * 0.02).to(dt) for _ in range(10)]
qkv = torch.empty(M, NQKV, dtype=dt, device=dev)
cos, sin = torch.rand(256, 32, device=dev), torch.rand(256, 32, device=dev)
tt = torch.randn(M, 64, device=dev).to(dt)
b_ext = (torch.randn(NQKV, 64, device=dev) * 0.02).to(dt)
kw = dict(A2=tt, lda2=64, W2=b_ext, ldw2=64, K2=64, rope_cos=cos, rope_sin=sin, rope_L=256, rope_cols=2560, **rs)
for t in (0, 257, 271, 256):
    res[("qkv", t)] = timeit(lambda i: raw(x, w[i % 10], qkv, capi.EPI_ROPE | capi.EPI_ROWSCALE, t, **kw))
del w
w = [(torch.randn(H, H, device=dev) * 0.02).to(dt) for _ in range(10)]
for t in (0, 257, 272, 256, 128):
    res[("o", t)] = timeit(lambda i: raw(x, w[i % 10], h, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, t, residual=h, ldr=H, norm_h16=h16, norm_part=pout))
    # 16-bit residual stream (C == NULL): in place on h16
    res[("o16", t)] = timeit(lambda i: raw(x, w[i % 10], None, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, t, ldc=H, norm_h16=h16, norm_part=pout))
del w
a_d = torch.randn(M, I, device=dev).to(dt)
w = [(torch.randn(H, I, device=dev) * 0.02).to(dt) for _ in range(10)]
for t in (0, 257, 272, 256):
    res[("down", t)] = timeit(lambda i: raw(a_d, w[i % 10], h, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, t, residual=h, ldr=H, norm_h16=h16, norm_part=pout))
    res[("down16", t)] = timeit(lambda i: raw(a_d, w[i % 10], None, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, t, ldc=H, norm_h16=h16, norm_part=pout))
for k, v in res.items():
    print(f"{k[0]:8s} tile {k[1]:3d}: {v:7.1f} us", flush=True)
