"""Tile-form A/B of the four decoder projections (in-model forms, fp16, rotating weights) at SMALL M (BASELINE config 4's low end:
M = B * L = 1024, 2048): tile 0 (auto: includes the two-launch split K when a workspace is lent), 64, 128, 257.  Measurement only.
usage: python tools/ab_tiles_small.py [M ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops

capi.init(0)
dev = torch.device("cuda:0")
dt = torch.float16
H, I, NQKV = 2048, 8192, 3072


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


def run(M):
    x = (torch.randn(M, H, device=dev) * 0.05).to(dt)
    part = torch.rand(M, H // 64, device=dev) + 0.5
    h16 = torch.zeros(M, H, dtype=dt, device=dev)
    pout = torch.empty(M, H // 64, device=dev)
    skws = torch.zeros((16 << 10) + 8 * M * H * 4, dtype=torch.uint8, device=dev)
    rs = dict(rowscale_part=part, rowscale_npart=H // 64, rowscale_h=H, rowscale_eps=1e-5)
    tiles = [0, 64, 128] + ([257] if M % 256 == 0 else [])
    shapes = {}
    w_gu = [(torch.randn(2 * I, H, device=dev) * 0.02).to(dt) for _ in range(10)]
    act = torch.empty(M, I, dtype=dt, device=dev)
    shapes["gateup"] = lambda t, i: raw(x, w_gu[i % 10], act, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, t, **rs)
    w_q = [(torch.randn(NQKV, H, device=dev) * 0.02).to(dt) for _ in range(10)]
    qkv = torch.empty(M, NQKV, dtype=dt, device=dev)
    cos, sin = torch.rand(256, 32, device=dev), torch.rand(256, 32, device=dev)
    shapes["qkv"] = lambda t, i: raw(x, w_q[i % 10], qkv, capi.EPI_ROPE | capi.EPI_ROWSCALE, t, rope_cos=cos, rope_sin=sin, rope_L=128,
                                     rope_cols=2560, **rs)
    w_o = [(torch.randn(H, H, device=dev) * 0.02).to(dt) for _ in range(10)]
    shapes["o"] = lambda t, i: raw(x, w_o[i % 10], None, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, t, ldc=H, norm_h16=h16, norm_part=pout)
    shapes["o+ws"] = lambda t, i: raw(x, w_o[i % 10], None, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, t, ldc=H, norm_h16=h16, norm_part=pout,
                                      splitk_ws=skws, splitk_ws_bytes=skws.numel())
    a_d = (torch.randn(M, I, device=dev) * 0.05).to(dt)
    w_d = [(torch.randn(H, I, device=dev) * 0.02).to(dt) for _ in range(10)]
    shapes["down"] = lambda t, i: raw(a_d, w_d[i % 10], None, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, t, ldc=H, norm_h16=h16, norm_part=pout)
    shapes["down+ws"] = lambda t, i: raw(a_d, w_d[i % 10], None, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, t, ldc=H, norm_h16=h16, norm_part=pout,
                                         splitk_ws=skws, splitk_ws_bytes=skws.numel())
    for name, fn in shapes.items():
        line = f"M={M:5d} {name:8s}"
        for t in tiles:
            if name.endswith("+ws") and t != 0:
                continue
            try:
                us = timeit(lambda i: fn(t, i))
                line += f" | tile {t:3d}: {us:6.1f} us"
            except Exception as e:  # (a form the tile code does not have)
                line += f" | tile {t:3d}: n/a"
                capi.lib()  # keep going
        print(line, flush=True)


for M in [int(a) for a in sys.argv[1:]] or [1024, 2048]:
    run(M)
