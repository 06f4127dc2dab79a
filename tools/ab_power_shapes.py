"""The four decoder GEMMs (in their in-model forms, fp16) on random and on all-zero operands: how much of each launch is the
data-dependent clock (power) and how much the instruction stream."""
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


def raw(a, w, out, epi, **kw):
    g = capi.GemmArgs()
    g.A, g.lda, g.W, g.ldw = a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0)
    g.C, g.ldc = (None, kw.pop("ldc")) if out is None else (out.data_ptr(), out.stride(0))
    g.M, g.N, g.K, g.tile = a.shape[0], w.shape[0], a.shape[1], 0
    g.in_dtype, g.out_dtype, g.epilogue = ops._DT[a.dtype], capi.F32 if out is None else ops._DT[out.dtype], epi
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(g), capi.stream_ptr()), "gemm")


part = torch.rand(M, H // 64, device=dev) + 0.5
pout = torch.empty(M, H // 64, device=dev)
rs = dict(rowscale_part=part, rowscale_npart=H // 64, rowscale_h=H, rowscale_eps=1e-5)
cos, sin = torch.rand(256, 32, device=dev), torch.rand(256, 32, device=dev)
for rep in range(2):
    for data in ("randn", "zeros"):
        mk = (lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(dt)) if data == "randn" else (lambda *s, sc=1.0: torch.zeros(*s, device=dev, dtype=dt))
        x = mk(M, H, sc=0.05)
        act_in = mk(M, I, sc=0.05)
        res = {}
        w = [mk(2 * I, H, sc=0.02) for _ in range(4)]
        act = torch.empty(M, I, dtype=dt, device=dev)
        res["gateup"] = timeit(lambda i: raw(x, w[i % 4], act, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, **rs))
        del w
        w = [mk(NQKV, H, sc=0.02) for _ in range(4)]
        qkv = torch.empty(M, NQKV, dtype=dt, device=dev)
        tt, b_ext = mk(M, 64), mk(NQKV, 64, sc=0.02)
        kw = dict(A2=tt, lda2=64, W2=b_ext, ldw2=64, K2=64, rope_cos=cos, rope_sin=sin, rope_L=256, rope_cols=2560, **rs)
        res["qkv"] = timeit(lambda i: raw(x, w[i % 4], qkv, capi.EPI_ROPE | capi.EPI_ROWSCALE, **kw))
        del w
        w = [mk(H, H, sc=0.02) for _ in range(4)]
        h16 = mk(M, H)
        res["o"] = timeit(lambda i: raw(x, w[i % 4], None, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, ldc=H, norm_h16=h16, norm_part=pout))
        del w
        w = [mk(H, I, sc=0.02) for _ in range(4)]
        res["down"] = timeit(lambda i: raw(act_in, w[i % 4], None, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, ldc=H, norm_h16=h16, norm_part=pout))
        del w
        fl = {"gateup": 2.0 * M * 2 * I * H, "qkv": 2.0 * M * NQKV * (H + 64), "o": 2.0 * M * H * H, "down": 2.0 * M * H * I}
        print(f"rep {rep} {data:6s}: " + "  ".join(f"{k} {v:6.1f} us ({fl[k] / v / 1e6 / 2500:.2f})" for k, v in res.items()), flush=True)
