"""Timing of the forms of the causal GQA attention backward at the decoder's shape (B=32, T=256, 32 query / 8 key-value heads):
the two-kernel MFMA form (product path), its GEMM-composed cross-checks and the scalar cross-check kernel, plus their agreement."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops
from tcavt_amd.config import LlamaShape
from tcavt_amd.llm_backward import attn_bwd_composed
from tcavt_amd.rope import rope_tables

capi.init(0)
dev = torch.device("cuda:0")
B, T, nq, nkv = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 256, 32, 8)
ncols, M = (nq + 2 * nkv) * 64, B * T
g = torch.Generator(device="cpu").manual_seed(1)
qkv = torch.zeros(M + 64, ncols, dtype=torch.bfloat16, device=dev)
qkv[:M] = torch.randn(M, ncols, generator=g).to(torch.bfloat16).to(dev)
dO = torch.randn(M, nq * 64, generator=g).to(torch.bfloat16).to(dev)
kv_len = torch.full((B,), T, dtype=torch.int32, device=dev)
kv_len[::3] = max(1, T - 37)
cos, sin = (t.to(dev) for t in rope_tables(LlamaShape(), T))
pool = {}


def buf(name, shape, dtype, zero=False):
    key = (name, tuple(shape), dtype)
    if key not in pool:
        pool[key] = torch.zeros(shape, dtype=dtype, device=dev)
    return pool[key]


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


outs = {}
for form in ("fused", "scores+gemm", "gemm"):
    out = torch.empty(M, ncols, dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: attn_bwd_composed(buf, qkv, dO, kv_len, B, T, nq, nkv, 0.125, cos, sin, out, scores=form))
    outs[form] = out.float()
    print(f"{form:12s} {us:8.1f} us", flush=True)
if T <= 280:
    g32 = torch.zeros(M, ncols, dtype=torch.float32, device=dev)
    out = torch.empty(M, ncols, dtype=torch.bfloat16, device=dev)

    def scalar():
        g32.zero_()
        ops.attn_causal_gqa_bwd(qkv[:M], dO, g32, kv_len, B, T, nq, nkv, 0.125)
        ops.rope_bwd_pack(g32, out, cos, sin, (nq + nkv) * 64, T)

    us = timeit(scalar, n=3, warm=1)
    outs["scalar"] = out.float()
    print(f"{'scalar':12s} {us:8.1f} us", flush=True)
ref = outs.get("scalar", outs["gemm"])
for k, v in outs.items():
    print(f"{k:12s} vs reference form: rel {((v - ref).norm() / ref.norm()).item():.2e}")
