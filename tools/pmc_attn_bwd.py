"""Driver for rocprofv3 --pmc passes over the two attention-backward kernels at the decoder's shape (tools/pmc_attn_bwd.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi
from tcavt_amd.config import LlamaShape
from tcavt_amd.llm_backward import attn_bwd_composed
from tcavt_amd.rope import rope_tables

capi.init(0)
dev = torch.device("cuda:0")
B, T, nq, nkv = 32, 256, 32, 8
ncols, M = (nq + 2 * nkv) * 64, B * T
g = torch.Generator(device="cpu").manual_seed(1)
qkv = torch.zeros(M + 64, ncols, dtype=torch.bfloat16, device=dev)
qkv[:M] = torch.randn(M, ncols, generator=g).to(torch.bfloat16).to(dev)
dO = torch.randn(M, nq * 64, generator=g).to(torch.bfloat16).to(dev)
kv_len = torch.randint(144, 257, (B,), generator=g).to(torch.int32).to(dev)
cos, sin = (t.to(dev) for t in rope_tables(LlamaShape(), T))
pool = {}


def buf(name, shape, dtype, zero=False):
    key = (name, tuple(shape), dtype)
    if key not in pool:
        pool[key] = torch.zeros(shape, dtype=dtype, device=dev)
    return pool[key]


out = torch.empty(M, ncols, dtype=torch.bfloat16, device=dev)
for _ in range(8):
    attn_bwd_composed(buf, qkv, dO, kv_len, B, T, nq, nkv, 0.125, cos, sin, out)
torch.cuda.synchronize()
