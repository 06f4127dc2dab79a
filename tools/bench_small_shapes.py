import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops
capi.init(0)
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K, dt in ((960, 64, 2048, torch.float32), (960, 64, 2048, torch.bfloat16), (8192, 64, 2048, torch.bfloat16), (960, 2048, 64, torch.bfloat16), (960, 2048, 2048, torch.bfloat16)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=dt, device=dev)
    for tile in (0, 64, 128):
        print(M, N, K, dt, "tile", tile, round(t(lambda: ops.gemm_bf16(a, w, out=out, tile=tile)), 1), "us", flush=True)
