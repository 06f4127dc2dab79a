"""Run only the gate|up GEMM (2-buffer 256x256 kernel) a few times: target for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops
capi.init(0)
dev = torch.device("cuda:0")
M, N, K = 8192, 16384, 2048
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 255
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
for _ in range(6):
    ops.gemm_bf16(a, w, out=out, tile=tile, silu_mul=True)
torch.cuda.synchronize()
print("done")
