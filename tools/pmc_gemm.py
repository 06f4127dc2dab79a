"""Run only the gate|up GEMM a few times: target for rocprofv3 --pmc passes.
   argv[1] = tile code for tcavt_gemm_bf16 (0 = auto), or "lib" for the vendor library (torch F.linear -> hipBLASLt,
   plain GEMM, measurement only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops
capi.init(0)
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "0"
shape = sys.argv[2] if len(sys.argv) > 2 else "gateup"
if shape == "attn":  # the decoder attention kernel on the bench shape: B=32, L=256, 32 q heads / 8 kv heads x 64
    B, L, nq, nkv = 32, 256, 32, 8
    qkv = torch.randn(B * L, (nq + 2 * nkv) * 64, device=dev).to(torch.bfloat16)
    out = torch.empty(B * L, nq * 64, dtype=torch.bfloat16, device=dev)
    lens = torch.randint(144, 257, (B,), device=dev, dtype=torch.int32)
    for i in range(8):
        ops.attn_causal_gqa(qkv, out, lens, B, L, nq, nkv, 0.125)
    torch.cuda.synchronize()
    print("done")
    sys.exit(0)
M, N, K = {"gateup": (8192, 16384, 2048), "down": (8192, 2048, 8192), "o": (8192, 2048, 2048)}[shape]
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(4)]
out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev) if shape == "gateup" else torch.randn(M, N, device=dev)
for i in range(8):
    if which == "lib":
        torch.nn.functional.linear(a, ws[i % 4])
    else:
        if shape == "gateup":
            ops.gemm_bf16(a, ws[i % 4], out=out, tile=int(which), silu_mul=True)
        else:
            ops.gemm_bf16(a, ws[i % 4], out=out, residual=out, tile=int(which))
torch.cuda.synchronize()
print("done")
