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
    qkv = (torch.randn(B * L, (nq + 2 * nkv) * 64, device=dev) * 0.5).to(torch.float16)  # (the production storage type)
    out = torch.empty(B * L, nq * 64, dtype=torch.float16, device=dev)
    lens = torch.randint(144, 257, (B,), device=dev, dtype=torch.int32)
    for i in range(8):
        ops.attn_causal_gqa(qkv, out, lens, B, L, nq, nkv, 0.125)
    torch.cuda.synchronize()
    print("done")
    sys.exit(0)
M, N, K = {"gateup": (8192, 16384, 2048), "down": (8192, 2048, 8192), "o": (8192, 2048, 2048)}[shape]
# round 2: the decoder's forms -- fp16 operands, gate|up with the fused-RMSNorm row scale, o / down with the NORM_OUT
# epilogue on the 16-bit residual stream (C == NULL: in place on h16, + partial sums of squares); argv[3:]: "res32" = the
# fp32-stream NORM_OUT form (LoRA-trainable variant), "plain" = round 1's forms, "bf16" = bf16 operands
import ctypes
dt = torch.bfloat16 if "bf16" in sys.argv[3:] else torch.float16
plain = "plain" in sys.argv[3:]
res32 = "res32" in sys.argv[3:] or dt == torch.bfloat16
a = (torch.randn(M, K, device=dev) * (0.2 if shape != "gateup" else 1.0)).to(dt)
ws = [(torch.randn(N, K, device=dev) * 0.02).to(dt) for _ in range(4)]
out = torch.empty(M, N // 2, dtype=dt, device=dev) if shape == "gateup" else torch.randn(M, N, device=dev)
part = torch.rand(M, K // 64, device=dev) + 0.5
h16 = torch.randn(M, N, device=dev).to(dt)
pout = torch.empty(M, N // 64, device=dev)


def raw(w, epi, **kw):
    g = capi.GemmArgs()
    g.A, g.lda, g.W, g.ldw, g.C, g.ldc = a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), out.stride(0)
    if kw.pop("stream16", False):
        g.C = None
    g.M, g.N, g.K, g.tile, g.epilogue = M, N, K, int(which), epi
    g.in_dtype, g.out_dtype = ops._DT[dt], ops._DT[out.dtype]
    for k_, v_ in kw.items():
        setattr(g, k_, v_.data_ptr() if torch.is_tensor(v_) else v_)
    capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(g), capi.stream_ptr()), "gemm")


for i in range(8):
    if which == "lib":
        torch.nn.functional.linear(a, ws[i % 4])
    elif shape == "gateup":
        if plain:
            raw(ws[i % 4], capi.EPI_SILU_MUL)
        else:
            raw(ws[i % 4], capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, rowscale_part=part, rowscale_npart=K // 64, rowscale_h=K, rowscale_eps=1e-5)
    elif plain:
        raw(ws[i % 4], capi.EPI_RESIDUAL, residual=out, ldr=N)
    elif res32:
        raw(ws[i % 4], capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, residual=out, ldr=N, norm_h16=h16, norm_part=pout)
    else:
        raw(ws[i % 4], capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, stream16=True, norm_h16=h16, norm_part=pout)
torch.cuda.synchronize()
print("done")
