"""A/B of the 4-wave GEMM kernel (tile codes 257-259) against the 8-wave production kernel (256): bit-equality of
the results on every epilogue, then timing on the decoder shapes with rotating (HBM-cold) weights."""
import os
import sys

os.environ.setdefault("TCAVT_LIB", "exp")  # the -DTCAVT_EXPERIMENTS build: python -m tcavt_amd.build --experiments
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops, rope as rope_mod

capi.init(0)
dev = torch.device("cuda:0")
codes = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [257, 258, 259]
M = 8192


def check():
    g = torch.Generator(device="cpu").manual_seed(1)
    Mc, N, K = 512, 768, 448
    a = torch.randn(Mc, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(Mc, N, generator=g).to(dev)
    pos = torch.arange(128, dtype=torch.float32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))
    ang = pos[:, None] * inv[None, :]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    for c in codes:
        a2 = torch.randn(Mc, 64, generator=g).to(torch.bfloat16).to(dev)
        w2 = (torch.randn(N, 64, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        for kw in (dict(), dict(bias=bias, relu=True, residual=res), dict(silu_mul=True), dict(rope=(cos, sin, 512)),
                   dict(rope=(cos, sin, 512), a2=a2, w2=w2)):
            dt = torch.bfloat16 if "rope" in kw else torch.float32
            r0 = ops.gemm_bf16(a, w, out_dtype=dt, tile=256, **kw)
            r1 = ops.gemm_bf16(a, w, out_dtype=dt, tile=c, **kw)
            torch.cuda.synchronize()
            ok = torch.equal(r0, r1)
            print(f"check tile={c} {sorted(kw)}: {'bit-equal' if ok else 'MISMATCH max %.3e' % (r0 - r1).abs().max().item()}", flush=True)
            if not ok:
                bad = (r0 != r1).nonzero()
                print("  mismatching elements:", bad.shape[0], "rows", bad[:, 0].min().item(), "-", bad[:, 0].max().item(),
                      "cols", bad[:, 1].min().item(), "-", bad[:, 1].max().item(), "first", bad[:8].tolist(), flush=True)
                global failed
                failed = True
    # full-size shapes, repeated (races are timing dependent)
    for c in codes:
        for name, N, K in (("gateup", 16384, 2048), ("down", 2048, 8192), ("n3072", 3072, 2048)):
            if c == 271 and N % 192:
                continue
            a = torch.randn(M, K, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
            kw = dict(silu_mul=True) if name == "gateup" else {}
            r0 = ops.gemm_bf16(a, w, out_dtype=torch.bfloat16, tile=256, **kw)
            nbad = 0
            for rep in range(6):
                r1 = ops.gemm_bf16(a, w, out_dtype=torch.bfloat16, tile=c, **kw)
                nbad += int((r0 != r1).sum().item())
            print(f"check tile={c} full-size {name} x6: {'bit-equal' if nbad == 0 else 'MISMATCH in %d elements' % nbad}", flush=True)
            if nbad:
                failed = True


def timeit(fn, n=32, warm=4):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


failed = False
if os.environ.get("TCAVT_AB_NOCHECK", "0") != "1":  # (timing-only experiment codes compute wrong results: skip the equality check)
    check()
if failed:
    sys.exit(1)
DT = torch.float16 if os.environ.get("TCAVT_AB_DTYPE", "bf16") == "fp16" else torch.bfloat16
ROUNDS = int(os.environ.get("TCAVT_AB_ROUNDS", "1"))
for name, N, K in (("qkv", 3072, 2048), ("o", 2048, 2048), ("gateup", 16384, 2048), ("down", 2048, 8192)):
    a = torch.randn(M, K, device=dev).to(DT)
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(DT) for _ in range(16)]
    out = torch.empty(M, N // 2 if name == "gateup" else N, dtype=DT, device=dev)
    kw = dict(silu_mul=True) if name == "gateup" else {}
    if name == "qkv":
        pos = torch.arange(256, dtype=torch.float32)
        inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))
        ang = pos[:, None] * inv[None, :]
        kw = dict(rope=(ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev), 2560),
                  a2=torch.randn(M, 64, device=dev).to(DT), w2=(torch.randn(N, 64, device=dev) * 0.02).to(DT))
    for rnd in range(ROUNDS):  # (interleaved rounds in one process: the chip's clock state drifts, rank by the distribution)
        line = f"{name:7s}"
        for tile in [256] + codes:
            if (tile == 271 and N % 192) or (tile == 273 and name == "qkv"):
                line += f" | {tile}: n/a"
                continue
            ms = timeit(lambda i: ops.gemm_bf16(a, ws[i % 16], out=out, tile=tile, **kw))
            line += f" | {tile}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF"
        print(line, flush=True)
