"""A/B of the decoder GEMMs' epilogues ACROSS TWO BUILDS in one process, interleaved on one device: the library under test
(libtcavt_hip.so) against a baseline build (tools/ab/libtcavt_hip_old.so, built from an earlier commit with the same ABI).
The four projections run in their in-model forms (fp16, fused-norm epilogues, rotating weights); outputs of the two builds
are also compared bit for bit (the 16-byte epilogues change instruction selection, not arithmetic)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tcavt_amd import capi, ops

capi.init(0)
dev = torch.device("cuda:0")
new = capi.lib()
old = ctypes.CDLL(os.path.join(ROOT, "tools", "ab", "libtcavt_hip_old.so"))
old.tcavt_last_error.restype = ctypes.c_char_p
n = ctypes.c_int(0)
assert old.tcavt_init(0, ctypes.byref(n)) == 0 and old.tcavt_abi_version() == capi.ABI_VERSION
old.tcavt_gemm_bf16.argtypes = [ctypes.POINTER(capi.GemmArgs), ctypes.c_void_p]
dt = torch.float16
M, H, I, NQKV = 8192, 2048, 8192, 3072
if len(sys.argv) > 1:
    M = int(sys.argv[1])


def args(a, w, out, epi, tile, **kw):
    g = capi.GemmArgs()
    g.A, g.lda, g.W, g.ldw = a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0)
    g.C, g.ldc = (None, kw.pop("ldc")) if out is None else (out.data_ptr(), out.stride(0))
    g.M, g.N, g.K, g.tile = a.shape[0], w.shape[0], a.shape[1], tile
    g.in_dtype, g.out_dtype, g.epilogue = ops._DT[a.dtype], capi.F32 if out is None else ops._DT[out.dtype], epi
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g


def call(lib, g):
    rc = lib.tcavt_gemm_bf16(ctypes.byref(g), capi.stream_ptr())
    if rc != 0:
        raise RuntimeError(lib.tcavt_last_error().decode())


def ab(name, make, nrep=6, per=20):
    """make(i) -> GemmArgs for weight set i.  Interleaved rounds: old x per, new x per, ..."""
    res = {"old": [], "new": []}
    for lib in (old, new):
        for i in range(5):
            call(lib, make(i))
    torch.cuda.synchronize()
    for r in range(nrep):
        for tag, lib in (("old", old), ("new", new)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(per):
                call(lib, make(i))
            b.record()
            torch.cuda.synchronize()
            res[tag].append(a.elapsed_time(b) / per * 1e3)
    o, n_ = sorted(res["old"]), sorted(res["new"])
    print(f"{name:10s} old median {o[len(o) // 2]:7.1f} us (min {o[0]:7.1f})   new median {n_[len(n_) // 2]:7.1f} us (min {n_[0]:7.1f})   "
          f"delta {n_[len(n_) // 2] - o[len(o) // 2]:+6.1f} us", flush=True)


x = (torch.randn(M, H, device=dev) * 0.05).to(dt)
part = torch.rand(M, H // 64, device=dev) + 0.5
rs = dict(rowscale_part=part, rowscale_npart=H // 64, rowscale_h=H, rowscale_eps=1e-5)
NW = 10

# ---- gate|up
w = [(torch.randn(2 * I, H, device=dev) * 0.02).to(dt) for _ in range(NW)]
act_o, act_n = torch.empty(M, I, dtype=dt, device=dev), torch.empty(M, I, dtype=dt, device=dev)
call(old, args(x, w[0], act_o, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, 0, **rs))
call(new, args(x, w[0], act_n, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, 0, **rs))
torch.cuda.synchronize()
print("gate|up outputs bit-equal:", torch.equal(act_o, act_n), flush=True)
ab("gate|up", lambda i: args(x, w[i % NW], act_n, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, 0, **rs))
del w, act_o
# ---- q|k|v with RoPE + LoRA second K source
w = [(torch.randn(NQKV, H, device=dev) * 0.02).to(dt) for _ in range(NW)]
q_o, q_n = torch.empty(M, NQKV, dtype=dt, device=dev), torch.empty(M, NQKV, dtype=dt, device=dev)
cos, sin = torch.rand(256, 32, device=dev), torch.rand(256, 32, device=dev)
tt = torch.randn(M, 64, device=dev).to(dt)
b_ext = (torch.randn(NQKV, 64, device=dev) * 0.02).to(dt)
kw = dict(A2=tt, lda2=64, W2=b_ext, ldw2=64, K2=64, rope_cos=cos, rope_sin=sin, rope_L=256, rope_cols=2560, **rs)
call(old, args(x, w[0], q_o, capi.EPI_ROPE | capi.EPI_ROWSCALE, 0, **kw))
call(new, args(x, w[0], q_n, capi.EPI_ROPE | capi.EPI_ROWSCALE, 0, **kw))
torch.cuda.synchronize()
print("q|k|v outputs bit-equal:", torch.equal(q_o, q_n), flush=True)
ab("q|k|v", lambda i: args(x, w[i % NW], q_n, capi.EPI_ROPE | capi.EPI_ROWSCALE, 0, **kw))
del w, q_o
# ---- o / down: in-place 16-bit residual stream + partial sums
pout_o, pout_n = torch.zeros(M, H // 64, device=dev), torch.zeros(M, H // 64, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
for name, K in (("o", H), ("down", I)):
    a_ = torch.randn(M, K, device=dev).to(dt) * 0.05
    w = [(torch.randn(H, K, device=dev) * 0.02).to(dt) for _ in range(NW)]
    h0 = (torch.randn(M, H, device=dev)).to(dt)
    h_o, h_n = h0.clone(), h0.clone()
    mk = lambda i, h16, po: args(a_, w[i % NW], None, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, 0, ldc=H, norm_h16=h16, norm_part=po,
                                 nonfinite_flag=flag, nonfinite_tag=1)
    call(old, mk(0, h_o, pout_o))
    call(new, mk(0, h_n, pout_n))
    torch.cuda.synchronize()
    print(f"{name} stream bit-equal: {torch.equal(h_o, h_n)}, partial sums bit-equal: {torch.equal(pout_o, pout_n)}", flush=True)
    h_n.zero_()  # (the timing loop accumulates in place: start from zero, small products)
    ab(name, lambda i: mk(i, h_n, pout_n))
    del w, a_
print("flag", flag.item())
