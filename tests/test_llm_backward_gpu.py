"""Decoder-layer backward pieces for the LoRA-trainable variant (modify_scripts/modify_train.py:512-528)
against torch autograd of the same fp32 expressions (floating-point kernels: tolerance stated per test)."""
import math

import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


def test_silu_mul_bwd(gpu):
    from tcavt_amd import ops
    from tcavt_amd.layout import interleave_gate_up

    dev = gpu["device"]
    g = torch.Generator().manual_seed(3)
    M, I = 37, 128
    gate = torch.randn(M, I, generator=g).to(torch.bfloat16).float().requires_grad_(True)
    up = torch.randn(M, I, generator=g).to(torch.bfloat16).float().requires_grad_(True)
    d = torch.randn(M, I, generator=g).to(torch.bfloat16)
    (torch.nn.functional.silu(gate) * up).backward(d.float())
    # the interleave acts on the feature axis (weight rows = output columns)
    gu = interleave_gate_up(gate.detach().T.contiguous(), up.detach().T.contiguous()).T.contiguous().to(torch.bfloat16).to(dev)
    out = torch.empty_like(gu)
    ops.silu_mul_bwd(gu, d.to(dev), out)
    want = interleave_gate_up(gate.grad.T.contiguous(), up.grad.T.contiguous()).T
    assert rel_err(out.float().cpu(), want) < 4e-3  # bf16 output rounding


@pytest.mark.parametrize("two", [False, True])
def test_rmsnorm_bwd(gpu, two):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(4)
    M, H, eps = 37, 384, 1e-5
    x = torch.randn(M, H, generator=g).requires_grad_(True)
    gamma = torch.rand(H, generator=g) + 0.5
    gy = torch.randn(M, H, generator=g).to(torch.bfloat16)
    gy2 = torch.randn(M, H, generator=g).to(torch.bfloat16) if two else None
    y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * gamma
    y.backward(gy.float() + (gy2.float() if two else 0.0))
    base = torch.randn(M, H, generator=g)
    gx = base.clone().to(dev)
    ops.rmsnorm_bwd(x.detach().to(dev), gamma.to(dev), gy.to(dev), gx, eps, gy2=gy2.to(dev) if two else None, accumulate=True)
    assert rel_err(gx.cpu() - base, x.grad) < 1e-5
    ops.rmsnorm_bwd(x.detach().to(dev), gamma.to(dev), gy.to(dev), gx, eps, gy2=gy2.to(dev) if two else None)
    assert rel_err(gx.cpu(), x.grad) < 1e-5


def test_rope_bwd_pack(gpu):
    from tcavt_amd import ops
    from tcavt_amd.config import LlamaShape
    from tcavt_amd.rope import rope_tables

    dev = gpu["device"]
    g = torch.Generator().manual_seed(5)
    B, L, nq, nkv = 3, 20, 4, 2
    ncols, rc = (nq + 2 * nkv) * 64, (nq + nkv) * 64
    cos, sin = rope_tables(LlamaShape(), L)
    t = torch.randn(B, L, ncols // 64, 64, generator=g).requires_grad_(True)
    t1, t2 = t[..., :32], t[..., 32:]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    rot = torch.cat([t1 * c - t2 * s, t2 * c + t1 * s], dim=-1)
    out = torch.cat([rot[:, :, : nq + nkv], t[:, :, nq + nkv:]], dim=2)
    go = torch.randn(B, L, ncols // 64, 64, generator=g)
    out.backward(go)
    res = torch.empty(B * L, ncols, dtype=torch.bfloat16, device=dev)
    ops.rope_bwd_pack(go.reshape(B * L, ncols).contiguous().to(dev), res, cos.to(dev), sin.to(dev), rc, L)
    assert rel_err(res.float().cpu(), t.grad.reshape(B * L, ncols)) < 4e-3  # bf16 output rounding


def _attn_ref(qkv, dO, kv_len, B, T, nq, nkv):
    hd = 64
    x = qkv.float().clone().requires_grad_(True)
    v3 = x.view(B, T, nq + 2 * nkv, hd)
    q = v3[:, :, :nq].permute(0, 2, 1, 3)
    k = v3[:, :, nq:nq + nkv].permute(0, 2, 1, 3).repeat_interleave(nq // nkv, dim=1)
    v = v3[:, :, nq + nkv:].permute(0, 2, 1, 3).repeat_interleave(nq // nkv, dim=1)
    i = torch.arange(T, device=qkv.device)
    allowed = (i[None, :] <= i[:, None])[None] & (i[None, None, :] < kv_len.to(qkv.device).long()[:, None, None])
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(~allowed[:, None], float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, nq * hd)
    o.backward(dO.float())
    return x.grad


@pytest.mark.parametrize("B,T,nq,nkv,lens", [(2, 40, 4, 1, [40, 23]), (3, 64, 8, 2, [64, 1, 33]), (2, 256, 32, 8, [256, 170])])
def test_attn_causal_gqa_bwd(gpu, B, T, nq, nkv, lens):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(6)
    ncols = (nq + 2 * nkv) * 64
    qkv = torch.randn(B * T, ncols, generator=g).to(torch.bfloat16).to(dev)
    dO = torch.randn(B * T, nq * 64, generator=g).to(torch.bfloat16).to(dev)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=dev)
    g32 = torch.zeros(B * T, ncols, dtype=torch.float32, device=dev)
    ops.attn_causal_gqa_bwd(qkv, dO, g32, kv_len, B, T, nq, nkv, 1.0 / 8.0)
    want = _attn_ref(qkv, dO, kv_len, B, T, nq, nkv)
    for name, lo, hi in (("dq", 0, nq * 64), ("dk", nq * 64, (nq + nkv) * 64), ("dv", (nq + nkv) * 64, ncols)):
        assert rel_err(g32[:, lo:hi].cpu(), want[:, lo:hi].cpu()) < 2e-5, name  # fp32 arithmetic on both sides


def test_attn_bwd_refuses_long_sequences(gpu):
    from tcavt_amd import capi, ops

    dev = gpu["device"]
    T = 512
    qkv = torch.zeros(T, 6 * 64, dtype=torch.bfloat16, device=dev)
    with pytest.raises(capi.TcavtError):
        ops.attn_causal_gqa_bwd(qkv, torch.zeros(T, 4 * 64, dtype=torch.bfloat16, device=dev),
                                torch.zeros(T, 6 * 64, dtype=torch.float32, device=dev),
                                torch.tensor([T], dtype=torch.int32, device=dev), 1, T, 4, 1, 0.125)
