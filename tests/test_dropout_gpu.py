"""Train-mode dropout (MC-dropout K-candidate protocol, scripts/test.py:1301-1342).

Bit-matching torch's RNG is not a goal (SURVEY.md 7 "Hard parts"); the contract is: (1) masks are a pure function of
(seed, site, element) -- the numpy restatement oracle/philox.py reproduces them bit for bit, so every fused dropout
site can be checked EXACTLY against `reference_op * mask / (1 - p)`; (2) keep rate and independence are right;
(3) eval mode is untouched; (4) K train-mode passes give K different candidates and min-over-K <= each single pass."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mask(n, p, seed, site, dev):
    from oracle import philox

    return torch.from_numpy(philox.keep_mask(n, p, seed, site)).to(dev)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_elementwise_dropout_matches_philox_oracle(gpu, dtype):
    from tcavt_amd import ops

    dev = gpu["device"]
    n, p, seed, site = 100003, 0.1, 0x1234567890ABCDEF, 7
    x = torch.randn(n, device=dev).to(dtype)
    y = torch.empty_like(x)
    ops.dropout(x, y, p, seed, site)
    keep = _mask(n, p, seed, site, dev)
    ref = (x.float() * keep / (1 - p)).to(dtype)
    assert torch.equal(y, ref)
    rate = keep.float().mean().item()
    assert abs(rate - 0.9) < 5e-3
    other = _mask(n, p, seed, site + 1, dev)
    both = (keep & other).float().mean().item()
    assert abs(both - 0.81) < 5e-3  # independent sites


def test_gemm_epilogue_dropout_is_exact(gpu):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(3)
    M, N, K, p, seed, site = 300, 256, 128, 0.1, 99, 3
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    keep = _mask(M * N, p, seed, site, dev).view(M, N)
    plain = ops.gemm_bf16(a, w, out_dtype=torch.float32, bias=bias, relu=True)
    out = ops.gemm_bf16(a, w, out_dtype=torch.float32, bias=bias, relu=True, residual=res, dropout=(p, seed, site))
    # placement: bias -> ReLU -> dropout -> residual
    assert torch.allclose(out, plain * keep / (1 - p) + res, rtol=1e-6, atol=1e-6)
    a32, w32 = torch.randn(70, 50, generator=g).to(dev), torch.randn(33, 50, generator=g).to(dev)
    b32, r32 = torch.randn(33, generator=g).to(dev), torch.randn(70, 33, generator=g).to(dev)
    keep2 = _mask(70 * 33, p, seed, site + 1, dev).view(70, 33)
    plain2 = ops.gemm_f32(a32, w32, bias=b32, relu=True)
    out2 = ops.gemm_f32(a32, w32, bias=b32, relu=True, residual=r32, dropout=(p, seed, site + 1))
    assert torch.allclose(out2, plain2 * keep2 / (1 - p) + r32, rtol=1e-6, atol=1e-6)


def test_attention_weight_dropout_is_exact(gpu):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(5)
    B, Lq, Lk, nh, dh, p, seed, site = 2, 18, 18, 4, 32, 0.1, 1234, 9
    E = nh * dh
    q, k, v = (torch.randn(B, L, E, generator=g).to(dev) for L in (Lq, Lk, Lk))
    out = torch.empty(B, Lq, E, device=dev)
    scale = 1 / math.sqrt(dh)
    ops.mha(q, k, v, out, B, Lq, Lk, nh, dh, scale, dropout=(p, seed, site))
    qh, kh, vh = (t.view(B, -1, nh, dh).transpose(1, 2) for t in (q, k, v))
    P = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1)  # [B,nh,Lq,Lk]; flat index ((b*nh+h)*Lq+i)*Lk+j
    keep = _mask(P.numel(), p, seed, site, dev).view_as(P)
    ref = ((P * keep / (1 - p)) @ vh).transpose(1, 2).reshape(B, Lq, E)
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)
    # softmax_rows (cross-attention probabilities)
    rows, L, Lp = 20, 40, 64
    S = torch.randn(rows, Lp, generator=g).to(dev)
    Pm = torch.empty(rows, Lp, dtype=torch.float16, device=dev)
    ops.softmax_rows(S, Pm, rows, L, Lp, Lp, Lp, dropout=(p, seed, site + 1))
    keep2 = _mask(rows * Lp, p, seed, site + 1, dev).view(rows, Lp)[:, :L]
    ref2 = torch.softmax(S[:, :L], -1) * keep2 / (1 - p)
    assert (Pm[:, :L].float() - ref2).abs().max().item() < 1e-3 and (Pm[:, L:] == 0).all()


def test_mc_dropout_candidates(gpu):
    from tests.util import batch_tensors, load_case
    from tcavt_amd import evaluate, model

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    g = {k: v.to(dev) for k, v in t.items()}
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    kw = dict(input_ids=g["input_ids"], attention_mask=g["attention_mask"])
    with torch.no_grad():
        ev1 = m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], **kw).clone()
        m.train()
        m._fwd_count = 0
        c = [m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], **kw).clone() for _ in range(3)]
        m._fwd_count = 0
        c0_again = m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], **kw).clone()
        m.eval()
        ev2 = m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], **kw).clone()
    # bit-equal: the forward has no order-dependent reduction (forward split-K is capped at two addends, small.hip)
    assert torch.equal(ev1, ev2)       # eval untouched by the dropout machinery, and reproducible
    assert torch.equal(c[0], c0_again)  # same seed -> same masks -> same candidate
    assert all(torch.isfinite(x).all() for x in c)
    spread = min((c[0] - c[1]).abs().max().item(), (c[1] - c[2]).abs().max().item(), (c[0] - ev1).abs().max().item())
    assert 1e-3 < spread < 5.0                          # different seeds -> genuinely different candidates
    r1 = evaluate.evaluate_model(m, [g], num_candidates=1)
    r5 = evaluate.evaluate_model(m, [g], num_candidates=5, mc_dropout=True)
    assert not m.training and r5["n"] == r1["n"] and all(math.isfinite(r5[k]) for k in ("ADE", "FDE", "RMSE"))


def test_rmsnorm_fused_dropout_matches_two_pass(gpu):
    """tcavt_rmsnorm's optional dropped output (LoRA branch input) is bit-identical to tcavt_dropout on its bf16 output."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(2)
    for M, H in ((37, 2048), (64, 384)):  # exact-template width and the generic two-pass width
        x = torch.randn(M, H, generator=g).to(dev)
        gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
        spec = (0.1, 99, 7)
        xn = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
        xd = torch.empty_like(xn)
        ops.rmsnorm(x, gamma, 1e-5, out_bf16=xn, out_drop=xd, dropout=spec)
        xn2 = torch.empty_like(xn)
        ops.rmsnorm(x, gamma, 1e-5, out_bf16=xn2)
        ref = torch.empty_like(xn)
        ops.dropout(xn2, ref, *spec)
        assert torch.equal(xn, xn2) and torch.equal(xd, ref)
        assert 0.05 < (xd == 0).float().mean().item() < 0.15


def test_k_candidates_reuse_llm_prefix(gpu):
    """with no dropout site inside the MLLM, the K MC-dropout candidates share one MLLM pass; the metrics
    equal those of K full passes exactly (same per-module mask numbering), and it is refused when the MLLM has dropout."""
    from tests.util import batch_tensors, load_case
    from tcavt_amd import evaluate, model

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    g = {k: v.to(dev) for k, v in t.items()}
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    with pytest.raises(ValueError):
        evaluate.evaluate_model(m, [g], num_candidates=3, mc_dropout=True, reuse_prefix=True)
    m.mllm.qformer.dropout_p = 0.0
    m.mllm.llama_wrapper.lora_dropout = 0.0
    m._fwd_count = 0
    full = evaluate.evaluate_model(m, [g, g], num_candidates=4, mc_dropout=True)
    m._fwd_count = 0
    fast = evaluate.evaluate_model(m, [g, g], num_candidates=4, mc_dropout=True, reuse_prefix=True)
    assert m._llm_cache is None and not m.training
    for k in ("ADE", "FDE", "RMSE"):
        assert abs(full[k] - fast[k]) <= 1e-6 * abs(full[k]), (k, full[k], fast[k])


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_lora_down_fused_masks(gpu, dt, p):
    """tcavt_lora_down: both adapters' down-projections in one pass, each under its own Philox site (PEFT: one lora_dropout
    module per adapted Linear, scripts/train.py:433-439) -- against dropout(x) rounded to 16 bits, then the fp32 product."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(8)
    M, H, r, s = 100, 256, 8, 4.0
    x = torch.randn(M, H, generator=g).to(dt).to(dev)
    a_cat = torch.zeros(64, H, dtype=dt, device=dev)
    a_cat[:r] = (torch.randn(r, H, generator=g) * 0.05).to(dt).to(dev)
    a_cat[16:16 + r] = (torch.randn(r, H, generator=g) * 0.05).to(dt).to(dev)
    t = torch.zeros(M, 64, dtype=dt, device=dev)
    seed, site = 0xABCDEF12345, 77
    ops.lora_down(x, a_cat, t, s, dropout=(p, seed, site) if p > 0 else None)
    torch.cuda.synchronize()
    for blk, st in ((0, site), (16, site + 1)):
        if p > 0:
            keep = _mask(M * H, p, seed, st, dev).view(M, H)
            xd = (x.float() * keep / (1 - p)).to(dt).float()
        else:
            xd = x.float()
        ref = s * (xd @ a_cat[blk:blk + 16].float().T)
        got = t[:, blk:blk + 16].float()
        assert (got - ref.to(dt).float()).abs().max().item() <= 2e-2 * ref.abs().max().item() * (1 if dt == torch.bfloat16 else 0.2)
        assert ((got - ref).norm() / ref.norm()).item() < (4e-3 if dt == torch.bfloat16 else 5e-4)
    assert (t[:, 32:] == 0).all()
    if p > 0:  # the two adapters really see different masks
        assert not torch.equal(_mask(M * H, p, seed, site, dev), _mask(M * H, p, seed, site + 1, dev))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_lora_dgrad_fused_masks(gpu, dt, p):
    """tcavt_lora_dgrad (backward of tcavt_lora_down w.r.t. its input; modify_scripts/modify_train.py:512-528 trains the adapters):
    out = mask_q * (g_t[:, :16] . A_q) + mask_v * (g_t[:, 16:32] . A_v), each adapter under its own Philox site -- against the
    two fp32 products masked with the oracle-equal masks; M not a multiple of 16 exercises the row guard."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(9)
    M, H, r = 100, 256, 8
    g_t = torch.zeros(M, 64, dtype=dt, device=dev)
    g_t[:, :r] = torch.randn(M, r, generator=g).to(dt).to(dev)
    g_t[:, 16:16 + r] = torch.randn(M, r, generator=g).to(dt).to(dev)
    Aq = (torch.randn(r, H, generator=g) * 0.05).to(dt)
    Av = (torch.randn(r, H, generator=g) * 0.05).to(dt)
    aqT = torch.zeros(H, 64, dtype=dt, device=dev)
    avT = torch.zeros(H, 64, dtype=dt, device=dev)
    aqT[:, :r] = Aq.T.to(dev)
    avT[:, 16:16 + r] = Av.T.to(dev)
    out = torch.full((M, H), float("nan"), dtype=dt, device=dev)
    seed, site = 0x5EEDF00D, 1031
    ops.lora_dgrad(g_t, aqT, avT, out, dropout=(p, seed, site) if p > 0 else None, site_v=site + 1)
    torch.cuda.synchronize()
    pq = g_t[:, :r].float() @ Aq.float().to(dev)
    pv = g_t[:, 16:16 + r].float() @ Av.float().to(dev)
    if p > 0:
        kq = _mask(M * H, p, seed, site, dev).view(M, H)
        kv = _mask(M * H, p, seed, site + 1, dev).view(M, H)
        ref = pq * kq / (1 - p) + pv * kv / (1 - p)
        both_dropped = ~(kq.bool() | kv.bool())
        assert both_dropped.any() and (out.float()[both_dropped] == 0).all()   # masked elements are exact zeros
    else:
        ref = pq + pv
    assert torch.isfinite(out.float()).all()
    e = ((out.float() - ref).norm() / ref.norm()).item()
    assert e < (4e-3 if dt == torch.bfloat16 else 5e-4), e  # one rounding to the 16-bit type
