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
    gxb = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
    ops.rmsnorm_bwd(x.detach().to(dev), gamma.to(dev), gy.to(dev), gx, eps, gy2=gy2.to(dev) if two else None, gx_bf16=gxb)
    assert rel_err(gx.cpu(), x.grad) < 1e-5
    assert torch.equal(gxb, gx.to(torch.bfloat16))  # the fused bf16 copy = a cast of the fp32 result


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


def _lora_keys(weights):
    return [k for k in weights if ".lora_A." in k or ".lora_B." in k]


@pytest.mark.parametrize("ragged,lora_drop", [(False, False), (True, False), (True, True)])
def test_decoder_backward_lora_grads_match_autograd(gpu, ragged, lora_drop):
    """Stage-level: identical decoder input and an arbitrary gradient of the final hidden states; the adapter gradients of
    LoraBackward vs torch autograd through the oracle's decoder (fp16 contract, straight-through casts).  The HIP walk is
    IEEE half throughout -- tapes and, under the device-chosen power-of-two scale, gradients."""
    from oracle import forward as O
    from tcavt_amd import model, training
    from tests.util import load_case

    dev = gpu["device"]
    cfg, weights, _ = load_case("tiny_6_12_lora_ragged")
    ll = cfg.llama
    B, L, H = 3, 40, ll.hidden
    g = torch.Generator().manual_seed(11)
    embeds = torch.randn(B, L, H, generator=g) * 0.5
    lens = [L, 17, 29] if ragged else [L] * B
    mask = torch.zeros(B, L, dtype=torch.int64)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    G = torch.randn(B, L, H, generator=g).to(torch.bfloat16)

    W = {k: torch.from_numpy(v).clone() for k, v in weights.items()}
    keys = _lora_keys(W)
    assert len(keys) == 4 * ll.layers
    for k in keys:
        W[k].requires_grad_(True)
    # train mode: LoRA dropout with the HIP path's Philox masks (block 2 of the model's site numbering, one site per layer)
    seed = 0xD0C
    drop = O.DropTape(seed, cfg.lora_dropout, first_site=(2 << 16) + 1) if lora_drop else O._ident
    out = O.llama_decoder(W, cfg, embeds, mask, O._rounder("fp16"), drop=drop)  # (the model's default storage contract)
    (out * G.float()).sum().backward()

    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    tr = training.Trainer(m, lora_trainable=True)
    lw = m.mllm.llama_wrapper
    with torch.no_grad():
        lw.dctx = model.DropoutCtx(seed).sub(2) if lora_drop else None
        res = lw(embeds.to(dev), mask.to(dev)).last_hidden_state
        lw.dctx = None
        assert m.storage == torch.float16 and rel_err(res.cpu(), out.detach()) < 2e-3
        assert (lw.tape.layers[0].dspec is not None) == lora_drop
        tr.lbw.run(G.reshape(B * L, H).contiguous().to(dev))
    torch.cuda.synchronize()
    worst = 0.0
    for k in keys:
        ref = W[k].grad
        got = tr.book.g[k].cpu()
        assert ref.abs().max() > 0, k
        e = rel_err(got, ref)
        worst = max(worst, e)
        assert e < 7.5e-3, (k, e)  # fp16 gradient chain through the layers vs fp32 autograd of the fp16-contract graph: 3 x the measured 2.4e-3 (bf16 chain: 1.2e-2)
    print(f"[lora grads ragged={ragged} lora_dropout={lora_drop}] worst relative error {worst:.2e}")


def test_lora_trainable_step(gpu):
    """End to end (modify_scripts/modify_train.py:512-528,1192): LoRA gradients are produced next to the train.py set, agree
    with autograd through the whole oracle graph, and an optimizer step changes the adapters and the next forward."""
    from oracle import forward as O
    from tcavt_amd import model, training
    from tcavt_amd.weights import trainable_keys
    from tests.util import batch_tensors, load_case

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    W = {k: torch.from_numpy(v).clone() for k, v in weights.items()}
    keys = _lora_keys(W)
    for k in keys + trainable_keys(W):
        W[k].requires_grad_(True)
    loss, _ = O.model_forward(W, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                              t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                              contract="fp16")
    loss.backward()

    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    tr = training.Trainer(m, lora_trainable=True, max_grad_norm=1.0)
    assert m.storage == torch.float16
    assert set(tr.book.g) == set(keys) | set(trainable_keys(W))
    gq = {k: v.to(dev) for k, v in t.items()}
    args = (gq["traj_emb"], gq["vision_emb"], gq["lane_polygon"], gq["lane_polygon_len"], gq["target_traj"], gq["norm_stat"],
            gq["input_ids"], gq["attention_mask"], gq["labels"])
    l0, _ = tr.forward_backward(*args)
    torch.cuda.synchronize()
    flat_ref = torch.cat([W[k].grad.reshape(-1).double() for k in keys])
    flat_got = torch.cat([tr.book.g[k].cpu().reshape(-1).double() for k in keys])
    assert torch.isfinite(flat_got).all() and flat_ref.norm() > 0
    rel = ((flat_got - flat_ref).norm() / flat_ref.norm()).item()
    cos = (flat_got @ flat_ref / (flat_got.norm() * flat_ref.norm())).item()
    print(f"[lora step] flat adapter gradient vs fp16-contract autograd: rel {rel:.2e}, cosine {cos:.5f}")
    assert rel < 1e-2 and cos > 0.9999  # end to end on a tiny case; 3 x the measured 3.0e-3
    before = {k: tr.book.g[k].clone() for k in keys[:2]}
    a0 = m.mllm.llama_wrapper.llama_model.model.layers[0].self_attn.q_proj.lora_A.weight.detach().clone()
    gn = torch.linalg.vector_norm(tr.book.grads).item()
    tr.optimizer_step()
    torch.cuda.synchronize()
    if gn > 1.0:  # clipped to max_grad_norm before AdamW
        assert abs(torch.linalg.vector_norm(tr.book.grads).item() - 1.0) < 1e-3
    a1 = m.mllm.llama_wrapper.llama_model.model.layers[0].self_attn.q_proj.lora_A.weight.detach()
    assert (a1 - a0).abs().max().item() > 0
    # the packed 16-bit operands (and their transposes for the dgrad GEMMs) follow the updated parameters, every layer
    lw, r = m.mllm.llama_wrapper, cfg.lora_r
    st = m.storage
    from tcavt_amd.model import LORA_V as LV
    nqh, nkvh = cfg.llama.n_q_heads * 64, cfg.llama.n_kv_heads * 64
    for li, lyr in enumerate(lw.llama_model.model.layers):
        d, dT, sa = lw._prepared().layers[li], lw.prepared_T()[li], lyr.self_attn
        # the forward's packed adapters carry the folded RMSNorm gain (A * input_layernorm.weight, rounded once) ...
        g1 = lyr.input_layernorm.weight.detach()[None, :]
        assert torch.equal(d.a_cat[:r], (sa.q_proj.lora_A.weight.detach() * g1).to(st))
        assert torch.equal(d.a_cat[LV:LV + r], (sa.v_proj.lora_A.weight.detach() * g1).to(st))
        assert torch.equal(d.b_ext[:nqh, :r], sa.q_proj.lora_B.weight.detach().to(st))
        assert torch.equal(d.b_ext[nqh + nkvh:, LV:LV + r], sa.v_proj.lora_B.weight.detach().to(st))
        assert d.a_cat[r:LV].abs().max().item() == 0 and d.a_cat[LV + r:].abs().max().item() == 0
        assert d.b_ext[nqh:nqh + nkvh].abs().max().item() == 0
        # ... the backward's copies are the plain matrices
        assert torch.equal(dT.a_plain[:r], sa.q_proj.lora_A.weight.detach().to(st))
        assert torch.equal(dT.a_plain[LV:LV + r], sa.v_proj.lora_A.weight.detach().to(st))
        assert torch.equal(dT.a_q[:, :LV], dT.a_plain[:LV].t()) and dT.a_q[:, LV:].abs().max().item() == 0
        assert torch.equal(dT.a_v[:, LV:], dT.a_plain[LV:].t()) and dT.a_v[:, :LV].abs().max().item() == 0
        assert torch.equal(dT.a_cat, dT.a_plain.t()) and torch.equal(dT.b_ext, d.b_ext.t())
    l1, _ = tr.forward_backward(*args)
    torch.cuda.synchronize()
    assert torch.isfinite(l1).item() and l1.item() != l0.item()
    for _ in range(8):
        tr.optimizer_step()
        l1, _ = tr.forward_backward(*args)
    assert l1.item() < l0.item()


@pytest.mark.parametrize("scores", ["fused", "scores+gemm", "gemm"])
@pytest.mark.parametrize("B,T,nq,nkv,lens", [(2, 40, 4, 1, [40, 23]), (3, 64, 8, 2, [64, 1, 33]), (2, 256, 32, 8, [256, 170]),
                                             (1, 300, 4, 2, [211])])
def test_attn_bwd_composed_matches_autograd_and_scalar_kernel(gpu, B, T, nq, nkv, lens, scores):
    """The production attention backward (batched MFMA products + row kernel) vs fp32 autograd of the same bf16 inputs,
    and vs the scalar cross-check kernel."""
    from tcavt_amd import ops
    from tcavt_amd.config import LlamaShape
    from tcavt_amd.llm_backward import attn_bwd_composed
    from tcavt_amd.rope import rope_tables

    dev = gpu["device"]
    g = torch.Generator().manual_seed(7)
    ncols = (nq + 2 * nkv) * 64
    M = B * T
    qkv_p = torch.zeros(M + 64, ncols, dtype=torch.bfloat16, device=dev)
    qkv_p[:M] = torch.randn(M, ncols, generator=g).to(torch.bfloat16).to(dev)
    dO = torch.randn(M, nq * 64, generator=g).to(torch.bfloat16).to(dev)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=dev)
    cos, sin = (t.to(dev) for t in rope_tables(LlamaShape(), T))
    pool = {}

    def buf(name, shape, dtype, zero=False):
        if name not in pool:
            pool[name] = torch.zeros(shape, dtype=dtype, device=dev)
        return pool[name]

    out = torch.empty(M, ncols, dtype=torch.bfloat16, device=dev)
    # a first call with full-length samples leaves its values in the scratch; the real call (shorter samples) must not see them
    attn_bwd_composed(buf, qkv_p, dO, torch.full((B,), T, dtype=torch.int32, device=dev), B, T, nq, nkv, 0.125, cos, sin, out,
                      scores=scores)
    attn_bwd_composed(buf, qkv_p, dO, kv_len, B, T, nq, nkv, 0.125, cos, sin, out, scores=scores)
    want32 = _attn_ref(qkv_p[:M], dO, kv_len, B, T, nq, nkv)
    want = torch.empty_like(out)
    ops.rope_bwd_pack(want32.contiguous(), want, cos, sin, (nq + nkv) * 64, T)  # (tested above against autograd)
    if T <= 280:  # (the scalar cross-check kernel keeps a whole score block in LDS)
        g32 = torch.zeros(M, ncols, dtype=torch.float32, device=dev)
        ops.attn_causal_gqa_bwd(qkv_p[:M], dO, g32, kv_len, B, T, nq, nkv, 0.125)
        assert rel_err(g32.cpu(), want32.cpu()) < 2e-5
    for name, lo, hi in (("dq", 0, nq * 64), ("dk", nq * 64, (nq + nkv) * 64), ("dv", (nq + nkv) * 64, ncols)):
        e = rel_err(out[:, lo:hi].float().cpu(), want[:, lo:hi].float().cpu())
        assert e < 1e-2, (name, e)  # P and dS pass through bf16 on their way into the MFMA products


@pytest.mark.parametrize("M,N,K", [(120, 1024, 256), (512, 1024, 256), (8192, 16384, 2048)])
def test_silu_epilogue_saves_preactivations(gpu, M, N, K):
    """tcavt_gemm_args.silu_preact: the SiLU epilogue's bf16 copy of gate|up = the plain GEMM's output, and the activated
    output is bit-identical to the call without it (small-launch kernels, 4-wave kernel, persistent 4-wave kernel)."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(8)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    ref_act = ops.gemm_bf16(a, w, silu_mul=True)
    plain = ops.gemm_bf16(a, w)
    pre = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    act = ops.gemm_bf16(a, w, silu_mul=True, silu_preact=pre)
    torch.cuda.synchronize()
    assert torch.equal(act, ref_act)
    assert rel_err(pre.float().cpu(), plain.float().cpu()) < 1e-3  # same products; tile forms may order the K sum differently


@pytest.mark.parametrize("scale", [1e-3, 3.0])
def test_clip_grad_norm(gpu, scale):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(9)
    v = (torch.randn(1_234_567, generator=g) * scale / 1000.0)
    p = torch.nn.Parameter(torch.zeros_like(v))
    p.grad = v.clone()
    norm = torch.nn.utils.clip_grad_norm_([p], 1.0)
    gd = v.to(dev)
    scratch = torch.zeros(1026, dtype=torch.float32, device=dev)
    ops.clip_grad_norm(gd, 1.0, scratch)
    norm64 = v.double().norm().item()  # (fp32 sums of 1.2 M squares differ by ~1e-5 between any two orders)
    assert abs(scratch[1025].item() - norm64) / norm64 < 2e-5 and abs(norm.item() - norm64) / norm64 < 1e-4
    assert rel_err(gd.cpu(), p.grad) < 1e-4
    want = v.double() * min(1.0, 1.0 / (norm64 + 1e-6))
    assert rel_err(gd.cpu().double(), want) < 2e-5
    again = v.to(dev)
    ops.clip_grad_norm(again, 1.0, scratch)
    assert torch.equal(again, gd)  # fixed summation order: bit-reproducible


def test_clip_grad_norm_of_the_dp_mean(gpu):
    """Data parallel: buckets are SUM-all-reduced, the reference clips DDP's MEAN (modify_train.py:1192 after the averaging
    all-reduce).  clip(grad_scale = 1 / world) on the summed gradient == clip_grad_norm_ on the mean."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(11)
    world = 4
    mean = torch.randn(500_000, generator=g) * 3e-3  # norm ~ 2.1 > 1: clipped; the SUM's norm is 4x that
    p = torch.nn.Parameter(torch.zeros_like(mean))
    p.grad = mean.clone()
    torch.nn.utils.clip_grad_norm_([p], 1.0)
    summed = (mean * world).to(dev)
    scratch = torch.zeros(1026, dtype=torch.float32, device=dev)
    ops.clip_grad_norm(summed, 1.0, scratch, grad_scale=1.0 / world)
    assert rel_err(summed.cpu(), p.grad) < 1e-4
    assert abs(scratch[1025].item() - mean.double().norm().item()) < 1e-4


def test_adamw_gated_skips_nonfinite_loss(gpu):
    """modify_scripts/modify_train.py:1190-1196: step only `if torch.isfinite(loss)`; decided on the device, and the
    bias corrections follow the number of APPLIED updates (torch.optim.AdamW's step count does not advance on a skip)."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(12)
    n = 100_003
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=5e-4, weight_decay=1e-4)
    p, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    ctl = torch.zeros(8, dtype=torch.int32, device=dev)
    losses = [1.0, float("nan"), 2.0, float("inf"), 3.0]
    for i, lv in enumerate(losses):
        grad = torch.randn(n, generator=g)
        loss = torch.tensor([lv], dtype=torch.float32, device=dev)
        before = p.clone()
        ops.adamw_gated(p, grad.to(dev), m, v, 5e-4, 0.9, 0.999, 1e-8, 1e-4, loss, ctl)
        if lv == lv and abs(lv) != float("inf"):
            ref.grad = grad.clone()
            opt.step()
        else:
            assert torch.equal(p, before)  # skipped: parameters untouched
    torch.cuda.synchronize()
    assert ctl[:3].tolist() == [3, 2, 0]
    assert rel_err(p.cpu(), ref.detach()) < 1e-6
    # a non-finite gradient norm skips as well (keeps data-parallel ranks in step)
    before = p.clone()
    ops.adamw_gated(p, torch.ones(n, device=dev), m, v, 5e-4, 0.9, 0.999, 1e-8, 1e-4,
                    torch.ones(1, device=dev), ctl, grad_norm=torch.full((1,), float("nan"), device=dev))
    assert torch.equal(p, before) and ctl[:3].tolist() == [3, 3, 1]
    # dynamic loss scaling of the fp16 backward: every skipped update bought four binary orders of headroom (ctl[6], capped at 24) ...
    assert ctl[6].item() == 12 and ctl[7].item() == 0
    gfin = (torch.randn(4096, generator=g) * 3.0).to(torch.bfloat16).to(dev)
    scale, scratch = torch.zeros(2, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    ops.grad_scale_pick(gfin, None, scale, scratch)
    s_plain = scale[0].item()
    ops.grad_scale_pick(gfin, None, scale, scratch, backoff=ctl[6:7])
    assert scale[0].item() == s_plain / 4096 and scale[1].item() == 4096 / s_plain
    amax = gfin.float().abs().max().item()
    assert 128.0 <= amax * s_plain <= 256.0
    # ... and 256 applied updates in a row give one back
    one = torch.ones(1, device=dev)
    for _ in range(256):
        ops.adamw_gated(p, torch.zeros(n, device=dev), m, v, 0.0, 0.9, 0.999, 1e-8, 0.0, one, ctl)
    assert ctl[6].item() == 11 and ctl[7].item() == 0 and ctl[1].item() == 3


@pytest.mark.parametrize("xdt", [torch.bfloat16, torch.float16])
def test_wgrad_tn_matches_transposed_product(gpu, xdt):
    """tcavt_wgrad_tn: C[i, h] += sum_m G[m, g0 + i] X[m, h] on token-major operands (no physical transposes), both
    output layouts, accumulation into a non-zero C, ragged M / H."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(21)
    M, H = 1000, 712
    G = torch.randn(M, 64, generator=g).to(torch.bfloat16).to(dev)
    X = torch.randn(M, H, generator=g).to(xdt).to(dev)
    Xb = X.float().to(torch.bfloat16).float() if xdt == torch.float16 else X.float()
    for g0, n in ((0, 32), (16, 16), (0, 64)):
        ref = G[:, g0:g0 + n].float().T @ Xb
        base = torch.randn(n, H, generator=g).to(dev)
        out = base.clone()
        ops.wgrad_tn(G, g0, n, X, out)
        assert rel_err(out.cpu(), (base + ref).cpu()) < 2e-6
        outT = torch.zeros(H, 64, device=dev)
        ops.wgrad_tn(G, g0, n, X, outT, trans_out=True)
        assert rel_err(outT[:, :n].cpu(), ref.T.cpu()) < 2e-6 and (outT[:, n:] == 0).all()


def test_allreduce_flat_on_a_raw_rccl_communicator(gpu):
    """tcavt_allreduce_flat on an ncclComm_t the caller made itself (here: a one-rank communicator created through ctypes on
    librccl): the sum over one rank is the buffer itself; a null communicator is an argument error, not a crash."""
    import ctypes

    from tcavt_amd import capi, ops

    dev = gpu["device"]
    buf = torch.randn(1 << 20, device=dev)
    want = buf.clone()
    with pytest.raises(capi.TcavtError):
        ops.allreduce_flat(buf, None)
    try:
        rccl = ctypes.CDLL("librccl.so.1")
    except OSError:
        pytest.skip("librccl.so.1 not loadable on this box")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        ops.allreduce_flat(buf, comm)
        torch.cuda.synchronize()
        assert torch.equal(buf, want)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


def _front_keys(weights):
    return [k for k in weights if k.startswith(("mllm.qformer.", "mllm.q_proj.")) or k in ("mllm.vision_modality_embedding",
                                                                                           "mllm.text_modality_embedding")]


@pytest.mark.parametrize("train_mode", [False, True])
def test_full_modify_train_gradients_match_autograd(gpu, train_mode):
    """Trainer(lora_trainable=True, train_mllm_front=True) = the whole trainable set of modify_scripts/modify_train.py
    (:523-528 freeze only the non-LoRA Llama weights): gradients of the Q-Former, mllm.q_proj and the modality embeddings
    against torch autograd through the oracle graph (fp16 contract), in eval arithmetic and in train mode with the same
    Philox masks; an optimizer step then moves them and the next forward."""
    from oracle import forward as O
    from tcavt_amd import model, training
    from tcavt_amd.weights import trainable_keys
    from tests.util import batch_tensors, load_case

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    W = {k: torch.from_numpy(v).clone() for k, v in weights.items()}
    lora, front = _lora_keys(W), _front_keys(W)
    assert any("qformer.decoder.layers.0.multihead_attn.in_proj_weight" in k for k in front) and "mllm.qformer.query_tokens" in front
    for k in lora + front + trainable_keys(W):
        W[k].requires_grad_(True)
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev)
    m.train(train_mode)
    m._fwd_count = 0
    tr = training.Trainer(m, lora_trainable=True, train_mllm_front=True, max_grad_norm=1.0)
    assert set(tr.book.g) == set(lora) | set(front) | set(trainable_keys(W))
    loss, _ = O.model_forward(W, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                              t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                              contract="fp16", dropout_seed=m.dropout_seed if train_mode else None)
    loss.backward()
    gq = {k: v.to(dev) for k, v in t.items()}
    args = (gq["traj_emb"], gq["vision_emb"], gq["lane_polygon"], gq["lane_polygon_len"], gq["target_traj"], gq["norm_stat"],
            gq["input_ids"], gq["attention_mask"], gq["labels"])
    l0, _ = tr.forward_backward(*args)
    torch.cuda.synchronize()
    assert abs(l0.item() - loss.item()) < 2e-2 * abs(loss.item())
    groups = {"modality embeddings": [k for k in front if "modality" in k], "q_proj": [k for k in front if ".q_proj." in k],
              "query tokens": ["mllm.qformer.query_tokens"],
              "qformer decoder": [k for k in front if "qformer.decoder" in k],
              "qformer encoder + vision_proj": [k for k in front if "qformer.encoder" in k or "vision_proj" in k]}
    for name, ks in groups.items():
        ref = torch.cat([W[k].grad.reshape(-1).double() for k in ks])
        got = torch.cat([tr.book.g[k].cpu().reshape(-1).double() for k in ks])
        assert torch.isfinite(got).all() and ref.norm() > 0, name
        rel = ((got - ref).norm() / ref.norm()).item()
        cos = (got @ ref / (got.norm() * ref.norm())).item()
        print(f"[modify_train set, train_mode={train_mode}] {name}: rel {rel:.2e}, cosine {cos:.5f}")
        # fp16 (scaled) gradient chain through the decoder layers, bf16 through the Q-Former layers, vs fp32 autograd of the
        # fp16-contract graph: 1.5 x the measured 9.6e-3 (eval arithmetic) / 3.1e-2 (train mode, same masks)
        assert rel < (5e-2 if train_mode else 1.5e-2) and cos > 0.999, (name, rel, cos)
        # ... and PER PARAMETER: inside a concatenated group a small tensor with a wrong mask site or a missing 1 / rms would
        # hide behind the large ones (a wrong site decorrelates the gradient: cosine ~ 0.9 at p = 0.1; a missing factor shows in rel)
        worst = (0.0, 1.0, None)
        for k in ks:
            r_, g_ = W[k].grad.reshape(-1).double(), tr.book.g[k].cpu().reshape(-1).double()
            if r_.norm() == 0:
                assert g_.norm() == 0, k
                continue
            rel_k = ((g_ - r_).norm() / r_.norm()).item()
            cos_k = (g_ @ r_ / (g_.norm() * r_.norm())).item()
            if rel_k > worst[0]:
                worst = (rel_k, cos_k, k)
            assert rel_k < (0.12 if train_mode else 4e-2) and cos_k > 0.993, (k, rel_k, cos_k)
        print(f"    worst parameter: {worst[2]}: rel {worst[0]:.2e}, cosine {worst[1]:.5f}")
    w0 = m.mllm.qformer.query_tokens.detach().clone()
    tr.optimizer_step()
    l1, _ = tr.forward_backward(*args)
    torch.cuda.synchronize()
    assert (m.mllm.qformer.query_tokens.detach() - w0).abs().max().item() > 0
    assert torch.isfinite(l1).item() and l1.item() != l0.item()
    if not train_mode:
        for _ in range(6):
            tr.optimizer_step()
            l1, _ = tr.forward_backward(*args)
        assert l1.item() < l0.item()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_silu_bwd_epilogue_matches_unfused(gpu, dt):
    """TCAVT_EPI_SILU_BWD: the dgrad GEMM of down_proj with d(silu(gate) * up) in its epilogue (in place over the taped
    pre-activations) against the unfused pair it replaces -- gemm, then tcavt_silu_mul_bwd -- and against fp32 autograd of
    HF's LlamaMLP arithmetic (modeling_llama.py:174-176).  The fused form does not round dL/d(act) to 16 bits in between."""
    from tcavt_amd import ops
    from tcavt_amd.layout import interleave_gate_up

    dev = gpu["device"]
    g = torch.Generator().manual_seed(21)
    M, I, H = 512, 768, 256
    assert ops.silu_bwd_fusable(M, I, H)
    gh = (torch.randn(M, H, generator=g) * 0.5).to(dt)
    w_t = (torch.randn(I, H, generator=g) * 0.06).to(dt)          # down_proj.weight^T
    gate = torch.randn(M, I, generator=g).to(dt).float().requires_grad_(True)
    up = torch.randn(M, I, generator=g).to(dt).float().requires_grad_(True)
    d_act = gh.float() @ w_t.float().T
    (torch.nn.functional.silu(gate) * up * d_act).sum().backward()
    want = interleave_gate_up(gate.grad.T.contiguous(), up.grad.T.contiguous()).T.contiguous()
    gu = interleave_gate_up(gate.detach().T.contiguous(), up.detach().T.contiguous()).T.contiguous().to(dt).to(dev)
    gu_fused = gu.clone()
    ops.gemm_silu_bwd(gh.to(dev), w_t.to(dev), gu_fused)                                   # in place
    g_act = ops.gemm_bf16(gh.to(dev), w_t.to(dev))
    gu_unf = gu.clone()
    ops.silu_mul_bwd(gu_unf, g_act, gu_unf)
    torch.cuda.synchronize()
    tol = 6e-3 if dt == torch.bfloat16 else 8e-4
    assert rel_err(gu_fused.float().cpu(), want) < tol
    assert rel_err(gu_unf.float().cpu(), want) < 1.5 * tol
    assert rel_err(gu_fused.float().cpu(), gu_unf.float().cpu()) < 1.5 * tol
    out2 = torch.empty_like(gu)
    ops.gemm_silu_bwd(gh.to(dev), w_t.to(dev), gu, out=out2)                               # out of place: the same bits
    assert torch.equal(out2, gu_fused)
    with pytest.raises(Exception):
        ops.gemm_silu_bwd(gh[:100].to(dev), w_t.to(dev), gu[:100].clone())                 # not whole tiles: refused


def test_grad_scale_pick_and_scaled_rmsnorm_bwd(gpu):
    """tcavt_grad_scale_pick: S = 2^k with max|g| * S in [target / 2, target], decided on the device; all-zero and non-finite
    inputs give S = 1.  tcavt_rmsnorm_bwd applies *gy_scale to the incoming gradient: its result under the scale, divided by S,
    equals the unscaled call (the backward is linear)."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(3)
    scratch = torch.zeros(1, dtype=torch.int32, device=dev)
    scale = torch.zeros(2, device=dev)
    for amp in (3e-7, 1.0, 7.3e4):
        a = (torch.randn(64, 256, generator=g) * amp).to(torch.bfloat16).to(dev)
        b = (torch.randn(64, 256, generator=g) * amp * 0.5).to(torch.bfloat16).to(dev)
        ops.grad_scale_pick(a, b, scale, scratch)
        S, invS = scale.tolist()
        mx = max(a.float().abs().max().item(), b.float().abs().max().item())
        assert S == 2.0 ** round(torch.log2(torch.tensor(S)).item()) and abs(S * invS - 1.0) < 1e-6
        assert 128.0 <= mx * S <= 256.0, (amp, mx, S)
        assert scratch.item() == 0  # re-armed
    ops.grad_scale_pick(torch.zeros(8, 8, dtype=torch.bfloat16, device=dev), None, scale, scratch)
    assert scale.tolist() == [1.0, 1.0]
    bad = torch.ones(8, 8, dtype=torch.bfloat16, device=dev)
    bad[0, 0] = float("inf")
    ops.grad_scale_pick(bad, None, scale, scratch)
    assert scale.tolist() == [1.0, 1.0]
    # the scale enters through the RMSNorm backward; x may be fp32 or the 16-bit stream of the tape
    M, H = 40, 256
    x32 = torch.randn(M, H, generator=g)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    gy = (torch.randn(M, H, generator=g) * 1e-3).to(torch.bfloat16).to(dev)
    ops.grad_scale_pick(gy, None, scale, scratch)
    for x in (x32.to(dev), x32.to(torch.float16).to(dev)):
        g0 = torch.empty(M, H, device=dev)
        g1 = torch.empty(M, H, device=dev)
        gb = torch.empty(M, H, dtype=torch.float16, device=dev)
        ops.rmsnorm_bwd(x, gamma, gy, g0, 1e-5)
        ops.rmsnorm_bwd(x, gamma, gy, g1, 1e-5, gx_bf16=gb, gy_scale=scale[0:1])
        torch.cuda.synchronize()
        assert rel_err((g1 * scale[1]).cpu(), g0.cpu()) < 1e-6
        assert torch.equal(gb, g1.to(torch.float16)) and gb.float().abs().max().item() > 1.0   # in the half range thanks to S
        xr = x.float().cpu().clone().requires_grad_(True)   # (autograd on the values the kernel read)
        y = xr * torch.rsqrt((xr * xr).mean(-1, keepdim=True) + 1e-5) * gamma.cpu()
        (y * gy.float().cpu()).sum().backward()
        assert rel_err(g0.cpu(), xr.grad) < 1e-5


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,T,nq,nkv,lens", [(2, 40, 4, 1, [40, 23]), (3, 64, 8, 2, [64, 1, 33]), (2, 256, 32, 8, [256, 170]),
                                             (1, 300, 4, 2, [211])])
def test_attn_bwd_one_sweep_from_the_forwards_row_statistics(gpu, B, T, nq, nkv, lens, dt):
    """tcavt_attn_causal_gqa_lse leaves the log-sum-exp of every query row; with it and the forward's output the scores
    kernel of the backward needs no first sweep for the row maximum / sum / sum(P dP) (P = exp(s - lse), sum(P dP) = dO . O).
    The lse against torch.logsumexp of the masked scores; the one-sweep gradients against fp32 autograd (same bar as the
    two-sweep form) and against the two-sweep form itself."""
    from tcavt_amd import ops
    from tcavt_amd.config import LlamaShape
    from tcavt_amd.llm_backward import attn_bwd_composed
    from tcavt_amd.rope import rope_tables

    dev = gpu["device"]
    g = torch.Generator().manual_seed(17)
    ncols = (nq + 2 * nkv) * 64
    M = B * T
    qkv_p = torch.zeros(M + 64, ncols, dtype=dt, device=dev)
    qkv_p[:M] = torch.randn(M, ncols, generator=g).to(dt).to(dev)
    dO = torch.randn(M, nq * 64, generator=g).to(dt).to(dev)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=dev)
    cos, sin = (t.to(dev) for t in rope_tables(LlamaShape(), T))
    att = torch.empty(M, nq * 64, dtype=dt, device=dev)
    lse = torch.full((B * nq * T,), float("nan"), device=dev)
    ops.attn_causal_gqa(qkv_p[:M], att, kv_len, B, T, nq, nkv, 0.125, lse=lse)
    att0 = torch.empty_like(att)
    ops.attn_causal_gqa(qkv_p[:M], att0, kv_len, B, T, nq, nkv, 0.125)
    assert torch.equal(att, att0)  # (the statistics are a by-product)
    v3 = qkv_p[:M].float().view(B, T, nq + 2 * nkv, 64)
    q = v3[:, :, :nq].permute(0, 2, 1, 3)
    k = v3[:, :, nq:nq + nkv].permute(0, 2, 1, 3).repeat_interleave(nq // nkv, dim=1)
    i = torch.arange(T, device=dev)
    allowed = (i[None, :] <= i[:, None])[None] & (i[None, None, :] < kv_len.long()[:, None, None])
    s = ((q @ k.transpose(-1, -2)) * 0.125).masked_fill(~allowed[:, None], float("-inf"))
    want_lse = torch.logsumexp(s, dim=-1).reshape(-1)
    assert torch.isfinite(lse).all() and (lse - want_lse).abs().max().item() < 2e-4

    pool = {}

    def buf(name, shape, dtype, zero=False):
        if name not in pool:
            pool[name] = torch.zeros(shape, dtype=dtype, device=dev)
        return pool[name]

    two = torch.empty(M, ncols, dtype=dt, device=dev)
    attn_bwd_composed(buf, qkv_p, dO, kv_len, B, T, nq, nkv, 0.125, cos, sin, two)
    stats2 = pool["at.stats"].clone()
    one = torch.empty(M, ncols, dtype=dt, device=dev)
    pool["at.stats"].fill_(float("nan"))
    attn_bwd_composed(buf, qkv_p, dO, kv_len, B, T, nq, nkv, 0.125, cos, sin, one, lse=lse, att=att)
    stats1 = pool["at.stats"]
    # the statistics the key-major kernel reads: log(sum) + max == lse, sum(P dP) == dO . O up to the rounding of O
    lse2 = stats2[:, 0] - torch.log(stats2[:, 1])
    assert (stats1[:, 0] - lse2).abs().max().item() < 2e-4 and torch.equal(stats1[:, 1], torch.ones_like(stats1[:, 1]))
    assert rel_err(stats1[:, 2].cpu(), stats2[:, 2].cpu()) < (3e-3 if dt == torch.float16 else 2e-2)
    want32 = _attn_ref(qkv_p[:M], dO, kv_len, B, T, nq, nkv)
    want = torch.empty_like(one)
    ops.rope_bwd_pack(want32.contiguous(), want, cos, sin, (nq + nkv) * 64, T)
    for name, lo, hi in (("dq", 0, nq * 64), ("dk", nq * 64, (nq + nkv) * 64), ("dv", (nq + nkv) * 64, ncols)):
        e1 = rel_err(one[:, lo:hi].float().cpu(), want[:, lo:hi].float().cpu())
        e2 = rel_err(two[:, lo:hi].float().cpu(), want[:, lo:hi].float().cpu())
        assert e1 < (3e-3 if dt == torch.float16 else 1e-2), (name, e1, e2)
        assert e1 < 1.5 * e2 + 1e-4, (name, e1, e2)  # no worse than the form that recomputes the statistics


@pytest.mark.parametrize("dt,small_rms", [(torch.float16, False), (torch.bfloat16, False), (torch.float16, True)])
@pytest.mark.parametrize("drop", [False, True])
def test_lora_wgrad_a_one_pass_matches_the_composed_leaf(gpu, drop, dt, small_rms):
    """tcavt_lora_wgrad_a (dA of both adapters from the taped input stream: norm, masks and both products in one pass) and
    tcavt_wgrad_tn's row scale (dB from the taped un-normalised t) against the composed form they replace -- rmsnorm16, two
    mask kernels, lora_down, three wgrad_tn -- and against fp32 arithmetic on the same 16-bit operands."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(31)
    M, H, nqkv, eps, s = 200, 512, 384, 1e-5, 4.0
    # small_rms: the stream of decoder layer 0 (embedding rows, rms ~0.02 -> 1 / rms ~50) under a gradient near the top of the
    # half range: 1 / rms must meet the stream operand, not the gradient (g_t / rms leaves fp16: inf, NaN in the product)
    h16 = (torch.randn(M, H, generator=g) * (0.02 if small_rms else 1.5)).to(dt).to(dev)
    gamma = (torch.rand(H, generator=g) + 0.5).to(dev)
    g_t = torch.zeros(M, 64)
    g_t[:, :32] = torch.randn(M, 32, generator=g) * (600 if small_rms else 3)
    g_t = g_t.to(dt).to(dev)
    g_qkv = torch.randn(M, nqkv, generator=g).to(dt).to(dev)
    a_plain = torch.zeros(64, H)
    a_plain[:8] = torch.randn(8, H, generator=g) * 0.05
    a_plain[16:24] = torch.randn(8, H, generator=g) * 0.05
    a_plain = a_plain.to(dt).to(dev)
    npart = 8
    part = h16.float().pow(2).view(M, npart, H // npart).sum(-1).contiguous()
    rs = torch.rsqrt(part.sum(1) / H + eps)
    spec = (0.1, 0xABCDE, 77) if drop else None
    site_v = 91 if drop else None
    # composed leaf (the round-3 form before the fusion)
    xn = torch.empty(M, H, dtype=dt, device=dev)
    ops.rmsnorm16(h16, gamma, eps, out16=xn)
    xq, xv = xn, xn
    if drop:
        xq, xv = torch.empty_like(xn), torch.empty_like(xn)
        ops.dropout(xn, xq, *spec)
        ops.dropout(xn, xv, spec[0], spec[1], site_v)
    dA0 = torch.zeros(64, H, device=dev)
    ops.wgrad_tn(g_t, 0, 16, xq, dA0)
    ops.wgrad_tn(g_t, 16, 16, xv, dA0[16:])
    # one pass
    dA1 = torch.zeros(64, H, device=dev)
    ops.lora_wgrad_a(h16, part, gamma, g_t, dA1, eps, dropout=spec, site_v=site_v)
    # fp32 arithmetic on the same operands and masks (mask = where the composed form's dropped copy is non-zero or xn is zero)
    keep_q = (xq.float() != 0) | (xn.float() == 0)
    keep_v = (xv.float() != 0) | (xn.float() == 0)
    inv_keep = 1.0 / (1.0 - spec[0]) if drop else 1.0
    xn32 = h16.float() * rs[:, None] * gamma[None, :]
    want_q = g_t[:, :16].float().T @ (xn32 * keep_q * inv_keep)
    want_v = g_t[:, 16:32].float().T @ (xn32 * keep_v * inv_keep)
    tol = 2e-3 if dt == torch.float16 else 1.2e-2
    assert torch.isfinite(dA1).all()
    assert rel_err(dA1[:16].cpu(), want_q.cpu()) < tol and rel_err(dA1[16:32].cpu(), want_v.cpu()) < tol
    assert rel_err(dA0[:16].cpu(), want_q.cpu()) < tol  # (the composed form rounds xn instead of g_t * rs: the same class of error)
    assert rel_err(dA1[:32].cpu(), dA0[:32].cpu()) < 2 * tol and float(dA1[32:].abs().max()) == 0.0
    # dB: taped un-normalised t (as tcavt_lora_down leaves it in the forward: no row scale), scaled by rs while staged
    t_tape = torch.zeros(M, 64, dtype=dt, device=dev)
    a_cat = (a_plain.float() * gamma[None, :]).to(dt)  # the forward's operand: gain folded in
    ops.lora_down(h16, a_cat, t_tape, s, dropout=spec, site_v=site_v)
    dB1 = torch.zeros(nqkv, 64, device=dev)
    ops.wgrad_tn(t_tape, 0, 32, g_qkv, dB1, trans_out=True, rs_part=part, rs_h=H, rs_eps=eps)
    want_b = g_qkv.float().T @ (t_tape.float()[:, :32] * rs[:, None])
    assert rel_err(dB1[:, :32].cpu(), want_b.cpu()) < tol
    t_re = torch.zeros(M, 64, dtype=dt, device=dev)
    ops.lora_down(xn, a_plain, t_re, s, dropout=spec, site_v=site_v)  # the composed form's recomputed t
    dB0 = torch.zeros(nqkv, 64, device=dev)
    ops.wgrad_tn(t_re, 0, 32, g_qkv, dB0, trans_out=True)
    assert rel_err(dB1[:, :32].cpu(), dB0[:, :32].cpu()) < 3 * tol


@pytest.mark.parametrize("train_mode,front", [(True, False), (False, True)])
def test_stage_level_decoder_backward_matches_the_python_composition(gpu, monkeypatch, train_mode, front):
    """tcavt_llama_stack_backward (the whole walk through the frozen layers as one C call: dgrad GEMMs, norm / attention /
    adapter kernels, the leaf stream's weight gradients) against the per-launch Python composition of the same entry points
    (TCAVT_PY_LLM_BACKWARD=1), at a shape the fused forms serve (midi: H = 512, I = 1536, 8 / 2 heads, B = 2, L = 256), with the
    forward's LoRA dropout masks (train mode) and with the input gradient continuing into the Q-Former (front)."""
    from tcavt_amd import config, model, synth, training
    from tcavt_amd.weights import make_weights

    dev = gpu["device"]
    cfg = config.midi()
    with torch.device(dev):
        m = model.MultiModalTrajectoryModel.from_config(cfg)
    m.load_weights(make_weights(cfg, seed=5, backend="torch", device=dev))
    m.train(train_mode)
    b = synth.make_batch(cfg, 2, text_len=240, seed=21, ragged=True, min_text=100)
    g = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
    keys = ["traj_emb", "vision_emb", "lane_polygon", "lane_polygon_len", "target_traj", "norm_stat", "input_ids", "attention_mask", "labels"]
    tr = training.Trainer(m, lr=1e-4, lora_trainable=True, train_mllm_front=front)
    calls = []
    real = tr.lbw._stage_call
    monkeypatch.setattr(tr.lbw, "_stage_call", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def grads():
        m._fwd_count = 0  # (the same dropout masks in both runs)
        tr.forward_backward(*[g[k] for k in keys])
        torch.cuda.synchronize()
        return tr.book.grads.detach().clone()

    g_c = grads()
    assert calls, "the stage call must serve this shape"
    monkeypatch.setenv("TCAVT_PY_LLM_BACKWARD", "1")
    n = len(calls)
    g_py = grads()
    assert len(calls) == n
    assert torch.isfinite(g_c).all()
    lora = [nm for nm in tr.book.names if ".lora_" in nm]
    assert len(lora) == 4 * cfg.llama.layers
    for nm in tr.book.names:
        o, cnt, _ = tr.book.offsets[nm]
        a_, b_ = g_c[o:o + cnt], g_py[o:o + cnt]
        if b_.abs().max() == 0:
            assert a_.abs().max() == 0, nm
            continue
        assert rel_err(a_.cpu(), b_.cpu()) < 2e-4, nm  # (atomics' summation order in the weight gradients)
