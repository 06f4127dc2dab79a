"""Training step (scripts/train.py:1168-1183) on the HIP path vs torch autograd on the CPU oracle.

The oracle graph (oracle/forward.py, fp32) with requires_grad on exactly the parameters train.py
trains (everything outside `mllm`, train.py:1140-1145) gives reference gradients; the HIP backward
(tcavt_amd/backward.py) must reproduce them.  Tolerances: fp32-only parameters (lane-polygon encoder,
LTSF front / self-attention / post-MLP / fusion / head) see the forward's bf16 cross-attention only
through small perturbations of their inputs; the cross-attention projections themselves run their
backward contractions in bf16 MFMA -> a few 1e-3 relative (Frobenius) per tensor is the contract.
"""
import numpy as np
import pytest
import torch

from tests.util import batch_tensors, load_case, rel_err

pytestmark = pytest.mark.gpu


def _oracle_grads(cfg, weights, t, contract="fp32"):
    from oracle import forward as O
    from tcavt_amd.weights import trainable_keys

    W = {k: torch.from_numpy(v).clone() for k, v in weights.items()}
    names = trainable_keys(W)
    for k in names:
        W[k].requires_grad_(True)
    loss, _ = O.model_forward(W, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                              t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                              contract=contract)
    loss.backward()
    return loss.item(), {k: W[k].grad for k in names}


@pytest.mark.parametrize("name", ["tiny_6_12_lora_ragged", "tiny_18_30_nolora_ragged"])
def test_gradients_match_autograd(gpu, name):
    from tcavt_amd import model, training

    dev = gpu["device"]
    cfg, weights, fx = load_case(name)
    t = batch_tensors(fx)
    ref_loss, ref = _oracle_grads(cfg, weights, t)
    _, ref16 = _oracle_grads(cfg, weights, t, contract="fp16")  # casts are straight-through for autograd
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    tr = training.Trainer(m)
    g = {k: v.to(dev) for k, v in t.items()}
    loss, _ = tr.forward_backward(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"],
                                  g["target_traj"], g["norm_stat"], g["input_ids"], g["attention_mask"], g["labels"])
    torch.cuda.synchronize()
    assert abs(loss.item() - ref_loss) / ref_loss < 1e-2
    assert set(tr.book.g) == set(ref)
    e16, e32, c32 = [], [], []
    for k, gref in ref.items():
        got = tr.book.g[k].cpu()
        assert torch.isfinite(got).all(), k
        if gref.abs().max() == 0:
            assert got.abs().max().item() == 0, k
            continue
        e16.append((rel_err(got, ref16[k]), k))
        e32.append(rel_err(got, gref))
        c32.append(rel_err(ref16[k], gref))
    e16.sort(reverse=True)
    med16 = float(np.median([e for e, _ in e16]))
    print(f"[grads {name}] HIP vs bf16-contract autograd: max {e16[0][0]:.2e} ({e16[0][1]}), median {med16:.2e}; "
          f"HIP vs fp32 autograd: max {max(e32):.2e}, median {float(np.median(e32)):.2e}; "
          f"bf16-contract autograd vs fp32 autograd: max {max(c32):.2e}, median {float(np.median(c32)):.2e}")
    # End-to-end the comparison is ill-conditioned on these tiny cases: the HIP forward and the oracle's
    # bf16 forward are two different realisations of bf16 rounding noise (they differ by ~4e-3 on the
    # final hidden states, DESIGN.md "Precision contract"); a handful of ReLU / LayerNorm inputs near zero
    # then flip, and the discrepancy grows along the backward chain (1e-3 at the head, up to ~1e-1 at the
    # lane-polygon encoder, measured).  The rigorous checks are the stage-level tests below (identical
    # inputs: <= 2e-4 fp32, <= 5e-3 bf16).  Here: every tensor within 50 %, and the whole flat gradient
    # within 10 % of both autograd references (cosine >= 0.995).
    names = [k for k, g in ref.items() if g.abs().max() > 0]
    for e, k in e16:
        assert e < 0.5, (k, e)
    flat = lambda d: torch.cat([d[k].reshape(-1).double() for k in names])
    got_flat = torch.cat([tr.book.g[k].cpu().reshape(-1).double() for k in names])
    for refd, nm in ((ref16, "bf16-contract"), (ref, "fp32")):
        r = flat(refd)
        rel = ((got_flat - r).norm() / r.norm()).item()
        cos = (got_flat @ r / (got_flat.norm() * r.norm())).item()
        print(f"[grads {name}] flat gradient vs {nm} autograd: rel {rel:.2e}, cosine {cos:.5f}")
        assert rel < 0.1 and cos > 0.995


def test_polygon_encoder_backward_is_exact_fp32(gpu):
    """fp32-only stage: hand-written backward vs autograd through the oracle, identical forward -> tight."""
    from oracle import forward as O
    from tcavt_amd import model, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    W = {k: torch.from_numpy(v).clone() for k, v in weights.items()}
    names = [k for k in W if k.startswith("lane_polygon_encoder.")]
    for k in names:
        W[k].requires_grad_(True)
    emb = O.lane_polygon_encoder(W, cfg, t["lane_polygon"], t["lane_polygon_len"])
    g_emb = torch.randn(emb.shape, generator=torch.Generator().manual_seed(9))
    (emb * g_emb).sum().backward()
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    tr = training.Trainer(m)
    with torch.no_grad():
        m.lane_polygon_encoder(t["lane_polygon"].to(dev), t["lane_polygon_len"].to(dev))
        tr.book.grads.zero_()
        tr.bw.polygon(g_emb.to(dev))
    torch.cuda.synchronize()
    for k in names:
        e = rel_err(tr.book.g[k].cpu(), W[k].grad)
        assert e < 2e-4, (k, e)


def test_ltsf_backward_from_fixed_hidden_states(gpu):
    """TransformerLTSF backward alone, final hidden states given: HIP (bf16 cross-attention forward AND
    backward contractions) vs autograd through the oracle's bf16-contract graph from identical inputs."""
    from oracle import forward as O
    from tcavt_amd import model, ops, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    B = t["traj_emb"].shape[0]
    gen = torch.Generator().manual_seed(10)
    L, H = 48, cfg.llama.hidden
    fh = torch.randn(B, L, H, generator=gen)
    poly = torch.randn(B, cfg.lane_polygon_d_model, generator=gen)
    W = {k: torch.from_numpy(v).clone() for k, v in weights.items()}
    names = [k for k in W if k.startswith("ltsf.")]
    for k in names:
        W[k].requires_grad_(True)
    poly_r = poly.clone().requires_grad_(True)
    out = O.ltsf_forward(W, cfg, t["traj_emb"], poly_r, fh, O._rounder("fp16")) + t["traj_emb"][:, :, -1:]
    dp, dg = O.denorm(out, t["norm_stat"]), O.denorm(t["target_traj"], t["norm_stat"])
    loss = torch.nn.functional.mse_loss(dp[:, 0], dg[:, 0]) + torch.nn.functional.mse_loss(dp[:, 1], dg[:, 1])
    loss.backward()

    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    tr = training.Trainer(m)
    with torch.no_grad():
        x = t["traj_emb"].to(dev)
        fh_b = torch.zeros(B * L + 64, H, dtype=m.storage, device=dev)
        ops.cast16(fh.to(dev).view(B * L, H), out=fh_b)
        dec = m.ltsf(x, poly.to(dev), fh.to(dev), final_hidden_bf16=fh_b, _fuse_last_residual=True)
        tr.book.grads.zero_()
        tr.bw._poly_emb, tr.bw._fh_b, tr.bw._L = poly.to(dev), fh_b, L
        g_out = torch.empty_like(dec)
        ops.mse_grad(dec, t["target_traj"].to(dev), t["norm_stat"].to(dev), g_out, B, cfg.out_len)
        g_poly = tr.bw.ltsf(g_out, x)
    torch.cuda.synchronize()
    errs = sorted(((rel_err(tr.book.g[k].cpu(), W[k].grad), k) for k in names), reverse=True)
    e_poly = rel_err(g_poly.cpu(), poly_r.grad)
    print(f"[ltsf grads] max {errs[0][0]:.2e} ({errs[0][1]}), median {float(np.median([e for e, _ in errs])):.2e}, "
          f"g_poly_emb {e_poly:.2e}")
    for e, k in errs:
        assert e < 3e-2, (k, e)
    assert float(np.median([e for e, _ in errs])) < 5e-3 and e_poly < 2e-2


def test_adamw_matches_torch(gpu):
    from tcavt_amd import ops

    dev = gpu["device"]
    g_ = torch.Generator().manual_seed(1)
    n = 10007
    p0 = torch.randn(n, generator=g_)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=5e-4, weight_decay=1e-4)
    p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g_) * 10
        ref.grad = grad.clone()
        opt.step()
        ops.adamw(p, (grad * 2).to(dev), m, v, 5e-4, 0.9, 0.999, 1e-8, 1e-4, step, grad_scale=0.5)
    assert (p.cpu() - ref.detach()).abs().max().item() < 1e-6


def test_training_step_reduces_loss(gpu):
    """A few real optimisation steps on one batch: loss must go down, shadows must be refreshed."""
    from tcavt_amd import model, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    tr = training.Trainer(m, lr=5e-4)
    g = {k: v.to(dev) for k, v in t.items()}
    losses = []
    for _ in range(8):
        loss, _ = tr.step(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"],
                          g["norm_stat"], g["input_ids"], g["attention_mask"], g["labels"])
        losses.append(loss.item())
    print("[train] losses:", [f"{l:.1f}" for l in losses])
    assert losses[-1] < losses[0]
    frozen = m.mllm.q_proj.weight.detach().cpu()
    assert torch.equal(frozen, torch.from_numpy(weights["mllm.q_proj.weight"]))  # MLLM untouched


# ---------------------------------------------------------------------------------------------------
# train-mode dropout: the backward regenerates the forward's Philox masks (seed, site) -> gradients of the
# SAME stochastic graph; the oracle draws identical masks (oracle.forward.DropTape / oracle.philox)
# ---------------------------------------------------------------------------------------------------
def test_mha_bwd_with_attention_dropout(gpu):
    from oracle import philox
    from tcavt_amd import ops

    dev = gpu["device"]
    B, nh, Lq, Lk, dh, p, seed, site = 3, 2, 10, 12, 8, 0.25, 77, 5
    g = torch.Generator().manual_seed(3)
    E = nh * dh
    q, k, v = (torch.randn(B, L_, E, generator=g).requires_grad_(True) for L_ in (Lq, Lk, Lk))
    go = torch.randn(B, Lq, E, generator=g)
    keep = torch.from_numpy(philox.keep_mask(B * nh * Lq * Lk, p, seed, site).reshape(B, nh, Lq, Lk).astype("float32"))
    qh, kh, vh = (t.view(B, -1, nh, dh).transpose(1, 2) for t in (q, k, v))
    P = torch.softmax(qh @ kh.transpose(-1, -2) / dh ** 0.5, -1) * keep / (1 - p)
    ((P @ vh).transpose(1, 2).reshape(B, Lq, E) * go).sum().backward()
    d = lambda t: t.detach().to(dev).contiguous()
    gq, gk, gv = (torch.empty_like(d(t)) for t in (q, k, v))
    ops.mha_bwd(d(q).view(-1, E), d(k).view(-1, E), d(v).view(-1, E), d(go).view(-1, E), gq.view(-1, E), gk.view(-1, E),
                gv.view(-1, E), B, Lq, Lk, nh, dh, 1.0 / dh ** 0.5, dropout=(p, seed, site))
    for a, b, nm in ((gq, q.grad, "q"), (gk, k.grad, "k"), (gv, v.grad, "v")):
        assert rel_err(a.cpu(), b) < 2e-5, nm


def test_polygon_encoder_backward_with_dropout(gpu):
    from oracle import forward as O
    from tcavt_amd import model, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    W = {k: torch.from_numpy(v).clone() for k, v in weights.items()}
    names = [k for k in W if k.startswith("lane_polygon_encoder.")]
    for k in names:
        W[k].requires_grad_(True)
    seed = 1234
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    enc = m.lane_polygon_encoder
    emb = O.lane_polygon_encoder(W, cfg, t["lane_polygon"], t["lane_polygon_len"], drop=O.DropTape(seed, enc.dropout_p))
    emb_eval = O.lane_polygon_encoder(W, cfg, t["lane_polygon"], t["lane_polygon_len"])
    assert rel_err(emb.detach(), emb_eval.detach()) > 1e-2  # the masks really bite
    g_emb = torch.randn(emb.shape, generator=torch.Generator().manual_seed(9))
    (emb * g_emb).sum().backward()
    tr = training.Trainer(m)
    with torch.no_grad():
        enc.dctx = model.DropoutCtx(seed)
        out = enc(t["lane_polygon"].to(dev), t["lane_polygon_len"].to(dev))
        enc.dctx = None
        tr.book.grads.zero_()
        tr.bw.polygon(g_emb.to(dev))
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), emb.detach()) < 1e-4  # same stochastic forward
    for k in names:
        e = rel_err(tr.book.g[k].cpu(), W[k].grad)
        assert e < 3e-4, (k, e)


def test_ltsf_backward_with_dropout(gpu):
    from oracle import forward as O
    from tcavt_amd import model, ops, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    B = t["traj_emb"].shape[0]
    gen = torch.Generator().manual_seed(10)
    L, H = 48, cfg.llama.hidden
    fh = torch.randn(B, L, H, generator=gen)
    poly = torch.randn(B, cfg.lane_polygon_d_model, generator=gen)
    W = {k: torch.from_numpy(v).clone() for k, v in weights.items()}
    names = [k for k in W if k.startswith("ltsf.")]
    for k in names:
        W[k].requires_grad_(True)
    poly_r = poly.clone().requires_grad_(True)
    seed = 4321
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    out = O.ltsf_forward(W, cfg, t["traj_emb"], poly_r, fh, O._rounder("fp16"),
                         drop=O.DropTape(seed, m.ltsf.dropout_p)) + t["traj_emb"][:, :, -1:]
    dp, dg = O.denorm(out, t["norm_stat"]), O.denorm(t["target_traj"], t["norm_stat"])
    loss = torch.nn.functional.mse_loss(dp[:, 0], dg[:, 0]) + torch.nn.functional.mse_loss(dp[:, 1], dg[:, 1])
    loss.backward()
    tr = training.Trainer(m)
    with torch.no_grad():
        x = t["traj_emb"].to(dev)
        fh_b = torch.zeros(B * L + 64, H, dtype=m.storage, device=dev)
        ops.cast16(fh.to(dev).view(B * L, H), out=fh_b)
        m.ltsf.dctx = m.ltsf.attn_block.dctx = model.DropoutCtx(seed)
        dec = m.ltsf(x, poly.to(dev), fh.to(dev), final_hidden_bf16=fh_b, _fuse_last_residual=True)
        m.ltsf.dctx = m.ltsf.attn_block.dctx = None
        tr.book.grads.zero_()
        tr.bw._poly_emb, tr.bw._fh_b, tr.bw._L = poly.to(dev), fh_b, L
        g_out = torch.empty_like(dec)
        ops.mse_grad(dec, t["target_traj"].to(dev), t["norm_stat"].to(dev), g_out, B, cfg.out_len)
        g_poly = tr.bw.ltsf(g_out, x)
    torch.cuda.synchronize()
    assert rel_err(dec.cpu(), out.detach()) < 5e-3  # same masks in both forwards
    errs = sorted(((rel_err(tr.book.g[k].cpu(), W[k].grad), k) for k in names), reverse=True)
    e_poly = rel_err(g_poly.cpu(), poly_r.grad)
    print(f"[ltsf grads, dropout] max {errs[0][0]:.2e} ({errs[0][1]}), median "
          f"{float(np.median([e for e, _ in errs])):.2e}, g_poly_emb {e_poly:.2e}")
    for e, k in errs:
        assert e < 3e-2, (k, e)
    assert float(np.median([e for e, _ in errs])) < 5e-3 and e_poly < 2e-2


def test_training_step_in_train_mode(gpu):
    """ddp_model.train() semantics (train.py:1160): dropout active in forward AND backward.  Per seed the step is
    reproducible (bit-equal loss; gradients up to the float-atomic order of the weight-gradient reductions), a different
    seed gives a different stochastic graph, and optimisation makes progress."""
    from tcavt_amd import model, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    g = {k: v.to(dev) for k, v in t.items()}
    args = (g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"], g["norm_stat"],
            g["input_ids"], g["attention_mask"], g["labels"])

    def first_step(seed):
        m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).train()
        m.dropout_seed = seed
        tr = training.Trainer(m, lr=1e-4)
        loss, _ = tr.forward_backward(*args)
        torch.cuda.synchronize()
        return loss.item(), tr.book.grads.clone(), tr

    la, ga, _ = first_step(11)
    lb, gb, tr = first_step(11)
    lc, gc, _ = first_step(12)
    # same seed: same masks in forward and backward (the scalar loss is a float-atomic sum over samples: last-bit noise)
    assert abs(la - lb) <= 1e-6 * abs(la) and rel_err(ga.cpu(), gb.cpu()) < 1e-5
    assert abs(lc - la) > 1e-3 * abs(la) and rel_err(gc.cpu(), ga.cpu()) > 1e-2  # another seed: another graph
    m_eval = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    le, _ = training.Trainer(m_eval).forward_backward(*args)
    assert abs(le.item() - la) > 1e-3 * abs(la)                     # and train mode is not the eval arithmetic
    losses = [la]
    tr.optimizer_step()
    for _ in range(8):
        loss, _ = tr.step(*args)
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and min(losses[4:]) < losses[0]


def test_step_with_qformer_prefetch_is_equivalent(gpu):
    """Trainer.step(..., next_vision_embs=...) only re-orders when the frozen Q-Former of the next batch runs."""
    from tcavt_amd import model, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    g = {k: v.to(dev) for k, v in t.items()}
    args = (g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"], g["norm_stat"],
            g["input_ids"], g["attention_mask"], g["labels"])

    def run(prefetch):
        m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
        tr = training.Trainer(m, lr=1e-4)
        losses = []
        for _ in range(4):
            loss, _ = tr.step(*args, next_vision_embs=g["vision_emb"] if prefetch else None)
            losses.append(loss.item())
        torch.cuda.synchronize()
        return losses, tr.book.params.clone()

    la, pa = run(False)
    lb, pb = run(True)
    assert abs(la[0] - lb[0]) <= 1e-6 * abs(la[0])  # same forward (the loss is a float-atomic sum over the samples)
    # later steps: equal up to the float-atomic summation order of the weight gradients, which the first AdamW step
    # (lr * g / (|g| + eps): a sign decision where g ~ 0) and the 16-bit forward amplify step by step -- two runs WITHOUT
    # prefetch differ by the same amount.  (With bf16 storage the forward's coarser rounding hid it on step 2.)
    assert np.allclose(la[:2], lb[:2], rtol=2e-4) and np.allclose(la, lb, rtol=5e-3) and rel_err(pa.cpu(), pb.cpu()) < 1e-3


def test_dropout_epoch_shifts_the_seed(gpu):
    """tcavt_set_dropout_epoch: a device-resident value added to every dropout seed at run time (fresh masks under hipGraph
    replay).  Mask with (seed, epoch e) == the oracle's mask of seed + e; epoch cleared -> the plain seed again."""
    from oracle import philox
    from tcavt_amd import ops

    dev = gpu["device"]
    n, p, seed, site = 40001, 0.1, 0xFEEDFACE12345, 5
    x = torch.ones(n, device=dev)
    epoch = torch.zeros(1, dtype=torch.int64, device=dev)
    try:
        ops.set_dropout_epoch(epoch)
        for e in (0, 1, 2):
            y = torch.empty_like(x)
            ops.dropout(x, y, p, seed, site)
            keep = torch.from_numpy(philox.keep_mask(n, p, seed + e, site)).to(dev)
            assert torch.equal(y != 0, keep)
            ops.dropout_epoch_advance(epoch)
        assert epoch.item() == 3
    finally:
        ops.set_dropout_epoch(None)
    y = torch.empty_like(x)
    ops.dropout(x, y, p, seed, site)
    assert torch.equal(y != 0, torch.from_numpy(philox.keep_mask(n, p, seed, site)).to(dev))


def test_captured_step_replays_like_eager_steps(gpu):
    """Trainer.capture: the whole train.py step as a hipGraph.  Eval arithmetic: N replays land where N eager steps land
    (optimizer step count on the device: the bias corrections follow the replays).  Train mode: every replay draws fresh
    masks (losses differ at lr = 0), and the epoch is released afterwards."""
    from tcavt_amd import model, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    g = {k: v.to(dev) for k, v in t.items()}
    args = (g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"], g["norm_stat"],
            g["input_ids"], g["attention_mask"], g["labels"])

    def make(train, lr):
        m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev)
        m.train(train)
        return m, training.Trainer(m, lr=lr)

    m1, tr1 = make(False, 1e-4)
    losses1 = [tr1.step(*args)[0].item() for _ in range(5)]
    m2, tr2 = make(False, 1e-4)
    graph, (gl, gd) = tr2.capture(*args)       # (capture's warm-up is step 1)
    losses2 = []
    for _ in range(4):
        graph.replay()
        torch.cuda.synchronize()
        losses2.append(gl.item())
    tr2.release_graph()
    assert tr2.optimizer_counters() == (5, 0)
    assert np.allclose(losses1[1:], losses2, rtol=5e-3)
    assert rel_err(tr2.book.params.cpu(), tr1.book.params.cpu()) < 1e-3
    m3, tr3 = make(True, 0.0)
    graph3, (gl3, _) = tr3.capture(*args)
    seen = []
    for _ in range(3):
        graph3.replay()
        torch.cuda.synchronize()
        seen.append(gl3.item())
    tr3.release_graph()
    assert len(set(seen)) == 3 and all(np.isfinite(seen))  # lr = 0: only the masks change from replay to replay


def test_pipelined_decoder_is_equivalent(gpu):
    """model.pipeline_decoder (the frozen MLLM pass on a stream of its own, next step's decoder over this step's head /
    backward / AdamW): same results as the one-stream order.  Two different batches alternate and no host sync happens
    between the steps, so the two output slots and the cross-stream events are all exercised; the frozen MLLM's output
    must be BIT-equal step by step, the trained parameters equal up to the float-atomic noise of the weight gradients."""
    from tcavt_amd import model, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    ga = {k: v.to(dev) for k, v in t.items()}
    gen = torch.Generator().manual_seed(11)
    perm = torch.randperm(t["traj_emb"].shape[0], generator=gen)
    gb = {k: v[perm].clone().to(dev) for k, v in t.items()}
    gb["vision_emb"] = gb["vision_emb"] * 0.5 + 0.1
    gb["input_ids"] = (gb["input_ids"] * 5 + 1) % cfg.llama.vocab

    def run(pipeline):
        m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).train()
        tr = training.Trainer(m, lr=1e-4)
        assert m.pipeline_decoder  # the frozen-MLLM variant turns it on
        m.pipeline_decoder = pipeline
        hid, dec, losses = [], [], []
        for i in range(6):
            g = ga if i % 2 == 0 else gb
            loss, d = tr.step(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"],
                              g["norm_stat"], g["input_ids"], g["attention_mask"], g["labels"],
                              inputs_ready=True if pipeline else None)
            hid.append(m.last.final_hidden_bf16.clone())
            dec.append(d.clone())
            losses.append(loss)
        torch.cuda.synchronize()
        m.mllm.check_flags()
        return hid, dec, [l.item() for l in losses], tr.book.params.clone()

    ha, da, la, pa = run(False)
    hb, db, lb, pb = run(True)
    for i in range(6):
        assert torch.equal(ha[i], hb[i]), f"step {i}: the frozen MLLM's output differs"
    assert torch.equal(da[0], db[0])
    assert np.allclose(la, lb, rtol=5e-3) and rel_err(pa.cpu(), pb.cpu()) < 1e-3
    assert not torch.equal(ha[0], ha[1])  # (the two batches do differ)


def test_data_parallel_stream_layout_stand_in(gpu, monkeypatch):
    """The stream layout of a data-parallel rank (all-reduce launched from the bucket's own stream, side channels folded onto
    two streams) with a stand-in for RCCL's stream (TCAVT_FAKE_DP: an in-place x1.0 kernel where the collective would run):
    same losses and parameters as the single-process layout, pipelined decoder included."""
    from tcavt_amd import model, streams, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    g = {k: v.to(dev) for k, v in t.items()}
    args = (g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"], g["norm_stat"],
            g["input_ids"], g["attention_mask"], g["labels"])

    def run(fake):
        if fake:
            monkeypatch.setenv("TCAVT_FAKE_DP", "1")
        else:
            monkeypatch.delenv("TCAVT_FAKE_DP", raising=False)
        m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
        tr = training.Trainer(m, lr=1e-4)
        assert (tr._fake_dp is not None) == fake
        losses = [tr.step(*args, next_vision_embs=g["vision_emb"], inputs_ready=True)[0] for _ in range(4)]
        torch.cuda.synchronize()
        return [l.item() for l in losses], tr.book.params.clone()

    try:
        la, pa = run(False)
        lb, pb = run(True)
        assert streams._ACTIVE == 2
    finally:
        streams._ACTIVE = None
    assert abs(la[0] - lb[0]) <= 1e-6 * abs(la[0])
    assert np.allclose(la, lb, rtol=5e-3) and rel_err(pa.cpu(), pb.cpu()) < 1e-3


@pytest.mark.parametrize("train", [False, True])
def test_cross_attn_backward_stage_matches_python_composition(gpu, train, monkeypatch):
    """tcavt_cross_attn_backward (one C call, single stream) against the per-launch Python composition on leaf streams
    (TCAVT_PY_TLAYERS=1): same in_proj weight / bias gradients and the same gradients upstream of the queries."""
    from tcavt_amd import model, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    g = {k: v.to(dev) for k, v in t.items()}
    args = (g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"], g["norm_stat"],
            g["input_ids"], g["attention_mask"], g["labels"])

    def run(py):
        if py:
            monkeypatch.setenv("TCAVT_PY_TLAYERS", "1")
        else:
            monkeypatch.delenv("TCAVT_PY_TLAYERS", raising=False)
        m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev)
        m.train(train)
        m._fwd_count = 0
        tr = training.Trainer(m, lr=1e-4)
        tr.forward_backward(*args)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in tr.book.g.items()}

    ga, gb = run(False), run(True)
    keys = [k for k in ga if "cross_attn.in_proj" in k or "dec_proj" in k or "post_mlp" in k or "lane_fc" in k]
    assert len(keys) >= 6
    for k in keys:
        assert rel_err(ga[k].cpu(), gb[k].cpu()) < 1e-4, k
    H = ga["ltsf.decoder.cross_attn.in_proj_bias"].numel() // 3
    assert ga["ltsf.decoder.cross_attn.in_proj_bias"][H:2 * H].abs().max().item() == 0.0  # key bias: softmax-invariant


@pytest.mark.parametrize("train", [False, True])
def test_ltsf_backward_stage_matches_python_composition(gpu, train, monkeypatch):
    """tcavt_ltsf_backward (SURVEY 8b ltsf_backward: one C call per phase, single stream) and tcavt_tlayer_stack_backward
    (the lane-polygon encoder's layers) against the per-launch Python composition on leaf streams (TCAVT_PY_TLAYERS=1): every
    gradient of the trainable set."""
    from tcavt_amd import backward, model, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    g = {k: v.to(dev) for k, v in t.items()}
    args = (g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"], g["norm_stat"],
            g["input_ids"], g["attention_mask"], g["labels"])
    used, used_poly = [], []
    orig = backward.Backward._ltsf_stage
    monkeypatch.setattr(backward.Backward, "_ltsf_stage", lambda self, *a, **k: (used.append(1), orig(self, *a, **k))[1])
    orig_p = backward.Backward._polygon_stage

    def poly_stage(self, *a, **k):
        r = orig_p(self, *a, **k)
        used_poly.append(bool(r))
        return r

    monkeypatch.setattr(backward.Backward, "_polygon_stage", poly_stage)

    def run(py):
        if py:
            monkeypatch.setenv("TCAVT_PY_TLAYERS", "1")
        else:
            monkeypatch.delenv("TCAVT_PY_TLAYERS", raising=False)
        m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev)
        m.train(train)
        m._fwd_count = 0
        tr = training.Trainer(m, lr=1e-4)
        tr.forward_backward(*args)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in tr.book.g.items()}

    ga = run(False)
    assert used == [1] and used_poly == [True]  # both stages ran (tcavt_ltsf_backward, tcavt_tlayer_stack_backward)
    gb = run(True)
    assert used == [1] and used_poly == [True, False]  # ... and the Python composition did not go through them
    assert len(ga) > 40
    for k in ga:  # same kernels on the same operands; the atomically accumulated reductions (LayerNorm / bias sums) differ in order
        assert rel_err(ga[k].cpu(), gb[k].cpu()) < 1e-4, k


def test_loss_trajectory_matches_torch_adamw_on_the_oracle(gpu):
    """Several whole optimisation steps (scripts/train.py:1168-1183: zero_grad, forward, backward, AdamW) on one batch, HIP
    path against torch autograd + torch.optim.AdamW on the CPU oracle from the same weights: the loss of every step and
    the trained parameters after the last one must agree.  AdamW's first updates are lr * sign(gradient), so this is a
    size-independent check that gradient signs, bias corrections, weight decay and the parameter refresh (16-bit shadows
    of the trained projections) all line up over consecutive steps -- not only for a single backward."""
    from oracle import forward as O
    from tcavt_amd import model, training
    from tcavt_amd.weights import trainable_keys

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    steps, lr, wd = 4, 2e-4, 1e-4  # (the loss falls 8x in these four steps; a fifth, at the bottom of the valley, tracks to 2e-3)
    # ---- reference: the oracle graph under autograd, torch's AdamW
    W = {k: torch.from_numpy(v).clone() for k, v in weights.items()}
    names = trainable_keys(W)
    for k in names:
        W[k].requires_grad_(True)
    opt = torch.optim.AdamW([W[k] for k in names], lr=lr, weight_decay=wd, betas=(0.9, 0.999), eps=1e-8)
    ref_losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss, _ = O.model_forward(W, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                                  t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                                  contract="fp32")
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    # ---- HIP path
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    tr = training.Trainer(m, lr=lr, weight_decay=wd)
    g = {k: v.to(dev) for k, v in t.items()}
    losses = []
    for _ in range(steps):
        loss, _ = tr.step(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"],
                          g["norm_stat"], g["input_ids"], g["attention_mask"], g["labels"])
        losses.append(loss.item())
    torch.cuda.synchronize()
    rel = [abs(a - b) / abs(b) for a, b in zip(losses, ref_losses)]
    print("[trajectory] HIP", [f"{x:.2f}" for x in losses], "torch", [f"{x:.2f}" for x in ref_losses], "rel", [f"{r:.1e}" for r in rel])
    assert ref_losses[-1] < ref_losses[0]
    # step 0 is the forward alone; afterwards every loss sits on the previous AdamW updates, whose first steps are
    # lr * sign-like (m / sqrt(v) ~ +-1): a gradient entry near zero that lands on the other side under the atomics'
    # summation order moves a parameter by 2 lr, so the later losses scatter from run to run (seen: 1.5e-4 ... 1.1e-3 at step 3)
    assert rel[0] < 2e-4 and max(rel[:3]) < 1e-3 and max(rel) < 5e-3
    # trained parameters: the distance HIP <-> torch is a small fraction of the distance either moved from the start
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    moved = torch.cat([(W[k].detach() - torch.from_numpy(weights[k])).reshape(-1) for k in names]).double()
    apart = torch.cat([(sd[k].float() - W[k].detach()).reshape(-1) for k in names]).double()
    print(f"[trajectory] parameters moved {moved.norm():.3e}, HIP vs torch {apart.norm():.3e}")
    assert apart.norm() < 0.1 * moved.norm()
