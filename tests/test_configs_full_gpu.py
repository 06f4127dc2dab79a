"""BASELINE.json configurations beyond configs[1], at their FULL sizes on one MI355X (round-2 verdict: they had only run through
tools/):

  configs[4] "test_10.py inference-only, batch 512, hipGraph-captured decode"  -> test_config5_*   (SURVEY 8d "Config 5":
             T_in 6 / T_out 12, L = 256, B = 512; scripts/test_10.py:1301-1342 K-candidate loop, :1409-1410 lengths)
  configs[3] "ablation_study_without_lora.py ... MFMA-util sweep"               -> test_config4_*   (SURVEY 8d "Config 4":
             no adapters, T_in 6 / T_out 30, corners of the B x L sweep: B = 64, L = 512 and B = 8, L = 128;
             scripts/ablation_study_without_lora.py:413-421,1257-1258)

At these sizes the oracle runs on two samples (seconds); the whole batch is held to size-independent properties (finite,
bit-reproducible, padded ids cannot reach valid rows, samples independent, hipGraph replay == eager, min over K candidates
<= any one candidate).
"""
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _fwd(m, g, with_loss=False):
    kw = dict(input_ids=g["input_ids"], attention_mask=g["attention_mask"], labels=g.get("labels"))
    if with_loss:
        kw.update(y=g["target_traj"], norm_stat=g["norm_stat"])
    return m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], **kw)


def _build(dev, seq_len, out_len, use_lora, host_weights):
    from tcavt_amd import config, model
    from tcavt_amd.weights import make_weights

    cfg = config.PRESETS["llama32_1b"](seq_len=seq_len, out_len=out_len, use_lora=use_lora)
    torch.set_num_threads(16)
    W = make_weights(cfg, seed=1, backend="torch", device="cpu" if host_weights else dev)
    with torch.device(dev):
        m = model.MultiModalTrajectoryModel.from_config(cfg)
    m.load_weights(W).eval()
    return cfg, (W if host_weights else None), m


def _batch(cfg, B, text_len, seed, dev=None):
    from tcavt_amd import synth

    b = synth.make_batch(cfg, B, text_len=text_len, seed=seed, ragged=True, min_text=max(1, text_len // 2))
    t = {k: torch.from_numpy(v) for k, v in b.items()}
    return t if dev is None else {k: v.to(dev) for k, v in t.items()}


def _properties(m, cfg, g, tag):
    """The size-independent property set of tests/test_configs_gpu.py::test_config2_full_size_properties."""
    dev = g["traj_emb"].device
    B, Lt, Nq = g["input_ids"].shape[0], g["input_ids"].shape[1], cfg.q_num_query_tokens
    with torch.no_grad():
        loss1, dec1 = _fwd(m, g, with_loss=True)
        fh1 = m.last.final_hidden.clone()
        loss2, dec2 = _fwd(m, g, with_loss=True)
        torch.cuda.synchronize()
        m.mllm.check_flags()
        assert dec1.shape == (B, 2, cfg.out_len) and fh1.shape == (B, Nq + Lt, cfg.llama.hidden)
        assert torch.isfinite(dec1).all() and torch.isfinite(fh1.float()).all() and torch.isfinite(loss1)
        assert torch.equal(dec1, dec2) and torch.equal(loss1, loss2) and torch.equal(fh1, m.last.final_hidden)
        mask = g["attention_mask"].bool()
        assert (~mask).any()
        g2 = dict(g)
        g2["input_ids"] = torch.where(mask, g["input_ids"], (g["input_ids"] + 12345) % cfg.llama.vocab)
        _fwd(m, g2)
        valid = torch.cat([torch.ones(B, Nq, dtype=torch.bool, device=dev), mask], dim=1)
        fh2 = m.last.final_hidden
        assert torch.equal(fh1[valid], fh2[valid]) and not torch.equal(fh1[~valid], fh2[~valid])
        g3 = dict(g)
        am = g["attention_mask"].clone()
        am[3, Lt // 3:] = 0
        g3["attention_mask"] = am
        dec3 = _fwd(m, g3)
        keep = torch.arange(B, device=dev) != 3
        assert torch.equal(dec3[keep], dec1[keep]) and not torch.equal(dec3[3], dec1[3])
        g4 = {k: v[:2].contiguous() for k, v in g.items()}
        dec4 = _fwd(m, g4)
        e = rel_err(dec4.cpu(), dec1[:2].cpu())
        print(f"[{tag}] B=2 vs rows 0-1 of B={B}: rel {e:.2e}")
        assert e < 1e-3
    return loss1, dec1


def _two_samples_vs_oracle(m, cfg, W, t, dev, tag):
    from oracle import forward as O

    t2 = {k: v[:2].contiguous() for k, v in t.items()}
    g2 = {k: v.to(dev) for k, v in t2.items()}
    with torch.no_grad():
        dec = _fwd(m, g2).float().cpu()
        torch.cuda.synchronize()
        _, d32 = O.model_forward(W, cfg, t2["traj_emb"], t2["vision_emb"], t2["lane_polygon"], t2["lane_polygon_len"],
                                 t2["input_ids"], t2["attention_mask"], y=t2["target_traj"], norm_stat=t2["norm_stat"], contract="fp32")
    mg, m32 = (O.traj_metrics(d, t2["target_traj"], t2["norm_stat"]) for d in (dec, d32))
    e = rel_err(dec, d32)
    ade = abs(mg["ade_sum"] - m32["ade_sum"]) / m32["ade_sum"]
    fde = abs(mg["fde_sum"] - m32["fde_sum"]) / m32["fde_sum"]
    print(f"[{tag} vs fp32 oracle] decoded {e:.2e}, ADE {ade:.2e}, FDE {fde:.2e}")
    assert e < 1e-3 and ade < 1e-3 and fde < 1e-3  # BASELINE.json's bar against the reference's fp32 arithmetic


# ------------------------------------------------------------------------------------------------
# configs[4] / "Config 5": test_10.py, B = 512, T 6 -> 12, L = 256, LoRA model
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config5(gpu):
    dev = gpu["device"]
    cfg, _, m = _build(dev, 6, 12, True, host_weights=False)
    g = _batch(cfg, 512, 240, seed=3, dev=dev)
    yield cfg, m, g
    del m, g
    torch.cuda.empty_cache()


@pytest.mark.timeout(900)
def test_config5_b512_properties(gpu, config5):
    cfg, m, g = config5
    _properties(m, cfg, g, "config 5, B=512")


@pytest.mark.timeout(900)
def test_config5_b512_hipgraph_replay_bit_equal_eager(gpu, config5):
    """The inference pass of the test_10.py loop body at B = 512 captured once and replayed on new buffer contents."""
    cfg, m, g = config5
    dev = gpu["device"]
    static = {k: v.clone() for k, v in g.items()}
    other = _batch(cfg, 512, 240, seed=4, dev=dev)
    with torch.no_grad():
        _fwd(m, static, with_loss=True)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            _fwd(m, static, with_loss=True)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g_loss, g_dec = _fwd(m, static, with_loss=True)
        for batch in (other, g, other):
            for k in static:
                static[k].copy_(batch[k])
            graph.replay()
            torch.cuda.synchronize()
            got_loss, got_dec = g_loss.clone(), g_dec.clone()
            e_loss, e_dec = _fwd(m, batch, with_loss=True)
            torch.cuda.synchronize()
            assert torch.equal(got_dec, e_dec) and torch.equal(got_loss, e_loss)
    m.mllm.check_flags()
    del graph


@pytest.mark.timeout(1200)
def test_config5_b512_k10_candidates(gpu, config5):
    """test_10.py:1301-1382 at its own size: K = 10 train-mode passes under no_grad, min over the candidates.  The dropout
    seed of a pass is (model seed + pass count), so with the counter reset the K = 1 run IS candidate 0 of the K = 10 run:
    min over K <= that single pass, metric by metric.  With the MLLM's dropout sites off, the shared-MLLM-pass form
    (reuse_prefix) must reproduce the K full passes."""
    from tcavt_amd import evaluate

    cfg, m, g = config5
    m._fwd_count = 0
    r1 = evaluate.evaluate_model(m, [g], num_candidates=1, mc_dropout=True)
    m._fwd_count = 0
    r10 = evaluate.evaluate_model(m, [g], num_candidates=10, mc_dropout=True)
    r_eval = evaluate.evaluate_model(m, [g], num_candidates=1)
    print(f"[config 5] K=1 (MC) {r1}  K=10 (MC) {r10}  K=1 eval {r_eval}")
    assert r10["n"] == r1["n"] == 512
    for k in ("ADE", "FDE", "RMSE"):
        assert r10[k] <= r1[k] * (1 + 1e-6) and r10[k] > 0 and r10[k] == r10[k]
    assert r10["ADE"] < r1["ADE"]  # ten different candidates: the minimum is strictly better somewhere
    qd, ld = m.mllm.qformer.dropout_p, m.mllm.llama_wrapper.lora_dropout
    try:
        m.mllm.qformer.dropout_p, m.mllm.llama_wrapper.lora_dropout = 0.0, 0.0
        m._fwd_count = 0
        full = evaluate.evaluate_model(m, [g], num_candidates=10, mc_dropout=True)
        m._fwd_count = 0
        shared = evaluate.evaluate_model(m, [g], num_candidates=10, mc_dropout=True, reuse_prefix=True)
    finally:
        m.mllm.qformer.dropout_p, m.mllm.llama_wrapper.lora_dropout = qd, ld
    print(f"[config 5] K=10, MLLM without dropout: ten full passes {full}  shared MLLM pass {shared}")
    for k in ("ADE", "FDE", "RMSE"):
        assert abs(full[k] - shared[k]) <= 1e-6 * abs(full[k])


# ------------------------------------------------------------------------------------------------
# configs[3] / "Config 4": ablation_study_without_lora.py, corners of the B x L sweep
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config4(gpu):
    dev = gpu["device"]
    cfg, W, m = _build(dev, 6, 30, False, host_weights=True)
    yield cfg, W, m
    del m
    torch.cuda.empty_cache()


@pytest.mark.timeout(1500)
def test_config4_nolora_b64_L512(gpu, config4):
    cfg, W, m = config4
    dev = gpu["device"]
    t = _batch(cfg, 64, 496, seed=7)   # L = 16 + 496 = 512, the longest fused sequence the reference can build (train.py:235-238)
    g = {k: v.to(dev) for k, v in t.items()}
    assert not m.mllm.llama_wrapper.use_lora
    _properties(m, cfg, g, "config 4, no LoRA, B=64 L=512")
    _two_samples_vs_oracle(m, cfg, W, t, dev, "config 4, B=64 L=512")


@pytest.mark.timeout(900)
def test_config4_nolora_b8_L128(gpu, config4):
    cfg, W, m = config4
    dev = gpu["device"]
    t = _batch(cfg, 8, 112, seed=8)    # L = 128
    g = {k: v.to(dev) for k, v in t.items()}
    _properties(m, cfg, g, "config 4, no LoRA, B=8 L=128")
    _two_samples_vs_oracle(m, cfg, W, t, dev, "config 4, B=8 L=128")
