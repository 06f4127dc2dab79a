"""Pins the CPU oracle (oracle/forward.py) to the fixtures produced by the reference's own
modules (tests/golden/make_golden.py).  fp32 both sides; differences are summation order only."""
import numpy as np
import pytest
import torch

from tests.util import MODEL_CASES, batch_tensors, load_case, rel_err


@pytest.mark.parametrize("name", MODEL_CASES)
def test_oracle_matches_reference_fixture(name):
    from oracle import forward as O

    cfg, weights, fx = load_case(name)
    t = batch_tensors(fx)
    extras = {}
    with torch.no_grad():
        loss, decoded = O.model_forward(weights, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"],
                                        t["lane_polygon_len"], t["input_ids"], t["attention_mask"],
                                        y=t["target_traj"], norm_stat=t["norm_stat"], contract="fp32",
                                        extras=extras)
        W = O.as_torch(weights)
        img = O.qformer(W, cfg, t["vision_emb"], O._rounder("fp32"))
    assert rel_err(extras["poly_emb"], fx["exp_poly_emb"]) < 2e-5
    assert rel_err(img, fx["exp_img_tokens"]) < 2e-5
    assert rel_err(extras["final_hidden"], fx["exp_final_hidden"]) < 5e-5
    assert np.abs(decoded.numpy() - fx["exp_decoded"]).max() < 2e-5
    assert abs(loss.item() - float(fx["exp_loss"])) / float(fx["exp_loss"]) < 1e-4


def test_oracle_padded_rows_follow_hf_semantics():
    """Padded query rows are computed (causal AND key-valid) and feed the unmasked cross-attention
    (train.py:798): the fixture's final_hidden at padded positions must match too."""
    from oracle import forward as O

    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    mask = t["attention_mask"]
    assert (mask == 0).any()
    extras = {}
    with torch.no_grad():
        O.model_forward(weights, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                        t["input_ids"], mask, extras=extras)
    nq = cfg.q_num_query_tokens
    pad = torch.cat([torch.zeros(mask.shape[0], nq, dtype=torch.bool), mask == 0], dim=1)
    got = extras["final_hidden"][pad]
    exp = torch.from_numpy(fx["exp_final_hidden"])[pad]
    assert torch.isfinite(got).all()
    assert rel_err(got, exp) < 5e-5


def test_oracle_bf16_contract_is_close_to_fp32():
    """The bf16-contract mode is the same graph with bf16 rounding points: it must stay within a
    few 1e-3 (relative, Frobenius) of the fp32 reference result on the decoded trajectories."""
    from oracle import forward as O

    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    with torch.no_grad():
        d16 = O.model_forward(weights, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"],
                              t["lane_polygon_len"], t["input_ids"], t["attention_mask"], contract="bf16")
    assert rel_err(d16, fx["exp_decoded"]) < 2e-2


def test_metrics_known_answers():
    """Hand-computable ADE/FDE/RMSE/min-over-K (formulas: train.py:1318-1321, test.py:1365-1372)."""
    from oracle import forward as O

    # one sample, K=2, To=2, unit ranges offset by (10, 20): errors are 3-4-5 triangles
    gt = torch.tensor([[[0.0, 0.0], [0.0, 0.0]]])                      # [B=1,2,To=2]
    pred = torch.tensor([[[[3.0, 6.0], [4.0, 8.0]],                     # k=0: err = 5, 10
                          [[0.0, 0.6], [0.0, 0.8]]]])                   # k=1: err = 0, 1
    ns = torch.tensor([[10.0, 11.0, 20.0, 21.0]])
    m = O.traj_metrics(pred, gt, ns)
    assert torch.allclose(m["ade"], torch.tensor([[7.5, 0.5]]), atol=1e-5)
    assert torch.allclose(m["fde"], torch.tensor([[10.0, 1.0]]))
    assert m["ade_argmin"].tolist() == [1] and m["fde_argmin"].tolist() == [1]
    assert abs(m["rmse"][0, 1].item() - (1.0 / 4) ** 0.5) < 1e-6
    assert abs(m["ade_sum"] - 0.5) < 1e-6
