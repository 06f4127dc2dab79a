"""Train-mode forward of the WHOLE model on the HIP path against the oracle drawing the same Philox masks
(oracle.forward.dropout_tapes: same seed, same per-module site numbering as model.DropoutCtx).  The oracle's site
sequence is pinned to the reference's own train-mode forward on CPU (tests/test_oracle_train_golden.py::
test_dropout_sites_match_reference_train_mode), so agreement here means every dropout of ddp_model.train()
(scripts/train.py:1152) sits where the reference has it -- a misplaced or missing site moves the output by O(10 %)."""
import pytest
import torch

from tests.util import batch_tensors, load_case, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["tiny_6_12_lora_ragged", "tiny_18_30_nolora_ragged"])
def test_train_mode_forward_matches_oracle_with_same_masks(gpu, name):
    from oracle import forward as O
    from tcavt_amd import model

    dev = gpu["device"]
    cfg, weights, fx = load_case(name)
    t = batch_tensors(fx)
    g = {k: v.to(dev) for k, v in t.items()}
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).train()
    m._fwd_count = 0
    seed = m.dropout_seed
    with torch.no_grad():
        loss, dec = m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], y=g["target_traj"],
                      norm_stat=g["norm_stat"], input_ids=g["input_ids"], attention_mask=g["attention_mask"])
        torch.cuda.synchronize()
        ex = {}
        loss_o, dec_o = O.model_forward(weights, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"],
                                        t["lane_polygon_len"], t["input_ids"], t["attention_mask"], y=t["target_traj"],
                                        norm_stat=t["norm_stat"], contract="fp16", extras=ex, dropout_seed=seed)
        _, dec_e = O.model_forward(weights, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                                   t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                                   contract="fp16")
    e_fh = rel_err(m.last.final_hidden.cpu(), ex["final_hidden"])
    e_poly = rel_err(m.last.poly_emb.cpu(), ex["poly_emb"])
    e_dec = rel_err(dec.cpu(), dec_o)
    spread = rel_err(dec_e, dec_o)  # what dropout does to the output: the scale a misplaced site would show up at
    print(f"[train-mode {name}] vs oracle with the same masks: poly_emb {e_poly:.2e}, final_hidden {e_fh:.2e}, decoded {e_dec:.2e} "
          f"(dropout moves decoded by {spread:.2e})")
    assert e_poly < 1e-4 and e_fh < 3e-3 and e_dec < 3e-3
    assert spread > 10 * e_dec
    assert abs(loss.item() - loss_o.item()) < 1e-2 * abs(loss_o.item())
