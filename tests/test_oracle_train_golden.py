"""The oracle's BACKWARD and TRAIN-MODE structure pinned to the reference's own modules (CPU, no GPU).

tests/golden/<case>_train.npz (tests/golden/make_golden.py: run_train_case) holds, from the reference model itself:
  * loss.backward() gradients of the train.py trainable set (scripts/train.py:1140-1145) and of the LoRA adapters
    (modify_scripts/modify_train.py:512-528) in eval arithmetic -- strided samples + norms per tensor;
  * the parameters after one torch.optim.AdamW(lr 5e-4, wd 1e-4) step (train.py:1145,1182-1183);
  * the (kind, p, shape) sequence of dropout calls of one train-mode forward (train.py:1152).
Autograd through oracle/forward.py is what every HIP gradient test compares against, and oracle.forward.DropTape is what
places the HIP path's dropout sites in the oracle -- both are builder-written, so they are held to the reference here.
"""
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN, batch_tensors, load_case

CASES = ["tiny_6_12_lora_ragged", "tiny_18_30_nolora_ragged"]


def _sample(t, cap=512):
    flat = t.detach().reshape(-1)
    stride = -(-flat.numel() // cap)
    return flat[::stride].to(torch.float32).numpy()


def _load(name):
    cfg, weights, fx = load_case(name)
    tr = dict(np.load(os.path.join(GOLDEN, name + "_train.npz"), allow_pickle=False))
    return cfg, weights, fx, tr


def _oracle_backward(cfg, weights, t, lora_too):
    from oracle import forward as O

    W = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in weights.items()}
    want = [k for k in W if not k.startswith("mllm.") or (lora_too and ".lora_" in k)]
    for k in want:
        W[k].requires_grad_(True)
    loss, _ = O.model_forward(W, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                              t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                              contract="fp32")
    loss.backward()
    return W, want, loss


@pytest.mark.parametrize("name", CASES)
def test_oracle_gradients_match_reference_autograd(name):
    cfg, weights, fx, tr = _load(name)
    t = batch_tensors(fx)
    W, want, loss = _oracle_backward(cfg, weights, t, lora_too=bool(tr["use_lora"]))
    assert abs(loss.item() - float(tr["loss"])) <= 2e-6 * abs(float(tr["loss"]))
    trainable = [str(k) for k in tr["trainable"]]
    assert sorted(trainable) == sorted(k for k in want if ".lora_" not in k)  # train.py:1140-1145: all outside mllm
    gmax = max(float(tr["gnorm." + k]) for k in trainable)
    worst = (0.0, None)
    for k in trainable:
        g, ref, nrm = W[k].grad, tr["grad." + k], float(tr["gnorm." + k])
        assert abs(g.double().norm().item() - nrm) <= 2e-4 * nrm + 1e-7 * gmax, k
        # fp32 summation-order noise only: relative to the tensor's own scale
        scale = max(nrm / np.sqrt(g.numel()), 1e-30)
        err = float(np.abs(_sample(g) - ref).max()) / scale
        worst = max(worst, (err, k))
        assert err < 5e-3, (k, err)
    print(f"[oracle grads {name}] worst sampled deviation {worst[0]:.2e} x rms ({worst[1]})")
    if bool(tr["use_lora"]):
        for k in (k for k in want if ".lora_" in k):
            ref = torch.from_numpy(tr["grad." + k])
            g = W[k].grad
            assert g.shape == ref.shape
            assert (g - ref).norm().item() <= 2e-4 * ref.norm().item() + 1e-12, k


@pytest.mark.parametrize("name", CASES)
def test_adamw_step_matches_reference_optimizer(name):
    """torch.optim.AdamW over the oracle's leaf tensors with the oracle's gradients lands where the reference model's
    optimizer lands (same hyper-parameters as train.py:1145)."""
    cfg, weights, fx, tr = _load(name)
    t = batch_tensors(fx)
    W, want, _ = _oracle_backward(cfg, weights, t, lora_too=False)
    trainable = [str(k) for k in tr["trainable"]]
    before = {k: W[k].detach().clone() for k in trainable}
    opt = torch.optim.AdamW([W[k] for k in trainable], lr=5e-4, weight_decay=1e-4)
    opt.step()
    n_bad = n_all = 0
    for k in trainable:
        got, ref, g = _sample(W[k]), tr["adamw." + k], np.abs(_sample(W[k].grad))
        # the first AdamW step moves an element by lr * g / (|g| + 1e-8): where |g| is far above eps this is lr * sign(g)
        # and agrees to fp32 rounding; where the gradient is numerically zero (e.g. the key bias of an attention, whose
        # exact gradient is 0: softmax is invariant to it) two fp32 summation orders give different signs / fractions of
        # a step -- elements below 1e-3 of the tensor's largest gradient are only held to the two-step bound below
        sig = g > max(1e-3 * float(g.max()), 1e-12)
        bad = (np.abs(got - ref) > 1e-6 + 1e-6 * np.abs(ref)) & sig
        n_bad += int(bad.sum())
        n_all += int(sig.sum())
        assert np.abs(got - ref).max() <= 2.0 * 5e-4 * 1.001 + 1e-6, k  # never further than the two possible steps apart
        assert not np.array_equal(got, _sample(before[k])) or float(tr["gnorm." + k]) == 0.0, k
    assert n_all > 1000 and n_bad <= max(3, int(1e-3 * n_all)), (n_bad, n_all)


@pytest.mark.parametrize("name", CASES)
def test_dropout_sites_match_reference_train_mode(name):
    """The oracle's DropTapes (whose (seed, site) numbering the HIP kernels share) visit, module by module, the same
    dropout sites as the reference's train-mode forward: same count, order, probability and element count."""
    from oracle import forward as O

    cfg, weights, fx, tr = _load(name)
    t = batch_tensors(fx)
    ex = {}
    with torch.no_grad():
        O.model_forward(weights, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                        t["input_ids"], t["attention_mask"], contract="fp32", extras=ex, dropout_seed=7)
    tapes = ex["tapes"]
    got = []
    for mod in ("poly", "qf", "lora", "ltsf"):  # the reference's execution order (train.py:914-940)
        if hasattr(tapes[mod], "log"):
            got += [(p, int(np.prod(shape))) for p, shape in tapes[mod].log]
    ref = list(zip(tr["drop_p"].tolist(), tr["drop_numel"].tolist()))
    assert len(got) == len(ref), (len(got), len(ref))
    for i, (g, r_) in enumerate(zip(got, ref)):
        assert abs(g[0] - r_[0]) < 1e-12 and g[1] == r_[1], (i, g, r_, tr["drop_shape"][i].tolist())
    # per-adapter LoRA dropout: two sites per decoder layer (q_proj, v_proj), as PEFT instantiates them
    n_lora = len(tapes["lora"].log) if hasattr(tapes["lora"], "log") else 0
    assert n_lora == (2 * cfg.llama.layers if cfg.use_lora else 0)


@pytest.mark.parametrize("name", CASES)
def test_oracle_front_end_gradients_match_reference_autograd(name):
    """The MLLM's front end (Q-Former, mllm.q_proj, modality embeddings) is trainable in modify_scripts/modify_train.py
    (:523-528 freeze only the non-LoRA Llama weights): the oracle's autograd for those parameters against the reference
    model's own loss.backward() (fixture entry "front")."""
    from oracle import forward as O

    cfg, weights, fx, tr = _load(name)
    t = batch_tensors(fx)
    front = [str(k) for k in tr["front"]]
    assert front and all(k.startswith(("mllm.qformer.", "mllm.q_proj.")) or "modality_embedding" in k for k in front)
    W = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in weights.items()}
    assert sorted(front) == sorted(k for k in W if k.startswith(("mllm.qformer.", "mllm.q_proj.")) or
                                   k in ("mllm.vision_modality_embedding", "mllm.text_modality_embedding"))
    for k in front:
        W[k].requires_grad_(True)
    loss, _ = O.model_forward(W, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                              t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"], contract="fp32")
    loss.backward()
    gmax = max(float(tr["gnorm." + k]) for k in front)
    worst = (0.0, None)
    for k in front:
        g, ref, nrm = W[k].grad, tr["grad." + k], float(tr["gnorm." + k])
        assert abs(g.double().norm().item() - nrm) <= 5e-4 * nrm + 1e-6 * gmax, k
        scale = max(nrm / np.sqrt(g.numel()), 1e-6 * gmax / np.sqrt(g.numel()), 1e-30)
        err = float(np.abs(_sample(g) - ref).max()) / scale
        worst = max(worst, (err, k))
        assert err < 5e-3, (k, err)  # (fp32 summation-order noise only, as for the train.py set)
    print(f"[oracle front-end grads {name}] worst sampled deviation {worst[0]:.2e} x rms ({worst[1]})")
