"""fp16 range is part of the storage contract: a 16-bit operand beyond +-65504 becomes inf.  That must be
OBSERVABLE, not silent: every residual epilogue (o_proj / down_proj, TCAVT_EPI_NORM_OUT) forms the partial sums of squares of
the rows it writes, and a non-finite partial stores the layer's tag into a device flag word that check_flags() turns into a
FloatingPointError naming the layer (include/tcavt.h: tcavt_gemm_args.nonfinite_flag).  Reference arithmetic is fp32
(HF modeling_llama.py:317-323 residual adds) and has no such limit -- real Llama checkpoints carry far larger activations than
the synthetic N(0, 0.02)-like weights of the other tests, so the limit is driven here on purpose."""
import numpy as np
import pytest
import torch

from tests.util import batch_tensors, load_case

pytestmark = pytest.mark.gpu

CASE = "tiny_6_12_lora_ragged"


def _model(dev, scales=None):
    from tcavt_amd import model

    cfg, weights, fx = load_case(CASE)
    if scales:
        weights = dict(weights)
        for k, sc in scales.items():
            weights[k] = weights[k] * np.float32(sc)
            assert np.abs(weights[k]).max() < 65504 / 4  # the WEIGHTS stay well inside the range
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    g = {k: v.to(dev) for k, v in batch_tensors(fx).items()}
    return cfg, m, g


def _fwd(m, g, with_loss):
    kw = dict(input_ids=g["input_ids"], attention_mask=g["attention_mask"], labels=g["labels"])
    if with_loss:
        kw.update(y=g["target_traj"], norm_stat=g["norm_stat"])
    return m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], **kw)


def test_clean_model_raises_nothing_and_reports_its_range(gpu):
    cfg, m, g = _model(gpu["device"])
    with torch.no_grad():
        _fwd(m, g, True)
        torch.cuda.synchronize()
    m.mllm.check_flags()
    lw = m.mllm.llama_wrapper
    M = g["input_ids"].shape[0] * (cfg.q_num_query_tokens + g["input_ids"].shape[1])
    h16, _ = lw.norm_inputs(M, gpu["device"])
    print(f"[range] tiny synthetic model: max |residual stream| after the last layer {h16.float().abs().max().item():.2f} "
          f"(fp16 max 65504)")
    assert h16.float().abs().max().item() < 65504 / 16


def _overflow_scales(layer, site):
    """Both operands of the projection stay finite in fp16 (weights |w| < 16 376 asserted in _model; `act` ~ 3e2 and the
    attention output ~ 3e1), its OUTPUT -- a sum of 128 ... 256 such products added to the stream -- leaves the range."""
    from tcavt_amd.weights import LLAMA_PREFIX

    L = f"{LLAMA_PREFIX}layers.{layer}."
    if site == "down_proj":
        return {L + "mlp.down_proj.weight": 2e4, L + "mlp.up_proj.weight": 1e3}
    return {L + "self_attn.o_proj.weight": 2e4, L + "self_attn.v_proj.weight": 1e2}


@pytest.mark.parametrize("layer,site", [(0, "down_proj"), (1, "o_proj"), (1, "down_proj")])
@pytest.mark.parametrize("with_loss", [True, False])
def test_overflowing_layer_is_named(gpu, layer, site, with_loss):
    cfg, m, g = _model(gpu["device"], _overflow_scales(layer, site))
    with torch.no_grad():
        out = _fwd(m, g, with_loss)
        torch.cuda.synchronize()
    res = out[0] if with_loss else out
    assert not torch.isfinite(res.float()).all()   # nothing is clamped silently: the result itself is not finite ...
    with pytest.raises(FloatingPointError) as ei:   # ... and the flag says where it started
        m.mllm.check_flags()
    msg = str(ei.value)
    assert f"layer {layer}" in msg and site in msg, msg
    m.mllm.check_flags()  # cleared by the check
    # the same model is usable again with in-range weights
    with torch.no_grad():
        m.load_weights(load_case(CASE)[1], device=gpu["device"])
        out = _fwd(m, g, with_loss)
        torch.cuda.synchronize()
    m.mllm.check_flags()
    assert torch.isfinite((out[0] if with_loss else out).float()).all()


def test_evaluate_model_checks_the_flag(gpu):
    """evaluate_model ends with check_flags(): an overflow in ANY batch of the loop is reported (no loss to go non-finite on
    the y=None branch it runs)."""
    from tcavt_amd import evaluate

    cfg, m, g = _model(gpu["device"], _overflow_scales(0, "down_proj"))
    with pytest.raises(FloatingPointError):
        evaluate.evaluate_model(m, [g], num_candidates=1)
