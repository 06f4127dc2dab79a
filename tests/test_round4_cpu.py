"""CPU-side checks of round 4's host logic and oracle additions (no GPU):
  * SyntheticTokenizer's batch form (the call surface of the reference's tokenizer branches, scripts/train.py:557, 590-598)
  * the oracle's scaled-stream contract (tcavt_llama_stack_args.stream_scale restated): a power-of-two scale changes nothing
    on the clean model and rescues an outlier model the plain fp16 contract loses
  * set_storage's argument checks."""
import numpy as np
import pytest
import torch

from tests.util import batch_tensors, load_case, rel_err


def test_synthetic_tokenizer_batch_form_pads_right_and_matches_the_single_form():
    from tcavt_amd.synth import SyntheticTokenizer

    tok = SyntheticTokenizer(vocab=512)
    texts = ["A1: vehicle 7 drives in lane A2.", "speed 33.5 km/h", ""]
    enc = tok(texts, return_tensors="pt", padding=True, truncation=True)
    ids, mask = enc["input_ids"], enc["attention_mask"]
    assert ids.dtype == torch.long and ids.shape == mask.shape and ids.shape[0] == 3
    for i, t in enumerate(texts):
        one = tok(t, return_tensors="pt")["input_ids"][0]
        n = one.numel()
        assert torch.equal(ids[i, :n], one) and int(mask[i].sum()) == n
        assert (ids[i, n:] == tok.pad_token_id).all() and (mask[i, n:] == 0).all()
        assert (mask[i, :n] == 1).all()  # right padded: a prefix of ones (what mask_to_kvlen requires)
    assert ids.shape[1] == max(len(tok.encode(t)) for t in texts)
    with pytest.raises(ValueError):
        tok(texts, return_tensors="pt")  # ragged rows without padding
    assert tok(texts, padding=True, truncation=True, max_length=3)["input_ids"].shape[1] == 3


def _oracle(weights, cfg, t, contract):
    from oracle import forward as O

    ex = {}
    with torch.no_grad():
        _, dec = O.model_forward(weights, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                                 t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                                 contract=contract, extras=ex)
    return dec, ex["final_hidden"]


def test_oracle_scaled_stream_contract():
    from tcavt_amd.weights import plant_outliers

    cfg, w, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    scaled = {"default": "fp16", "gamma": "fp32", "stream_scale": 2.0 ** -4}
    d32, _ = _oracle(w, cfg, t, "fp32")
    d16, _ = _oracle(w, cfg, t, "fp16")
    d16s, _ = _oracle(w, cfg, t, scaled)
    assert rel_err(d16s, d16) < 3e-4 and rel_err(d16s, d32) < 1e-3   # clean model: the scale is rounding noise
    wo = plant_outliers(w, cfg, 2.0e5)
    o32, f32 = _oracle(wo, cfg, t, "fp32")
    o16, _ = _oracle(wo, cfg, t, "fp16")
    o16s, f16s = _oracle(wo, cfg, t, scaled)
    assert not torch.isfinite(o16).all()                                # plain fp16: the stream leaves the range
    assert rel_err(o16s, o32) < 1e-3 and rel_err(f16s, f32) < 2e-3      # held at 2^-4 it does not
    # the planted channels really are the stream's outliers
    H = cfg.llama.hidden
    v = np.asarray(wo["mllm.text_modality_embedding"]).reshape(-1)
    assert abs(v[H // 5]) == 2.0e5 and np.abs(np.delete(v, [H // 5, (3 * H) // 4 + 1])).max() < 10


def test_set_storage_argument_checks():
    from tcavt_amd import config, model

    cfg = config.PRESETS["tiny"](seq_len=6, out_len=12, use_lora=True)
    m = model.MultiModalTrajectoryModel.from_config(cfg)
    with pytest.raises(ValueError):
        m.set_storage(torch.float32)
    with pytest.raises(ValueError):
        m.set_storage(torch.float16, stream_scale=0.3)   # not a power of two
    with pytest.raises(ValueError):
        m.set_storage(torch.float16, stream_scale=2.0)
    m.set_storage("auto")
    assert m.storage == torch.float16 and m._auto_range == "pending" and m.mllm.llama_wrapper.stream_scale == 1.0
    m.set_storage(torch.float16, stream_scale=2.0 ** -6, wide_stream=True)
    lw = m.mllm.llama_wrapper
    assert lw.stream_scale == 2.0 ** -6 and lw.wide_stream and not lw.stream16 and m._auto_range is None
    m.set_storage(torch.bfloat16)
    assert not lw.wide_stream and lw.stream_scale == 1.0 and not lw.stream16
