"""Checkpoint key remap (reference PEFT layout <-> package layout) round-trips and loads strictly."""
import torch


def test_peft_roundtrip_and_strict_load():
    from tcavt_amd import checkpoint, config, model
    from tcavt_amd.weights import make_weights

    cfg = config.tiny()
    w = make_weights(cfg, 3)
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(w)
    sd = m.state_dict()
    ref = checkpoint.to_reference(sd, peft=True)
    q = "mllm.llama_wrapper.llama_model.base_model.model.model.layers.0.self_attn.q_proj"
    assert q + ".base_layer.weight" in ref and q + ".lora_A.default.weight" in ref and q + ".lora_B.default.weight" in ref
    assert "mllm.llama_wrapper.llama_model.base_model.model.model.layers.0.self_attn.k_proj.weight" in ref
    assert "mllm.llama_wrapper.llama_model.base_model.model.lm_head.weight" in ref
    assert "mllm.qformer.query_tokens" in ref and "ltsf.pos_encoding" in ref
    back = checkpoint.from_reference(ref)
    assert set(back) == set(sd)
    for k in sd:
        assert torch.equal(back[k], sd[k])
    # bare MLLM dict, as train.py:1137-1138 loads it into `.mllm`
    mllm_ref = {k[len("mllm."):]: v for k, v in ref.items() if k.startswith("mllm.")}
    m2 = model.MultiModalTrajectoryModel.from_config(cfg)
    m2.mllm.load_state_dict(checkpoint.from_reference(mllm_ref), strict=True)
    assert torch.equal(m2.mllm.q_proj.weight, m.mllm.q_proj.weight)


def test_no_lora_layout_drops_adapters():
    from tcavt_amd import checkpoint, config, model
    from tcavt_amd.weights import make_weights

    cfg = config.tiny()
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(make_weights(cfg, 3))
    plain = checkpoint.to_reference(m.state_dict(), peft=False)
    assert not any("lora" in k for k in plain)
    cfg0 = config.tiny(use_lora=False)
    m0 = model.MultiModalTrajectoryModel.from_config(cfg0)
    m0.load_state_dict(plain, strict=True)
    # and a PEFT checkpoint loads into the no-LoRA model the way adjust_state_dict does it
    peft = checkpoint.to_reference(m.state_dict(), peft=True)
    m0.load_state_dict(checkpoint.from_reference(peft, keep_lora=False), strict=True)
