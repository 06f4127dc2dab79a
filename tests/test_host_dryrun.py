"""Host-logic dry run on CPU: the whole model.forward call sequence is executed against a stub
library that accepts every call, so the Python side's buffer shapes, leading dimensions and
argument orders go through the wrappers' host-side guards (ops._need) without a GPU.
No arithmetic happens here; numerics are the GPU tests' job."""
import ctypes

import pytest
import torch

from tests.util import batch_tensors, load_case


class _StubLib:
    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        if not name.startswith("tcavt_"):
            raise AttributeError(name)

        def fn(*args):
            self.calls.append(name)
            return 0

        return fn


@pytest.fixture
def dry(monkeypatch):
    from tcavt_amd import ops

    stub = _StubLib()
    monkeypatch.setattr(ops, "lib", lambda: stub)
    monkeypatch.setattr(ops, "stream_ptr", lambda: None)
    monkeypatch.setattr(ops, "_ALLOW_CPU", True)
    return stub


@pytest.mark.parametrize("name", ["tiny_6_12_lora_ragged", "tiny_18_30_nolora_ragged"])
def test_forward_call_sequence_passes_host_guards(dry, name):
    from tcavt_amd import model

    cfg, weights, fx = load_case(name)
    t = batch_tensors(fx)
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights).eval()
    with torch.no_grad():
        loss, decoded = m(t["traj_emb"], t["vision_emb"], None, t["lane_polygon"], t["lane_polygon_len"],
                          y=t["target_traj"], norm_stat=t["norm_stat"], input_ids=t["input_ids"],
                          attention_mask=t["attention_mask"], labels=t["labels"])
    assert decoded.shape == (t["traj_emb"].shape[0], 2, cfg.out_len)
    # the whole decoder stack is ONE C call (tcavt_llama_stack_forward issues every layer's launches itself)
    assert dry.calls.count("tcavt_llama_stack_forward") == 1
    assert dry.calls.count("tcavt_attn_causal_gqa") == 0 and dry.calls.count("tcavt_rmsnorm") == 0
    assert dry.calls.count("tcavt_embed_fuse") == 1


def test_guard_catches_short_buffer(dry):
    from tcavt_amd import capi, ops

    table = torch.zeros(16, 64, dtype=torch.bfloat16)
    ids = torch.zeros(2, 5, dtype=torch.int64)
    img = torch.zeros(2 * 3, 64)
    h_ok = torch.zeros(2 * 8, 64)
    flag = torch.zeros(1, dtype=torch.int32)
    ops.embed_fuse(table, ids, img, torch.zeros(64), torch.zeros(64), h_ok, flag)
    with pytest.raises(capi.TcavtError, match="kernel needs"):
        ops.embed_fuse(table, ids, img, torch.zeros(64), torch.zeros(64), torch.zeros(2 * 7, 64), flag)


def test_embed_fuse_extent_with_flattened_image_tokens(monkeypatch):
    """The one GPU memory-access fault on record (round 1, gpurun_out/k2.log): ops.embed_fuse took Nq from img.shape[1]
    while the model hands the image tokens over flattened as [B*Nq, H], so Nq became H and the kernel's grid covered
    B*(H+Lt) rows of `h` (allocated B*(Nq+Lt)) and B*H rows of `img` (allocated B*Nq).  Nq must come from the element
    count, and both extents are guarded on the host."""
    from tcavt_amd import capi, ops

    seen = []

    class Stub:
        def tcavt_embed_fuse(self, *a):
            seen.append(a)
            return 0

    monkeypatch.setattr(ops, "lib", lambda: Stub())
    monkeypatch.setattr(ops, "stream_ptr", lambda: None)
    monkeypatch.setattr(ops, "_ALLOW_CPU", True)
    B, Nq, Lt, H, V = 2, 16, 5, 64, 32
    table, ids = torch.zeros(V, H, dtype=torch.bfloat16), torch.zeros(B, Lt, dtype=torch.int64)
    mods, flag = torch.zeros(H), torch.zeros(1, dtype=torch.int32)
    h = torch.zeros(B * (Nq + Lt), H)
    for img in (torch.zeros(B * Nq, H), torch.zeros(B, Nq, H)):  # flattened (what the model passes) and 3-D
        ops.embed_fuse(table, ids, img, mods, mods, h, flag)
        assert seen[-1][6:11] == (B, Nq, Lt, H, V)  # B, Nq, Lt, H, V as the kernel sees them: Nq = 16, not H
    with pytest.raises(capi.TcavtError, match="kernel needs"):  # h sized for fewer image tokens than img holds
        ops.embed_fuse(table, ids, torch.zeros(B * Nq, H), mods, mods, torch.zeros(B * (Nq + Lt) - 1, H), flag)


@pytest.mark.parametrize("lora_trainable", [False, True])
def test_training_step_call_sequence_passes_host_guards(dry, lora_trainable):
    """train.py step (and its LoRA-trainable variant, modify_scripts/modify_train.py:512-528) end to end against the stub:
    every wrapper's host-side guard sees the real buffer shapes of forward, both backwards and the optimizer."""
    from tcavt_amd import model, training

    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights).eval()
    tr = training.Trainer(m, lora_trainable=lora_trainable, max_grad_norm=1.0 if lora_trainable else None)
    loss, decoded = tr.step(t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"], t["target_traj"],
                            t["norm_stat"], t["input_ids"], t["attention_mask"], t["labels"])
    assert decoded.shape == (t["traj_emb"].shape[0], 2, cfg.out_len)
    L = cfg.llama.layers
    # the LoRA-trainable loop gates the update on a finite loss on the device (modify_train.py:1190-1196)
    assert dry.calls.count("tcavt_adamw_gated" if lora_trainable else "tcavt_adamw") == 1
    assert dry.calls.count("tcavt_adamw" if lora_trainable else "tcavt_adamw_gated") == 0
    assert dry.calls.count("tcavt_attn_bwd_scores") == (L if lora_trainable else 0)
    assert dry.calls.count("tcavt_silu_mul_bwd") == (L if lora_trainable else 0)
    assert dry.calls.count("tcavt_attn_bwd_dkv") == (L if lora_trainable else 0)
    assert dry.calls.count("tcavt_rope_bwd_pack") == (L if lora_trainable else 0)
    assert dry.calls.count("tcavt_rmsnorm_bwd") == (2 * L if lora_trainable else 0)  # final norm + two per layer, none below layer 0
    assert dry.calls.count("tcavt_llama_stack_forward") == 1
    if lora_trainable:
        assert m.mllm.llama_wrapper.tape is not None and len(m.mllm.llama_wrapper.tape.layers) == L


def test_lora_trainable_needs_adapters(dry):
    from tcavt_amd import model, training

    cfg, weights, _ = load_case("tiny_18_30_nolora_ragged")
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights).eval()
    with pytest.raises(ValueError, match="no LoRA adapters"):
        training.Trainer(m, lora_trainable=True)
