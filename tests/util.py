"""Shared helpers for the tests (fixture loading, config/weights reconstruction)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_CASES = ["tiny_6_12_lora_ragged", "tiny_18_30_nolora_ragged", "tiny_6_30_lora_full"]


def load_case(name):
    """-> (cfg, weights dict of numpy arrays, fixture dict of numpy arrays)"""
    from tcavt_amd import config as tconfig
    from tcavt_amd.weights import make_weights

    fx = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    cfg = tconfig.PRESETS[str(fx["preset"])](seq_len=int(fx["seq_len"]), out_len=int(fx["out_len"]),
                                              use_lora=bool(fx["use_lora"]))
    return cfg, make_weights(cfg, int(fx["seed"])), fx


def batch_tensors(fx, device="cpu"):
    keys = ["traj_emb", "target_traj", "vision_emb", "lane_polygon", "lane_polygon_len", "norm_stat", "input_ids",
            "attention_mask", "labels"]
    return {k: torch.from_numpy(np.asarray(fx[k])).to(device) for k in keys}


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def scale_generation_weights(weights, text_mod_scale, out_gain):
    """The weight variant of the text-generation fixture: the text-modality embedding scaled down and the o_proj / down_proj
    weights scaled up, so that a position's hidden state is dominated by what the layers computed from the context rather
    than by the (tied) embedding of its own token -- with the plain synthetic weights the arg-max continuation of every
    prompt is one token repeated.  Applied identically by tests/golden/make_golden.py to the reference model."""
    out = dict(weights)
    out["mllm.text_modality_embedding"] = weights["mllm.text_modality_embedding"] * np.float32(text_mod_scale)
    for k, v in weights.items():
        if k.endswith("self_attn.o_proj.weight") or k.endswith("mlp.down_proj.weight"):
            out[k] = v * np.float32(out_gain)
    return out


def load_generation_case():
    """-> (fixture, cfg, weights, tensors) of tests/golden/tiny_generation.npz"""
    from tcavt_amd import config as tconfig
    from tcavt_amd.weights import make_weights

    fx = dict(np.load(os.path.join(GOLDEN, "tiny_generation.npz"), allow_pickle=False))
    cfg = tconfig.PRESETS[str(fx["preset"])](seq_len=int(fx["seq_len"]), out_len=int(fx["out_len"]), use_lora=bool(fx["use_lora"]))
    w = scale_generation_weights(make_weights(cfg, int(fx["seed"])), float(fx["gen_text_mod_scale"]), float(fx["gen_out_gain"]))
    t = {k: torch.from_numpy(np.asarray(fx[k])) for k in ("vision_emb", "input_ids", "attention_mask")}
    return fx, cfg, w, t
