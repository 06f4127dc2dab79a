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
