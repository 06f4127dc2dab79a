"""RoPE cos/sin tables for the decoder (host side, fp32).

HF computes ``inv_freq`` once on the host and cos/sin per forward in fp32
(modeling_llama.py:113-160; rope_type "llama3": modeling_rope_utils
``_compute_llama3_parameters``).  Positions are ``arange(L)`` regardless of padding
(modeling_llama.py:386-389), so the tables depend on L only; they are built on the host
once per L and uploaded ([L][head_dim/2] fp32 each), never recomputed on the device.
"""
import math

import torch


def llama3_inv_freq(ll):
    d = ll.head_dim
    inv = 1.0 / (ll.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    if ll.rope_factor is None or ll.rope_factor == 1.0:
        return inv
    low_wl = ll.rope_original_max_pos / ll.rope_low_freq_factor
    high_wl = ll.rope_original_max_pos / ll.rope_high_freq_factor
    wl = 2 * math.pi / inv
    inv_l = torch.where(wl > low_wl, inv / ll.rope_factor, inv)
    smooth = (ll.rope_original_max_pos / wl - ll.rope_low_freq_factor) / (
        ll.rope_high_freq_factor - ll.rope_low_freq_factor)
    smoothed = (1 - smooth) * inv_l / ll.rope_factor + smooth * inv_l
    mid = ~(wl < high_wl) & ~(wl > low_wl)
    return torch.where(mid, smoothed, inv_l)


def rope_tables(ll, L):
    """-> (cos, sin), each [L, head_dim/2] fp32 CPU tensors."""
    inv = llama3_inv_freq(ll)
    ang = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
    return ang.cos().contiguous(), ang.sin().contiguous()
