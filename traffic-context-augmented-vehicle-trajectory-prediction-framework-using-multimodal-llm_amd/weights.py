"""Seeded synthetic weights with the reference's state-dict key layout.

There are no real checkpoints in this environment (no network), so every run uses
weights generated here.  Keys follow ``MultiModalTrajectoryModel.state_dict()`` of the
reference's no-LoRA module (scripts/ablation_study_without_lora.py, same layout as
scripts/train.py minus the PEFT prefixes):

    lane_polygon_encoder.*, mllm.qformer.*, mllm.q_proj.*, mllm.*_modality_embedding,
    mllm.llama_wrapper.llama_model.model.*, mllm.llama_wrapper.llama_model.lm_head.weight,
    ltsf.*

LoRA adapters (scripts/train.py:433-440; q_proj/v_proj, modify_scripts/modify_train.py:518)
are stored next to their base layer as ``...self_attn.{q,v}_proj.lora_A.weight`` [r, in]
and ``...lora_B.weight`` [out, r].

Each tensor is drawn from its own numpy PCG64 stream keyed by (seed, crc32(name)), so the
same (config, seed) gives the same weights on every machine, in any generation order.
Scales are "trained-like" (fan-in scaled), not framework defaults: e.g. the polygon
``input_proj`` is small because it multiplies raw pixel coordinates (up to 3839).
"""
import zlib

import numpy as np

from .config import ModelConfig

LLAMA_PREFIX = "mllm.llama_wrapper.llama_model.model."
LM_HEAD = "mllm.llama_wrapper.llama_model.lm_head.weight"


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


class _Gen:
    """backend "numpy": machine-independent PCG64 streams (fixtures, parity tests).
    backend "torch": torch.randn on `device` (fast init of the 1.2 B-parameter bench model; same
    scales, different stream -- never used where values are compared against fixtures)."""

    def __init__(self, seed, backend="numpy", device=None):
        self.seed = seed
        self.out = {}
        self.backend, self.device = backend, device

    def normal(self, name, shape, std, mean=0.0):
        if self.backend == "numpy":
            a = _rng(self.seed, name).standard_normal(shape, dtype=np.float32) * np.float32(std) + np.float32(mean)
            self.out[name] = a.astype(np.float32)
        else:
            import torch

            g = torch.Generator(device=self.device)
            g.manual_seed((int(self.seed) << 32) ^ zlib.crc32(name.encode()))
            t = torch.randn(tuple(shape), generator=g, device=self.device, dtype=torch.float32)
            self.out[name] = t.mul_(float(std)).add_(float(mean))

    def linear(self, prefix, n_out, n_in, bias=True, gain=0.7):
        self.normal(prefix + ".weight", (n_out, n_in), gain / np.sqrt(n_in))
        if bias:
            self.normal(prefix + ".bias", (n_out,), 0.02)

    def layernorm(self, prefix, d):
        self.normal(prefix + ".weight", (d,), 0.1, mean=1.0)
        self.normal(prefix + ".bias", (d,), 0.05)

    def mha(self, prefix, e, gain=0.7):
        self.normal(prefix + ".in_proj_weight", (3 * e, e), gain / np.sqrt(e))
        self.normal(prefix + ".in_proj_bias", (3 * e,), 0.02)
        self.linear(prefix + ".out_proj", e, e)

    def enc_layer(self, prefix, d, ff):
        self.mha(prefix + ".self_attn", d, gain=1.2)
        self.linear(prefix + ".linear1", ff, d)
        self.linear(prefix + ".linear2", d, ff)
        self.layernorm(prefix + ".norm1", d)
        self.layernorm(prefix + ".norm2", d)

    def dec_layer(self, prefix, d, ff):
        self.mha(prefix + ".self_attn", d, gain=1.2)
        self.mha(prefix + ".multihead_attn", d, gain=1.2)
        self.linear(prefix + ".linear1", ff, d)
        self.linear(prefix + ".linear2", d, ff)
        self.layernorm(prefix + ".norm1", d)
        self.layernorm(prefix + ".norm2", d)
        self.layernorm(prefix + ".norm3", d)


def make_weights(cfg: ModelConfig, seed: int = 0, backend: str = "numpy", device=None):
    """Return {state-dict key: float32 ndarray (numpy backend) or tensor (torch backend)}."""
    g = _Gen(seed, backend, device)
    ll = cfg.llama
    H = ll.hidden
    d = cfg.d_model
    T, To = cfg.seq_len, cfg.out_len

    # --- LanePolygonEncoder (train.py:352-360)
    pd = cfg.lane_polygon_d_model
    g.normal("lane_polygon_encoder.pos_embedding", (1, cfg.max_polygon_points, pd), 0.1)
    g.normal("lane_polygon_encoder.input_proj.weight", (pd, 2), 2e-3)
    g.normal("lane_polygon_encoder.input_proj.bias", (pd,), 0.02)
    for i in range(cfg.lane_polygon_layers):
        g.enc_layer(f"lane_polygon_encoder.encoder.layers.{i}", pd, cfg.transformer_ff)

    # --- MLLM: modality embeddings, Q-Former, q_proj (train.py:388-406, 493-498)
    g.normal("mllm.vision_modality_embedding", (1, 1, H), 0.5)
    g.normal("mllm.text_modality_embedding", (1, 1, H), 0.5)
    qh = cfg.q_hidden_size
    g.normal("mllm.qformer.query_tokens", (cfg.q_num_query_tokens, qh), 1.0)
    g.linear("mllm.qformer.vision_proj", qh, cfg.vision_dim)
    for i in range(cfg.q_enc_layers):
        g.enc_layer(f"mllm.qformer.encoder.layers.{i}", qh, cfg.transformer_ff)
    for i in range(cfg.q_dec_layers):
        g.dec_layer(f"mllm.qformer.decoder.layers.{i}", qh, cfg.transformer_ff)
    g.linear("mllm.q_proj", H, qh)

    # --- Llama decoder (HF LlamaForCausalLM layout)
    P = LLAMA_PREFIX
    g.normal(P + "embed_tokens.weight", (ll.vocab, H), 0.5)
    nq, nkv, hd = ll.n_q_heads, ll.n_kv_heads, ll.head_dim
    for i in range(ll.layers):
        L = f"{P}layers.{i}."
        g.linear(L + "self_attn.q_proj", nq * hd, H, bias=False, gain=1.5)
        g.linear(L + "self_attn.k_proj", nkv * hd, H, bias=False, gain=1.5)
        g.linear(L + "self_attn.v_proj", nkv * hd, H, bias=False)
        g.linear(L + "self_attn.o_proj", H, nq * hd, bias=False)
        g.linear(L + "mlp.gate_proj", ll.inter, H, bias=False, gain=1.0)
        g.linear(L + "mlp.up_proj", ll.inter, H, bias=False, gain=1.0)
        g.linear(L + "mlp.down_proj", H, ll.inter, bias=False)
        g.normal(L + "input_layernorm.weight", (H,), 0.1, mean=1.0)
        g.normal(L + "post_attention_layernorm.weight", (H,), 0.1, mean=1.0)
        if cfg.use_lora:
            r = cfg.lora_r
            # PEFT initialises B = 0; non-zero here so the adapter path is exercised
            g.normal(L + "self_attn.q_proj.lora_A.weight", (r, H), 1.0 / np.sqrt(H))
            g.normal(L + "self_attn.q_proj.lora_B.weight", (nq * hd, r), 0.1)
            g.normal(L + "self_attn.v_proj.lora_A.weight", (r, H), 1.0 / np.sqrt(H))
            g.normal(L + "self_attn.v_proj.lora_B.weight", (nkv * hd, r), 0.1)
    g.normal(P + "norm.weight", (H,), 0.1, mean=1.0)
    g.out[LM_HEAD] = g.out[P + "embed_tokens.weight"]  # tied embeddings (public model card)

    # --- TransformerLTSF (train.py:808-834)
    g.normal("ltsf.pos_encoding", (1, d, T), 0.1)
    g.normal("ltsf.token_proj.weight", (d, cfg.feature_size, 1), 0.7 / np.sqrt(cfg.feature_size))
    g.normal("ltsf.token_proj.bias", (d,), 0.02)
    for c in range(d):
        g.linear(f"ltsf.nlinear_encoder.encoder_linears.{c}", T, T)
        g.linear(f"ltsf.decoder.decoder_linears.{c}", To, T)
    g.layernorm("ltsf.attn_block.norm1", d)
    g.mha("ltsf.attn_block.mha", d, gain=1.2)
    g.linear("ltsf.attn_block.ffn.0", 4 * d, d)
    g.linear("ltsf.attn_block.ffn.3", d, 4 * d)
    g.layernorm("ltsf.attn_block.norm2", d)
    g.linear("ltsf.decoder.lane_fc", d * To, cfg.lane_polygon_d_model)
    g.linear("ltsf.decoder.post_mlp.0", cfg.post_mlp_hidden_dim, d * To)
    g.linear("ltsf.decoder.post_mlp.3", d * To, cfg.post_mlp_hidden_dim)
    g.mha("ltsf.decoder.cross_attn", H, gain=1.0)
    g.linear("ltsf.decoder.dec_proj", H, d)
    g.linear("ltsf.decoder.dec_unproj", d, H)
    g.layernorm("ltsf.decoder.fusion_layer.0", d)
    g.linear("ltsf.decoder.fusion_layer.1", d, d)
    g.linear("ltsf.decoder.fusion_layer.3", d, d)
    g.linear("ltsf.decoder.out_proj", cfg.feature_size, d)
    return g.out


def plant_outliers(weights, cfg: ModelConfig, magnitude: float, channels=None):
    """Weight variant with two OUTLIER hidden channels in the decoder's residual stream.

    Real Llama checkpoints (scripts/train.py:427-431 loads one) carry a few hidden channels whose activations sit orders of
    magnitude above the bulk of the stream, nearly constant over the tokens; the N(0, sigma) weights of make_weights never
    do (largest |stream| ~ 10).  Here both modality embeddings (train.py:497-498: added to EVERY fused-sequence token) get
    +magnitude in one channel and -0.7 magnitude in another, so every token's residual stream starts -- and, the residual
    adds being small beside it, stays -- at that level in those channels.  The parameters themselves are fp32 in the HIP path
    too (embed_fuse adds them in fp32), so a magnitude beyond fp16's 65504 is a statement about ACTIVATION range only.
    Used by the range tests (tests/test_range_gpu.py: 3e4 fits the plain fp16 stream, 2e5 needs set_storage("auto"))."""
    H = cfg.llama.hidden
    c0, c1 = channels or (H // 5, (3 * H) // 4 + 1)
    out = dict(weights)
    for k in ("mllm.vision_modality_embedding", "mllm.text_modality_embedding"):
        v = out[k].clone() if hasattr(out[k], "clone") else np.array(out[k], copy=True)
        v[..., c0] = magnitude
        v[..., c1] = -0.7 * magnitude
        out[k] = v
    return out


def is_lora_key(name):
    return ".lora_A." in name or ".lora_B." in name


def trainable_keys(weights):
    """Parameters scripts/train.py trains: everything outside ``mllm.`` (train.py:1141-1145)."""
    return [k for k in weights if not k.startswith("mllm.")]
