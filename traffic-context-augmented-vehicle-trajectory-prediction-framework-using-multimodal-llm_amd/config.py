"""Shape/config description of the hot path.

Mirrors the hard-coded ``args`` dict of the reference (scripts/train.py:1331-1366)
plus the Llama-3.2-1B architecture constants the reference obtains from
``AutoModelForCausalLM.from_pretrained`` (scripts/train.py:427; public model card).
"""
from dataclasses import asdict, dataclass, field
from typing import Optional


@dataclass
class LlamaShape:
    hidden: int = 2048
    inter: int = 8192
    layers: int = 16
    n_q_heads: int = 32
    n_kv_heads: int = 8
    head_dim: int = 64
    vocab: int = 128256
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    # llama3 rope scaling (HF modeling_rope_utils `_compute_llama3_parameters`)
    rope_factor: float = 32.0
    rope_low_freq_factor: float = 1.0
    rope_high_freq_factor: float = 4.0
    rope_original_max_pos: int = 8192


@dataclass
class ModelConfig:
    """Constructor kwargs of ``MultiModalTrajectoryModel`` (scripts/train.py:848-872)."""

    seq_len: int = 18
    out_len: int = 30
    individual: bool = True
    feature_size: int = 2
    d_model: int = 64
    lane_polygon_d_model: int = 64
    lane_polygon_nhead: int = 4
    lane_polygon_layers: int = 2
    max_polygon_points: int = 64
    use_post_mlp: bool = True
    post_mlp_hidden_dim: int = 64
    use_lora: bool = True
    lora_r: int = 8
    lora_alpha: int = 32
    lora_dropout: float = 0.1
    vision_dim: int = 512
    q_hidden_size: int = 768
    q_nhead: int = 8
    q_enc_layers: int = 4
    q_dec_layers: int = 4
    q_num_query_tokens: int = 16
    ltsf_nhead: int = 2
    ltsf_dropout: float = 0.1
    transformer_ff: int = 2048  # nn.Transformer*Layer default dim_feedforward
    cross_nhead: int = 2        # scripts/train.py:908
    llama: LlamaShape = field(default_factory=LlamaShape)

    def to_dict(self):
        return asdict(self)


def llama32_1b(seq_len=18, out_len=30, use_lora=True) -> ModelConfig:
    """The configuration of scripts/train.py:1331-1366 (Llama-3.2-1B, LoRA r=8 alpha=32)."""
    return ModelConfig(seq_len=seq_len, out_len=out_len, use_lora=use_lora)


def tiny(seq_len=6, out_len=12, use_lora=True) -> ModelConfig:
    """Small shape for fixtures / parity tests: same structure, head_dim 64, GQA 4:1."""
    return ModelConfig(
        seq_len=seq_len,
        out_len=out_len,
        use_lora=use_lora,
        llama=LlamaShape(hidden=256, inter=512, layers=2, n_q_heads=4, n_kv_heads=1, vocab=512),
    )


def midi(seq_len=18, out_len=30, use_lora=True) -> ModelConfig:
    """Intermediate shape (4 layers, hidden 512) the CPU oracle still finishes in seconds."""
    return ModelConfig(
        seq_len=seq_len,
        out_len=out_len,
        use_lora=use_lora,
        llama=LlamaShape(hidden=512, inter=1536, layers=4, n_q_heads=8, n_kv_heads=2, vocab=1024),
    )


PRESETS = {"llama32_1b": llama32_1b, "tiny": tiny, "midi": midi}
