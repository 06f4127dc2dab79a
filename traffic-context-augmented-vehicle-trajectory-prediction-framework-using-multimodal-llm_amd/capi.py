"""ctypes binding of the C-ABI library declared in ``include/tcavt.h``.

This is the only way the Python host side reaches the GPU: there is no eager /
PyTorch fallback for any op on the hot path.  If ``libtcavt_hip.so`` is missing
or fails to load, importing :func:`lib` raises -- loudly, by design.

torch is imported first so that the HIP runtime already mapped by PyTorch
(``libamdhip64.so.7``) is the one our library binds to; device pointers and
streams then belong to the same runtime instance.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtcavt_hip.so")
if os.environ.get("TCAVT_LIB") == "exp":  # tools/ only: the -DTCAVT_EXPERIMENTS build (python -m tcavt_amd.build --experiments)
    LIB_PATH = os.path.join(_HERE, "libtcavt_hip_exp.so")

ABI_VERSION = 4  # TCAVT_ABI_VERSION of include/tcavt.h
F32, BF16, F16 = 0, 1, 2
EPI_BIAS, EPI_RELU, EPI_RESIDUAL, EPI_SILU_MUL, EPI_ROPE, EPI_BIAS_ROW, EPI_ACCUM = 1, 2, 4, 8, 16, 32, 64
EPI_NORM_OUT, EPI_ROWSCALE, EPI_SILU_BWD = 128, 256, 512
ACT_A_FRAG16, ACT_OUT_FRAG16, ACT_BLOCK8 = 1, 2, 4  # tcavt_gemm_args.act_layout
W_FRAG16 = 1  # tcavt_gemm_args.w_layout / tcavt_decode_args.w_layout: tcavt_pack_weight16 copy

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class GemmArgs(ctypes.Structure):
    """Mirror of ``tcavt_gemm_args`` (include/tcavt.h)."""

    _fields_ = [
        ("A", c_void_p), ("lda", c_int64),
        ("W", c_void_p), ("ldw", c_int64),
        ("A2", c_void_p), ("lda2", c_int64),
        ("W2", c_void_p), ("ldw2", c_int64),
        ("C", c_void_p), ("ldc", c_int64),
        ("bias", c_void_p),
        ("residual", c_void_p), ("ldr", c_int64),
        ("rope_cos", c_void_p),
        ("rope_sin", c_void_p),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("K2", ctypes.c_int32),
        ("out_dtype", ctypes.c_int32),
        ("epilogue", ctypes.c_int32),
        ("rope_L", ctypes.c_int32), ("rope_cols", ctypes.c_int32),
        ("tile", ctypes.c_int32),
        ("acc_scale", ctypes.c_float),
        ("in_dtype", ctypes.c_int32),
        ("batch", ctypes.c_int32), ("batch_inner", ctypes.c_int32),
        ("sAo", c_int64), ("sAi", c_int64), ("sWo", c_int64), ("sWi", c_int64), ("sCo", c_int64), ("sCi", c_int64),
        ("dropout_p", ctypes.c_float), ("dropout_site", ctypes.c_uint32), ("dropout_seed", ctypes.c_uint64),
        ("batch_w_group", ctypes.c_int32), ("w_layout", ctypes.c_int32),
        ("silu_preact", c_void_p), ("ld_preact", c_int64),
        ("norm_h16", c_void_p), ("norm_part", c_void_p), ("rowscale_part", c_void_p),
        ("rowscale_npart", ctypes.c_int32), ("rowscale_h", ctypes.c_int32), ("rowscale_eps", ctypes.c_float),
        ("norm_scale", ctypes.c_float),
        ("rope_pos", c_void_p),
        ("nonfinite_flag", c_void_p), ("nonfinite_tag", ctypes.c_int32), ("act_layout", ctypes.c_int32),
        ("norm_res16", c_void_p), ("splitk_ws", c_void_p), ("splitk_ws_bytes", c_int64),
        ("lora_part", c_void_p), ("lora_part_a", c_void_p), ("lora_part_lda", c_int64), ("lora_part_np", c_int),
        ("lora_part_scale", c_float),
    ]


class LlamaLayer(ctypes.Structure):
    """Mirror of ``tcavt_llama_layer`` (include/tcavt.h)."""

    _fields_ = [(n, c_void_p) for n in ("w_qkv", "a_cat", "b_ext", "w_o", "w_gu", "w_d", "tape_h_mid", "tape_h_out",
                                        "tape_qkv", "tape_gu", "tape_t", "tape_att", "tape_lse", "tape_part")]


TLAYER_FIELDS = ("w_in", "b_in", "w_out", "b_out", "w_q", "b_q", "w_kv", "b_kv", "w_co", "b_co", "w1", "b1", "w2", "b2",
                 "n1_w", "n1_b", "n2_w", "n2_b", "n3_w", "n3_b",
                 "qkv", "att", "y", "x1", "x1b", "cq", "ckv", "catt", "cy", "x2", "x2b", "ffh", "y2", "out", "outb")


class TLayer(ctypes.Structure):
    """Mirror of ``tcavt_tlayer`` (include/tcavt.h)."""

    _fields_ = [(n, c_void_p) for n in TLAYER_FIELDS]


class TStackArgs(ctypes.Structure):
    """Mirror of ``tcavt_tstack_args`` (include/tcavt.h)."""

    _fields_ = [
        ("layers", ctypes.POINTER(TLayer)),
        ("x", c_void_p), ("xb", c_void_p), ("mem", c_void_p), ("memb", c_void_p), ("key_len", c_void_p),
        ("n_layers", ctypes.c_int32), ("B", ctypes.c_int32), ("L", ctypes.c_int32), ("Lk", ctypes.c_int32),
        ("E", ctypes.c_int32), ("FF", ctypes.c_int32), ("nhead", ctypes.c_int32), ("dtype16", ctypes.c_int32),
        ("dropout_p", ctypes.c_float), ("first_site", ctypes.c_uint32), ("dropout_seed", ctypes.c_uint64),
    ]


class CrossAttnArgs(ctypes.Structure):
    """Mirror of ``tcavt_cross_attn_args`` (include/tcavt.h)."""

    _fields_ = [(n, c_void_p) for n in ("q", "wk_t", "w_v", "b_v", "fh", "fh_t", "qp", "scores", "probs", "ctx", "att")] + [
        (n, ctypes.c_int32) for n in ("B", "To", "L", "Lp", "H", "nhead", "dtype16")] + [
        ("dropout_p", ctypes.c_float), ("dropout_site", ctypes.c_uint32), ("reserved0", ctypes.c_uint32),
        ("dropout_seed", ctypes.c_uint64)]


class TLayerGrads(ctypes.Structure):
    """Mirror of ``tcavt_tlayer_grads`` (include/tcavt.h)."""

    _fields_ = [(n, c_void_p) for n in ("g_w_in", "g_b_in", "g_w_out", "g_b_out", "g_w1", "g_b1", "g_w2", "g_b2", "g_n1_w", "g_n1_b",
                                        "g_n2_w", "g_n2_b")]


class TStackBwdArgs(ctypes.Structure):
    """Mirror of ``tcavt_tstack_bwd_args`` (include/tcavt.h)."""

    _fields_ = [("fwd", ctypes.POINTER(TStackArgs)), ("grads", ctypes.POINTER(TLayerGrads))] + [(n, c_void_p) for n in (
        "g_out", "g_x", "g_tmp", "g_y2", "g_y2d", "g_x1", "g_y", "g_yd", "g_att", "g_f", "g_qkv")]


class CrossAttnBwdArgs(ctypes.Structure):
    """Mirror of ``tcavt_cross_attn_bwd_args`` (include/tcavt.h)."""

    _fields_ = [("fwd", ctypes.POINTER(CrossAttnArgs))] + [(n, c_void_p) for n in (
        "g_att", "w_in", "gw_in", "gb_in", "g_q", "fh_tb", "fh_b", "ga_t", "g_ctx", "w_t", "x_t", "d_p", "d_s", "g_qp", "p_undropped")]


class LtsfArgs(ctypes.Structure):
    """Mirror of ``tcavt_ltsf_args`` (include/tcavt.h)."""

    _fields_ = [(n, c_void_p) for n in ('x', 'conv_w', 'conv_b', 'enc_w', 'enc_b', 'pos', 'tok', 'xp_tok', 'sa_n1_w', 'sa_n1_b', 'sa_in_w', 'sa_in_b', 'sa_out_w', 'sa_out_b', 'sa_n2_w', 'sa_n2_b', 'sa_f0_w', 'sa_f0_b', 'sa_f3_w', 'sa_f3_b', 'sa_xn', 'sa_qkv', 'sa_att', 'sa_res1', 'sa_rn', 'sa_f', 'e', 'poly_emb', 'lane_w', 'lane_b', 'dec_w', 'dec_b', 'lane', 'd0', 'pm0_w', 'pm0_b', 'pm3_w', 'pm3_b', 'hid', 'd1', 'dec_t', 'dec_tb', 'w_dp', 'b_dp', 'proj', 'w_q', 'b_q')] + [("xattn", CrossAttnArgs)] + [(n, c_void_p) for n in ('w_co', 'b_co', 'cross', 'w_un', 'b_un', 'fused', 'fl_n_w', 'fl_n_b', 'fn', 'fl1_w', 'fl1_b', 'f1', 'fl3_w', 'fl3_b', 'f2', 'out_w', 'out_b', 'out')] + [
        (n, ctypes.c_int32) for n in ("B", "C", "T", "To", "F", "H", "nhead_sa", "poly_dim", "post_hidden", "add_last")] + [
        ("dropout_p", ctypes.c_float), ("first_site", ctypes.c_uint32), ("dropout_seed", ctypes.c_uint64)]


class LtsfBwdArgs(ctypes.Structure):
    """Mirror of ``tcavt_ltsf_bwd_args`` (include/tcavt.h)."""

    _fields_ = [("fwd", ctypes.POINTER(LtsfArgs)), ("xattn", CrossAttnBwdArgs)] + [(n, c_void_p) for n in (
        "g_out", "w_un", "w_co", "w_dp",
        "g_out_w", "g_out_b", "g_fl3_w", "g_fl3_b", "g_fl1_w", "g_fl1_b", "g_fl_n_w", "g_fl_n_b",
        "g_un_w", "g_un_b", "g_co_w", "g_co_b", "g_dp_w", "g_dp_b",
        "g_pm3_w", "g_pm3_b", "g_pm0_w", "g_pm0_b", "g_lane_w", "g_lane_b", "g_dec_w", "g_dec_b",
        "g_sa_n1_w", "g_sa_n1_b", "g_sa_in_w", "g_sa_in_b", "g_sa_out_w", "g_sa_out_b", "g_sa_n2_w", "g_sa_n2_b", "g_sa_f0_w",
        "g_sa_f0_b", "g_sa_f3_w", "g_sa_f3_b", "g_enc_w", "g_enc_b", "g_pos", "g_conv_w", "g_conv_b", "g_poly",
        "g_f2", "g_f1", "g_fn", "g_dec_t", "g_dt2", "g_cross", "g_proj", "g_d1", "g_hid", "g_d0", "g_dw", "g_db", "g_e",
        "g_e_d", "g_res1_d", "g_ff", "g_rn", "g_res1", "g_att_sa", "g_qkv", "g_xn", "g_tok", "g_ew", "g_eb", "g_xp",
        "s_gyb", "s_gyt", "s_xt", "s_wt")] + [("dec_stride", c_int64), ("enc_stride", c_int64), ("pos_ld", ctypes.c_int32),
                                              ("reserved0", ctypes.c_int32)]


class LlamaStackArgs(ctypes.Structure):
    """Mirror of ``tcavt_llama_stack_args`` (include/tcavt.h)."""

    _fields_ = [
        ("layers", ctypes.POINTER(LlamaLayer)),
        ("gamma_final", c_void_p), ("rope_cos", c_void_p), ("rope_sin", c_void_p),
        ("h", c_void_p), ("h16", c_void_p), ("part", c_void_p), ("kv_len", c_void_p),
        ("qkv", c_void_p), ("att", c_void_p), ("act", c_void_p), ("t", c_void_p), ("xq", c_void_p), ("xv", c_void_p),
        ("out_f32", c_void_p), ("out16", c_void_p),
        ("k_cache", c_void_p), ("v_cache", c_void_p),
        ("events", ctypes.POINTER(c_void_p)),
        ("n_layers", ctypes.c_int32), ("B", ctypes.c_int32), ("L", ctypes.c_int32), ("H", ctypes.c_int32),
        ("I", ctypes.c_int32), ("nq", ctypes.c_int32), ("nkv", ctypes.c_int32), ("dtype16", ctypes.c_int32),
        ("kv_lmax", ctypes.c_int32), ("gemm_tile", ctypes.c_int32), ("npart_in", ctypes.c_int32), ("stream_scale", ctypes.c_float),
        ("rms_eps", ctypes.c_float), ("lora_scale", ctypes.c_float), ("lora_dropout_p", ctypes.c_float),
        ("lora_first_site", ctypes.c_uint32), ("dropout_seed", ctypes.c_uint64),
        ("nonfinite_flag", c_void_p), ("splitk_ws", c_void_p), ("splitk_ws_bytes", c_int64),
    ]


class LlamaBwdLayer(ctypes.Structure):
    """Mirror of ``tcavt_llama_bwd_layer`` (include/tcavt.h)."""

    _fields_ = [(n, c_void_p) for n in ("w_dT", "w_guT", "w_oT", "w_qkvT", "b_extT", "a_qT", "a_vT", "g1", "g2", "h_in", "h_mid",
                                        "qkv", "gu", "att", "lse", "part", "t", "g_Aq", "g_Av", "g_Bq", "g_Bv")]


class LlamaBackwardArgs(ctypes.Structure):
    """Mirror of ``tcavt_llama_backward_args`` (include/tcavt.h)."""

    _fields_ = [
        ("layers", ctypes.POINTER(LlamaBwdLayer)),
        ("h_last", c_void_p), ("gamma_final", c_void_p), ("g_final_a", c_void_p), ("g_final_b", c_void_p),
        ("rope_cos", c_void_p), ("rope_sin", c_void_p), ("kv_len", c_void_p), ("scale", c_void_p), ("scale_scratch", c_void_p),
        ("g_h", c_void_p), ("g_hb", c_void_p), ("g_xn", c_void_p), ("g_xl", c_void_p), ("g_att", c_void_p),
        ("g_qkv0", c_void_p), ("g_qkv1", c_void_p), ("g_t0", c_void_p), ("g_t1", c_void_p),
        ("dA", c_void_p), ("dB", c_void_p), ("stats", c_void_p),
        ("leaf_stream", c_void_p), ("events", ctypes.POINTER(c_void_p)),
        ("n_layers", ctypes.c_int32), ("B", ctypes.c_int32), ("L", ctypes.c_int32), ("H", ctypes.c_int32),
        ("I", ctypes.c_int32), ("nq", ctypes.c_int32), ("nkv", ctypes.c_int32), ("dtype16", ctypes.c_int32),
        ("npart", ctypes.c_int32), ("lora_rank", ctypes.c_int32), ("input_grad", ctypes.c_int32), ("reserved0", ctypes.c_int32),
        ("rms_eps", ctypes.c_float), ("lora_scale", ctypes.c_float), ("lora_dropout_p", ctypes.c_float),
        ("lora_first_site", ctypes.c_uint32), ("dropout_seed", ctypes.c_uint64), ("scale_backoff", c_void_p),
    ]


class SampleParams(ctypes.Structure):
    """Mirror of ``tcavt_sample_params`` (include/tcavt.h)."""

    _fields_ = [("temperature", c_float), ("top_p", c_float), ("repetition_penalty", c_float), ("top_k", ctypes.c_int32),
                ("no_repeat_ngram_size", ctypes.c_int32), ("do_sample", ctypes.c_int32), ("eos_token_id", c_int64),
                ("pad_token_id", c_int64), ("seed", ctypes.c_uint64)]


class DecodeArgs(ctypes.Structure):
    """Mirror of ``tcavt_decode_args`` (include/tcavt.h)."""

    _fields_ = [("layers", ctypes.POINTER(LlamaLayer))] + [(n, c_void_p) for n in (
        "gamma_final", "rope_cos", "rope_sin", "table", "txt_mod", "cur_tok", "pos", "h", "h16", "part", "qkv", "att", "act",
        "t", "k_cache", "v_cache", "x16", "logits", "bad_id_flag")] + [(n, ctypes.c_int32) for n in (
            "n_layers", "B", "H", "I", "nq", "nkv", "V", "dtype16", "kv_lmax", "rope_L")] + [
        ("rms_eps", c_float), ("lora_scale", c_float), ("nonfinite_flag", c_void_p), ("splitk_ws", c_void_p),
        ("splitk_ws_bytes", c_int64), ("lora_part", c_void_p), ("lora_rank", c_int), ("stream_scale", ctypes.c_float),
        ("w_layout", ctypes.c_int32), ("act_layout", ctypes.c_int32), ("table_packed", c_void_p)]


# name -> argtypes (return type is always int unless listed in _RESTYPES)
_SIGNATURES = {
    "tcavt_abi_version": [],
    "tcavt_last_error": [],
    "tcavt_init": [c_int, ctypes.POINTER(c_int)],
    "tcavt_gemm_bf16": [ctypes.POINTER(GemmArgs), c_void_p],
    "tcavt_rmsnorm": [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, ctypes.c_uint64,
                      ctypes.c_uint32, c_int, c_void_p],
    "tcavt_layernorm": [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tcavt_cast_f32_16": [c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "tcavt_copy_batch": [c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tcavt_embed_fuse": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                         c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p],
    "tcavt_softmax_rows": [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, ctypes.c_uint64,
                           ctypes.c_uint32, c_void_p],
    "tcavt_set_dropout_epoch": [c_void_p],
    "tcavt_dropout_epoch_advance": [c_void_p, c_void_p],
    "tcavt_dropout": [c_void_p, c_void_p, c_int64, c_int, c_float, ctypes.c_uint64, ctypes.c_uint32, c_void_p, c_void_p],
    "tcavt_mask_to_kvlen": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "tcavt_attn_causal_gqa": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p],
    "tcavt_attn_causal_gqa_lse": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p],
    "tcavt_mha": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int,
                  c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_float, ctypes.c_uint64, ctypes.c_uint32, c_void_p],
    "tcavt_gemm_f32": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                       c_int, c_int, c_int, c_int, c_float, ctypes.c_uint64, ctypes.c_uint32, c_void_p],
    "tcavt_poly_embed": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tcavt_masked_mean": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tcavt_ltsf_front": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                         c_int, c_void_p],
    "tcavt_ltsf_decode": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "tcavt_transpose_ct": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "tcavt_out_head": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                       c_int, c_void_p],
    "tcavt_traj_metrics": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                           c_void_p],
    "tcavt_gemm_f32_strided": [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                               c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p],
    "tcavt_transpose16": [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int, c_void_p],
    "tcavt_transpose_f32_bf16": [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p],
    "tcavt_colsum": [c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "tcavt_relu_bwd": [c_void_p, c_void_p, c_int, c_int64, c_void_p],
    "tcavt_add_inplace": [c_void_p, c_void_p, c_int64, c_void_p],
    "tcavt_silu_mul_bwd": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p],
    "tcavt_rmsnorm_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                          c_void_p, c_int, c_void_p],
    "tcavt_lora_dgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, ctypes.c_uint64, ctypes.c_uint32,
                         ctypes.c_uint32, c_int, c_void_p],
    "tcavt_grad_scale_pick": [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "tcavt_rope_bwd_pack": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p],
    "tcavt_attn_causal_gqa_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                  c_void_p],
    "tcavt_causal_softmax_bwd_rows": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                      c_void_p],
    "tcavt_causal_softmax_bwd_tiles": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_float, c_void_p],
    "tcavt_attn_bwd_scores": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int,
                              c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p],
    "tcavt_attn_bwd_resident_ok": [c_int, c_int, c_int],
    "tcavt_attn_bwd_resident": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                c_int, c_int, c_int, c_int, c_float, c_int, c_void_p],
    "tcavt_attn_bwd_dkv": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                           c_int, c_void_p],
    "tcavt_gqa_rope_bwd_pack": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p],
    "tcavt_wgrad_tn": [c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                       c_void_p, c_int, c_int, c_float, c_void_p],
    "tcavt_lora_wgrad_a": [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float,
                           ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_int, c_void_p],
    "tcavt_clip_grad_norm": [c_void_p, c_int64, c_float, c_float, c_void_p, c_void_p],
    "tcavt_layernorm_bwd": [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "tcavt_mha_bwd": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                      c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, ctypes.c_uint64,
                      ctypes.c_uint32, c_void_p],
    "tcavt_softmax_bwd_rows": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_float, c_int, c_int, c_int,
                               c_void_p],
    "tcavt_mse_grad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "tcavt_out_head_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                           c_void_p],
    "tcavt_nlinear_bwd": [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                          c_int, c_int, c_int, c_void_p],
    "tcavt_conv1x1_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "tcavt_poly_embed_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tcavt_masked_mean_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tcavt_adamw": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                    c_int, c_float, c_void_p],
    "tcavt_llama_stack_forward": [ctypes.POINTER(LlamaStackArgs), c_void_p],
    "tcavt_sample_logits": [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, ctypes.POINTER(SampleParams), c_void_p, c_void_p,
                            c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p],
    "tcavt_sample_workspace_bytes": [c_int],
    "tcavt_gather_last": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tcavt_llama_decode_step": [ctypes.POINTER(DecodeArgs), c_void_p],
    "tcavt_norm_npart": [c_int, c_int, c_int],
    "tcavt_pack_weight16": [c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p],
    "tcavt_lora_down": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, ctypes.c_uint64, ctypes.c_uint32,
                        ctypes.c_uint32, c_int, c_void_p],
    "tcavt_rownorm_prep": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "tcavt_allreduce_flat": [c_void_p, c_int64, c_void_p, c_void_p],
    "tcavt_tlayer_stack_forward": [ctypes.POINTER(TStackArgs), c_void_p],
    "tcavt_cross_attn_forward": [ctypes.POINTER(CrossAttnArgs), c_void_p],
    "tcavt_ltsf_forward": [ctypes.POINTER(LtsfArgs), c_int, c_void_p],
    "tcavt_cross_attn_backward": [ctypes.POINTER(CrossAttnBwdArgs), c_void_p],
    "tcavt_ltsf_backward": [ctypes.POINTER(LtsfBwdArgs), c_int, c_void_p],
    "tcavt_tlayer_stack_backward": [ctypes.POINTER(TStackBwdArgs), c_void_p],
    "tcavt_rmsnorm16": [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tcavt_llama_stack_backward": [ctypes.POINTER(LlamaBackwardArgs), c_void_p],
    "tcavt_events_create": [ctypes.POINTER(c_void_p), c_int],
    "tcavt_events_destroy": [ctypes.POINTER(c_void_p), c_int],
    "tcavt_event_elapsed_ms": [c_void_p, c_void_p, ctypes.POINTER(c_float)],
    "tcavt_adamw_gated": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                          c_float, c_void_p, c_void_p, c_void_p, c_void_p],
}
_RESTYPES = {"tcavt_last_error": ctypes.c_char_p, "tcavt_sample_workspace_bytes": c_int64}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class TcavtError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the ctypes handle of libtcavt_hip.so."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TcavtError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `python -m tcavt_amd.build`). "
            "There is no CPU/PyTorch fallback for the hot path."
        )
    handle = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    if handle.tcavt_abi_version() != ABI_VERSION:
        raise TcavtError("libtcavt_hip.so ABI version mismatch")
    _lib = handle
    return handle


def check(rc, what=""):
    if rc != 0:
        msg = lib().tcavt_last_error()
        raise TcavtError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


_initialised = set()


def init(device_index=0):
    """Verify the device is gfx950 and return its CU count."""
    n = c_int(0)
    check(lib().tcavt_init(int(device_index), ctypes.byref(n)), "tcavt_init")
    _initialised.add(int(device_index))
    return n.value


def stream_ptr():
    """hipStream_t of torch's current stream (0 = the null stream)."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())
