"""Tensor-level wrappers over the C ABI (one Python function per entry point).

torch is used here for device memory and the current stream only; every
computation happens inside libtcavt_hip.so.  All tensors must be CUDA(HIP)
tensors, contiguous in the layout the header documents.
"""
import ctypes

import torch

from . import capi
from .capi import BF16, EPI_BIAS, EPI_RELU, EPI_RESIDUAL, EPI_ROPE, EPI_SILU_MUL, F32, check, lib, ptr, stream_ptr

__all__ = [
    "gemm_bf16", "rmsnorm", "layernorm", "cast_bf16", "embed_fuse", "attn_causal_gqa", "mha", "gemm_f32",
    "poly_embed", "masked_mean", "ltsf_front", "ltsf_decode", "transpose_ct", "out_head", "traj_metrics",
]


def _req(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise capi.TcavtError(f"{name}: tensor must live on the GPU (no CPU fallback)")
    if t.dtype != dtype:
        raise capi.TcavtError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise capi.TcavtError(f"{name}: tensor must be contiguous")


def gemm_bf16(a, w, out=None, *, out_dtype=torch.bfloat16, bias=None, relu=False, residual=None, a2=None,
              w2=None, silu_mul=False, rope=None, tile=0):
    """C = a @ w.T (+ a2 @ w2.T) with fused epilogue.  a [M,K] bf16, w [N,K] bf16.

    rope = (cos [L,32] f32, sin [L,32] f32, rope_cols) applies RoPE with position m % L.
    """
    _req(a, torch.bfloat16, "gemm_bf16.a")
    _req(w, torch.bfloat16, "gemm_bf16.w")
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    n_out = N // 2 if silu_mul else N
    if out is None:
        out = torch.empty((M, n_out), dtype=out_dtype, device=a.device)
    _req(out, out.dtype, "gemm_bf16.out")
    args = capi.GemmArgs()
    args.A, args.lda = a.data_ptr(), a.stride(0)
    args.W, args.ldw = w.data_ptr(), w.stride(0)
    if a2 is not None:
        _req(a2, torch.bfloat16, "gemm_bf16.a2")
        _req(w2, torch.bfloat16, "gemm_bf16.w2")
        args.A2, args.lda2 = a2.data_ptr(), a2.stride(0)
        args.W2, args.ldw2 = w2.data_ptr(), w2.stride(0)
        args.K2 = a2.shape[1]
    args.C, args.ldc = out.data_ptr(), out.stride(0)
    epi = 0
    if bias is not None:
        _req(bias, torch.float32, "gemm_bf16.bias")
        args.bias = bias.data_ptr()
        epi |= EPI_BIAS
    if relu:
        epi |= EPI_RELU
    if residual is not None:
        _req(residual, torch.float32, "gemm_bf16.residual")
        args.residual, args.ldr = residual.data_ptr(), residual.stride(0)
        epi |= EPI_RESIDUAL
    if silu_mul:
        epi |= EPI_SILU_MUL
    if rope is not None:
        cos, sin, cols = rope
        _req(cos, torch.float32, "gemm_bf16.rope_cos")
        _req(sin, torch.float32, "gemm_bf16.rope_sin")
        args.rope_cos, args.rope_sin = cos.data_ptr(), sin.data_ptr()
        args.rope_L, args.rope_cols = cos.shape[0], cols
        epi |= EPI_ROPE
    args.M, args.N, args.K = M, N, K
    args.out_dtype = BF16 if out.dtype == torch.bfloat16 else F32
    args.epilogue = epi
    args.tile = tile
    check(lib().tcavt_gemm_bf16(ctypes.byref(args), stream_ptr()), "tcavt_gemm_bf16")
    return out


def rmsnorm(x, gamma, eps, out_bf16=None, out_f32=None):
    _req(x, torch.float32, "rmsnorm.x")
    _req(gamma, torch.float32, "rmsnorm.gamma")
    M, H = x.shape
    check(lib().tcavt_rmsnorm(ptr(x), ptr(gamma), eps, ptr(out_bf16), ptr(out_f32), M, H, stream_ptr()),
          "tcavt_rmsnorm")


def layernorm(x, gamma, beta, eps=1e-5, residual=None, out_f32=None, out_bf16=None):
    _req(x, torch.float32, "layernorm.x")
    _req(residual, torch.float32, "layernorm.residual")
    M, D = x.shape
    check(lib().tcavt_layernorm(ptr(x), ptr(residual), ptr(gamma), ptr(beta), eps, ptr(out_f32), ptr(out_bf16), M,
                                D, stream_ptr()), "tcavt_layernorm")


def cast_bf16(x, out=None):
    _req(x, torch.float32, "cast.x")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib().tcavt_cast_f32_bf16(ptr(x), ptr(out), x.numel(), stream_ptr()), "tcavt_cast_f32_bf16")
    return out


def embed_fuse(table, ids, img, vis_mod, txt_mod, h, bad_flag):
    _req(table, torch.bfloat16, "embed.table")
    _req(ids, torch.int64, "embed.ids")
    _req(img, torch.float32, "embed.img")
    B, Lt = ids.shape
    Nq = img.shape[1]
    V, H = table.shape
    check(lib().tcavt_embed_fuse(ptr(table), ptr(ids), ptr(img), ptr(vis_mod), ptr(txt_mod), ptr(h), B, Nq, Lt, H,
                                 V, ptr(bad_flag), stream_ptr()), "tcavt_embed_fuse")


def attn_causal_gqa(qkv, out, kv_len, B, L, nq, nkv, scale):
    _req(qkv, torch.bfloat16, "attn.qkv")
    _req(out, torch.bfloat16, "attn.out")
    _req(kv_len, torch.int32, "attn.kv_len")
    check(lib().tcavt_attn_causal_gqa(ptr(qkv), ptr(out), ptr(kv_len), B, L, nq, nkv, scale, stream_ptr()),
          "tcavt_attn_causal_gqa")


def mha(q, k, v, out, B, Lq, Lk, nh, dh, scale, key_len=None, ldq=None, ldk=None, ldv=None, ldo=None):
    """q/k/v may be column slices of wider row-major buffers: pass the slice's data_ptr tensor and ld."""
    in_dt = BF16 if q.dtype == torch.bfloat16 else F32
    out_dt = BF16 if out.dtype == torch.bfloat16 else F32
    check(lib().tcavt_mha(ptr(q), ldq if ldq else q.stride(-2), ptr(k), ldk if ldk else k.stride(-2), ptr(v),
                          ldv if ldv else v.stride(-2), ptr(out), ldo if ldo else out.stride(-2), ptr(key_len), B,
                          Lq, Lk, nh, dh, scale, in_dt, out_dt, stream_ptr()), "tcavt_mha")


def gemm_f32(a, w, out=None, bias=None, relu=False, residual=None):
    _req(a, torch.float32, "gemm_f32.a")
    _req(w, torch.float32, "gemm_f32.w")
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    flags = (EPI_BIAS if bias is not None else 0) | (EPI_RELU if relu else 0) | (
        EPI_RESIDUAL if residual is not None else 0)
    check(lib().tcavt_gemm_f32(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(residual),
                               residual.stride(0) if residual is not None else 0, ptr(out), out.stride(0), M, N, K,
                               flags, stream_ptr()), "tcavt_gemm_f32")
    return out


def poly_embed(polygon, w_in, b_in, pos, out):
    B, P, _ = polygon.shape
    D = w_in.shape[0]
    check(lib().tcavt_poly_embed(ptr(polygon), ptr(w_in), ptr(b_in), ptr(pos), ptr(out), B, P, D, stream_ptr()),
          "tcavt_poly_embed")


def masked_mean(enc, lens, out, B, P, D):
    check(lib().tcavt_masked_mean(ptr(enc), ptr(lens), ptr(out), B, P, D, stream_ptr()), "tcavt_masked_mean")


def ltsf_front(x, conv_w, conv_b, enc_w, enc_b, pos, out, B, C, T):
    check(lib().tcavt_ltsf_front(ptr(x), ptr(conv_w), ptr(conv_b), ptr(enc_w), ptr(enc_b), ptr(pos), ptr(out), B, C,
                                 T, stream_ptr()), "tcavt_ltsf_front")


def ltsf_decode(e_tok, dec_w, dec_b, lane_adj, out, B, C, T, To):
    check(lib().tcavt_ltsf_decode(ptr(e_tok), ptr(dec_w), ptr(dec_b), ptr(lane_adj), ptr(out), B, C, T, To,
                                  stream_ptr()), "tcavt_ltsf_decode")


def transpose_ct(x, out_f32, out_bf16, B, C, To):
    check(lib().tcavt_transpose_ct(ptr(x), ptr(out_f32), ptr(out_bf16), B, C, To, stream_ptr()),
          "tcavt_transpose_ct")


def out_head(fused, w, bias, x, out, B, To, C, F, T):
    check(lib().tcavt_out_head(ptr(fused), ptr(w), ptr(bias), ptr(x), ptr(out), B, To, C, F, T, stream_ptr()),
          "tcavt_out_head")


def traj_metrics(pred, gt, norm_stat, sums, argmin, per_sample, B, K, To):
    check(lib().tcavt_traj_metrics(ptr(pred), ptr(gt), ptr(norm_stat), ptr(sums), ptr(argmin), ptr(per_sample), B,
                                   K, To, stream_ptr()), "tcavt_traj_metrics")
