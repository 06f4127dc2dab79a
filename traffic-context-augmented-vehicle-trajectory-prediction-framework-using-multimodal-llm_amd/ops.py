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
    "gemm_bf16", "rmsnorm", "layernorm", "cast_bf16", "embed_fuse", "mask_to_kvlen", "attn_causal_gqa", "mha", "gemm_f32",
    "poly_embed", "masked_mean", "ltsf_front", "ltsf_decode", "transpose_ct", "out_head", "traj_metrics",
]


_ALLOW_CPU = False  # tests only: host-logic dry run with a stubbed library (tests/test_host_dryrun.py)


def _req(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda and not _ALLOW_CPU:
        raise capi.TcavtError(f"{name}: tensor must live on the GPU (no CPU fallback)")
    if t.dtype != dtype:
        raise capi.TcavtError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise capi.TcavtError(f"{name}: tensor must be contiguous")


_H16 = (torch.float16, torch.bfloat16)


def _req16(t, name, like=None, rows_ok=False):
    """16-bit storage tensor (fp16 = the forward path's default, or bf16); `like`: must share that tensor's dtype.
    rows_ok: a 2-D view with unit column stride (a column slice of a wider matrix) is accepted as well."""
    if t is None:
        return
    if not t.is_cuda and not _ALLOW_CPU:
        raise capi.TcavtError(f"{name}: tensor must live on the GPU (no CPU fallback)")
    if t.dtype not in _H16 or (like is not None and t.dtype != like.dtype):
        raise capi.TcavtError(f"{name}: expected {'fp16 or bf16' if like is None else like.dtype}, got {t.dtype}")
    if not t.is_contiguous() and not (rows_ok and t.dim() == 2 and t.stride(1) == 1):
        raise capi.TcavtError(f"{name}: tensor must be contiguous")


def _need(t, numel, name):
    """Host-side guard: the buffer must hold at least what the kernel's grid will touch."""
    if t is not None and t.numel() < numel:
        raise capi.TcavtError(f"{name}: buffer has {t.numel()} elements, kernel needs {numel}")


def _drop(d):
    """dropout spec (p, seed, site) or None -> (p, seed, site) ctypes-ready."""
    if d is None:
        return 0.0, 0, 0
    p, seed, site = d
    return float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(site) & 0xFFFFFFFF


def gemm_bf16(a, w, out=None, *, out_dtype=None, bias=None, relu=False, residual=None, a2=None,
              w2=None, silu_mul=False, rope=None, tile=0, acc_scale=1.0, dropout=None, silu_preact=None, splitk_ws=None):
    """C = a @ w.T (+ a2 @ w2.T) with fused epilogue.  a [M,K], w [N,K]: both fp16 or both bf16 (the name is
    historical); out: fp32 or either 16-bit type (default: the operands' type).

    silu_preact (with silu_mul): bf16 [M, N] buffer that receives the gate|up pre-activations (for silu_mul_bwd).

    rope = (cos [L,32] f32, sin [L,32] f32, rope_cols) applies RoPE with position m % L.

    splitk_ws (M <= 32 only): uint8 device workspace (>= 64 KiB, first 16 KiB zeroed once) that lets the skinny form split K
    across workgroups -- tcavt_gemm_args.splitk_ws.
    """
    _req16(a, "gemm_bf16.a", rows_ok=True)  # (A may be a column slice: lda = its row stride)
    _req16(w, "gemm_bf16.w", like=a, rows_ok=True)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    n_out = N // 2 if silu_mul else N
    if out is None:
        out = torch.empty((M, n_out), dtype=out_dtype or a.dtype, device=a.device)
    if (not out.is_cuda and not _ALLOW_CPU) or out.dim() != 2 or out.stride(1) != 1:  # a column slice of a wider buffer is fine
        raise capi.TcavtError("gemm_bf16.out: 2-D GPU tensor with unit column stride required")
    args = capi.GemmArgs()
    args.A, args.lda = a.data_ptr(), a.stride(0)
    args.W, args.ldw = w.data_ptr(), w.stride(0)
    if a2 is not None:
        _req16(a2, "gemm_bf16.a2", like=a)
        _req16(w2, "gemm_bf16.w2", like=a)
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
        if silu_preact is not None:
            _req16(silu_preact, "gemm_bf16.silu_preact", like=a)
            if silu_preact.shape[0] < M or silu_preact.shape[1] < N:
                raise capi.TcavtError("gemm_bf16.silu_preact: smaller than (M, N)")
            args.silu_preact, args.ld_preact = silu_preact.data_ptr(), silu_preact.stride(0)
    if rope is not None:
        cos, sin, cols = rope
        _req(cos, torch.float32, "gemm_bf16.rope_cos")
        _req(sin, torch.float32, "gemm_bf16.rope_sin")
        args.rope_cos, args.rope_sin = cos.data_ptr(), sin.data_ptr()
        args.rope_L, args.rope_cols = cos.shape[0], cols
        epi |= EPI_ROPE
    if out.shape[0] < M or out.shape[1] < n_out:
        raise capi.TcavtError(f"gemm_bf16.out: shape {tuple(out.shape)} smaller than ({M}, {n_out})")
    if a2 is not None and (a2.shape[0] < M or w2.shape[0] < N or w2.shape[1] != a2.shape[1]):
        raise capi.TcavtError("gemm_bf16: a2/w2 shapes do not match (M, K2) / (N, K2)")
    if bias is not None:
        _need(bias, N, "gemm_bf16.bias")
    if residual is not None and (residual.shape[0] < M or residual.shape[1] < N):
        raise capi.TcavtError("gemm_bf16.residual: smaller than (M, N)")
    args.M, args.N, args.K = M, N, K
    args.out_dtype = _DT[out.dtype]
    args.in_dtype = _DT[a.dtype]
    args.epilogue = epi
    args.tile = tile
    args.acc_scale = acc_scale
    args.dropout_p, args.dropout_seed, args.dropout_site = _drop(dropout)
    if splitk_ws is not None:
        _req(splitk_ws, torch.uint8, "gemm_bf16.splitk_ws")
        args.splitk_ws, args.splitk_ws_bytes = splitk_ws.data_ptr(), splitk_ws.numel()
    check(lib().tcavt_gemm_bf16(ctypes.byref(args), stream_ptr()), "tcavt_gemm_bf16")
    return out


def silu_bwd_fusable(M, I, H):
    """Shapes for which gemm_silu_bwd exists (whole 256 x 256 tiles of the 4-wave kernel)."""
    return M % 256 == 0 and I % 256 == 0 and H % 64 == 0 and H >= 128


def gemm_silu_bwd(g, w_t, gu, out=None):
    """d(gate|up) = silu_mul_bwd(gu, g @ w_t.T) with the activation gradient never leaving the GEMM (TCAVT_EPI_SILU_BWD):
    g 16-bit [M, H] = dL/d(down_proj output), w_t [I, H] = down_proj.weight^T, gu [M, 2I] the forward's gate|up
    pre-activations (interleaved layout); out [M, 2I] (default: gu itself, in place)."""
    _req16(g, "gemm_silu_bwd.g", rows_ok=True)
    _req16(w_t, "gemm_silu_bwd.w_t", like=g, rows_ok=True)
    _req16(gu, "gemm_silu_bwd.gu", like=g, rows_ok=True)
    M, H = g.shape
    I = w_t.shape[0]
    out = gu if out is None else out
    _req16(out, "gemm_silu_bwd.out", like=g, rows_ok=True)
    if w_t.shape[1] != H or gu.shape[0] < M or gu.shape[1] < 2 * I or out.shape[0] < M or out.shape[1] < 2 * I or not silu_bwd_fusable(M, I, H):
        raise capi.TcavtError("gemm_silu_bwd: shapes (g [M,H], w_t [I,H], gu / out [M,2I]; M, I multiples of 256)")
    args = capi.GemmArgs()
    args.A, args.lda, args.W, args.ldw = g.data_ptr(), g.stride(0), w_t.data_ptr(), w_t.stride(0)
    args.C, args.ldc = out.data_ptr(), out.stride(0)
    args.silu_preact, args.ld_preact = gu.data_ptr(), gu.stride(0)
    args.M, args.N, args.K = M, I, H
    args.out_dtype = args.in_dtype = _DT[g.dtype]
    args.epilogue = capi.EPI_SILU_BWD
    check(lib().tcavt_gemm_bf16(ctypes.byref(args), stream_ptr()), "tcavt_gemm_bf16(SILU_BWD)")
    return out


def _avail(t):
    """Elements addressable from t.data_ptr() to the end of its storage."""
    return t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()


_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: capi.F16}


def gemm_batched(a, w, out, *, M, N, K, lda, ldw, ldc, batch, inner, sA, sW, sC, bias_row=None, acc_scale=1.0,
                 tile=128, w_group=1):
    """`batch` independent products C_i = A_i . W_i^T with two-level strides (outer, inner) in elements:
    product i uses a + (i // inner) * sA[0] + (i % inner) * sA[1], likewise w and out.  a/w share a
    16-bit dtype (bf16 or fp16); out may be fp32, bf16 or fp16.  See tcavt_gemm_args (include/tcavt.h)."""
    if a.dtype != w.dtype or a.dtype not in (torch.bfloat16, torch.float16):
        raise capi.TcavtError("gemm_batched: a and w must both be bf16 or both fp16")
    outer = batch // inner
    if batch < 1 or batch % inner:
        raise capi.TcavtError("gemm_batched: batch must be a positive multiple of inner")
    for t, s, rows, ld, cols, nm in ((a, sA, M, lda, K, "a"), (w, sW, N, ldw, K, "w"), (out, sC, M, ldc, N, "out")):
        if not t.is_cuda and not _ALLOW_CPU:
            raise capi.TcavtError(f"gemm_batched.{nm}: tensor must live on the GPU")
        last_inner = (inner - 1) // w_group if nm == "w" else inner - 1  # w_group products share one W
        need = (outer - 1) * s[0] + last_inner * s[1] + (rows - 1) * ld + cols
        if _avail(t) < need:
            raise capi.TcavtError(f"gemm_batched.{nm}: buffer has {_avail(t)} elements past its pointer, needs {need}")
    args = capi.GemmArgs()
    args.A, args.lda = a.data_ptr(), lda
    args.W, args.ldw = w.data_ptr(), ldw
    args.C, args.ldc = out.data_ptr(), ldc
    args.M, args.N, args.K = M, N, K
    args.out_dtype = _DT[out.dtype]
    args.in_dtype = _DT[a.dtype]
    args.acc_scale = acc_scale
    args.tile = tile
    if bias_row is not None:
        _req(bias_row, torch.float32, "gemm_batched.bias_row")
        _need(bias_row, M, "gemm_batched.bias_row")
        args.bias = bias_row.data_ptr()
        args.epilogue = capi.EPI_BIAS_ROW
    args.batch, args.batch_inner = batch, inner
    args.batch_w_group = w_group
    args.sAo, args.sAi = sA
    args.sWo, args.sWi = sW
    args.sCo, args.sCi = sC
    check(lib().tcavt_gemm_bf16(ctypes.byref(args), stream_ptr()), "tcavt_gemm_bf16(batched)")
    return out


def softmax_rows(s, p, rows, n_valid, n_out, lds, ldp, dropout=None):
    _req(s, torch.float32, "softmax_rows.s")
    if p.dtype not in (torch.bfloat16, torch.float16):
        raise capi.TcavtError("softmax_rows.p: must be bf16 or fp16")
    if _avail(s) < (rows - 1) * lds + n_valid or _avail(p) < (rows - 1) * ldp + n_out:
        raise capi.TcavtError("softmax_rows: buffer too small")
    dp, dseed, dsite = _drop(dropout)
    check(lib().tcavt_softmax_rows(ptr(s), lds, ptr(p), ldp, _DT[p.dtype], rows, n_valid, n_out, dp, dseed, dsite,
                                   stream_ptr()), "tcavt_softmax_rows")


def set_dropout_epoch(epoch):
    """Register (or, with None, clear) the device-resident int64 that every dropout kernel adds to its seed when it runs
    (fresh masks under hipGraph replay, tcavt_set_dropout_epoch).  The tensor must outlive its registration."""
    if epoch is not None:
        _req(epoch, torch.int64, "set_dropout_epoch.epoch")
        _need(epoch, 1, "set_dropout_epoch.epoch")
    check(lib().tcavt_set_dropout_epoch(ptr(epoch)), "tcavt_set_dropout_epoch")


def dropout_epoch_advance(epoch):
    _req(epoch, torch.int64, "dropout_epoch_advance.epoch")
    check(lib().tcavt_dropout_epoch_advance(ptr(epoch), stream_ptr()), "tcavt_dropout_epoch_advance")


def dropout_(x, spec):
    """In-place dropout with a (p, seed, site) spec, no-op for None: how the backward re-applies a forward mask to a
    gradient of the same shape."""
    if spec is not None:
        dropout(x, x, *spec)
    return x


def dropout(x, out, p, seed, site, add=None):
    """out = x * keep / (1 - p) (+ add) with the Philox mask of (seed, site); x/out/add share fp32, bf16 or fp16; in
    place allowed."""
    if x.dtype != out.dtype or x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise capi.TcavtError("dropout: x and out must share one of fp32 / bf16 / fp16")
    _need(out, x.numel(), "dropout.out")
    if add is not None and (add.dtype != x.dtype or add.numel() < x.numel()):
        raise capi.TcavtError("dropout.add: same dtype and at least as many elements as x")
    check(lib().tcavt_dropout(ptr(x), ptr(out), x.numel(), _DT[x.dtype], float(p), int(seed) & 0xFFFFFFFFFFFFFFFF,
                              int(site) & 0xFFFFFFFF, ptr(add), stream_ptr()), "tcavt_dropout")
    return out


def rmsnorm(x, gamma, eps, out_bf16=None, out_f32=None, out_drop=None, dropout=None):
    """out_drop (bf16, with dropout=(p, seed, site)): dropout(out_bf16) from the same pass (LoRA branch input)."""
    _req(x, torch.float32, "rmsnorm.x")
    _req(gamma, torch.float32, "rmsnorm.gamma")
    M, H = x.shape
    _need(gamma, H, "rmsnorm.gamma")
    _need(out_bf16, M * H, "rmsnorm.out_bf16")
    _need(out_f32, M * H, "rmsnorm.out_f32")
    if (out_drop is None) != (dropout is None):
        raise capi.TcavtError("rmsnorm: out_drop and dropout go together")
    _req16(out_bf16, "rmsnorm.out_bf16")
    if out_drop is not None:
        _req16(out_drop, "rmsnorm.out_drop", like=out_bf16)
        _need(out_drop, M * H, "rmsnorm.out_drop")
    dt16 = _DT[out_bf16.dtype] if out_bf16 is not None else (_DT[out_drop.dtype] if out_drop is not None else BF16)
    check(lib().tcavt_rmsnorm(ptr(x), ptr(gamma), eps, ptr(out_bf16), ptr(out_f32), M, H, ptr(out_drop),
                              *_drop(dropout), dt16, stream_ptr()), "tcavt_rmsnorm")


def layernorm(x, gamma, beta, eps=1e-5, residual=None, out_f32=None, out_bf16=None):
    _req(x, torch.float32, "layernorm.x")
    _req(residual, torch.float32, "layernorm.residual")
    M, D = x.shape
    for t, n, nm in ((gamma, D, "gamma"), (beta, D, "beta"), (residual, M * D, "residual"), (out_f32, M * D, "out_f32"),
                     (out_bf16, M * D, "out_bf16")):
        _need(t, n, "layernorm." + nm)
    _req16(out_bf16, "layernorm.out_bf16")
    check(lib().tcavt_layernorm(ptr(x), ptr(residual), ptr(gamma), ptr(beta), eps, ptr(out_f32), ptr(out_bf16), M,
                                D, _DT[out_bf16.dtype] if out_bf16 is not None else BF16, stream_ptr()), "tcavt_layernorm")


def cast16(x, out=None, dtype=torch.float16):
    """fp32 -> fp16 / bf16 copy (round to nearest even); the type is `out`'s when given."""
    _req(x, torch.float32, "cast.x")
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _req16(out, "cast.out")
    _need(out, x.numel(), "cast.out")
    check(lib().tcavt_cast_f32_16(ptr(x), ptr(out), x.numel(), _DT[out.dtype], stream_ptr()), "tcavt_cast_f32_16")
    return out


def cast_bf16(x, out=None):
    """fp32 -> bf16 (gradient-side tensors); out may also be an fp16 buffer, whose type then wins."""
    return cast16(x, out, torch.bfloat16)


def pack_weight16(w, out=None):
    """Fragment-major copy of a 16-bit weight matrix [N, K] for the skinny (decode-step) form of gemm_bf16
    (tcavt_pack_weight16; tcavt_gemm_args.w_layout = W_FRAG16): same N * K elements, every 16-row x 32-column fragment a run
    of 1 KiB in the consuming wave's lane order."""
    assert w.dim() == 2 and w.dtype in (torch.float16, torch.bfloat16) and w.stride(1) == 1
    N, K = w.shape
    if out is None:
        out = torch.empty(N * K, dtype=w.dtype, device=w.device)
    assert out.numel() == N * K and out.dtype == w.dtype and out.is_contiguous()
    capi.check(capi.lib().tcavt_pack_weight16(w.data_ptr(), w.stride(0), out.data_ptr(), N, K, capi.stream_ptr()), "pack_weight16")
    return out


def norm_npart(M, N, K):
    """Partial sums of squares per row that a NORM_OUT product [M, N] over K writes (tcavt_norm_npart)."""
    return int(lib().tcavt_norm_npart(int(M), int(N), int(K)))


def embed_fuse(table, ids, img, vis_mod, txt_mod, h, bad_flag, h16=None, part=None, npart=None, stream_scale=1.0):
    """h16 / part (both or neither): the 16-bit copy of h and its rows' partial sums of squares [rows, H / 64] -- the
    inputs of the first decoder layer's fused RMSNorm (tcavt_llama_stack_forward).  stream_scale: h16 and part hold
    stream_scale * h (tcavt_llama_stack_args.stream_scale)."""
    _req16(table, "embed.table")
    _req(ids, torch.int64, "embed.ids")
    _req(img, torch.float32, "embed.img")
    B, Lt = ids.shape
    V, H = table.shape
    Nq = img.numel() // (B * H)
    _need(img, B * Nq * H, "embed.img")
    if h is None and h16 is None:
        raise capi.TcavtError("embed: h (fp32 stream) or h16 (16-bit residual stream) required")
    if h is not None:
        _req(h, torch.float32, "embed.h")
        _need(h, B * (Nq + Lt) * H, "embed.h")
    _need(vis_mod, H, "embed.vis_mod")
    _need(txt_mod, H, "embed.txt_mod")
    _need(bad_flag, 1, "embed.bad_flag")
    if (h16 is None) != (part is None):
        raise capi.TcavtError("embed.h16 and embed.part go together")
    if h16 is not None:
        _req16(h16, "embed.h16", like=table)
        _req(part, torch.float32, "embed.part")
        npart = int(npart or H // 64)
        _need(h16, B * (Nq + Lt) * H, "embed.h16")
        _need(part, B * (Nq + Lt) * npart, "embed.part")
    check(lib().tcavt_embed_fuse(ptr(table), ptr(ids), ptr(img), ptr(vis_mod), ptr(txt_mod), ptr(h), B, Nq, Lt, H,
                                 V, ptr(bad_flag), _DT[table.dtype], ptr(h16), ptr(part), npart if h16 is not None else 0, float(stream_scale),
                                 stream_ptr()),
          "tcavt_embed_fuse")


def mask_to_kvlen(mask, Nq, kv_len, flag):
    _req(mask, torch.int64, "mask_to_kvlen.mask")
    B, Lt = mask.shape
    _need(kv_len, B, "mask_to_kvlen.kv_len")
    _need(flag, 1, "mask_to_kvlen.flag")
    check(lib().tcavt_mask_to_kvlen(ptr(mask), B, Lt, Nq, ptr(kv_len), ptr(flag), stream_ptr()),
          "tcavt_mask_to_kvlen")


def attn_causal_gqa(qkv, out, kv_len, B, L, nq, nkv, scale, lse=None):
    """lse (optional, fp32 [B, nq, L]): receives the log-sum-exp of every query row's scaled scores (for attn_bwd_scores)."""
    _req16(qkv, "attn.qkv")
    _req16(out, "attn.out", like=qkv)
    _req(kv_len, torch.int32, "attn.kv_len")
    _need(qkv, B * L * (nq + 2 * nkv) * 64, "attn.qkv")
    _need(out, B * L * nq * 64, "attn.out")
    _need(kv_len, B, "attn.kv_len")
    if lse is not None:
        _req(lse, torch.float32, "attn.lse")
        _need(lse, B * nq * L, "attn.lse")
    check(lib().tcavt_attn_causal_gqa_lse(ptr(qkv), ptr(out), ptr(lse) if lse is not None else None, ptr(kv_len), B, L, nq, nkv,
                                          scale, _DT[qkv.dtype], stream_ptr()), "tcavt_attn_causal_gqa")


def mha(q, k, v, out, B, Lq, Lk, nh, dh, scale, key_len=None, ldq=None, ldk=None, ldv=None, ldo=None, dropout=None):
    """q/k/v may be column slices of wider row-major buffers: pass the slice's data_ptr tensor and ld."""
    in_dt, out_dt = _DT[q.dtype], _DT[out.dtype]
    if k.dtype != q.dtype or v.dtype != q.dtype:
        raise capi.TcavtError("mha: q, k, v must share a dtype")
    lq, lk, lv, lo = (ldq or q.stride(-2)), (ldk or k.stride(-2)), (ldv or v.stride(-2)), (ldo or out.stride(-2))
    E = nh * dh
    for t, rows, ld, nm in ((q, B * Lq, lq, "q"), (k, B * Lk, lk, "k"), (v, B * Lk, lv, "v"), (out, B * Lq, lo, "out")):
        if ld < E:
            raise capi.TcavtError(f"mha.{nm}: leading dimension {ld} < nh*dh = {E}")
        avail = t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()
        if avail < (rows - 1) * ld + E:
            raise capi.TcavtError(f"mha.{nm}: buffer too small for {rows} rows of stride {ld}")
    _need(key_len, B, "mha.key_len")
    check(lib().tcavt_mha(ptr(q), ldq if ldq else q.stride(-2), ptr(k), ldk if ldk else k.stride(-2), ptr(v),
                          ldv if ldv else v.stride(-2), ptr(out), ldo if ldo else out.stride(-2), ptr(key_len), B,
                          Lq, Lk, nh, dh, scale, in_dt, out_dt, *_drop(dropout), stream_ptr()), "tcavt_mha")


def gemm_f32(a, w, out=None, bias=None, relu=False, residual=None, dropout=None):
    _req(a, torch.float32, "gemm_f32.a")
    _req(w, torch.float32, "gemm_f32.w")
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    flags = (EPI_BIAS if bias is not None else 0) | (EPI_RELU if relu else 0) | (
        EPI_RESIDUAL if residual is not None else 0)
    if w.shape[1] != K or out.shape[0] < M or out.shape[1] < N:
        raise capi.TcavtError(f"gemm_f32: shape mismatch a{tuple(a.shape)} w{tuple(w.shape)} out{tuple(out.shape)}")
    if bias is not None:
        _need(bias, N, "gemm_f32.bias")
    if residual is not None and (residual.shape[0] < M or residual.shape[1] < N):
        raise capi.TcavtError("gemm_f32.residual: smaller than (M, N)")
    check(lib().tcavt_gemm_f32(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(residual),
                               residual.stride(0) if residual is not None else 0, ptr(out), out.stride(0), M, N, K,
                               flags, *_drop(dropout), stream_ptr()), "tcavt_gemm_f32")
    return out


def poly_embed(polygon, w_in, b_in, pos, out):
    B, P, _ = polygon.shape
    D = w_in.shape[0]
    _need(pos, P * D, "poly_embed.pos")
    _need(out, B * P * D, "poly_embed.out")
    check(lib().tcavt_poly_embed(ptr(polygon), ptr(w_in), ptr(b_in), ptr(pos), ptr(out), B, P, D, stream_ptr()),
          "tcavt_poly_embed")


def masked_mean(enc, lens, out, B, P, D):
    _need(enc, B * P * D, "masked_mean.enc")
    _need(lens, B, "masked_mean.lens")
    _need(out, B * D, "masked_mean.out")
    check(lib().tcavt_masked_mean(ptr(enc), ptr(lens), ptr(out), B, P, D, stream_ptr()), "tcavt_masked_mean")


def ltsf_front(x, conv_w, conv_b, enc_w, enc_b, pos, out, B, C, T, xp_tok=None):
    _need(xp_tok, B * T * C, "ltsf_front.xp_tok")
    for t, n, nm in ((x, B * 2 * T, "x"), (conv_w, C * 2, "conv_w"), (conv_b, C, "conv_b"), (enc_w, C * T * T, "enc_w"),
                     (enc_b, C * T, "enc_b"), (pos, C * T, "pos"), (out, B * T * C, "out")):
        _need(t, n, "ltsf_front." + nm)
    check(lib().tcavt_ltsf_front(ptr(x), ptr(conv_w), ptr(conv_b), ptr(enc_w), ptr(enc_b), ptr(pos), ptr(out),
                                 ptr(xp_tok), B, C, T, stream_ptr()), "tcavt_ltsf_front")


def ltsf_decode(e_tok, dec_w, dec_b, lane_adj, out, B, C, T, To):
    for t, n, nm in ((e_tok, B * T * C, "e_tok"), (dec_w, C * To * T, "dec_w"), (dec_b, C * To, "dec_b"),
                     (lane_adj, B * C * To, "lane_adj"), (out, B * C * To, "out")):
        _need(t, n, "ltsf_decode." + nm)
    check(lib().tcavt_ltsf_decode(ptr(e_tok), ptr(dec_w), ptr(dec_b), ptr(lane_adj), ptr(out), B, C, T, To,
                                  stream_ptr()), "tcavt_ltsf_decode")


def transpose_ct(x, out_f32, out_bf16, B, C, To):
    for t, nm in ((x, "x"), (out_f32, "out_f32"), (out_bf16, "out_bf16")):
        _need(t, B * C * To, "transpose_ct." + nm)
    _req16(out_bf16, "transpose_ct.out_bf16")
    check(lib().tcavt_transpose_ct(ptr(x), ptr(out_f32), ptr(out_bf16), B, C, To,
                                   _DT[out_bf16.dtype] if out_bf16 is not None else BF16, stream_ptr()), "tcavt_transpose_ct")


def out_head(fused, w, bias, x, out, B, To, C, F, T, add_last=True):
    for t, n, nm in ((fused, B * To * C, "fused"), (w, F * C, "w"), (bias, F, "bias"), (x, B * F * T, "x"),
                     (out, B * F * To, "out")):
        _need(t, n, "out_head." + nm)
    check(lib().tcavt_out_head(ptr(fused), ptr(w), ptr(bias), ptr(x), ptr(out), B, To, C, F, T, int(add_last),
                               stream_ptr()),
          "tcavt_out_head")


def traj_metrics(pred, gt, norm_stat, sums, argmin, per_sample, B, K, To):
    for t, n, nm in ((pred, B * K * 2 * To, "pred"), (gt, B * 2 * To, "gt"), (norm_stat, B * 4, "norm_stat"),
                     (sums, 5, "sums"), (argmin, B * 3, "argmin"), (per_sample, B * 3, "per_sample")):
        _need(t, n, "traj_metrics." + nm)
    check(lib().tcavt_traj_metrics(ptr(pred), ptr(gt), ptr(norm_stat), ptr(sums), ptr(argmin), ptr(per_sample), B,
                                   K, To, stream_ptr()), "tcavt_traj_metrics")


# ----------------------------------------------------------------------------------------------
# training-step kernels (include/tcavt.h, second half)
# ----------------------------------------------------------------------------------------------
def gemm_f32_strided(a, rsA, csA, w, rsW, csW, out, M, N, K, bias=None, relu=False, residual=None, accumulate=False):
    for t, rs, cs, rows, nm in ((a, rsA, csA, M, "a"), (w, rsW, csW, N, "w")):
        if _avail(t) < (rows - 1) * rs + (K - 1) * cs + 1:
            raise capi.TcavtError(f"gemm_f32_strided.{nm}: buffer too small")
    if _avail(out) < (M - 1) * out.stride(0) + N:
        raise capi.TcavtError("gemm_f32_strided.out: buffer too small")
    flags = (EPI_BIAS if bias is not None else 0) | (EPI_RELU if relu else 0) | (
        EPI_RESIDUAL if residual is not None else 0) | (capi.EPI_ACCUM if accumulate else 0)
    check(lib().tcavt_gemm_f32_strided(ptr(a), rsA, csA, ptr(w), rsW, csW, ptr(bias), ptr(residual),
                                       residual.stride(0) if residual is not None else 0, ptr(out), out.stride(0), M, N,
                                       K, flags, stream_ptr()), "tcavt_gemm_f32_strided")
    return out


def transpose16(x, out, rows, cols, rows_pad, ld_in=None, ld_out=None, batch=1, s_in=0, s_out=0):
    """out[c][r] = x[r][c] (16-bit), zero-filled for r in [rows, rows_pad); batched with strides.  An fp16 source with a
    bf16 destination is converted on the way (forward activations entering a gradient-side contraction)."""
    ld_in = ld_in or x.stride(-2)
    ld_out = ld_out or out.stride(-2)
    if _avail(x) < (batch - 1) * s_in + (rows - 1) * ld_in + cols or \
            _avail(out) < (batch - 1) * s_out + (cols - 1) * ld_out + rows_pad:
        raise capi.TcavtError("transpose16: buffer too small")
    if x.element_size() != 2 or out.element_size() != 2:
        raise capi.TcavtError("transpose16: 16-bit tensors only")
    if x.dtype != out.dtype and not (x.dtype == torch.float16 and out.dtype == torch.bfloat16):
        raise capi.TcavtError("transpose16: same 16-bit type on both sides, or fp16 -> bf16")
    check(lib().tcavt_transpose16(ptr(x), ld_in, ptr(out), ld_out, rows, cols, rows_pad, batch, s_in, s_out,
                                  int(x.dtype != out.dtype), stream_ptr()), "tcavt_transpose16")
    return out


def transpose_f32_bf16(x, out, rows, cols, rows_pad):
    _req(x, torch.float32, "transpose_f32_bf16.x")
    if _avail(x) < (rows - 1) * x.stride(0) + cols or _avail(out) < (cols - 1) * out.stride(0) + rows_pad:
        raise capi.TcavtError("transpose_f32_bf16: buffer too small")
    check(lib().tcavt_transpose_f32_bf16(ptr(x), x.stride(0), ptr(out), out.stride(0), rows, cols, rows_pad,
                                         stream_ptr()), "tcavt_transpose_f32_bf16")
    return out


def colsum(g, out, M, N, accumulate=False):
    _need(out, N, "colsum.out")
    if _avail(g) < (M - 1) * g.stride(0) + N:
        raise capi.TcavtError("colsum.g: buffer too small")
    check(lib().tcavt_colsum(ptr(g), g.stride(0), _DT[g.dtype], ptr(out), M, N, int(accumulate), stream_ptr()),
          "tcavt_colsum")


def relu_bwd(g, y):
    _req(g, torch.float32, "relu_bwd.g")
    _need(y, g.numel(), "relu_bwd.y")
    check(lib().tcavt_relu_bwd(ptr(g), ptr(y), _DT[y.dtype], g.numel(), stream_ptr()), "tcavt_relu_bwd")


def add_inplace(a, b):
    _req(a, torch.float32, "add_inplace.a")
    _req(b, torch.float32, "add_inplace.b")
    _need(b, a.numel(), "add_inplace.b")
    check(lib().tcavt_add_inplace(ptr(a), ptr(b), a.numel(), stream_ptr()), "tcavt_add_inplace")


def silu_mul_bwd(gu, g_act, g_gu):
    """d(silu(gate) * up) in the interleaved gate|up layout (layout.interleave_gate_up)."""
    M, I = g_act.shape
    for t, n, nm in ((gu, 2 * M * I, "gu"), (g_act, M * I, "g_act"), (g_gu, 2 * M * I, "g_gu")):
        _req(t, gu.dtype, "silu_mul_bwd." + nm)  # one 16-bit type for all three (bf16, or fp16 with scaled gradients)
        _need(t, n, "silu_mul_bwd." + nm)
    check(lib().tcavt_silu_mul_bwd(ptr(gu), ptr(g_act), ptr(g_gu), M, I, _DT16(gu), stream_ptr()), "tcavt_silu_mul_bwd")


def _DT16(t):
    if t.dtype not in _H16:
        raise capi.TcavtError(f"16-bit tensor (fp16 / bf16) required, got {t.dtype}")
    return _DT[t.dtype]


def grad_scale_pick(g_a, g_b, scale, scratch, target=256.0, backoff=None):
    """scale[0] = 2^k with max|g_a, g_b| * 2^k in [target / 2, target], scale[1] = its inverse -- on the device
    (tcavt_grad_scale_pick).  scratch: one zero-initialised int32."""
    n = g_a.numel()
    if g_b is not None and (g_b.dtype != g_a.dtype or g_b.numel() != n):
        raise capi.TcavtError("grad_scale_pick: g_b must match g_a")
    _req(scale, torch.float32, "grad_scale_pick.scale")
    _need(scale, 2, "grad_scale_pick.scale")
    if scratch.dtype != torch.int32 or scratch.numel() < 1:
        raise capi.TcavtError("grad_scale_pick.scratch: int32 [1] required")
    if backoff is not None:
        _req(backoff, torch.int32, "grad_scale_pick.backoff")
    check(lib().tcavt_grad_scale_pick(ptr(g_a), ptr(g_b) if g_b is not None else None, n, _DT16(g_a), float(target), ptr(scale),
                                      ptr(scratch), ptr(backoff), stream_ptr()), "tcavt_grad_scale_pick")


def rmsnorm_bwd(x, gamma, gy, gx, eps, gy2=None, accumulate=False, gx_bf16=None, gy_scale=None):
    M, H = x.shape
    if x.dtype not in (torch.float32,) + _H16:
        raise capi.TcavtError("rmsnorm_bwd.x: fp32 or a 16-bit residual stream required")
    _need(x, M * H, "rmsnorm_bwd.x")
    _req(gx, torch.float32, "rmsnorm_bwd.gx")
    _req(gamma, torch.float32, "rmsnorm_bwd.gamma")
    _need(gamma, H, "rmsnorm_bwd.gamma")
    _need(gx, M * H, "rmsnorm_bwd.gx")
    for t, nm in ((gy, "gy"), (gy2, "gy2"), (gx_bf16, "gx_bf16")):
        if t is not None:
            _DT16(t)
            _need(t, M * H, "rmsnorm_bwd." + nm)
    if gy2 is not None and gy2.dtype != gy.dtype:
        raise capi.TcavtError("rmsnorm_bwd: gy and gy2 must have one 16-bit type")
    if gy_scale is not None:
        _req(gy_scale, torch.float32, "rmsnorm_bwd.gy_scale")
    check(lib().tcavt_rmsnorm_bwd(ptr(x), ptr(gamma), ptr(gy), ptr(gy2) if gy2 is not None else None, eps, ptr(gx),
                                  ptr(gx_bf16) if gx_bf16 is not None else None, int(accumulate), M, H, _DT16(gy),
                                  _DT16(gx_bf16) if gx_bf16 is not None else 0,
                                  ptr(gy_scale) if gy_scale is not None else None, _DT[x.dtype], stream_ptr()),
          "tcavt_rmsnorm_bwd")


def rope_bwd_pack(g32, out, cos, sin, rope_cols, L):
    M, ncols = g32.shape
    _req(g32, torch.float32, "rope_bwd_pack.g32")
    _need(out, M * ncols, "rope_bwd_pack.out")
    _need(cos, L * 32, "rope_bwd_pack.cos")
    _need(sin, L * 32, "rope_bwd_pack.sin")
    check(lib().tcavt_rope_bwd_pack(ptr(g32), ptr(out), ptr(cos), ptr(sin), M, ncols, rope_cols, L, _DT16(out), stream_ptr()),
          "tcavt_rope_bwd_pack")


def attn_causal_gqa_bwd(qkv, dO, g32, kv_len, B, T, nq, nkv, scale):
    """Backward of attn_causal_gqa; g32 (fp32, q|k|v layout) must be zeroed by the caller."""
    ncols = (nq + 2 * nkv) * 64
    _req(qkv, torch.bfloat16, "attn_causal_gqa_bwd.qkv")
    _req(dO, torch.bfloat16, "attn_causal_gqa_bwd.dO")
    _req(g32, torch.float32, "attn_causal_gqa_bwd.g32")
    _need(qkv, B * T * ncols, "attn_causal_gqa_bwd.qkv")
    _need(g32, B * T * ncols, "attn_causal_gqa_bwd.g32")
    _need(dO, B * T * nq * 64, "attn_causal_gqa_bwd.dO")
    _need(kv_len, B, "attn_causal_gqa_bwd.kv_len")
    check(lib().tcavt_attn_causal_gqa_bwd(ptr(qkv), ptr(dO), ptr(g32), ptr(kv_len), B, T, nq, nkv, 64, scale,
                                          stream_ptr()), "tcavt_attn_causal_gqa_bwd")


def causal_softmax_bwd_rows(S, dP, P, dS, kv_len, B, T, Tp, nq, scale):
    rows = B * nq * T
    for t, dt, nm in ((S, torch.float32, "S"), (dP, torch.float32, "dP"), (P, torch.bfloat16, "P"), (dS, torch.bfloat16, "dS")):
        _req(t, dt, "causal_softmax_bwd_rows." + nm)
        _need(t, rows * Tp, "causal_softmax_bwd_rows." + nm)
    _need(kv_len, B, "causal_softmax_bwd_rows.kv_len")
    check(lib().tcavt_causal_softmax_bwd_rows(ptr(S), ptr(dP), ptr(P), ptr(dS), ptr(kv_len), B, T, Tp, nq, scale,
                                              stream_ptr()), "tcavt_causal_softmax_bwd_rows")


def causal_softmax_bwd_tiles(S, dP, dS, PT, dST, kv_len, B, T, Tp, nq, scale):
    """Tiled form of causal_softmax_bwd_rows: dS row-major plus P^T, dS^T; the outputs must have been zero-initialised
    once (blocks above the causal diagonal are never written)."""
    rows = B * nq * T
    for t, dt, n, nm in ((S, torch.float32, rows * Tp, "S"), (dP, torch.float32, rows * Tp, "dP"),
                         (dS, torch.bfloat16, rows * Tp, "dS"), (PT, torch.bfloat16, B * nq * Tp * Tp, "PT"),
                         (dST, torch.bfloat16, B * nq * Tp * Tp, "dST")):
        _req(t, dt, "causal_softmax_bwd_tiles." + nm)
        _need(t, n, "causal_softmax_bwd_tiles." + nm)
    _need(kv_len, B, "causal_softmax_bwd_tiles.kv_len")
    check(lib().tcavt_causal_softmax_bwd_tiles(ptr(S), ptr(dP), ptr(dS), ptr(PT), ptr(dST), ptr(kv_len), B, T, Tp, nq, scale,
                                               stream_ptr()), "tcavt_causal_softmax_bwd_tiles")


def attn_bwd_dkv(qkv, dO, stats, g32, kv_len, B, T, Tp, nq, nkv, scale):
    """dK, dV of the causal GQA attention (key-major MFMA kernel) into the k / v columns of g32 (fp32, q|k|v layout)."""
    ncols = (nq + 2 * nkv) * 64
    for t, dt, n, nm in ((qkv, qkv.dtype, B * T * ncols, "qkv"), (dO, qkv.dtype, B * T * nq * 64, "dO"),
                         (stats, torch.float32, B * nq * T * 4, "stats"), (g32, torch.float32, B * T * ncols, "g32")):
        if t.dtype != dt or (not t.is_cuda and not _ALLOW_CPU) or _avail(t) < n:
            raise capi.TcavtError(f"attn_bwd_dkv.{nm}: {dt} GPU buffer with {n} elements required")
    _need(kv_len, B, "attn_bwd_dkv.kv_len")
    check(lib().tcavt_attn_bwd_dkv(ptr(qkv), ptr(dO), ptr(stats), ptr(g32), ptr(kv_len), B, T, Tp, nq, nkv, 64, scale,
                                   _DT16(qkv), stream_ptr()), "tcavt_attn_bwd_dkv")


def attn_bwd_scores(qkv, dO, dS, PT, dST, kv_len, B, T, Tp, nq, nkv, scale, dQ=None, stats=None, lse=None, att=None):
    """Scores + softmax backward in one kernel (MFMA inside): P^T, dS^T (zero-initialised once), plus dQ = dS K (fp32
    [B*T, >= nq*64], any leading dimension) computed in place and / or the row-major dS for an external dQ product.
    lse (fp32 [B*nq*T]) + att (the forward's output, 16-bit [B*T, nq*64]), from attn_causal_gqa(..., lse=...): the row
    statistics come from the forward and the kernel sweeps the keys once instead of twice."""
    rows, ncols = B * nq * T, (nq + 2 * nkv) * 64
    if dQ is not None:
        if dQ.dtype != torch.float32 or dQ.stride(-1) != 1 or _avail(dQ) < (B * T - 1) * dQ.stride(0) + nq * 64:
            raise capi.TcavtError("attn_bwd_scores.dQ: fp32 [B*T, >= nq*64] required")
    for t, n, nm in ((qkv, B * T * ncols, "qkv"), (dO, B * T * nq * 64, "dO"), (dS, rows * Tp, "dS"),
                     (PT, B * nq * Tp * Tp, "PT"), (dST, B * nq * Tp * Tp, "dST")):
        if t is None and ((nm == "dS" and dQ is not None) or (nm in ("PT", "dST") and stats is not None)):
            continue
        if t.dtype != qkv.dtype or t.dtype not in _H16 or (not t.is_cuda and not _ALLOW_CPU):
            raise capi.TcavtError(f"attn_bwd_scores.{nm}: 16-bit GPU tensor of the type of qkv required")
        if _avail(t) < n:
            raise capi.TcavtError(f"attn_bwd_scores.{nm}: buffer too small")
    _need(kv_len, B, "attn_bwd_scores.kv_len")
    if stats is not None:
        _req(stats, torch.float32, "attn_bwd_scores.stats")
        _need(stats, rows * 4, "attn_bwd_scores.stats")
    if (lse is None) != (att is None):
        raise capi.TcavtError("attn_bwd_scores: lse and att come together")
    if lse is not None:
        _req(lse, torch.float32, "attn_bwd_scores.lse")
        _need(lse, rows, "attn_bwd_scores.lse")
        if att.dtype != qkv.dtype or not att.is_contiguous() or _avail(att) < B * T * nq * 64:
            raise capi.TcavtError("attn_bwd_scores.att: contiguous 16-bit [B*T, nq*64] of the type of qkv required")
    opt = lambda t: ptr(t) if t is not None else None
    check(lib().tcavt_attn_bwd_scores(ptr(qkv), ptr(dO), opt(dS), opt(PT), opt(dST), opt(dQ),
                                      dQ.stride(0) if dQ is not None else 0, opt(stats), ptr(kv_len), B, T, Tp, nq, nkv, 64,
                                      scale, _DT16(qkv), opt(lse), opt(att), stream_ptr()), "tcavt_attn_bwd_scores")


def attn_bwd_resident_ok(T, nq, nkv):
    """Shapes the two-launch resident form of the attention backward serves (T <= 256, 16 % (nq / nkv) == 0)."""
    return bool(lib().tcavt_attn_bwd_resident_ok(T, nq, nkv))


def attn_bwd_resident(qkv, dO, att, lse, g_qkv, stats, cos, sin, kv_len, B, T, nq, nkv, scale):
    """The whole causal GQA attention backward in two launches (tcavt_attn_bwd_resident): g_qkv 16-bit [B*T, (nq+2nkv)*64]
    = gradient of the projections' outputs, RoPE undone; att / lse from attn_causal_gqa(..., lse=...)."""
    ncols = (nq + 2 * nkv) * 64
    for t, n, nm in ((qkv, B * T * ncols, "qkv"), (dO, B * T * nq * 64, "dO"), (att, B * T * nq * 64, "att"),
                     (g_qkv, B * T * ncols, "g_qkv")):
        if t.dtype != qkv.dtype or t.dtype not in _H16 or not t.is_contiguous() or (not t.is_cuda and not _ALLOW_CPU):
            raise capi.TcavtError(f"attn_bwd_resident.{nm}: contiguous 16-bit GPU tensor of the type of qkv required")
        if _avail(t) < n:
            raise capi.TcavtError(f"attn_bwd_resident.{nm}: buffer too small")
    for t, n, nm in ((lse, B * nq * T, "lse"), (stats, B * nq * T * 4, "stats"), (cos, T * 32, "cos"), (sin, T * 32, "sin")):
        _req(t, torch.float32, f"attn_bwd_resident.{nm}")
        _need(t, n, f"attn_bwd_resident.{nm}")
    _need(kv_len, B, "attn_bwd_resident.kv_len")
    check(lib().tcavt_attn_bwd_resident(ptr(qkv), ptr(dO), ptr(att), ptr(lse), ptr(g_qkv), ptr(stats), ptr(cos), ptr(sin),
                                        ptr(kv_len), B, T, nq, nkv, 64, scale, _DT16(qkv), stream_ptr()), "tcavt_attn_bwd_resident")
    return g_qkv


def gqa_rope_bwd_pack(G3, out, cos, sin, nq, nkv, L):
    M = G3.shape[0]
    _req(G3, torch.float32, "gqa_rope_bwd_pack.G3")
    _req(out, torch.bfloat16, "gqa_rope_bwd_pack.out")
    _need(G3, M * 3 * nq * 64, "gqa_rope_bwd_pack.G3")
    _need(out, M * (nq + 2 * nkv) * 64, "gqa_rope_bwd_pack.out")
    _need(cos, L * 32, "gqa_rope_bwd_pack.cos")
    _need(sin, L * 32, "gqa_rope_bwd_pack.sin")
    check(lib().tcavt_gqa_rope_bwd_pack(ptr(G3), ptr(out), ptr(cos), ptr(sin), M, nq, nkv, 64, L, stream_ptr()),
          "tcavt_gqa_rope_bwd_pack")


def layernorm_bwd(x, gamma, gy, gx, ggamma, gbeta, eps=1e-5):
    M, D = x.shape
    for t, n, nm in ((gamma, D, "gamma"), (gy, M * D, "gy"), (gx, M * D, "gx"), (ggamma, D, "ggamma"), (gbeta, D, "gbeta")):
        _req(t, torch.float32, "layernorm_bwd." + nm)
        _need(t, n, "layernorm_bwd." + nm)
    check(lib().tcavt_layernorm_bwd(ptr(x), ptr(gamma), ptr(gy), eps, ptr(gx), ptr(ggamma), ptr(gbeta), M, D,
                                    stream_ptr()), "tcavt_layernorm_bwd")


def mha_bwd(q, k, v, go, gq, gk, gv, B, Lq, Lk, nh, dh, scale, key_len=None, ldq=None, ldk=None, ldv=None, ldo=None,
            ldg=None, dropout=None):
    E = nh * dh
    lq, lk, lv, lo, lg = (ldq or q.stride(-2)), (ldk or k.stride(-2)), (ldv or v.stride(-2)), (ldo or go.stride(-2)), (
        ldg or gq.stride(-2))
    for t, rows, ld, nm in ((q, B * Lq, lq, "q"), (k, B * Lk, lk, "k"), (v, B * Lk, lv, "v"), (go, B * Lq, lo, "go"),
                            (gq, B * Lq, lg, "gq"), (gk, B * Lk, lg, "gk"), (gv, B * Lk, lg, "gv")):
        if t.dtype != torch.float32 or _avail(t) < (rows - 1) * ld + E:
            raise capi.TcavtError(f"mha_bwd.{nm}: fp32 buffer with {rows} rows of stride {ld} required")
    _need(key_len, B, "mha_bwd.key_len")
    check(lib().tcavt_mha_bwd(ptr(q), lq, ptr(k), lk, ptr(v), lv, ptr(go), lo, ptr(gq), ptr(gk), ptr(gv), lg,
                              ptr(key_len), B, Lq, Lk, nh, dh, scale, *_drop(dropout), stream_ptr()), "tcavt_mha_bwd")


def softmax_bwd_rows(p_f16, dP, dS, scale, rows, n_valid, n_out, ldp, ldd, lds):
    if p_f16.dtype != torch.float16 or dP.dtype != torch.float32 or dS.dtype != torch.bfloat16:
        raise capi.TcavtError("softmax_bwd_rows: P fp16, dP fp32, dS bf16 required")
    if _avail(p_f16) < (rows - 1) * ldp + n_valid or _avail(dP) < (rows - 1) * ldd + n_valid or \
            _avail(dS) < (rows - 1) * lds + n_out:
        raise capi.TcavtError("softmax_bwd_rows: buffer too small")
    check(lib().tcavt_softmax_bwd_rows(ptr(p_f16), ldp, ptr(dP), ldd, ptr(dS), lds, scale, rows, n_valid, n_out,
                                       stream_ptr()), "tcavt_softmax_bwd_rows")


def mse_grad(pred, gt, norm_stat, g, B, To):
    for t, n, nm in ((pred, B * 2 * To, "pred"), (gt, B * 2 * To, "gt"), (norm_stat, B * 4, "norm_stat"), (g, B * 2 * To, "g")):
        _req(t, torch.float32, "mse_grad." + nm)
        _need(t, n, "mse_grad." + nm)
    check(lib().tcavt_mse_grad(ptr(pred), ptr(gt), ptr(norm_stat), ptr(g), B, To, stream_ptr()), "tcavt_mse_grad")


def out_head_bwd(g, fused, w, gf, gw, gb, B, To, C, F):
    for t, n, nm in ((g, B * F * To, "g"), (fused, B * To * C, "fused"), (w, F * C, "w"), (gf, B * To * C, "gf"),
                     (gw, F * C, "gw"), (gb, F, "gb")):
        _need(t, n, "out_head_bwd." + nm)
    check(lib().tcavt_out_head_bwd(ptr(g), ptr(fused), ptr(w), ptr(gf), ptr(gw), ptr(gb), B, To, C, F, stream_ptr()),
          "tcavt_out_head_bwd")


def nlinear_bwd(in_tok, W, g, g_strides, gW, gbias, gin_tok, B, C, T, S):
    sb, sc, ss = g_strides
    for t, n, nm in ((in_tok, B * T * C, "in_tok"), (W, C * S * T, "W"), (gW, C * S * T, "gW"), (gbias, C * S, "gbias"),
                     (gin_tok, B * T * C, "gin_tok")):
        _need(t, n, "nlinear_bwd." + nm)
    if _avail(g) < (B - 1) * sb + (C - 1) * sc + (S - 1) * ss + 1:
        raise capi.TcavtError("nlinear_bwd.g: buffer too small")
    check(lib().tcavt_nlinear_bwd(ptr(in_tok), ptr(W), ptr(g), sb, sc, ss, ptr(gW), ptr(gbias), ptr(gin_tok), B, C, T, S,
                                  stream_ptr()), "tcavt_nlinear_bwd")


def conv1x1_bwd(gxp_tok, x, gw, gb, B, C, T, F):
    for t, n, nm in ((gxp_tok, B * T * C, "gxp_tok"), (x, B * F * T, "x"), (gw, C * F, "gw"), (gb, C, "gb")):
        _need(t, n, "conv1x1_bwd." + nm)
    check(lib().tcavt_conv1x1_bwd(ptr(gxp_tok), ptr(x), ptr(gw), ptr(gb), B, C, T, F, stream_ptr()), "tcavt_conv1x1_bwd")


def poly_embed_bwd(g, polygon, gw, gb, gpos, B, P, D):
    for t, n, nm in ((g, B * P * D, "g"), (polygon, B * P * 2, "polygon"), (gw, D * 2, "gw"), (gb, D, "gb"), (gpos, P * D, "gpos")):
        _need(t, n, "poly_embed_bwd." + nm)
    check(lib().tcavt_poly_embed_bwd(ptr(g), ptr(polygon), ptr(gw), ptr(gb), ptr(gpos), B, P, D, stream_ptr()),
          "tcavt_poly_embed_bwd")


def masked_mean_bwd(gemb, lens, genc, B, P, D):
    for t, n, nm in ((gemb, B * D, "gemb"), (lens, B, "lens"), (genc, B * P * D, "genc")):
        _need(t, n, "masked_mean_bwd." + nm)
    check(lib().tcavt_masked_mean_bwd(ptr(gemb), ptr(lens), ptr(genc), B, P, D, stream_ptr()), "tcavt_masked_mean_bwd")


def wgrad_tn(g, g_col0, n, x, out, trans_out=False, rs_part=None, rs_h=0, rs_eps=0.0):
    """out[i, h] += sum_m g[m, g_col0 + i] * x[m, h] (i < n; trans_out: out[h, i]) -- skinny weight gradient without
    transposes (tcavt_wgrad_tn).  g bf16 [M, >= g_col0 + n], x fp16 / bf16 [M, H], out fp32, accumulated into.
    rs_part (fp32 [M, npart], with rs_h, rs_eps): rows of g are scaled by 1 / rms of their token on the way in."""
    if g.dtype not in _H16 or x.dtype not in _H16 or out.dtype != torch.float32 or (g.dtype == torch.float16 and x.dtype != torch.float16):
        raise capi.TcavtError("wgrad_tn: g bf16 with x fp16 / bf16, or g fp16 with x fp16; out fp32")
    M, H = x.shape
    if g.shape[0] < M or g.shape[1] < g_col0 + n or g.stride(1) != 1 or x.stride(1) != 1 or out.stride(1) != 1:
        raise capi.TcavtError("wgrad_tn: operand shapes / strides")
    rows, cols = (H, n) if trans_out else (n, H)
    if out.shape[0] < rows or out.shape[1] < cols:
        raise capi.TcavtError(f"wgrad_tn.out: needs at least ({rows}, {cols})")
    npart = 0
    if rs_part is not None:
        _req(rs_part, torch.float32, "wgrad_tn.rs_part")
        if rs_part.dim() != 2 or rs_part.shape[0] < M or not rs_part.is_contiguous() or rs_h <= 0:
            raise capi.TcavtError("wgrad_tn.rs_part: contiguous fp32 [M, npart] and rs_h > 0 required")
        npart = rs_part.shape[1]
    check(lib().tcavt_wgrad_tn(ptr(g), g.stride(0), int(g_col0), int(n), ptr(x), x.stride(0), _DT[x.dtype], ptr(out), out.stride(0),
                               M, H, int(trans_out), _DT[g.dtype], ptr(rs_part) if rs_part is not None else None, npart, int(rs_h),
                               float(rs_eps), stream_ptr()), "tcavt_wgrad_tn")


def lora_wgrad_a(x16, part, gamma, g_t, dA, eps, dropout=None, site_v=None):
    """dA (fp32 [>= 32, H], accumulated into) of both adapters in one pass over the layer's taped input stream x16 [M, H]:
    rows 0..15 = (g_t[:, :16] * rs)^T drop_q(x16) * gamma, rows 16..31 the v adapter (tcavt_lora_wgrad_a).  part: the
    stream's partial sums of squares [M, npart]; dropout = (p, seed, site_q) with site_v as in lora_down."""
    _req16(x16, "lora_wgrad_a.x16")
    _req16(g_t, "lora_wgrad_a.g_t", like=x16)
    _req(part, torch.float32, "lora_wgrad_a.part")
    _req(gamma, torch.float32, "lora_wgrad_a.gamma")
    _req(dA, torch.float32, "lora_wgrad_a.dA")
    M, H = x16.shape
    if (part.dim() != 2 or part.shape[0] < M or not part.is_contiguous() or g_t.shape[0] < M or g_t.shape[1] != 64
            or not g_t.is_contiguous() or dA.dim() != 2 or dA.shape[0] < 32 or dA.shape[1] < H or dA.stride(1) != 1):
        raise capi.TcavtError("lora_wgrad_a: part [M, npart], g_t [M, 64], dA [>= 32, >= H] required")
    _need(gamma, H, "lora_wgrad_a.gamma")
    p, seed, site = _drop(dropout)
    check(lib().tcavt_lora_wgrad_a(ptr(x16), ptr(part), part.shape[1], float(eps), ptr(gamma), ptr(g_t), ptr(dA), dA.stride(0), M, H,
                                   p, seed, site, site + 1 if site_v is None else int(site_v), _DT16(x16), stream_ptr()),
          "tcavt_lora_wgrad_a")


def clip_grad_norm(g, max_norm, scratch, grad_scale=1.0):
    """In-place g *= grad_scale, then clip of the flat gradient vector to max_norm (torch.nn.utils.clip_grad_norm_
    semantics); scratch: fp32, >= 1026 elements; afterwards scratch[1025] holds the norm of the scaled gradient before
    clipping, scratch[1024] the total factor applied.  grad_scale = 1 / world: clip the data-parallel MEAN gradient."""
    _req(g, torch.float32, "clip_grad_norm.g")
    _req(scratch, torch.float32, "clip_grad_norm.scratch")
    _need(scratch, 1026, "clip_grad_norm.scratch")
    check(lib().tcavt_clip_grad_norm(ptr(g), g.numel(), float(max_norm), float(grad_scale), ptr(scratch), stream_ptr()),
          "tcavt_clip_grad_norm")


def adamw(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    n = p.numel()
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _req(t, torch.float32, "adamw." + nm)
        _need(t, n, "adamw." + nm)
    check(lib().tcavt_adamw(ptr(p), ptr(g), ptr(m), ptr(v), n, lr, beta1, beta2, eps, weight_decay, int(step),
                            grad_scale, stream_ptr()), "tcavt_adamw")


def adamw_gated(p, g, m, v, lr, beta1, beta2, eps, weight_decay, loss, ctl, grad_scale=1.0, grad_norm=None):
    """AdamW update that the device skips when `loss` (device fp32 scalar) or `grad_norm` (optional device fp32 scalar)
    is not finite (modify_scripts/modify_train.py:1190-1196); the step count lives in ctl (int32[8], device)."""
    n = p.numel()
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _req(t, torch.float32, "adamw_gated." + nm)
        _need(t, n, "adamw_gated." + nm)
    _req(loss, torch.float32, "adamw_gated.loss")
    _need(loss, 1, "adamw_gated.loss")
    _req(ctl, torch.int32, "adamw_gated.ctl")
    _need(ctl, 8, "adamw_gated.ctl")
    if grad_norm is not None:
        _req(grad_norm, torch.float32, "adamw_gated.grad_norm")
        _need(grad_norm, 1, "adamw_gated.grad_norm")
    check(lib().tcavt_adamw_gated(ptr(p), ptr(g), ptr(m), ptr(v), n, lr, beta1, beta2, eps, weight_decay, grad_scale,
                                  ptr(loss), ptr(grad_norm), ptr(ctl), stream_ptr()), "tcavt_adamw_gated")


# ----------------------------------------------------------------------------------------------
# stage-level entry points (include/tcavt.h "Stage-level entry points")
# ----------------------------------------------------------------------------------------------
def llama_stack_forward(args):
    """args: capi.LlamaStackArgs filled by model.LlamaWithCrossAttnPEFT (which owns and sizes every buffer it names)."""
    check(lib().tcavt_llama_stack_forward(ctypes.byref(args), stream_ptr()), "tcavt_llama_stack_forward")


def lora_down(x16, a_cat, t, scale, dropout=None, site_v=None):
    """t[:, 0:16] | t[:, 16:32] = scale * dropout_q / dropout_v (x16) . a_cat[0:16 | 16:32]^T in one pass (tcavt_lora_down);
    dropout = (p, seed, site_q) with site_v = site_q + 1 unless given."""
    _req16(x16, "lora_down.x16")
    _req16(a_cat, "lora_down.a_cat", like=x16)
    _req16(t, "lora_down.t", like=x16)
    M, H = x16.shape
    _need(a_cat, 32 * H, "lora_down.a_cat")
    _need(t, M * 64, "lora_down.t")
    p, seed, site = _drop(dropout)
    check(lib().tcavt_lora_down(ptr(x16), ptr(a_cat), ptr(t), M, H, float(scale), p, seed, site,
                                (site + 1) if site_v is None else int(site_v), _DT[x16.dtype], stream_ptr()), "tcavt_lora_down")


def lora_dgrad(g_t, a_qT, a_vT, out, dropout=None, site_v=None):
    """out = mask_q * (g_t[:, :16] . A_q) + mask_v * (g_t[:, 16:32] . A_v) in one pass (tcavt_lora_dgrad); a_qT / a_vT [H, 64]
    with the other adapter's columns zero; dropout = (p, seed, site_q), site_v = site_q + 1 unless given."""
    _req16(g_t, "lora_dgrad.g_t")
    for t, nm in ((a_qT, "a_qT"), (a_vT, "a_vT"), (out, "out")):
        _req16(t, "lora_dgrad." + nm, like=g_t)
    M, H = out.shape
    _need(g_t, M * 64, "lora_dgrad.g_t")
    _need(a_qT, H * 64, "lora_dgrad.a_qT")
    _need(a_vT, H * 64, "lora_dgrad.a_vT")
    p, seed, site = _drop(dropout)
    check(lib().tcavt_lora_dgrad(ptr(g_t), ptr(a_qT), ptr(a_vT), ptr(out), M, H, p, seed, site,
                                 (site + 1) if site_v is None else int(site_v), _DT[g_t.dtype], stream_ptr()), "tcavt_lora_dgrad")


def copy_batch(dsts, srcs):
    """dsts[i].copy_(srcs[i]) for up to 16 (device tensor, pinned host tensor) pairs of equal size, as ONE kernel launch on the
    current stream (tcavt_copy_batch: the copying lanes read the pinned memory over the host link; no copy-engine transfer)."""
    n = len(dsts)
    if n != len(srcs) or not 0 < n <= 16:
        raise capi.TcavtError("copy_batch: 1 .. 16 (dst, src) pairs")
    D, S, Bn = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_int64 * n)()
    for i, (d, s) in enumerate(zip(dsts, srcs)):
        nb = d.numel() * d.element_size()
        if s.numel() * s.element_size() != nb or d.dtype != s.dtype or not d.is_contiguous() or not s.is_contiguous():
            raise capi.TcavtError(f"copy_batch: pair {i}: contiguous tensors of one dtype and size required")
        if not (d.is_cuda or _ALLOW_CPU) or not (s.is_cuda or s.is_pinned() or _ALLOW_CPU):
            raise capi.TcavtError(f"copy_batch: pair {i}: dst on the GPU, src on the GPU or in pinned host memory")
        D[i], S[i], Bn[i] = d.data_ptr(), s.data_ptr(), nb
    check(lib().tcavt_copy_batch(D, S, Bn, n, stream_ptr()), "tcavt_copy_batch")


def sample_workspace(B, device):
    """Workspace of the two-stage token selection for B samples (tcavt_sample_workspace_bytes): uint8, zeroed once here; every
    call leaves its control words zero.  Calls that share it must be stream-ordered."""
    return torch.zeros(int(lib().tcavt_sample_workspace_bytes(int(B))), dtype=torch.uint8, device=device)


def sample_logits(logits, history, hist_len, params, step, cur_tok, pos, finished, out_tokens, advance_pos, workspace=None):
    """Logits processors + token selection (tcavt_sample_logits); every tensor is device state that the call advances.
    workspace (sample_workspace(B, device)): the two-stage form -- B x 16 workgroups instead of B; same tokens."""
    B, V = logits.shape
    if workspace is not None:
        _req(workspace, torch.uint8, "sample_logits.workspace")
    _req(logits, torch.float32, "sample_logits.logits")
    for t, dt, n, nm in ((history, torch.int64, B, "history"), (hist_len, torch.int32, B, "hist_len"), (step, torch.int32, 1, "step"),
                         (cur_tok, torch.int64, B, "cur_tok"), (pos, torch.int32, B, "pos"), (finished, torch.int32, B, "finished"),
                         (out_tokens, torch.int64, B, "out_tokens")):
        _req(t, dt, "sample_logits." + nm)
        _need(t, n, "sample_logits." + nm)
    if history.shape[0] != B or out_tokens.shape[0] != B:
        raise capi.TcavtError("sample_logits: history / out_tokens need one row per sample")
    check(lib().tcavt_sample_logits(ptr(logits), B, V, ptr(history), history.shape[1], ptr(hist_len), ctypes.byref(params),
                                    ptr(step), ptr(cur_tok), ptr(pos), ptr(finished), ptr(out_tokens), out_tokens.shape[1],
                                    int(advance_pos), ptr(workspace), workspace.numel() if workspace is not None else 0,
                                    stream_ptr()), "tcavt_sample_logits")


def gather_last(src16, kv_len, out16, B, L, H):
    _req16(src16, "gather_last.src16")
    _req16(out16, "gather_last.out16", like=src16)
    _req(kv_len, torch.int32, "gather_last.kv_len")
    _need(src16, B * L * H, "gather_last.src16")
    _need(out16, B * H, "gather_last.out16")
    _need(kv_len, B, "gather_last.kv_len")
    check(lib().tcavt_gather_last(ptr(src16), ptr(kv_len), ptr(out16), B, L, H, stream_ptr()), "tcavt_gather_last")


def llama_decode_step(args):
    """args: capi.DecodeArgs filled by model.LlamaMultiModal.generate_batch (which owns and sizes every buffer)."""
    check(lib().tcavt_llama_decode_step(ctypes.byref(args), stream_ptr()), "tcavt_llama_decode_step")


def tlayer_stack_forward(args):
    """args: capi.TStackArgs filled by model._TLayerRunner.run_stack (which owns and sizes every buffer)."""
    check(lib().tcavt_tlayer_stack_forward(ctypes.byref(args), stream_ptr()), "tcavt_tlayer_stack_forward")


def cross_attn_forward(args):
    """args: capi.CrossAttnArgs filled by model.TransformerLTSF.forward (which owns and sizes every buffer)."""
    check(lib().tcavt_cross_attn_forward(ctypes.byref(args), stream_ptr()), "tcavt_cross_attn_forward")


def cross_attn_backward(args):
    """args: capi.CrossAttnBwdArgs filled by backward.Backward._xattn_absorbed."""
    check(lib().tcavt_cross_attn_backward(ctypes.byref(args), stream_ptr()), "tcavt_cross_attn_backward")


def ltsf_forward(args, phase):
    """args: capi.LtsfArgs filled by model.TransformerLTSF (which owns and sizes every buffer); phase 1 / 2 / 3."""
    check(lib().tcavt_ltsf_forward(ctypes.byref(args), int(phase), stream_ptr()), "tcavt_ltsf_forward")


def tlayer_stack_backward(args):
    """args: capi.TStackBwdArgs filled by backward.Backward.polygon (fp32 encoder layers)."""
    check(lib().tcavt_tlayer_stack_backward(ctypes.byref(args), stream_ptr()), "tcavt_tlayer_stack_backward")


def ltsf_backward(args, phase):
    """args: capi.LtsfBwdArgs filled by backward.Backward._ltsf_stage; phase 1 (head .. g_poly) / 2 (the rest) / 3."""
    check(lib().tcavt_ltsf_backward(ctypes.byref(args), int(phase), stream_ptr()), "tcavt_ltsf_backward")


def allreduce_flat(buf, nccl_comm):
    """In-place SUM all-reduce of a flat fp32 buffer on a raw RCCL communicator (an ncclComm_t as int / c_void_p) and the
    current stream: tcavt_allreduce_flat, the C host's form of the gradient-bucket exchange (Trainer itself goes through
    torch.distributed, whose process group owns its communicator)."""
    _req(buf, torch.float32, "allreduce_flat.buf")
    comm = nccl_comm if isinstance(nccl_comm, ctypes.c_void_p) else ctypes.c_void_p(int(nccl_comm) if nccl_comm else None)
    check(lib().tcavt_allreduce_flat(ptr(buf), buf.numel(), comm, stream_ptr()), "tcavt_allreduce_flat")
    return buf


def rmsnorm16(x16, gamma, eps, out16=None, out_f32=None):
    """RMSNorm of 16-bit rows (the final norm of the 16-bit residual stream): out16 and / or out_f32."""
    _req16(x16, "rmsnorm16.x16")
    _req(gamma, torch.float32, "rmsnorm16.gamma")
    M, H = x16.shape
    _need(gamma, H, "rmsnorm16.gamma")
    if out16 is None and out_f32 is None:
        out_f32 = torch.empty((M, H), dtype=torch.float32, device=x16.device)
    if out16 is not None:
        _req16(out16, "rmsnorm16.out16", like=x16)
        _need(out16, M * H, "rmsnorm16.out16")
    if out_f32 is not None:
        _req(out_f32, torch.float32, "rmsnorm16.out_f32")
        _need(out_f32, M * H, "rmsnorm16.out_f32")
    check(lib().tcavt_rmsnorm16(ptr(x16), ptr(gamma), eps, ptr(out16), ptr(out_f32), M, H, _DT[x16.dtype], stream_ptr()),
          "tcavt_rmsnorm16")
    return out16 if out_f32 is None else out_f32


def rownorm_prep(x, x16, part, npart=None, rounded_sums=False, stream_scale=1.0):
    """x16 = 16-bit copy of x [M, H]; part [M, npart] = (row's sum of squares, 0, ...): inputs of a fused RMSNorm.
    rounded_sums: sums of the rounded values (16-bit residual stream)."""
    _req(x, torch.float32, "rownorm_prep.x")
    _req16(x16, "rownorm_prep.x16")
    _req(part, torch.float32, "rownorm_prep.part")
    M, H = x.shape
    npart = npart or H // 64
    _need(x16, M * H, "rownorm_prep.x16")
    _need(part, M * npart, "rownorm_prep.part")
    check(lib().tcavt_rownorm_prep(ptr(x), ptr(x16), ptr(part), M, H, npart, _DT[x16.dtype], int(bool(rounded_sums)), float(stream_scale),
                                   stream_ptr()),
          "tcavt_rownorm_prep")


class StackEvents:
    """hipEvents for the in-situ kernel timing of tcavt_llama_stack_forward (10 per layer: start / stop around q|k|v,
    attention, o, gate|up, down)."""

    STAGES = ("qkv", "attn", "o", "gateup", "down")

    def __init__(self, n_layers):
        self.n = 10 * n_layers
        self.arr = (ctypes.c_void_p * self.n)()
        check(lib().tcavt_events_create(self.arr, self.n), "tcavt_events_create")

    def summary(self):
        """{stage: (launches, mean ms)} of the LAST pass recorded (call after a stream synchronise)."""
        out = {}
        for si, name in enumerate(self.STAGES):
            tot, cnt = 0.0, 0
            for li in range(self.n // 10):
                ms = ctypes.c_float(0.0)
                check(lib().tcavt_event_elapsed_ms(self.arr[li * 10 + 2 * si], self.arr[li * 10 + 2 * si + 1], ctypes.byref(ms)),
                      "tcavt_event_elapsed_ms")
                tot += ms.value
                cnt += 1
            out[name] = (cnt, tot / max(cnt, 1))
        return out

    def close(self):
        if self.arr is not None:
            lib().tcavt_events_destroy(self.arr, self.n)
            self.arr = None
