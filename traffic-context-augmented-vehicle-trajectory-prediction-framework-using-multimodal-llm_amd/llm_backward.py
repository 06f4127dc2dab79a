"""Backward through the frozen decoder layers for the LoRA-trainable variant.

modify_scripts/modify_train.py:512-528 freezes every base weight of the LLM and leaves `lora_A` / `lora_B` of q_proj and
v_proj trainable; the loss reaches them through the LTSF cross-attention's keys and values, i.e. through the decoder's
final hidden states.  This module walks the activation gradient back through the layers (the decoder saved its per-layer
residual streams, rotated q|k|v and LoRA down-projections: LlamaWithCrossAttnPEFT.save_for_backward) and produces the
adapter gradients:

    final RMSNorm  <-  g_final (bf16, two addends: through the key and through the value projection)
    per layer, last to first:
        down^T (dgrad GEMM)  ->  d(silu*up)  ->  gate|up^T (dgrad GEMM)  ->  RMSNorm backward        (MLP half)
        o^T (dgrad GEMM)  ->  causal GQA attention backward  ->  RoPE^T  ->  g(q|k|v)
        LoRA:  g_t = s * g(q|k|v) B_ext ;  dB = g(q|k|v)^T t ;  dA_q = g_tq^T drop_q(xn), dA_v = g_tv^T drop_v(xn) ;
               g_xl = mask_q * (g_tq A_q) + mask_v * (g_tv A_v)      (one dropout site per adapter, as PEFT has them)
        q|k|v^T (dgrad GEMM)  ->  RMSNorm backward                                                    (attention half)

The dgrad GEMMs are the forward's MFMA kernel on transposed copies of the frozen weights (prepared_T); the gate|up
pre-activations come from the forward (its SiLU epilogue writes a 16-bit copy in this mode: tcavt_gemm_args.silu_preact), the
normed rows feeding the adapters are recomputed.
Layer 0's input gradient is not formed: nothing below it is trainable.
"""
import ctypes
import math
import os

import torch

from . import capi, ops, streams
from .model import LORA_V


def _rup(x, m):
    return (x + m - 1) // m * m


def lora_named_parameters(model):
    """(name, parameter) of every LoRA matrix, in the order their gradients become ready (last layer first)."""
    out = []
    lw = model.mllm.llama_wrapper
    pre = "mllm.llama_wrapper.llama_model.model.layers."
    for li in reversed(range(lw.shape.layers)):
        a = lw.llama_model.model.layers[li].self_attn
        for proj, mod in (("q_proj", a.q_proj), ("v_proj", a.v_proj)):
            out.append((f"{pre}{li}.self_attn.{proj}.lora_A.weight", mod.lora_A.weight))
            out.append((f"{pre}{li}.self_attn.{proj}.lora_B.weight", mod.lora_B.weight))
    return out


def attn_bwd_composed(buf, qkv, dO, kv_len, B, T, nq, nkv, scale, cos, sin, g_qkv, scores="fused", lse=None, att=None):
    """Backward of the causal grouped-query attention.  scores="fused" (the product path): two MFMA kernels + the RoPE
    pack; "scores+gemm" / "gemm": the same mathematics with dK, dV (or all five products) on the batched MFMA GEMM around
    a row kernel -- kept as cross-checks.
    qkv: the forward's rotated q|k|v, bf16, rows B*T (+ >= 63 readable pad rows: the score products address keys up to
    the next multiple of 64); dO bf16 [B*T, nq*64]; g_qkv (out) bf16 [B*T, (nq+2nkv)*64] = gradient of the
    projections' outputs (RoPE undone).  buf(name, shape, dtype, zero=False) hands out reusable scratch (zero: zero-filled
    when first allocated).

        S = scale q K^T, dP = dO V^T                      per (sample, query head); the heads of a group share K / V
        P^T, dS^T, dQ = dS K: all of the above in one kernel (ops.attn_bwd_scores; S, dP, dS never reach memory)
        dK_h = dS^T q,  dV_h = P^T dO                      per query head on the batched GEMM; contraction over queries
        g(q|k|v) = RoPE^T(dQ | sum_group dK_h) | sum_group dV_h
    """
    hd = 64
    Tp = _rup(T, 64)
    nqkv, grp, BH = (nq + 2 * nkv) * hd, nq // nkv, B * nq
    f32, b16 = torch.float32, qkv.dtype
    if scores == "fused":
        # two MFMA kernels and the pack: query-major (row statistics, dQ), key-major (dK, dV; group sum in registers);
        # S, dP, P, dS never reach memory
        stats = buf("at.stats", (BH * T, 4), f32)
        if (lse is not None and ops.attn_bwd_resident_ok(T, nq, nkv) and cos.shape[0] >= T
                and os.environ.get("TCAVT_ATTN_BWD_NO_RESIDENT") is None):
            # a head's keys / queries fit in LDS: two launches, 16-bit output with the rotation undone in their epilogues
            return ops.attn_bwd_resident(qkv, dO, att, lse, g_qkv, stats, cos, sin, kv_len, B, T, nq, nkv, scale)
        g32 = buf("at.g32", (B * T, nqkv), f32)
        # (lse / att from the forward's tape: the scores kernel sweeps the keys once, not twice)
        ops.attn_bwd_scores(qkv, dO, None, None, None, kv_len, B, T, Tp, nq, nkv, scale, dQ=g32, stats=stats, lse=lse, att=att)
        ops.attn_bwd_dkv(qkv, dO, stats, g32, kv_len, B, T, Tp, nq, nkv, scale)
        ops.rope_bwd_pack(g32, g_qkv, cos, sin, (nq + nkv) * hd, T)
        return g_qkv
    # zero-initialised once: the kernels never write the key blocks above the causal diagonal
    PT, dST = buf("at.PT", (BH * Tp, Tp), b16, True), buf("at.dST", (BH * Tp, Tp), b16, True)
    qT = buf("at.qT", (nq * hd, B * Tp), b16)
    gT = buf("at.gT", (nq * hd, B * Tp), b16)
    G3 = buf("at.G3", (B * T, 3 * nq * hd), f32)
    k, v = qkv[:, nq * hd:], qkv[:, (nq + nkv) * hd:]
    ld3 = 3 * nq * hd
    if scores == "scores+gemm":  # scores kernel (S, dP, dQ inside), dK / dV on the batched GEMM from its P^T, dS^T
        ops.attn_bwd_scores(qkv, dO, None, PT, dST, kv_len, B, T, Tp, nq, nkv, scale, dQ=G3)
    else:  # "gemm": every product on the batched GEMM, fp32 S / dP scratch + the tile kernel (cross-check of the fused kernel)
        dS = buf("at.dS", (BH * T, Tp), b16, True)
        kT = buf("at.kT", (nkv * hd, B * Tp), b16)
        S, dP = buf("at.S", (BH * T, Tp), f32), buf("at.dP", (BH * T, Tp), f32)
        ops.gemm_batched(qkv, k, S, M=T, N=Tp, K=hd, lda=nqkv, ldw=nqkv, ldc=Tp, batch=BH, inner=nq,
                         sA=(T * nqkv, hd), sW=(T * nqkv, hd), sC=(nq * T * Tp, T * Tp), acc_scale=scale, w_group=grp)
        ops.gemm_batched(dO, v, dP, M=T, N=Tp, K=hd, lda=nq * hd, ldw=nqkv, ldc=Tp, batch=BH, inner=nq,
                         sA=(T * nq * hd, hd), sW=(T * nqkv, hd), sC=(nq * T * Tp, T * Tp), w_group=grp)
        ops.causal_softmax_bwd_tiles(S, dP, dS, PT, dST, kv_len, B, T, Tp, nq, scale)
        ops.transpose16(k, kT, T, nkv * hd, Tp, ld_in=nqkv, ld_out=B * Tp, batch=B, s_in=T * nqkv, s_out=Tp)
        ops.gemm_batched(dS, kT, G3, M=T, N=hd, K=Tp, lda=Tp, ldw=B * Tp, ldc=ld3, batch=BH, inner=nq,
                         sA=(nq * T * Tp, T * Tp), sW=(Tp, hd * B * Tp), sC=(T * ld3, hd), w_group=grp, tile=64)
    ops.transpose16(qkv, qT, T, nq * hd, Tp, ld_in=nqkv, ld_out=B * Tp, batch=B, s_in=T * nqkv, s_out=Tp)
    ops.transpose16(dO, gT, T, nq * hd, Tp, ld_in=nq * hd, ld_out=B * Tp, batch=B, s_in=T * nq * hd, s_out=Tp)
    ops.gemm_batched(dST, qT, G3[:, nq * hd:], M=T, N=hd, K=Tp, lda=Tp, ldw=B * Tp, ldc=ld3, batch=BH, inner=nq,
                     sA=(nq * Tp * Tp, Tp * Tp), sW=(Tp, hd * B * Tp), sC=(T * ld3, hd), tile=64)
    ops.gemm_batched(PT, gT, G3[:, 2 * nq * hd:], M=T, N=hd, K=Tp, lda=Tp, ldw=B * Tp, ldc=ld3, batch=BH, inner=nq,
                     sA=(nq * Tp * Tp, Tp * Tp), sW=(Tp, hd * B * Tp), sC=(T * ld3, hd), tile=64)
    ops.gqa_rope_bwd_pack(G3, g_qkv, cos, sin, nq, nkv, T)
    return g_qkv


class LoraBackward:
    def __init__(self, model, book):
        self.m, self.book = model, book
        self.scale_backoff = None  # device int32 [1]: training.Trainer points it at the gated optimizer's ctl[6] (dynamic loss scaling)
        self.lw = model.mllm.llama_wrapper
        if not self.lw.use_lora:
            raise ValueError("LoraBackward: the model has no LoRA adapters (use_lora=False)")
        self.ws = self.lw._ws
        self.input_grad = False  # True: also walk through layer 0's projections and return dL/d(inputs_embeds)

    def _buf(self, name, shape, dtype=None, zero=False):  # noqa: D401 (scratch from the decoder's workspace)
        """16-bit scratch is of the model's storage type (fp16: the forward's contract; gradients then run under the scale of
        ops.grad_scale_pick) unless a type is given."""
        dtype = self.lw.storage if dtype is None else dtype
        return self.ws.get(f"llbw.{name}.{str(dtype)[6:]}", shape, dtype, self.book.grads.device, zero=zero)

    def _stage_call_ok(self, tape, st, B, L, M, H, I, nq, nkv, r, fuse_silu, dev):
        """Shapes / tapes the C++ stage call serves (everything the fp16-contract product path produces at L <= 256)."""
        if dev.type != "cuda" or os.environ.get("TCAVT_PY_LLM_BACKWARD", "0") == "1" or not fuse_silu or H % 128 or r > 16:
            return False
        if any(os.environ.get(k) for k in ("TCAVT_ATTN_BWD_TWO_SWEEPS", "TCAVT_ATTN_BWD_NO_RESIDENT", "TCAVT_LORA_LEAF_UNFUSED")):
            return False  # (A/B switches of the Python composition)
        if not ops.attn_bwd_resident_ok(L, nq, nkv) or st != torch.float16 or self.lw.shape.head_dim != 64:
            return False  # (the stage call walks fp16 stream tapes with head_dim 64; anything else: the composition below)
        return all(sv.h_in.dtype == st and getattr(sv, "lse", None) is not None and getattr(sv, "part", None) is not None
                   and sv.t is not None for sv in tape.layers)

    def _stage_call(self, tape, P, PT, g_final_a, g_final_b, scale, leaf, cos, sin, st, B, L, H, I, nq, nkv, r, s, eps, pre, buf):
        lw, G = self.lw, self.book.g
        nL = len(tape.layers)
        arr = (capi.LlamaBwdLayer * nL)()
        for li in range(nL):
            d, dT, sv, c = P.layers[li], PT[li], tape.layers[li], arr[li]
            c.w_dT, c.w_guT, c.w_oT, c.w_qkvT = dT.w_d.data_ptr(), dT.w_gu.data_ptr(), dT.w_o.data_ptr(), dT.w_qkv.data_ptr()
            c.b_extT, c.a_qT, c.a_vT = dT.b_ext.data_ptr(), dT.a_q.data_ptr(), dT.a_v.data_ptr()
            c.g1, c.g2 = d.g1.data_ptr(), d.g2.data_ptr()
            c.h_in, c.h_mid, c.qkv, c.gu = sv.h_in.data_ptr(), sv.h_mid.data_ptr(), sv.qkv_padded.data_ptr(), sv.gu.data_ptr()
            c.att, c.lse, c.part, c.t = sv.att.data_ptr(), sv.lse.data_ptr(), sv.part.data_ptr(), sv.t.data_ptr()
            p = f"{pre}{li}.self_attn."
            outs = [G[p + n] for n in ("q_proj.lora_A.weight", "v_proj.lora_A.weight", "q_proj.lora_B.weight", "v_proj.lora_B.weight")]
            if any(not o.is_contiguous() or o.dtype != torch.float32 for o in outs):
                raise capi.TcavtError("LoraBackward: the adapters' gradient tensors must be contiguous fp32")
            c.g_Aq, c.g_Av, c.g_Bq, c.g_Bv = (o.data_ptr() for o in outs)
        a = capi.LlamaBackwardArgs()
        a.layers = arr
        a.h_last, a.gamma_final = tape.h_last.data_ptr(), P.g_final.data_ptr()
        a.g_final_a, a.g_final_b = g_final_a.data_ptr(), None if g_final_b is None else g_final_b.data_ptr()
        a.rope_cos, a.rope_sin, a.kv_len = cos.data_ptr(), sin.data_ptr(), tape.kv_len.data_ptr()
        a.scale, a.scale_scratch = scale.data_ptr(), self._buf("scale_scratch", (1,), torch.int32, zero=True).data_ptr()
        a.scale_backoff = None if self.scale_backoff is None else self.scale_backoff.data_ptr()
        for k in ("g_h", "g_hb", "g_xn", "g_xl", "g_att", "dA", "dB"):
            setattr(a, k, buf[k].data_ptr())
        a.g_qkv0, a.g_qkv1, a.g_t0, a.g_t1 = buf["g_qkv"][0].data_ptr(), buf["g_qkv"][1].data_ptr(), buf["g_t"][0].data_ptr(), buf["g_t"][1].data_ptr()
        a.stats = self._buf("at.stats", (B * nq * L, 4), torch.float32).data_ptr()
        if leaf is not None:
            if getattr(self, "_ev", None) is None:
                self._ev = (ctypes.c_void_p * 4)()
                capi.check(capi.lib().tcavt_events_create(self._ev, 4), "tcavt_events_create")
            a.leaf_stream, a.events = leaf.cuda_stream, self._ev
        a.n_layers, a.B, a.L, a.H, a.I, a.nq, a.nkv = nL, B, L, H, I, nq, nkv
        a.dtype16, a.npart, a.lora_rank, a.input_grad = ops._DT[st], tape.layers[0].part.shape[1], r, int(self.input_grad)
        a.rms_eps, a.lora_scale = eps, s
        dspec = tape.layers[0].dspec
        if dspec is not None:  # ((p, seed, site_q), (p, seed, site_v)) of layer 0: sites advance by two per layer
            a.lora_dropout_p, a.dropout_seed, a.lora_first_site = dspec[0][0], dspec[0][1] & 0xFFFFFFFFFFFFFFFF, dspec[0][2]
        capi.check(capi.lib().tcavt_llama_stack_backward(ctypes.byref(a), capi.stream_ptr()), "tcavt_llama_stack_backward")
        self._keep = (arr, a)

    def run(self, g_final_a, g_final_b=None):
        """g_final_a (+ g_final_b): 16-bit [B*L, H] gradient of the post-final-norm hidden states (bf16 when the model's
        storage is fp16: they are what the backward's scale is picked from and must not overflow themselves).  Returns the
        fp32 gradient of the decoder's input embeddings [B*L, H] when `input_grad` is set (QFormerBackward continues from it).

        fp16 storage (the default contract): every 16-bit tensor of the walk -- the forward's tapes AND the gradients -- is
        IEEE half; the gradients carry one power-of-two factor S chosen on the device so that max |g_final| * S ~ 2^8
        (tcavt_grad_scale_pick), the fp32 residual-gradient stream carries it too, and the adapters' fp32 weight gradients
        (and the returned input gradient) are multiplied by 1 / S at the end.  A value that still leaves the half range
        becomes inf, reaches the weight gradients and makes the gated optimizer skip the step (Trainer.skip_nonfinite)."""
        lw, G = self.lw, self.book.g
        tape = lw.tape
        if tape is None:
            raise RuntimeError("LoraBackward.run: no tape (set llama_wrapper.save_for_backward before the forward)")
        ll = lw.shape
        P, PT = lw._prepared(), lw.prepared_T()
        B, L = tape.B, tape.L
        M, H, I = B * L, ll.hidden, ll.inter
        nq, nkv, hd = ll.n_q_heads, ll.n_kv_heads, ll.head_dim
        nqkv, r = (nq + 2 * nkv) * hd, lw.lora_r
        dev = tape.h_last.device
        cos, sin = lw._rope_tables(L, dev)
        s = lw.lora_alpha / lw.lora_r
        eps = ll.rms_eps
        pre = "mllm.llama_wrapper.llama_model.model.layers."

        st = lw.storage
        scaled = st == torch.float16
        scale = self._buf("scale", (2,), torch.float32)       # [S, 1 / S]
        if scaled:
            if g_final_a.dtype != torch.bfloat16 or (g_final_b is not None and g_final_b.dtype != torch.bfloat16):
                raise ValueError("LoraBackward.run: with fp16 storage the incoming gradient(s) must be bf16 (range)")
        elif g_final_a.dtype != st:
            raise ValueError("LoraBackward.run: the incoming gradient must have the model's 16-bit storage type")
        inv_s = scale[1:2]
        g_h = self._buf("g_h", (M, H), torch.float32)
        g_hb = self._buf("g_hb", (M, H))
        xn = self._buf("xn", (M, H))
        xl = self._buf("xl", (M, H))
        xl2 = self._buf("xl2", (M, H))
        t_re = self._buf("t_re", (M, 64), zero=True)
        g_xl2 = self._buf("g_xl2", (M, H))
        g_xn = self._buf("g_xn", (M, H))
        g_xl = self._buf("g_xl", (M, H))
        fuse_silu = ops.silu_bwd_fusable(M, I, H) and os.environ.get("TCAVT_NO_SILU_BWD_FUSION", "0") != "1"
        g_act = None if fuse_silu else self._buf("g_act", (M, I))
        g_att = self._buf("g_att", (M, nq * hd))
        # the adapters' weight gradients are leaf work (nothing downstream reads them): they run on a side stream while the
        # main stream walks on to the next layer; what they read alternates between two buffers (layer parity), and a
        # buffer is rewritten only after the leaf that read it has finished
        multi = dev.type == "cuda" and os.environ.get("TCAVT_BW_SERIAL", "0") != "1"
        leaf = streams.side_stream(dev, 2) if multi else None
        leaf_done = [None, None]
        g_qkv2 = [self._buf(f"g_qkv{i}", (M, nqkv)) for i in range(2)]
        g_t2 = [self._buf(f"g_t{i}", (M, 64)) for i in range(2)]
        dA = self._buf("dA", (64, H), torch.float32)
        dB = self._buf("dB", (nqkv, 64), torch.float32)

        if self._stage_call_ok(tape, st, B, L, M, H, I, nq, nkv, r, fuse_silu, dev):
            # the whole walk as ONE C call (tcavt_llama_stack_backward: the loop below, issued from C++)
            self._stage_call(tape, P, PT, g_final_a, g_final_b, scale, leaf, cos, sin, st, B, L, H, I, nq, nkv, r, s, eps, pre,
                             dict(g_h=g_h, g_hb=g_hb, g_xn=g_xn, g_xl=g_xl, g_att=g_att, g_qkv=g_qkv2, g_t=g_t2, dA=dA, dB=dB))
            if self.input_grad and scaled:
                g_h.mul_(inv_s)
            return g_h if self.input_grad else None
        if scaled:
            ops.grad_scale_pick(g_final_a, g_final_b, scale, self._buf("scale_scratch", (1,), torch.int32, zero=True),
                                backoff=self.scale_backoff)
        ops.rmsnorm_bwd(tape.h_last, P.g_final, g_final_a, g_h, eps, gy2=g_final_b, gx_bf16=g_hb,
                        gy_scale=scale[0:1] if scaled else None)
        for li in reversed(range(ll.layers)):
            d, dT, sv = P.layers[li], PT[li], tape.layers[li]
            # ---- MLP half: h_out = h_mid + (silu(gate) * up) W_d^T,  gate|up = rmsnorm(h_mid) W_gu^T
            gu = sv.gu  # gate|up pre-activations, saved by the forward's SiLU epilogue (silu_preact)
            if fuse_silu:  # d(silu(gate) * up) in the dgrad GEMM's epilogue: dL/d(act) never reaches memory
                ops.gemm_silu_bwd(g_hb, dT.w_d, gu)
            else:
                ops.gemm_bf16(g_hb, dT.w_d, out=g_act)  # g_hb: 16-bit copy of g_h, written by the RMSNorm backward before
                ops.silu_mul_bwd(gu, g_act, gu)  # in place: every thread reads its block of gate / up before writing it
            ops.gemm_bf16(gu, dT.w_gu, out=g_xn)
            ops.rmsnorm_bwd(sv.h_mid, d.g2, g_xn, g_h, eps, accumulate=True, gx_bf16=g_hb)
            # ---- attention half: h_mid = h_in + att W_o^T
            ops.gemm_bf16(g_hb, dT.w_o, out=g_att)
            par = li & 1
            g_qkv, g_t = g_qkv2[par], g_t2[par]
            if leaf_done[par] is not None:
                torch.cuda.current_stream().wait_event(leaf_done[par])  # the leaf of layer li + 2 has read them
            one_sweep = os.environ.get("TCAVT_ATTN_BWD_TWO_SWEEPS", "0") != "1" and getattr(sv, "lse", None) is not None  # (A/B switch)
            attn_bwd_composed(self._buf, sv.qkv_padded, g_att, tape.kv_len, B, L, nq, nkv, 1.0 / math.sqrt(hd), cos, sin,
                              g_qkv, lse=sv.lse if one_sweep else None, att=sv.att if one_sweep else None)
            # ---- adapters: q|k|v += t B_ext^T,  t = bf16(s * dropout(xn) A_cat^T)
            ops.gemm_bf16(g_qkv, dT.b_ext, out=g_t, acc_scale=s)

            def adapter_grads(li=li, d=d, sv=sv, g_qkv=g_qkv, g_t=g_t):
                # the adapter branches' inputs and down-projections in the un-fused form the backward walks (the forward
                # kept only the un-normalised t of its fused RMSNorm): xn = rmsnorm(h_in), t = s * dropout(xn) A^T
                dA.zero_()
                dB.zero_()
                fused_leaf = (sv.h_in.dtype == g_t.dtype and getattr(sv, "part", None) is not None and H % 16 == 0
                              and os.environ.get("TCAVT_LORA_LEAF_UNFUSED") is None)  # (A/B switch)
                if fused_leaf:
                    # one pass over the taped input stream for dA (norm and both masks recomputed on the way in), dB from
                    # the taped un-normalised t with 1 / rms applied while it is staged: two launches, ~85 MB instead of
                    # nine passes and ~350 MB per layer
                    ops.lora_wgrad_a(sv.h_in, sv.part, d.g1, g_t, dA, eps, dropout=None if sv.dspec is None else sv.dspec[0],
                                     site_v=None if sv.dspec is None else sv.dspec[1][2])
                    ops.wgrad_tn(sv.t, 0, 2 * LORA_V, g_qkv, dB, trans_out=True, rs_part=sv.part, rs_h=H, rs_eps=eps)
                elif sv.dspec is not None:  # ... with the forward's two masks
                    if sv.h_in.dtype == torch.float32:
                        ops.rmsnorm(sv.h_in, d.g1, eps, out_bf16=xn, out_drop=xl, dropout=sv.dspec[0])
                    else:  # 16-bit residual stream on the tape
                        ops.rmsnorm16(sv.h_in, d.g1, eps, out16=xn)
                        ops.dropout(xn, xl, *sv.dspec[0])
                    ops.dropout(xn, xl2, *sv.dspec[1])
                    ops.lora_down(xn, dT.a_plain, t_re, s, dropout=sv.dspec[0], site_v=sv.dspec[1][2])
                    ops.wgrad_tn(g_t, 0, LORA_V, xl, dA)               # dA_q = g_tq^T drop_q(xn): token-major operands as they are
                    ops.wgrad_tn(g_t, LORA_V, LORA_V, xl2, dA[LORA_V:])
                else:
                    if sv.h_in.dtype == torch.float32:
                        ops.rmsnorm(sv.h_in, d.g1, eps, out_bf16=xn)
                    else:
                        ops.rmsnorm16(sv.h_in, d.g1, eps, out16=xn)
                    ops.lora_down(xn, dT.a_plain, t_re, s)
                    ops.wgrad_tn(g_t, 0, 2 * LORA_V, xn, dA)
                if not fused_leaf:
                    ops.wgrad_tn(t_re, 0, 2 * LORA_V, g_qkv, dB, trans_out=True)  # dB = g_qkv^T t, stored as [nqkv, 64]
                p = f"{pre}{li}.self_attn."
                if scaled:  # out of the backward's scale (device scalar, no synchronisation)
                    torch.mul(dA[:r], inv_s, out=G[p + "q_proj.lora_A.weight"])
                    torch.mul(dA[LORA_V:LORA_V + r], inv_s, out=G[p + "v_proj.lora_A.weight"])
                    torch.mul(dB[: nq * hd, :r], inv_s, out=G[p + "q_proj.lora_B.weight"])
                    torch.mul(dB[(nq + nkv) * hd:, LORA_V:LORA_V + r], inv_s, out=G[p + "v_proj.lora_B.weight"])
                else:
                    G[p + "q_proj.lora_A.weight"].copy_(dA[:r])
                    G[p + "v_proj.lora_A.weight"].copy_(dA[LORA_V:LORA_V + r])
                    G[p + "q_proj.lora_B.weight"].copy_(dB[: nq * hd, :r])
                    G[p + "v_proj.lora_B.weight"].copy_(dB[(nq + nkv) * hd:, LORA_V:LORA_V + r])

            if leaf is None:
                adapter_grads()
            else:
                ready = torch.cuda.Event()
                ready.record()
                leaf.wait_event(ready)
                with torch.cuda.stream(leaf):  # (xn, xl, dA, dB and the transposed copies are touched by this stream only)
                    adapter_grads()
                    leaf_done[par] = torch.cuda.Event()
                    leaf_done[par].record(leaf)
            if li == 0 and not self.input_grad:
                break
            if H % 128 == 0:  # both adapters' input gradients under their own masks, summed: one kernel
                ops.lora_dgrad(g_t, dT.a_q, dT.a_v, g_xl, dropout=None if sv.dspec is None else sv.dspec[0],
                               site_v=None if sv.dspec is None else sv.dspec[1][2])
            elif sv.dspec is None:
                ops.gemm_bf16(g_t, dT.a_cat, out=g_xl)
            else:
                ops.gemm_bf16(g_t, dT.a_q, out=g_xl2)
                ops.dropout_(g_xl2, sv.dspec[0])
                ops.gemm_bf16(g_t, dT.a_v, out=g_xl)
                ops.dropout(g_xl, g_xl, *sv.dspec[1], add=g_xl2)
            ops.gemm_bf16(g_qkv, dT.w_qkv, out=g_xn)
            ops.rmsnorm_bwd(sv.h_in, d.g1, g_xn, g_h, eps, gy2=g_xl, accumulate=True, gx_bf16=g_hb)
        if leaf is not None:
            torch.cuda.current_stream().wait_stream(leaf)
        if self.input_grad and scaled:
            g_h.mul_(inv_s)  # what continues below the decoder (Q-Former backward) runs unscaled
        return g_h if self.input_grad else None
