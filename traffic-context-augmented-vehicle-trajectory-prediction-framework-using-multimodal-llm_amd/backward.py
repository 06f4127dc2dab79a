"""Backward pass of the part of the model that scripts/train.py trains, over the C ABI.

In the reference, ``loss.backward()`` (train.py:1182) walks an autograd graph that only covers the
lane-polygon encoder and ``TransformerLTSF``: every MLLM parameter has ``requires_grad=False``
(train.py:1140-1142), so the LLM's final hidden states are a constant.  This module spells that
graph out stage by stage, in reverse, on the activations the forward retained
(``save_for_backward`` workspaces of model.py).  Nothing here calls torch autograd or torch math.

Precision: fp32 wherever the forward is fp32; the cross-attention projections run their backward
contractions in bf16 MFMA with fp32 accumulation (standard mixed precision), writing fp32 gradients.
"""
import contextlib
import ctypes
import math
import os

import torch

from . import ops, streams
from .model import XATTN_PAD


def _rup(n, m):
    return (n + m - 1) // m * m


class GradBook:
    """Flat fp32 gradient vector with one view per trainable parameter (and the matching flat
    parameter vector), ordered by the time the gradient becomes ready in the backward so that
    contiguous buckets can be all-reduced while the rest of the backward is still running."""

    def __init__(self, named_params, device):
        self.names = [n for n, _ in named_params]
        sizes = [p.numel() for _, p in named_params]
        self.offsets = {}
        off = 0
        for (n, p), sz in zip(named_params, sizes):
            self.offsets[n] = (off, sz, tuple(p.shape))
            off += _rup(sz, 4)  # keep every view 16-byte aligned
        self.total = off
        self.params = torch.zeros(off, dtype=torch.float32, device=device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=device)
        self.g = {}
        for n, p in named_params:
            o, sz, shape = self.offsets[n]
            view = self.params[o:o + sz].view(shape)
            view.copy_(p.data)
            p.data = view  # the nn.Parameter now aliases the flat vector (state_dict() is unchanged)
            self.g[n] = self.grads[o:o + sz].view(shape)

    def end_of(self, name):
        o, sz, _ = self.offsets[name]
        return o + _rup(sz, 4)


class _Fork:
    """Run the enclosed launches on `stream`, ordered after everything enqueued so far on the current stream."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        ev = torch.cuda.Event()
        ev.record()
        self.stream.wait_event(ev)
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()

    def __exit__(self, *a):
        return self.ctx.__exit__(*a)


class Backward:
    """Stream layout.  The backward of the trainable part is ~200 small launches; most of them are latency bound
    and leave the chip idle.  Only the activation gradients form a dependency chain: every weight / bias gradient
    (g_y^T x, column sums, their transposes and memsets) is a LEAF that nothing downstream reads.  The chain stays
    on the calling stream, the leaves go to a second stream, and the lane-polygon encoder's backward (independent of
    the LTSF encoder's once d(poly_emb) is known) to a third; everything joins before `run` returns.
    TCAVT_BW_SERIAL=1 keeps all of it on one stream (A/B, debugging)."""

    def __init__(self, model, book: GradBook, prefix_ltsf="ltsf.", prefix_poly="lane_polygon_encoder."):
        self.m, self.book = model, book
        self.pl, self.pp = prefix_ltsf, prefix_poly
        self.ws = model.ltsf._ws  # scratch for gradient activations
        self._leaf_streams, self._poly_stream, self._leaf_i = None, None, 0
        self.poly_after_chain = False  # training.Trainer: True for the LoRA-trainable variant (see _ltsf_stage)
        self._v_ready = None
        self._serial = os.environ.get("TCAVT_BW_SERIAL", "") == "1"

    # ---- streams ---------------------------------------------------------------------------
    def _multi(self):
        return (not self._serial) and self.book.grads.device.type == "cuda"

    def _ensure_streams(self):
        if self._multi() and self._leaf_streams is None:
            dev = self.book.grads.device
            # shared pool (streams.py): more streams than hardware queues would serialise behind one another
            self._leaf_streams = [streams.side_stream(dev, 4), streams.side_stream(dev, 5)]  # = slots 1, 2 of a pool of 3
            self._poly_stream = streams.side_stream(dev, 3)  # = slot 0

    def _leaf(self, pin=None):
        """Context for leaf work (weight / bias gradients): a leaf stream, behind the current stream's queue.
        Leaves are independent of one another and alternate between the streams [1:]; the one chain of leaves that
        does depend on earlier leaves (cross-attention K/V branch: d(k), d(v) -> their in-projection weight gradients,
        sharing one transposed copy of the hidden states) is pinned to stream 0."""
        if not self._multi():
            return contextlib.nullcontext()
        self._ensure_streams()
        cur = torch.cuda.current_stream()
        if any(cur == s for s in self._leaf_streams):
            return contextlib.nullcontext()
        if pin is not None or len(self._leaf_streams) == 1:
            return _Fork(self._leaf_streams[0])
        self._leaf_i = self._leaf_i % (len(self._leaf_streams) - 1) + 1
        return _Fork(self._leaf_streams[self._leaf_i])

    # ---- helpers ---------------------------------------------------------------------------
    def _buf(self, name, shape, dtype=torch.float32, zero=False):
        dev = self.book.grads.device
        return self.ws.get("bw." + name, shape, dtype, dev, zero=zero)

    def _dropped(self, g, spec, name):
        """Gradient through a dropout site: g o mask / (1 - p) with the forward's (p, seed, site), in a buffer of its own
        (g itself usually continues along a residual path).  spec None (eval arithmetic): g unchanged."""
        if spec is None:
            return g
        out = self._buf(name, tuple(g.shape), g.dtype)
        ops.dropout(g, out, *spec)
        return out

    def lin_bwd_f32(self, x, W, gy, gW, gb, gx=None):
        """y = x W^T + b (fp32).  x [M,K], W [N,K], gy [M,N] -> gW [N,K], gb [N], optional gx [M,K]."""
        M, K = x.shape
        N = W.shape[0]
        with self._leaf():  # += into the flat gradient (zeroed by zero_grad): no per-launch memsets
            ops.gemm_f32_strided(gy, 1, gy.stride(0), x, 1, x.stride(0), gW, N, K, M, accumulate=True)
            if gb is not None:
                ops.colsum(gy, gb, M, N, accumulate=True)
        if gx is not None:
            ops.gemm_f32_strided(gy, gy.stride(0), 1, W, 1, W.stride(0), gx, M, K, N)
        return gx

    def lin_bwd_bf16(self, tag, x_b, W, gy, gW, gb, gx=None, xT=None, pin=None):
        """y = x16 W16^T + b on MFMA.  x_b: the forward's 16-bit activation [M,K] (fp16 or bf16; its transposed copy is bf16
        either way); W fp32 param [N,K]; gy bf16 or fp32 [M,N].
        gW fp32 [N,K] = gy^T x, gb = colsum(gy), optional gx [M,K] (dtype of the given buffer) = gy W."""
        M, K = x_b.shape
        N = W.shape[0]
        Mp = _rup(M, 64)
        if gy.dtype == torch.float32:
            gy_b = self._buf(tag + ".gyb", (M, N), torch.bfloat16)
            ops.cast_bf16(gy, out=gy_b)
        else:
            gy_b = gy
        with self._leaf(pin):
            gyT = self._buf(tag + ".gyT", (N, Mp), torch.bfloat16)
            ops.transpose16(gy_b, gyT, M, N, Mp)
            if xT is None:
                xT = self._buf(tag + ".xT", (K, Mp), torch.bfloat16)
                ops.transpose16(x_b, xT, M, K, Mp)
            ops.gemm_bf16(gyT, xT, out=gW)  # [N, K] fp32
            if gb is not None:
                ops.colsum(gy, gb, M, N, accumulate=True)
        if gx is not None:
            WT = self._buf(tag + ".WT", (K, N), torch.bfloat16)
            ops.transpose_f32_bf16(W.detach(), WT, N, K, N)
            ops.gemm_bf16(gy_b, WT, out=gx)
        return xT

    # ---- TransformerLTSF ----------------------------------------------------------------------
    def ltsf(self, g_out, x_in, on_poly_grad=None):
        """g_out [B,2,To] = dL/d decoded; returns g_poly_emb [B, D_poly]."""
        m, G, ws = self.m.ltsf, self.book.g, self.m.ltsf._ws
        dec, sab = m.decoder, m.attn_block
        pl = self.pl
        B, F, T = x_in.shape
        C, To = m.d_model, m.out_len
        dev = g_out.device
        M = B * To
        H = dec.cross_dim
        nh = dec.cross_nhead
        dh = H // nh
        if self._ltsf_stage_specs() is not None:  # one C call per phase: tcavt_ltsf_backward (csrc/tlayers.hip)
            return self._ltsf_stage(g_out, x_in, on_poly_grad)
        fwd = lambda name, shape, dt=torch.float32: ws.get("lt." + name, shape, dt, dev)
        st = m.storage  # 16-bit type of the forward's activations (fp16 by default); gradient-side tensors are bf16, and
        # a forward activation that enters a gradient-side contraction is converted while it is transposed (ops.transpose16)
        # saved forward activations
        f2 = fwd("f2", (M, C)); f1 = fwd("f1", (M, C)); fn = fwd("fn", (M, C)); fused = fwd("fused", (M, C))
        cross = fwd("cross", (M, H), st); att = fwd("att", (M, H), st)
        q = fwd("q", (M, H), st); proj = fwd("proj", (M, H), st)
        dec_tb = fwd("dectb", (M, C), st)
        fl = dec.fusion_layer

        # head: out = f2 W_out^T + b (+ last position, no gradient needed)
        g_f2 = self._buf("g_f2", (M, C))
        ops.out_head_bwd(g_out, f2, dec.out_proj.weight, g_f2, G[pl + "decoder.out_proj.weight"],
                         G[pl + "decoder.out_proj.bias"], B, To, C, F)
        # fusion layer: LN -> Linear -> ReLU -> Linear (no residual)
        g_f1 = self._buf("g_f1", (M, C))
        self.lin_bwd_f32(f1, fl[3].weight, g_f2, G[pl + "decoder.fusion_layer.3.weight"],
                         G[pl + "decoder.fusion_layer.3.bias"], gx=g_f1)
        ops.relu_bwd(g_f1, f1)
        g_fn = self._buf("g_fn", (M, C))
        self.lin_bwd_f32(fn, fl[1].weight, g_f1, G[pl + "decoder.fusion_layer.1.weight"],
                         G[pl + "decoder.fusion_layer.1.bias"], gx=g_fn)
        g_dec_t = self._buf("g_dec_t", (M, C))  # gradient w.r.t. fused == first term of g(dec_t)
        ops.layernorm_bwd(fused, fl[0].weight, g_fn, g_dec_t, G[pl + "decoder.fusion_layer.0.weight"],
                          G[pl + "decoder.fusion_layer.0.bias"])
        # fused = cross W_un^T + b_un + dec_t
        g_cross = self._buf("g_cross", (M, H), torch.bfloat16)
        self.lin_bwd_bf16("un", cross, dec.dec_unproj.weight, g_dec_t, G[pl + "decoder.dec_unproj.weight"],
                          G[pl + "decoder.dec_unproj.bias"], gx=g_cross)
        # cross = att W_co^T + b_co
        g_att = self._buf("g_att", (M, H), torch.bfloat16)
        ca = dec.cross_attn
        self.lin_bwd_bf16("co", att, ca.out_proj.weight, g_cross, G[pl + "decoder.cross_attn.out_proj.weight"],
                          G[pl + "decoder.cross_attn.out_proj.bias"], gx=g_att)
        # attention core + in-projection (packed [3H, H] weight / [3H] bias)
        gWin, gbin = G[pl + "decoder.cross_attn.in_proj_weight"], G[pl + "decoder.cross_attn.in_proj_bias"]
        g_proj = self._buf("g_proj", (M, H), torch.bfloat16)
        if m._absorbed:  # the forward ran the absorbed form: k / v never existed, their weight gradients come from q', ctx
            g_q = self._xattn_absorbed(g_att, q, B, To, H, nh, dh, gWin, gbin)
            self.lin_bwd_bf16("q", proj, ca.in_proj_weight[:H], g_q, gWin[:H], gbin[:H], gx=g_proj)
        else:
            g_q, g_k, g_v, L = self._xattn_core(g_att, B, To, H, nh, dh)
            self.lin_bwd_bf16("q", proj, ca.in_proj_weight[:H], g_q, gWin[:H], gbin[:H], gx=g_proj)
            fh_b = self._fh_b
            fhT = self.lin_bwd_bf16("k", fh_b[: B * L], ca.in_proj_weight[H:2 * H], g_k[: B * L], gWin[H:2 * H],
                                    gbin[H:2 * H], pin=0)
            self.lin_bwd_bf16("v", fh_b[: B * L], ca.in_proj_weight[2 * H:], g_v[: B * L], gWin[2 * H:], gbin[2 * H:],
                              xT=fhT, pin=0)
        # proj = dec_t W_dp^T + b_dp
        g_dt2 = self._buf("g_dt2", (M, C))
        self.lin_bwd_bf16("dp", dec_tb, dec.dec_proj.weight, g_proj, G[pl + "decoder.dec_proj.weight"],
                          G[pl + "decoder.dec_proj.bias"], gx=g_dt2)
        # total gradient of dec_t, accumulated into g_dt2 (NOT into g_dec_t: the dec_unproj bias-gradient leaf may still be
        # reading g_dec_t on a side stream -- a buffer handed to a leaf is never written again in the same backward)
        ops.add_inplace(g_dt2, g_dec_t)
        # dec_t [B,To,C] -> d1 [B,C,To]
        g_d1 = self._buf("g_d1", (B, C * To))
        ops.transpose_ct(g_dt2, g_d1, None, B, To, C)
        # post MLP: d1 = relu(d0 W0^T + b0) W3^T + b3
        d0 = fwd("dec0", (B, C * To))
        if dec.use_post_mlp:
            hid = fwd("hid", (B, dec.post_mlp[0].weight.shape[0]))
            g_hid = self._buf("g_hid", tuple(hid.shape))
            self.lin_bwd_f32(hid, dec.post_mlp[3].weight, g_d1, G[pl + "decoder.post_mlp.3.weight"],
                             G[pl + "decoder.post_mlp.3.bias"], gx=g_hid)
            ops.dropout_(g_hid, getattr(m, "drop_post", None))  # hid = drop(relu(.)): mask, then the ReLU gate
            ops.relu_bwd(g_hid, hid)
            g_d0 = self._buf("g_d0", (B, C * To))
            self.lin_bwd_f32(d0, dec.post_mlp[0].weight, g_hid, G[pl + "decoder.post_mlp.0.weight"],
                             G[pl + "decoder.post_mlp.0.bias"], gx=g_d0)
        else:
            g_d0 = g_d1
        # d0 = NLinear_dec(e) + lane_fc(poly_emb)
        poly_emb = self._poly_emb
        g_poly = self._buf("g_poly", tuple(poly_emb.shape))
        self.lin_bwd_f32(poly_emb, dec.lane_fc.weight, g_d0, G[pl + "decoder.lane_fc.weight"],
                         G[pl + "decoder.lane_fc.bias"], gx=g_poly)
        if on_poly_grad is not None:
            on_poly_grad(g_poly)  # the lane-polygon encoder's backward can start here, beside the rest of this one
        e = sab._ws.get("sab.out", (B * T, C), torch.float32, dev)
        P = m._prepared()
        g_dw = self._buf("g_dw", (C, To, T))
        g_db = self._buf("g_db", (C, To))
        g_e = self._buf("g_e", (B * T, C))
        ops.nlinear_bwd(e, P.dec_w, g_d0, (C * To, To, 1), g_dw, g_db, g_e, B, C, T, To)
        self._scatter_channels(g_dw, g_db, pl + "decoder.decoder_linears.", C)
        # SelfAttentionBlock
        g_tok = self._sab(g_e, B, T, C)
        # front: tok = NLinear_enc(conv(x)) + pos
        xp = ws.get("lt.xp", (B * T, C), torch.float32, dev)
        g_ew = self._buf("g_ew", (C, T, T))
        g_eb = self._buf("g_eb", (C, T))
        g_xp = self._buf("g_xp", (B * T, C))
        ops.nlinear_bwd(xp, P.enc_w, g_tok, (T * C, 1, C), g_ew, g_eb, g_xp, B, C, T, T)
        self._scatter_channels(g_ew, g_eb, pl + "nlinear_encoder.encoder_linears.", C)
        gpos = G[pl + "pos_encoding"]  # (1, C, seq_len): same reduction as the encoder bias
        with self._leaf():
            gpos.view(C, -1)[:, :T].copy_(g_eb)
        ops.conv1x1_bwd(g_xp, x_in, G[pl + "token_proj.weight"].view(C, F), G[pl + "token_proj.bias"], B, C, T, F)
        return g_poly

    def _ltsf_stage_specs(self):
        """(p, seed, first_site) of the forward's dropout sites when tcavt_ltsf_backward covers this backward: the default
        form (fp16 storage, absorbed cross-attention, C++ stage forward) with the sites numbered as the stage numbers them
        (self-attention block first_site .. + 3, post-MLP + 4); None -> the Python composition below."""
        m = self.m.ltsf
        if not (m._absorbed and m._stage_ok(self.book.grads.device)):
            return None
        sp = getattr(m.attn_block, "drop_specs", None) or [None] * 4
        post = getattr(m, "drop_post", None) if m.decoder.use_post_mlp else None
        xs = getattr(m, "drop_xattn", None)
        if all(s is None for s in sp) and post is None and xs is None:
            return (0.0, 0, 0)
        if any(s is None for s in sp) or xs is None or (m.decoder.use_post_mlp and post is None):
            return None
        p, seed, s0 = sp[0]
        sites = [s[2] for s in sp] + ([post[2]] if post is not None else [])
        same = all(s[0] == p and s[1] == seed for s in sp + ([post] if post is not None else []))
        if not same or sites != list(range(s0, s0 + len(sites))):
            return None
        return (p, seed, s0)

    def _ltsf_stage(self, g_out, x_in, on_poly_grad):
        """Backward.ltsf as tcavt_ltsf_backward: phase 1 (head .. g_poly), the lane-polygon hand-off, phase 2 (the rest);
        the same kernels on the same named buffers as the composition in ltsf()/_sab()/_xattn_absorbed(), on one stream."""
        from . import capi

        m, G, ws = self.m.ltsf, self.book.g, self.m.ltsf._ws
        dec, sab, pl = m.decoder, m.attn_block, self.pl
        B, F, T = x_in.shape
        C, To, dev = m.d_model, m.out_len, g_out.device
        Mo, Mt, H, nh = B * To, B * T, dec.cross_dim, dec.cross_nhead
        dh, L = H // nh, self._L
        Lp, Mp = _rup(L, XATTN_PAD), _rup(Mo, 64)
        f32, bf, st = torch.float32, torch.bfloat16, m.storage
        p, seed, s0 = self._ltsf_stage_specs()
        P = m._prepared()
        fl, ca = dec.fusion_layer, dec.cross_attn
        post = dec.post_mlp[0].weight.shape[0] if dec.use_post_mlp else 0
        keep = []

        def put(obj, **kw):
            for k_, t_ in kw.items():
                if t_ is not None:
                    setattr(obj, k_, t_.data_ptr())
                    keep.append(t_)

        lt = lambda n, shape, dt=f32: ws.get("lt." + n, shape, dt, dev)
        sa = lambda n, shape: sab._ws.get("sab." + n, shape, f32, dev)
        # ---- the forward's arguments (both phases): activations by their workspace names, weights as the forward used them
        f = capi.LtsfArgs()
        put(f, x=x_in, enc_w=P.enc_w, tok=lt("tok", (Mt, C)), xp_tok=lt("xp", (Mt, C)),
            sa_n1_w=sab.norm1.weight, sa_in_w=sab.mha.in_proj_weight, sa_out_w=sab.mha.out_proj.weight, sa_n2_w=sab.norm2.weight,
            sa_f0_w=sab.ffn[0].weight, sa_f3_w=sab.ffn[3].weight,
            sa_xn=sa("xn", (Mt, C)), sa_qkv=sa("qkv", (Mt, 3 * C)), sa_att=sa("att", (Mt, C)), sa_res1=sa("res1", (Mt, C)),
            sa_rn=sa("rn", (Mt, C)), sa_f=sa("f", (Mt, 4 * C)), e=sa("out", (Mt, C)),
            poly_emb=self._poly_emb.contiguous(), lane_w=dec.lane_fc.weight, dec_w=P.dec_w, d0=lt("dec0", (B, C * To)),
            dec_tb=lt("dectb", (Mo, C), st), proj=lt("proj", (Mo, H), st), cross=lt("cross", (Mo, H), st),
            fused=lt("fused", (Mo, C)), fl_n_w=fl[0].weight, fn=lt("fn", (Mo, C)), fl1_w=fl[1].weight, f1=lt("f1", (Mo, C)),
            fl3_w=fl[3].weight, f2=lt("f2", (Mo, C)), out_w=dec.out_proj.weight)
        if post:
            put(f, pm0_w=dec.post_mlp[0].weight, pm3_w=dec.post_mlp[3].weight, hid=lt("hid", (B, post)))
        put(f.xattn, q=lt("q", (Mo, H), st), fh=self._fh_b, scores=lt("S", (B * nh * To, Lp)),
            probs=lt("P", (B * nh * To, Lp), torch.float16), ctx=lt("ctx", (nh, Mo, H), st), att=lt("att", (Mo, H), st))
        fx = f.xattn
        fx.B, fx.To, fx.L, fx.Lp, fx.H, fx.nhead, fx.dtype16 = B, To, L, Lp, H, nh, capi.F16
        xsp = getattr(m, "drop_xattn", None)
        if xsp is not None:
            fx.dropout_p, fx.dropout_seed, fx.dropout_site = xsp[0], xsp[1] & 0xFFFFFFFFFFFFFFFF, xsp[2]
        f.B, f.C, f.T, f.To, f.F, f.H, f.nhead_sa = B, C, T, To, F, H, sab.nhead
        f.poly_dim, f.post_hidden = self._poly_emb.shape[1], post
        f.dropout_p, f.dropout_seed, f.first_site = p, seed & 0xFFFFFFFFFFFFFFFF, s0
        # ---- gradients, workspaces
        a = capi.LtsfBwdArgs()
        a.fwd = ctypes.pointer(f)
        a.xattn.fwd = ctypes.cast(ctypes.addressof(f) + capi.LtsfArgs.xattn.offset, ctypes.POINTER(capi.CrossAttnArgs))
        b = self._buf
        g_q = b("xa.g_q", (Mo, H), bf)
        put(a.xattn, g_att=b("g_att", (Mo, H), bf), w_in=ca.in_proj_weight.detach(), gw_in=G[pl + "decoder.cross_attn.in_proj_weight"],
            gb_in=G[pl + "decoder.cross_attn.in_proj_bias"], g_q=g_q, fh_tb=b("xa.fhTb", (H, B * Lp), bf),
            fh_b=b("xa.fhb", (B * Lp, H), bf), ga_t=b("xa.gaT2", (H, Mp), bf), g_ctx=b("xa.g_ctx", (nh, Mo, H), bf),
            w_t=b("xa.w_t", (H * dh,), bf), x_t=b("xa.x_t", (H, Mp), bf), d_p=b("xa.dP", (B * nh * To, Lp)),
            d_s=b("xa.dS", (B * nh * To, Lp), bf), g_qp=b("xa.g_qp", (nh, Mo, H), bf),
            p_undropped=b("xa.Pu", (B * nh * To, Lp), torch.float16) if xsp is not None else None)
        g = lambda n: G[pl + n]
        g_poly = b("g_poly", tuple(self._poly_emb.shape))
        put(a, g_out=g_out, w_un=dec.dec_unproj.weight, w_co=ca.out_proj.weight, w_dp=dec.dec_proj.weight,
            g_out_w=g("decoder.out_proj.weight"), g_out_b=g("decoder.out_proj.bias"),
            g_fl3_w=g("decoder.fusion_layer.3.weight"), g_fl3_b=g("decoder.fusion_layer.3.bias"),
            g_fl1_w=g("decoder.fusion_layer.1.weight"), g_fl1_b=g("decoder.fusion_layer.1.bias"),
            g_fl_n_w=g("decoder.fusion_layer.0.weight"), g_fl_n_b=g("decoder.fusion_layer.0.bias"),
            g_un_w=g("decoder.dec_unproj.weight"), g_un_b=g("decoder.dec_unproj.bias"),
            g_co_w=g("decoder.cross_attn.out_proj.weight"), g_co_b=g("decoder.cross_attn.out_proj.bias"),
            g_dp_w=g("decoder.dec_proj.weight"), g_dp_b=g("decoder.dec_proj.bias"),
            g_lane_w=g("decoder.lane_fc.weight"), g_lane_b=g("decoder.lane_fc.bias"),
            g_sa_n1_w=g("attn_block.norm1.weight"), g_sa_n1_b=g("attn_block.norm1.bias"),
            g_sa_in_w=g("attn_block.mha.in_proj_weight"), g_sa_in_b=g("attn_block.mha.in_proj_bias"),
            g_sa_out_w=g("attn_block.mha.out_proj.weight"), g_sa_out_b=g("attn_block.mha.out_proj.bias"),
            g_sa_n2_w=g("attn_block.norm2.weight"), g_sa_n2_b=g("attn_block.norm2.bias"),
            g_sa_f0_w=g("attn_block.ffn.0.weight"), g_sa_f0_b=g("attn_block.ffn.0.bias"),
            g_sa_f3_w=g("attn_block.ffn.3.weight"), g_sa_f3_b=g("attn_block.ffn.3.bias"),
            g_pos=g("pos_encoding"), g_conv_w=g("token_proj.weight"), g_conv_b=g("token_proj.bias"), g_poly=g_poly,
            g_f2=b("g_f2", (Mo, C)), g_f1=b("g_f1", (Mo, C)), g_fn=b("g_fn", (Mo, C)), g_dec_t=b("g_dec_t", (Mo, C)),
            g_dt2=b("g_dt2", (Mo, C)), g_cross=b("g_cross", (Mo, H), bf), g_proj=b("g_proj", (Mo, H), bf),
            g_d1=b("g_d1", (B, C * To)), g_dw=b("g_dw", (C, To, T)), g_db=b("g_db", (C, To)), g_e=b("g_e", (Mt, C)),
            g_ff=b("sab.g_ff", (Mt, 4 * C)), g_rn=b("sab.g_rn", (Mt, C)), g_res1=b("sab.g_res1", (Mt, C)),
            g_att_sa=b("sab.g_att", (Mt, C)), g_qkv=b("sab.g_qkv", (Mt, 3 * C)), g_xn=b("sab.g_xn", (Mt, C)),
            g_tok=b("sab.g_tok", (Mt, C)), g_ew=b("g_ew", (C, T, T)), g_eb=b("g_eb", (C, T)), g_xp=b("g_xp", (Mt, C)),
            s_gyb=b("st.gyb", (Mo, H), bf), s_gyt=b("st.gyT", (H, Mp), bf), s_xt=b("st.xT", (H, Mp), bf),
            s_wt=b("st.WT", (H, H), bf))
        if post:
            put(a, g_pm3_w=g("decoder.post_mlp.3.weight"), g_pm3_b=g("decoder.post_mlp.3.bias"),
                g_pm0_w=g("decoder.post_mlp.0.weight"), g_pm0_b=g("decoder.post_mlp.0.bias"),
                g_hid=b("g_hid", (B, post)), g_d0=b("g_d0", (B, C * To)))
        if p > 0.0:
            put(a, g_e_d=b("sab.g_e_d", (Mt, C)), g_res1_d=b("sab.g_res1_d", (Mt, C)))
        # per-channel nn.Linear gradients of the N-Linear blocks: contiguous in the flat book in channel order (weight, bias, ...)
        flat, off = self.book.grads, self.book.offsets
        for name, prefix in (("dec", pl + "decoder.decoder_linears."), ("enc", pl + "nlinear_encoder.encoder_linears.")):
            o0, ob = off[prefix + "0.weight"][0], off[prefix + "0.bias"][0]
            setattr(a, f"g_{name}_w", flat.data_ptr() + 4 * o0)
            setattr(a, f"g_{name}_b", flat.data_ptr() + 4 * ob)
            setattr(a, f"{name}_stride", off[prefix + "1.weight"][0] - o0 if C > 1 else 0)
        a.pos_ld = G[pl + "pos_encoding"].shape[-1]
        if on_poly_grad is None:
            ops.ltsf_backward(a, 3)
        elif self.poly_after_chain and torch.cuda.is_available():
            # the caller's stream has a long chain to go after this backward (the decoder's, LoRA-trainable variant) and the host
            # is NOT ahead of the card here: enqueuing the lane-polygon encoder's ~45 launches first left the caller's stream idle
            # for their whole enqueue time (585 us per step in the timeline).  The chain goes first; the side stream still starts
            # from the point where d(poly_emb) exists.
            ops.ltsf_backward(a, 1)
            ev = torch.cuda.Event()
            ev.record()
            ops.ltsf_backward(a, 2)
            on_poly_grad(g_poly, ev)
        else:
            ops.ltsf_backward(a, 1)
            on_poly_grad(g_poly)  # the lane-polygon encoder's backward can start here, beside the rest of this one
            ops.ltsf_backward(a, 2)
        del keep
        return g_poly

    def _scatter_channels(self, gW, gb, prefix, C):
        """Stacked [C, S, T] / [C, S] gradients -> the C separate nn.Linear gradient views.  The flat
        book lays these views out contiguously in channel order (weight, bias, weight, bias, ...), so
        this is two strided device copies (memory plumbing, no arithmetic)."""
        G = self.book.g
        o0, _, shape = self.book.offsets[prefix + "0.weight"]
        S, T = shape
        stride = self.book.offsets[prefix + "1.weight"][0] - o0 if C > 1 else 0
        flat = self.book.grads
        with self._leaf():
            wv = torch.as_strided(flat, (C, S * T), (stride, 1), o0)
            wv.copy_(gW.view(C, S * T))
            ob = self.book.offsets[prefix + "0.bias"][0]
            bv = torch.as_strided(flat, (C, S), (stride, 1), ob)
            bv.copy_(gb.view(C, S))

    def _sab(self, g_e, B, T, C):
        m, G = self.m.ltsf.attn_block, self.book.g
        pl = self.pl + "attn_block."
        ws, dev = m._ws, g_e.device
        Mt = B * T
        f = lambda n, shape: ws.get("sab." + n, shape, torch.float32, dev)
        tok = self.m.ltsf._ws.get("lt.tok", (Mt, C), torch.float32, dev)
        xn, qkv, att, res1, rn, ff = f("xn", (Mt, C)), f("qkv", (Mt, 3 * C)), f("att", (Mt, C)), f("res1", (Mt, C)), \
            f("rn", (Mt, C)), f("f", (Mt, 4 * C))
        # e = ff W3^T + b3 + rn ; ff = relu(rn W0^T + b0)
        sp = getattr(m, "drop_specs", None) or [None] * 4  # [attention weights, after out_proj, after ReLU, after ffn.3]
        g_ff = self._buf("sab.g_ff", (Mt, 4 * C))
        self.lin_bwd_f32(ff, m.ffn[3].weight, self._dropped(g_e, sp[3], "sab.g_e_d"), G[pl + "ffn.3.weight"],
                         G[pl + "ffn.3.bias"], gx=g_ff)
        ops.dropout_(g_ff, sp[2])
        ops.relu_bwd(g_ff, ff)
        g_rn = self._buf("sab.g_rn", (Mt, C))
        self.lin_bwd_f32(rn, m.ffn[0].weight, g_ff, G[pl + "ffn.0.weight"], G[pl + "ffn.0.bias"], gx=g_rn)
        ops.add_inplace(g_rn, g_e)
        g_res1 = self._buf("sab.g_res1", (Mt, C))
        ops.layernorm_bwd(res1, m.norm2.weight, g_rn, g_res1, G[pl + "norm2.weight"], G[pl + "norm2.bias"])
        # res1 = att Wo^T + bo + xn
        g_att = self._buf("sab.g_att", (Mt, C))
        self.lin_bwd_f32(att, m.mha.out_proj.weight, self._dropped(g_res1, sp[1], "sab.g_res1_d"),
                         G[pl + "mha.out_proj.weight"], G[pl + "mha.out_proj.bias"], gx=g_att)
        g_qkv = self._buf("sab.g_qkv", (Mt, 3 * C))
        dh = C // m.nhead
        ops.mha_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], g_att, g_qkv[:, :C], g_qkv[:, C:2 * C], g_qkv[:, 2 * C:],
                    B, T, T, m.nhead, dh, 1.0 / math.sqrt(dh), ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C, ldg=3 * C,
                    dropout=sp[0])
        g_xn = self._buf("sab.g_xn", (Mt, C))
        self.lin_bwd_f32(xn, m.mha.in_proj_weight, g_qkv, G[pl + "mha.in_proj_weight"], G[pl + "mha.in_proj_bias"],
                         gx=g_xn)
        ops.add_inplace(g_xn, g_res1)
        g_tok = self._buf("sab.g_tok", (Mt, C))
        ops.layernorm_bwd(tok, m.norm1.weight, g_xn, g_tok, G[pl + "norm1.weight"], G[pl + "norm1.bias"])
        return g_tok

    def _xattn_core(self, g_att, B, To, H, nh, dh):
        """Backward of softmax(q k^T / sqrt(dh)) v per (sample, head) on batched MFMA GEMMs."""
        ws, dev = self.m.ltsf._ws, g_att.device
        fh_b = self._fh_b
        L = self._L
        Lp = _rup(L, XATTN_PAD)
        Tp = _rup(To, 64)
        M = B * To
        P = self.m.ltsf._prepared()
        q = ws.get("lt.q", (M, H), self.m.ltsf.storage, dev)
        kx = ws.get("lt.k", (B * L + XATTN_PAD, H), self.m.ltsf.storage, dev)
        Pm = ws.get("lt.P", (B * nh * To, Lp), torch.float16, dev)
        scale = 1.0 / math.sqrt(dh)
        # v (non-transposed, bf16) is recomputed: the forward only kept v^T in fp16.  It depends on no gradient, so
        # run() already started it on a side stream; here the chain only waits for it.
        if self._v_ready is not None:
            torch.cuda.current_stream().wait_event(self._v_ready)
            self._v_ready = None
            v = self._buf("xa.v", (B * L + XATTN_PAD, H), torch.bfloat16)
        else:
            v = self._recompute_v(B, L, H)
        # dP = dO V^T   [B*nh*To, Lp] fp32 (columns >= L are never read)
        dP = self._buf("xa.dP", (B * nh * To, Lp))
        ops.gemm_batched(g_att, v, dP, M=To, N=Lp, K=dh, lda=H, ldw=H, ldc=Lp, batch=B * nh, inner=nh,
                         sA=(To * H, dh), sW=(L * H, dh), sC=(nh * To * Lp, To * Lp))
        # attention-weight dropout (train mode): the forward kept only the dropped probabilities; the softmax
        # backward needs the un-dropped ones (recomputed from the retained scores) and the masked dP
        xsp = getattr(self.m.ltsf, "drop_xattn", None)
        if xsp is not None:
            ops.dropout_(dP, xsp)
            Pu = self._buf("xa.Pu", (B * nh * To, Lp), torch.float16)
            ops.softmax_rows(ws.get("lt.S", (B * nh * To, Lp), torch.float32, dev), Pu, B * nh * To, L, Lp, Lp, Lp)
        else:
            Pu = Pm
        dS = self._buf("xa.dS", (B * nh * To, Lp), torch.bfloat16)
        ops.softmax_bwd_rows(Pu, dP, dS, scale, B * nh * To, L, Lp, Lp, Lp, Lp)
        # dQ_bh = dS_bh K_bh : contraction over keys -> needs K^T per sample  kT [H, B*Lp]
        kT = self._buf("xa.kT", (H, B * Lp), torch.bfloat16)
        ops.transpose16(kx, kT, L, H, Lp, ld_in=H, ld_out=B * Lp, batch=B, s_in=L * H, s_out=Lp)
        g_q = self._buf("xa.g_q", (M, H), torch.bfloat16)
        ops.gemm_batched(dS, kT, g_q, M=To, N=dh, K=Lp, lda=Lp, ldw=B * Lp, ldc=H, batch=B * nh, inner=nh,
                         sA=(nh * To * Lp, To * Lp), sW=(Lp, dh * B * Lp), sC=(To * H, dh))
        # dK_bh = dS_bh^T Q_bh ; dV_bh = P_bh^T dO_bh : contraction over the To queries (padded to Tp).
        # Only the k / v in-projection WEIGHT gradients consume these (the LLM's hidden states are a constant), so
        # the whole branch is leaf work.
        with self._leaf(pin=0):
            g_k, g_v = self._xattn_kv_grads(g_att, dS, q, B, To, H, nh, dh, L, Lp, Tp)
        return g_q, g_k, g_v, L

    def _xattn_absorbed(self, g_att, q, B, To, H, nh, dh, gWin, gbin):
        """Backward of the absorbed cross-attention (model.TransformerLTSF.forward), per head h with F = the sample's final
        hidden states (a constant: the MLLM is frozen), q' = q_h W_k[h], S = q' F^T / sqrt(dh), P = dropout(softmax(S)),
        ctx = P F, att_h = ctx W_v[h]^T + b_v:
            gW_v[h] = g_att_h^T ctx      gb_v = colsum(g_att)      g_ctx = g_att_h W_v[h]
            g_P = g_ctx F^T              g_S = softmax'(P, g_P)    g_q' = g_S F
            gW_k[h] = q_h^T g_q'         gb_k = 0 (softmax-invariant)                 g_q_h = g_q' W_k[h]^T
        Every contraction over the B*L hidden-state rows of the un-absorbed form (dW_k, dW_v, the V recompute: three
        chip-filling GEMMs + two transposes of the hidden states) becomes a contraction over the B*To query rows."""
        m = self.m.ltsf
        ws, dev = m._ws, g_att.device
        L = self._L
        Lp = _rup(L, XATTN_PAD)
        M = B * To
        Mp = _rup(M, 64)
        ca = m.decoder.cross_attn
        st = m.storage
        bf = torch.bfloat16
        scale = 1.0 / math.sqrt(dh)
        ctx = ws.get("lt.ctx", (nh, M, H), st, dev)
        Pm = ws.get("lt.P", (B * nh * To, Lp), torch.float16, dev)
        if os.environ.get("TCAVT_PY_TLAYERS", "0") != "1":  # one C call: tcavt_cross_attn_backward (csrc/tlayers.hip)
            from . import capi
            f = capi.CrossAttnArgs()
            fwd = dict(q=q, fh=self._fh_b, scores=ws.get("lt.S", (B * nh * To, Lp), torch.float32, dev), probs=Pm, ctx=ctx)
            for k_, t_ in fwd.items():
                setattr(f, k_, t_.data_ptr())
            f.B, f.To, f.L, f.Lp, f.H, f.nhead, f.dtype16 = B, To, L, Lp, H, nh, capi.F16
            xsp = getattr(m, "drop_xattn", None)
            if xsp is not None:
                f.dropout_p, f.dropout_seed, f.dropout_site = xsp[0], xsp[1] & 0xFFFFFFFFFFFFFFFF, xsp[2]
            a = capi.CrossAttnBwdArgs()
            a.fwd = ctypes.pointer(f)
            g_q = self._buf("xa.g_q", (M, H), bf)
            bufs = dict(g_att=g_att, w_in=ca.in_proj_weight.detach(), gw_in=gWin, gb_in=gbin, g_q=g_q,
                        fh_tb=self._buf("xa.fhTb", (H, B * Lp), bf), fh_b=self._buf("xa.fhb", (B * Lp, H), bf),
                        ga_t=self._buf("xa.gaT2", (H, Mp), bf), g_ctx=self._buf("xa.g_ctx", (nh, M, H), bf),
                        w_t=self._buf("xa.w_t", (H * dh,), bf), x_t=self._buf("xa.x_t", (H, Mp), bf),
                        d_p=self._buf("xa.dP", (B * nh * To, Lp)), d_s=self._buf("xa.dS", (B * nh * To, Lp), bf),
                        g_qp=self._buf("xa.g_qp", (nh, M, H), bf))
            if xsp is not None:
                bufs["p_undropped"] = self._buf("xa.Pu", (B * nh * To, Lp), torch.float16)
            for k_, t_ in bufs.items():
                setattr(a, k_, t_.data_ptr())
            ops.cross_attn_backward(a)
            del fwd, bufs
            return g_q
        # bf16 copies of the hidden states for the gradient-side contractions: per-sample transposed [H][B*Lp] (converted
        # while transposing, keys padded to Lp with zeros) and, transposed back, row-major [B*Lp][H]
        fhTb = self._buf("xa.fhTb", (H, B * Lp), bf)
        ops.transpose16(self._fh_b, fhTb, L, H, Lp, ld_in=H, ld_out=B * Lp, batch=B, s_in=L * H, s_out=Lp)
        fhb = self._buf("xa.fhb", (B * Lp, H), bf)
        ops.transpose16(fhTb, fhb, H, B * Lp, H)
        # ---- value side
        with self._leaf():
            ops.colsum(g_att, gbin[2 * H:], M, H, accumulate=True)
        gaT = self._buf("xa.gaT2", (H, Mp), bf)  # g_att^T: rows h*dh.. are head h's
        ops.transpose16(g_att, gaT, M, H, Mp)
        g_ctx = self._buf("xa.g_ctx", (nh, M, H), bf)
        for h in range(nh):
            Wv_h = ca.in_proj_weight[2 * H + h * dh: 2 * H + (h + 1) * dh]  # [dh, H] fp32 parameter rows
            WvT = self._buf(f"xa.WvT{h}", (H, dh), bf)
            ops.transpose_f32_bf16(Wv_h.detach(), WvT, dh, H, dh)
            ops.gemm_bf16(g_att[:, h * dh:(h + 1) * dh], WvT, out=g_ctx[h])
            with self._leaf():
                ctxT = self._buf(f"xa.ctxT{h}", (H, Mp), bf)
                ops.transpose16(ctx[h], ctxT, M, H, Mp)
                ops.gemm_bf16(gaT[h * dh:(h + 1) * dh], ctxT, out=gWin[2 * H + h * dh: 2 * H + (h + 1) * dh])
        # ---- scores
        dP = self._buf("xa.dP", (B * nh * To, Lp))
        ops.gemm_batched(g_ctx, fhb, dP, M=To, N=Lp, K=H, lda=H, ldw=H, ldc=Lp, batch=B * nh, inner=nh,
                         sA=(To * H, M * H), sW=(Lp * H, 0), sC=(nh * To * Lp, To * Lp), tile=64)
        xsp = getattr(m, "drop_xattn", None)
        if xsp is not None:  # the softmax backward needs the un-dropped probabilities and the masked dP
            ops.dropout_(dP, xsp)
            Pu = self._buf("xa.Pu", (B * nh * To, Lp), torch.float16)
            ops.softmax_rows(ws.get("lt.S", (B * nh * To, Lp), torch.float32, dev), Pu, B * nh * To, L, Lp, Lp, Lp)
        else:
            Pu = Pm
        dS = self._buf("xa.dS", (B * nh * To, Lp), bf)
        ops.softmax_bwd_rows(Pu, dP, dS, scale, B * nh * To, L, Lp, Lp, Lp, Lp)
        # ---- query side
        g_qp = self._buf("xa.g_qp", (nh, M, H), bf)
        ops.gemm_batched(dS, fhTb, g_qp, M=To, N=H, K=Lp, lda=Lp, ldw=B * Lp, ldc=H, batch=B * nh, inner=nh,
                         sA=(nh * To * Lp, To * Lp), sW=(Lp, 0), sC=(To * H, M * H), tile=64)
        g_q = self._buf("xa.g_q", (M, H), bf)
        for h in range(nh):
            Wk_h = ca.in_proj_weight[H + h * dh: H + (h + 1) * dh]
            Wkb = self._buf(f"xa.Wkb{h}", (dh, H), bf)
            ops.cast_bf16(Wk_h.detach(), out=Wkb)
            ops.gemm_bf16(g_qp[h], Wkb, out=g_q[:, h * dh:(h + 1) * dh])
            with self._leaf():
                qT = self._buf(f"xa.qT{h}", (dh, Mp), bf)
                ops.transpose16(q[:, h * dh:(h + 1) * dh], qT, M, dh, Mp, ld_in=H)
                gqpT = self._buf(f"xa.gqpT{h}", (H, Mp), bf)
                ops.transpose16(g_qp[h], gqpT, M, H, Mp)
                ops.gemm_bf16(qT, gqpT, out=gWin[H + h * dh: H + (h + 1) * dh])
        return g_q

    def _recompute_v(self, B, L, H):
        P = self.m.ltsf._prepared()
        v = self._buf("xa.v", (B * L + XATTN_PAD, H), torch.bfloat16, zero=True)
        ops.gemm_bf16(self._fh_b[: B * L], P.w_v, out=v, bias=P.b_v)
        return v

    def _xattn_kv_grads(self, g_att, dS, q, B, To, H, nh, dh, L, Lp, Tp):
        ws, dev = self.m.ltsf._ws, g_att.device
        dST = self._buf("xa.dST", (B * nh * Lp, Tp), torch.bfloat16)
        ops.transpose16(dS, dST, To, Lp, Tp, ld_in=Lp, ld_out=Tp, batch=B * nh, s_in=To * Lp, s_out=Lp * Tp)
        Pb = self._buf("xa.Pb", (B * nh * To, Lp), torch.bfloat16)
        ops.softmax_rows(ws.get("lt.S", (B * nh * To, Lp), torch.float32, dev), Pb, B * nh * To, L, Lp, Lp, Lp,
                         dropout=getattr(self.m.ltsf, "drop_xattn", None))  # dV uses the probabilities the forward used
        PT = self._buf("xa.PT", (B * nh * Lp, Tp), torch.bfloat16)
        ops.transpose16(Pb, PT, To, Lp, Tp, ld_in=Lp, ld_out=Tp, batch=B * nh, s_in=To * Lp, s_out=Lp * Tp)
        qT = self._buf("xa.qT", (H, B * Tp), torch.bfloat16)
        ops.transpose16(q, qT, To, H, Tp, ld_in=H, ld_out=B * Tp, batch=B, s_in=To * H, s_out=Tp)
        gaT = self._buf("xa.gaT", (H, B * Tp), torch.bfloat16)
        ops.transpose16(g_att, gaT, To, H, Tp, ld_in=H, ld_out=B * Tp, batch=B, s_in=To * H, s_out=Tp)
        g_k = self._buf("xa.g_k", (B * L + XATTN_PAD, H), torch.bfloat16, zero=True)
        g_v = self._buf("xa.g_v", (B * L + XATTN_PAD, H), torch.bfloat16, zero=True)
        ops.gemm_batched(dST, qT, g_k, M=L, N=dh, K=Tp, lda=Tp, ldw=B * Tp, ldc=H, batch=B * nh, inner=nh,
                         sA=(nh * Lp * Tp, Lp * Tp), sW=(Tp, dh * B * Tp), sC=(L * H, dh))
        ops.gemm_batched(PT, gaT, g_v, M=L, N=dh, K=Tp, lda=Tp, ldw=B * Tp, ldc=H, batch=B * nh, inner=nh,
                         sA=(nh * Lp * Tp, Lp * Tp), sW=(Tp, dh * B * Tp), sC=(L * H, dh))
        return g_k, g_v

    # ---- LanePolygonEncoder -------------------------------------------------------------------
    def polygon(self, g_emb):
        enc, G, pp = self.m.lane_polygon_encoder, self.book.g, self.pp
        sv = enc.saved
        B, P, D = sv.B, sv.P, enc.d_model
        M = B * P
        nh = enc.nhead
        dh = D // nh
        g_x = self._buf("po.g_x", (M, D))
        ops.masked_mean_bwd(g_emb, sv.lens, g_x, B, P, D)
        if self._polygon_stage(g_x, sv, B, P, D, nh):  # the layer loop as one C call (tcavt_tlayer_stack_backward)
            gpos = G[pp + "pos_embedding"]
            ops.poly_embed_bwd(self._buf("po.g_x0", (M, D)), sv.polygon, G[pp + "input_proj.weight"], G[pp + "input_proj.bias"],
                               gpos.view(-1, D)[:P], B, P, D)
            return
        for i in reversed(range(len(sv.layers))):
            s, lyr = sv.layers[i], enc.encoder.layers[i]
            pre = f"{pp}encoder.layers.{i}."
            g_y2 = self._buf(f"po.g_y2{i}", (M, D))
            ops.layernorm_bwd(s["y2"], lyr.norm2.weight, g_x, g_y2, G[pre + "norm2.weight"], G[pre + "norm2.bias"])
            ff = lyr.linear1.weight.shape[0]
            g_f = self._buf(f"po.g_f{i}", (M, ff))
            sp = s.get("drop") or [None] * 4  # [attention weights, after out_proj, after ReLU, after linear2]
            self.lin_bwd_f32(s["f"], lyr.linear2.weight, self._dropped(g_y2, sp[3], f"po.g_y2d{i}"),
                             G[pre + "linear2.weight"], G[pre + "linear2.bias"], gx=g_f)
            ops.dropout_(g_f, sp[2])
            ops.relu_bwd(g_f, s["f"])
            g_x1 = self._buf(f"po.g_x1{i}", (M, D))
            self.lin_bwd_f32(s["x1"], lyr.linear1.weight, g_f, G[pre + "linear1.weight"], G[pre + "linear1.bias"], gx=g_x1)
            ops.add_inplace(g_x1, g_y2)
            g_y = self._buf(f"po.g_y{i}", (M, D))
            ops.layernorm_bwd(s["y"], lyr.norm1.weight, g_x1, g_y, G[pre + "norm1.weight"], G[pre + "norm1.bias"])
            g_att = self._buf(f"po.g_att{i}", (M, D))
            sa = lyr.self_attn
            self.lin_bwd_f32(s["att"], sa.out_proj.weight, self._dropped(g_y, sp[1], f"po.g_yd{i}"),
                             G[pre + "self_attn.out_proj.weight"], G[pre + "self_attn.out_proj.bias"], gx=g_att)
            qkv = s["qkv"]
            g_qkv = self._buf(f"po.g_qkv{i}", (M, 3 * D))
            ops.mha_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], g_att, g_qkv[:, :D], g_qkv[:, D:2 * D],
                        g_qkv[:, 2 * D:], B, P, P, nh, dh, 1.0 / math.sqrt(dh), key_len=sv.lens, ldq=3 * D, ldk=3 * D,
                        ldv=3 * D, ldo=D, ldg=3 * D, dropout=sp[0])
            g_xin = self._buf(f"po.g_xin{i}", (M, D))
            self.lin_bwd_f32(s["x"], sa.in_proj_weight, g_qkv, G[pre + "self_attn.in_proj_weight"],
                             G[pre + "self_attn.in_proj_bias"], gx=g_xin)
            ops.add_inplace(g_xin, g_y)
            g_x = g_xin
        gpos = G[pp + "pos_embedding"]
        ops.poly_embed_bwd(g_x, sv.polygon, G[pp + "input_proj.weight"], G[pp + "input_proj.bias"],
                           gpos.view(-1, D)[:P], B, P, D)

    def _polygon_stage(self, g_out, sv, B, P, D, nh):
        """Backward.polygon's layer loop as tcavt_tlayer_stack_backward (fp32 encoder layers; csrc/tlayers.hip): the gradient
        of the stack input lands in the buffer "po.g_x0".  False -> the caller runs the Python composition (TCAVT_PY_TLAYERS=1,
        or dropout sites that are not the stage's numbering)."""
        if os.environ.get("TCAVT_PY_TLAYERS", "0") == "1":
            return False
        from . import capi

        enc, G, pp = self.m.lane_polygon_encoder, self.book.g, self.pp
        n, M = len(sv.layers), B * P
        specs = [s.get("drop") or [None] * 4 for s in sv.layers]
        flat = [x for sp in specs for x in sp[:4]]
        if all(x is None for x in flat):
            p, seed, s0 = 0.0, 0, 0
        elif any(x is None for x in flat):
            return False
        else:
            p, seed, s0 = flat[0]
            if any(x[0] != p or x[1] != seed for x in flat) or [x[2] for x in flat] != list(range(s0, s0 + 4 * n)):
                return False
        keep = []

        def put(obj, **kw):
            for k_, t_ in kw.items():
                setattr(obj, k_, t_.data_ptr())
                keep.append(t_)

        arr, grads = (capi.TLayer * n)(), (capi.TLayerGrads * n)()
        ff = enc.encoder.layers[0].linear1.weight.shape[0]
        for i, (s, lyr) in enumerate(zip(sv.layers, enc.encoder.layers)):
            sa, pre = lyr.self_attn, f"{pp}encoder.layers.{i}."
            put(arr[i], w_in=sa.in_proj_weight.detach(), w_out=sa.out_proj.weight.detach(), w1=lyr.linear1.weight.detach(),
                w2=lyr.linear2.weight.detach(), n1_w=lyr.norm1.weight.detach(), n2_w=lyr.norm2.weight.detach(), qkv=s["qkv"],
                att=s["att"], y=s["y"], x1=s["x1"], ffh=s["f"], y2=s["y2"])
            if i > 0:
                put(arr[i - 1], out=s["x"])  # a layer's input is its predecessor's output
            put(grads[i], g_w_in=G[pre + "self_attn.in_proj_weight"], g_b_in=G[pre + "self_attn.in_proj_bias"],
                g_w_out=G[pre + "self_attn.out_proj.weight"], g_b_out=G[pre + "self_attn.out_proj.bias"],
                g_w1=G[pre + "linear1.weight"], g_b1=G[pre + "linear1.bias"], g_w2=G[pre + "linear2.weight"],
                g_b2=G[pre + "linear2.bias"], g_n1_w=G[pre + "norm1.weight"], g_n1_b=G[pre + "norm1.bias"],
                g_n2_w=G[pre + "norm2.weight"], g_n2_b=G[pre + "norm2.bias"])
        f = capi.TStackArgs()
        f.layers, f.n_layers = arr, n
        put(f, x=sv.layers[0]["x"], key_len=sv.lens)
        f.B, f.L, f.E, f.FF, f.nhead, f.dtype16 = B, P, D, ff, nh, 0
        f.dropout_p, f.dropout_seed, f.first_site = p, seed & 0xFFFFFFFFFFFFFFFF, s0
        a = capi.TStackBwdArgs()
        a.fwd, a.grads = ctypes.pointer(f), grads
        b = self._buf
        put(a, g_out=g_out, g_x=b("po.g_x0", (M, D)), g_tmp=b("po.g_tmp", (M, D)), g_y2=b("po.g_y2", (M, D)),
            g_x1=b("po.g_x1", (M, D)), g_y=b("po.g_y", (M, D)), g_att=b("po.g_att", (M, D)), g_f=b("po.g_f", (M, ff)),
            g_qkv=b("po.g_qkv", (M, 3 * D)))
        if p > 0.0:
            put(a, g_y2d=b("po.g_y2d", (M, D)), g_yd=b("po.g_yd", (M, D)))
        ops.tlayer_stack_backward(a)
        del keep
        return True

    # ---- entry ------------------------------------------------------------------------------
    def run(self, decoded, y, norm_stat, x_in, poly_emb, fh_b, L, after_ltsf=None):
        """Fills book.grads (must be zeroed by the caller) from the retained forward activations.
        `after_ltsf` (callable) is invoked once the LTSF gradients are complete (bucket hand-off)."""
        B, F, To = decoded.shape
        self._poly_emb, self._fh_b, self._L = poly_emb, fh_b, L
        self._ensure_streams()
        g_out = self._buf("g_out", (B, F, To))
        ops.mse_grad(decoded, y, norm_stat, g_out, B, To)
        if not self._multi():
            g_poly = self.ltsf(g_out, x_in)
            if after_ltsf is not None:
                after_ltsf()
            self.polygon(g_poly)
            return
        main = torch.cuda.current_stream()
        # the cross-attention backward needs V row-major (the forward kept only V^T): a chip-filling GEMM that depends on
        # no gradient -> started now on a side stream, off the gradient chain
        if not self.m.ltsf._absorbed:
            with _Fork(self._leaf_streams[0]):
                self._recompute_v(B, L, fh_b.shape[1])
                self._v_ready = torch.cuda.Event()
                self._v_ready.record()

        def start_polygon(g_poly, after=None):
            if after is not None:  # (ordered after an event recorded earlier on the caller's stream, not after "now")
                self._poly_stream.wait_event(after)
                with torch.cuda.stream(self._poly_stream):
                    self.polygon(g_poly)
                return
            with _Fork(self._poly_stream):
                self.polygon(g_poly)

        self.ltsf(g_out, x_in, on_poly_grad=start_polygon)
        # LTSF gradients are complete once the leaf stream has drained what was queued up to here; the bucket
        # hand-off (all-reduce launch) is issued from the leaf stream so that the chain does not wait for it
        if after_ltsf is not None:
            for ls in self._leaf_streams[1:]:
                self._leaf_streams[0].wait_stream(ls)
            with _Fork(self._leaf_streams[0]):
                after_ltsf()
        # the leaves join here (the LoRA-trainable variant's decoder backward starts from dL/dk, dL/dv, which a leaf stream produced)
        for ls in self._leaf_streams:
            main.wait_stream(ls)
        self._join_pending = True
        if not self.poly_after_chain:
            self.join()

    def join(self):
        """The caller's stream waits for the lane-polygon encoder's backward (its own side stream).  run() does this itself unless
        poly_after_chain is set: then the caller continues its own chain first (the decoder's backward) and joins before it
        reads those gradients (training.Trainer)."""
        if not getattr(self, "_join_pending", False) or self._poly_stream is None:
            return
        torch.cuda.current_stream().wait_stream(self._poly_stream)
        self._join_pending = False
