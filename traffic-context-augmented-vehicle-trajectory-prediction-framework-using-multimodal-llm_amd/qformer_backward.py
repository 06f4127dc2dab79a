"""Backward of the MLLM's trainable front end for the full trainable set of modify_scripts/modify_train.py: that script
freezes only the non-LoRA Llama weights (:523-528), so the Q-Former (train.py:388-414), mllm.q_proj (:493,521) and the two
modality embeddings (:494-495,524-527) train next to the adapters.

Input: g_h0 = dL/d(inputs_embeds) [B*L, H] fp32, what llm_backward.LoraBackward leaves after walking through decoder layer 0.
    inputs_embeds[b] = [ q_proj(qformer(vision))[b] + vision_modality_embedding | embed_tokens(ids)[b] + text_modality_embedding ]
so the image rows carry the gradient of the image tokens, the modality embeddings receive column sums, embed_tokens is frozen.
The Q-Former is eight post-LN nn.Transformer{Encoder,Decoder}Layer blocks on 18 vision / 16 query tokens per sample: the
same stage arithmetic as the lane-polygon encoder's backward (backward.Backward.polygon) with 16-bit MFMA contractions
(Backward.lin_bwd_bf16) and the small-sequence attention backward (tcavt_mha_bwd); dropout masks are regenerated from the
forward's (p, seed, site) specs.
"""
import math

import torch

from . import ops


class QFormerBackward:
    def __init__(self, model, book, bw):
        self.m, self.book, self.bw = model, book, bw  # bw: backward.Backward (buffers, leaf streams, linear backward)

    # ---- pieces ---------------------------------------------------------------------------------
    def _ffn_block(self, tag, rec, lyr, pre, g_out, norm_name, y_key, xin_key, xinb_key, drop_i):
        """out = LN(y), y = xin + drop(linear2(drop(relu(linear1(xin16))))): returns g_xin (fp32)."""
        G = self.book.g
        bw = self.bw
        M, E = rec[xin_key].shape
        sp = rec.get("drop") or [None] * 8
        g_y = bw._buf(f"qf.{tag}.g_y", (M, E))
        norm = getattr(lyr, norm_name)
        ops.layernorm_bwd(rec[y_key], norm.weight, g_out, g_y, G[pre + norm_name + ".weight"], G[pre + norm_name + ".bias"])
        ff = lyr.linear1.weight.shape[0]
        g_f = bw._buf(f"qf.{tag}.g_f", (M, ff))
        bw.lin_bwd_bf16(f"qf.{tag}.l2", rec["f"], lyr.linear2.weight, bw._dropped(g_y, sp[drop_i + 1], f"qf.{tag}.g_yd"),
                        G[pre + "linear2.weight"], G[pre + "linear2.bias"], gx=g_f)
        ops.dropout_(g_f, sp[drop_i])
        ops.relu_bwd(g_f, rec["f"])
        g_x = bw._buf(f"qf.{tag}.g_x", (M, E))
        bw.lin_bwd_bf16(f"qf.{tag}.l1", rec[xinb_key], lyr.linear1.weight, g_f, G[pre + "linear1.weight"],
                        G[pre + "linear1.bias"], gx=g_x)
        ops.add_inplace(g_x, g_y)
        return g_x

    def _self_attn_block(self, tag, rec, lyr, pre, g_out, B, L, nh):
        """out = LN1(y), y = x + drop(out_proj(MHA(x16))): returns g_x (fp32)."""
        G, bw = self.book.g, self.bw
        M, E = rec["x"].shape
        sp = rec.get("drop") or [None] * 8
        g_y = bw._buf(f"qf.{tag}.g_ys", (M, E))
        ops.layernorm_bwd(rec["y"], lyr.norm1.weight, g_out, g_y, G[pre + "norm1.weight"], G[pre + "norm1.bias"])
        sa = lyr.self_attn
        g_att = bw._buf(f"qf.{tag}.g_att", (M, E))
        bw.lin_bwd_bf16(f"qf.{tag}.so", rec["att"], sa.out_proj.weight, bw._dropped(g_y, sp[1], f"qf.{tag}.g_ysd"),
                        G[pre + "self_attn.out_proj.weight"], G[pre + "self_attn.out_proj.bias"], gx=g_att)
        qkv = rec["qkv"]
        g_qkv = bw._buf(f"qf.{tag}.g_qkv", (M, 3 * E))
        dh = E // nh
        ops.mha_bwd(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], g_att, g_qkv[:, :E], g_qkv[:, E:2 * E], g_qkv[:, 2 * E:],
                    B, L, L, nh, dh, 1.0 / math.sqrt(dh), ldq=3 * E, ldk=3 * E, ldv=3 * E, ldo=E, ldg=3 * E, dropout=sp[0])
        g_x = bw._buf(f"qf.{tag}.g_xs", (M, E))
        bw.lin_bwd_bf16(f"qf.{tag}.si", rec["xb"], sa.in_proj_weight, g_qkv, G[pre + "self_attn.in_proj_weight"],
                        G[pre + "self_attn.in_proj_bias"], gx=g_x)
        ops.add_inplace(g_x, g_y)
        return g_x

    def _cross_attn_block(self, tag, rec, lyr, pre, g_out, memb, g_mem, B, Lq, Lk, nh):
        """out = LN2(y2), y2 = x1 + drop(out_proj(MHA(q = x1, k = v = mem))): returns g_x1; ADDS the memory's gradient to g_mem."""
        G, bw = self.book.g, self.bw
        M, E = rec["x1"].shape
        sp = rec["drop"] or [None] * 8
        g_y2 = bw._buf(f"qf.{tag}.g_y2", (M, E))
        ops.layernorm_bwd(rec["y2"], lyr.norm2.weight, g_out, g_y2, G[pre + "norm2.weight"], G[pre + "norm2.bias"])
        ca = lyr.multihead_attn
        g_att = bw._buf(f"qf.{tag}.g_catt", (M, E))
        bw.lin_bwd_bf16(f"qf.{tag}.co", rec["catt"], ca.out_proj.weight, bw._dropped(g_y2, sp[3], f"qf.{tag}.g_y2d"),
                        G[pre + "multihead_attn.out_proj.weight"], G[pre + "multihead_attn.out_proj.bias"], gx=g_att)
        q, kv = rec["cq"], rec["ckv"]
        dh = E // nh
        # one leading dimension for the three gradients (tcavt_mha_bwd): dQ | dK | dV side by side in a [rows, 3E] buffer
        rows = max(M, kv.shape[0])
        g3 = bw._buf(f"qf.{tag}.g_c3", (rows, 3 * E))
        ops.mha_bwd(q, kv[:, :E], kv[:, E:], g_att, g3[:, :E], g3[:, E:2 * E], g3[:, 2 * E:], B, Lq, Lk, nh, dh,
                    1.0 / math.sqrt(dh), ldq=E, ldk=2 * E, ldv=2 * E, ldo=E, ldg=3 * E, dropout=sp[2])
        Win, bin_ = ca.in_proj_weight, ca.in_proj_bias
        gWin, gbin = G[pre + "multihead_attn.in_proj_weight"], G[pre + "multihead_attn.in_proj_bias"]
        g_x1 = bw._buf(f"qf.{tag}.g_x1", (M, E))
        g_q = bw._buf(f"qf.{tag}.g_cq", (M, E))
        g_q.copy_(g3[:M, :E])
        bw.lin_bwd_bf16(f"qf.{tag}.cq", rec["x1b"], Win[:E], g_q, gWin[:E], gbin[:E], gx=g_x1)
        ops.add_inplace(g_x1, g_y2)
        Mk = kv.shape[0]
        g_kv = bw._buf(f"qf.{tag}.g_ckv", (Mk, 2 * E))
        g_kv.copy_(g3[:Mk, E:])
        g_m = bw._buf(f"qf.{tag}.g_m", (Mk, E))
        bw.lin_bwd_bf16(f"qf.{tag}.ckv", memb, Win[E:], g_kv, gWin[E:], gbin[E:], gx=g_m)
        ops.add_inplace(g_mem, g_m)
        return g_x1

    # ---- entry ----------------------------------------------------------------------------------
    def run(self, g_h0, B, L):
        """g_h0 fp32 [B*L, H]: gradient of the fused input embeddings."""
        m, G, bw = self.m.mllm, self.book.g, self.bw
        qf = m.qformer
        sv = qf.saved
        if sv is None:
            raise RuntimeError("QFormerBackward.run: no saved activations (qformer.save_for_backward before the forward)")
        H, Nq, E, nh, Tv = m.llama_hidden_size, qf.num_query_tokens, qf.hidden_size, qf.nhead, sv.Tv
        pre = "mllm."
        g3 = g_h0.view(B, L, H)
        # modality embeddings: column sums over their rows; image-token gradient: the first Nq rows of every sample
        g_img = bw._buf("qf.g_img", (B * Nq, H))
        g_img.view(B, Nq, H).copy_(g3[:, :Nq])
        g_txt = bw._buf("qf.g_txt", (B * (L - Nq), H))
        g_txt.view(B, L - Nq, H).copy_(g3[:, Nq:])
        ops.colsum(g_img, G[pre + "vision_modality_embedding"].view(-1), B * Nq, H, accumulate=True)
        ops.colsum(g_txt, G[pre + "text_modality_embedding"].view(-1), B * (L - Nq), H, accumulate=True)
        # q_proj: img = out16 W^T + b
        g_q = bw._buf("qf.g_out", (B * Nq, E))
        bw.lin_bwd_bf16("qf.qp", m._imgb, m.q_proj.weight, g_img, G[pre + "q_proj.weight"], G[pre + "q_proj.bias"], gx=g_q)
        # decoder layers, last first; the memory (encoder output) collects a gradient from every layer's cross-attention
        g_mem = bw._buf("qf.g_mem", (B * Tv, E), zero=True)
        g_mem.zero_()
        for i in reversed(range(len(sv.dec))):
            rec, lyr = sv.dec[i], qf.decoder.layers[i]
            p = f"{pre}qformer.decoder.layers.{i}."
            tag = f"D{i}"
            g_x2 = self._ffn_block(tag, rec, lyr, p, g_q, "norm3", "y3", "x2", "x2b", 4)
            g_x1 = self._cross_attn_block(tag, rec, lyr, p, g_x2, sv.memb, g_mem, B, Nq, Tv, nh)
            g_q = self._self_attn_block(tag, rec, lyr, p, g_x1, B, Nq, nh)
        # the learned queries are broadcast over the batch (train.py:412)
        gq = G[pre + "qformer.query_tokens"]
        ops.colsum(g_q.view(B, Nq * E), gq.view(-1), B, Nq * E, accumulate=True)
        # encoder layers
        g_x = g_mem
        for i in reversed(range(len(sv.enc))):
            rec, lyr = sv.enc[i], qf.encoder.layers[i]
            p = f"{pre}qformer.encoder.layers.{i}."
            tag = f"E{i}"
            g_x1 = self._ffn_block(tag, rec, lyr, p, g_x, "norm2", "y2", "x1", "x1b", 2)
            g_x = self._self_attn_block(tag, rec, lyr, p, g_x1, B, Tv, nh)
        # vision_proj: x0 = vision16 W^T + b
        bw.lin_bwd_bf16("qf.vp", sv.vb, qf.vision_proj.weight, g_x, G[pre + "qformer.vision_proj.weight"],
                        G[pre + "qformer.vision_proj.bias"])
