"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

A functional, fp32, CPU restatement (torch CPU tensors, plain ops) of the reference's
trajectory-prediction forward path ``MultiModalTrajectoryModel.forward``
(/root/reference/scripts/train.py:914-964) and everything under it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product path (``tcavt_amd``) must fail loudly without its HIP library.

Pinning: the reference has no tests or golden vectors of its own, so
this restatement is pinned by fixtures generated in the build container by importing the
reference's own modules (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``;
checked by ``tests/test_oracle_golden.py``).  Third-party arithmetic the reference calls
(HF ``LlamaForCausalLM`` as installed: transformers 5.15.0; ``torch.nn.MultiheadAttention``
/ ``nn.Transformer*Layer`` of torch 2.10; PEFT LoRA, absent here and restated from its
published definition ``y = W x + (alpha/r) B A dropout(x)``) is covered by the same
fixtures except LoRA, which is pinned by a 6-line wrapper inside the fixture generator.

Three numeric modes share one code path (``Rounder`` below; a dict toggles single rounding points):
  contract="fp32"  -- the reference's arithmetic (everything fp32); the <= 1e-3 parity bar of
                      BASELINE.json is asserted against THIS mode and the fixtures it is pinned to
  contract="fp16"  -- the HIP path's DEFAULT storage contract: same graph with values rounded to IEEE
                      half at exactly the points where the MI355X path stores 16 bits (GEMM operands,
                      the decoder's residual stream included; RMSNorm gains stay fp32); values beyond
                      +-65504 become inf, as on the device
  contract="bf16"  -- the round-1 contract (model.set_storage(torch.bfloat16)): bf16 at the same
                      points, attention probabilities fp16, fp32 residual stream
Weights are a flat dict {reference state-dict key: tensor}, see tcavt_amd/weights.py.
"""
import math

import torch
import torch.nn.functional as F

LLAMA = "mllm.llama_wrapper.llama_model.model."


def _ste(dtype):
    """Round the VALUE to `dtype` and let the gradient pass unchanged (straight-through): autograd through a plain
    .to(fp16).to(fp32) pair would push the gradient through fp16 as well, where the pixel-space loss overflows it."""
    def cast(t):
        q = t.detach().to(dtype).to(torch.float32)
        return t + (q - t.detach()) if t.requires_grad else q
    return cast


_CASTS = {
    "fp32": lambda t: t,
    "bf16": _ste(torch.bfloat16),
    # fp16 storage as the HIP path does it: IEEE half, round to nearest even; beyond +-65504 -> inf (outside the contract)
    "fp16": _ste(torch.float16),
}


class Rounder:
    """r(t, tag) rounds `t` to the storage type of the rounding point `tag`.  A contract is either one mode for every
    point ("fp32" | "bf16" | "fp16") or a dict {tag or scope or "scope.tag": mode, "default": mode}; the stage functions
    name their points (llama: w, xn, t, qkv, att, act; scopes: qf = Q-Former + q_proj, xa = LTSF cross-attention head,
    emb = embedding table, fh = final hidden states handed to the head) -- tests/tools/error_budget.py toggles them one by one."""

    def __init__(self, contract, scope=None):
        # Named contracts of the HIP path (model.set_storage): "fp16" = its default storage type, "bf16" = the round-1
        # contract kept for the LoRA-trainable variant.  In both, attention probabilities are carried in fp16 (P in
        # [0, 1]: csrc/attention.hip) and the RMSNorm gains stay fp32 (they are applied in fp32 inside the norm kernel).
        # "res" = the decoder's residual stream (the fused embeddings and the stream after each residual add): the fp16
        # contract carries it in fp16 (csrc/stack.hip, h == NULL: the residual epilogues add to the 16-bit stream in place),
        # the bf16 contract -- the LoRA-trainable variant, whose backward reads the streams -- in fp32.
        named = {"bf16": {"default": "bf16", "p": "fp16", "gamma": "fp32", "res": "fp32"},
                 "fp16": {"default": "fp16", "gamma": "fp32"},
                 "fp32": {"default": "fp32"}}
        self.modes = dict(contract) if isinstance(contract, dict) else dict(named[contract])
        self.scope = scope
        # "stream_scale" (dict contracts only): the 16-bit images of the decoder's residual stream hold stream_scale * x
        # (tcavt_llama_stack_args.stream_scale: a power of two; moves fp16's overflow limit, costs nothing else)
        self.stream_scale = float(self.modes.pop("stream_scale", 1.0))
        for m in self.modes.values():
            if m not in _CASTS:
                raise ValueError(m)

    def scoped(self, scope):
        return Rounder(dict(self.modes, stream_scale=self.stream_scale), scope)

    def mode(self, tag=None):
        for k in ((f"{self.scope}.{tag}" if self.scope and tag else None), tag, self.scope, "default"):
            if k is not None and k in self.modes:
                return self.modes[k]
        return "fp32"

    def __call__(self, t, tag=None):
        return _CASTS[self.mode(tag)](t)


def _rounder(contract):
    return contract if isinstance(contract, Rounder) else Rounder(contract)


def as_torch(weights):
    return {k: (torch.from_numpy(v) if not torch.is_tensor(v) else v) for k, v in weights.items()}


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
class DropTape:
    """Train-mode dropout for the oracle with the HIP path's masks: every call is the next dropout site (numbered in
    call order from `first_site`, as tcavt_amd.model.DropoutCtx numbers them) of one 64-bit seed; the keep mask of a
    site is oracle/philox.py's restatement of csrc/philox.hpp over the tensor's row-major element index.  `cols`
    (padded row length) reproduces sites whose kernel indexes a padded [rows, cols] matrix (cross-attention)."""

    def __init__(self, seed, p, first_site=1):
        self.seed, self.p, self.site = int(seed), float(p), int(first_site) - 1
        self.log = []  # (p, shape) of every site visited, in call order (compared with the reference's own sequence)

    def __call__(self, x, cols=None):
        from . import philox
        self.site += 1
        self.log.append((self.p, tuple(x.shape)))
        if cols is None:
            keep = philox.keep_mask(x.numel(), self.p, self.seed, self.site).reshape(tuple(x.shape))
        else:
            rows = x.numel() // x.shape[-1]
            keep = philox.keep_mask(rows * cols, self.p, self.seed, self.site).reshape(rows, cols)[:, : x.shape[-1]]
            keep = keep.reshape(tuple(x.shape))
        return x * torch.from_numpy(keep.astype("float32")) / (1.0 - self.p)


def _ident(x, **kw):
    return x


def linear(x, W, prefix, r, bias=True):
    """nn.Linear with both operands rounded per contract (r = identity in fp32 mode)."""
    y = r(x) @ r(W[prefix + ".weight"]).T
    if bias:
        y = y + W[prefix + ".bias"]
    return y


def mha_core(q, k, v, nhead, key_len=None, drop=_ident, drop_cols=None):
    """softmax(q k^T / sqrt(dh)) v per head; q [B,Lq,E], k/v [B,Lk,E]; key_len masks keys >= len
    (nn.MultiheadAttention key_padding_mask semantics, train.py:366-371)."""
    B, Lq, E = q.shape
    Lk = k.shape[1]
    dh = E // nhead
    qh = q.view(B, Lq, nhead, dh).transpose(1, 2)
    kh = k.view(B, Lk, nhead, dh).transpose(1, 2)
    vh = v.view(B, Lk, nhead, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    if key_len is not None:
        j = torch.arange(Lk)
        s = s.masked_fill((j[None, :] >= key_len[:, None])[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = drop(p, cols=drop_cols) if drop_cols is not None else drop(p)  # attention-weight dropout (train mode)
    return (p @ vh).transpose(1, 2).reshape(B, Lq, E)


def mha_module(xq, xkv, W, prefix, nhead, r, key_len=None, r_attn=None, drop=_ident):
    """nn.MultiheadAttention forward (eval): packed in_proj, heads, out_proj."""
    E = xq.shape[-1]
    Win, bin_ = W[prefix + ".in_proj_weight"], W[prefix + ".in_proj_bias"]
    q = r(xq) @ r(Win[:E]).T + bin_[:E]
    k = r(xkv) @ r(Win[E:2 * E]).T + bin_[E:2 * E]
    v = r(xkv) @ r(Win[2 * E:]).T + bin_[2 * E:]
    a = mha_core(q, k, v, nhead, key_len, drop=drop)
    a = (r_attn or r)(a)
    return a @ r(W[prefix + ".out_proj.weight"]).T + W[prefix + ".out_proj.bias"]


def layer_norm(x, W, prefix, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), W[prefix + ".weight"], W[prefix + ".bias"], eps)


def encoder_layer(x, W, prefix, nhead, r, key_len=None, drop=_ident):
    """nn.TransformerEncoderLayer defaults: post-LN, ReLU (train.py:358,402); `drop` = identity in eval, a DropTape
    in train mode (sites in the module's call order: attention weights, dropout1, dropout, dropout2)."""
    x = layer_norm(x + drop(mha_module(x, x, W, prefix + ".self_attn", nhead, r, key_len, drop=drop)), W, prefix + ".norm1")
    f = r(drop(torch.relu(linear(x, W, prefix + ".linear1", r))))
    x = layer_norm(x + drop(linear(f, W, prefix + ".linear2", r)), W, prefix + ".norm2")
    return x


def decoder_layer(x, mem, W, prefix, nhead, r, drop=_ident):
    """nn.TransformerDecoderLayer defaults: post-LN, ReLU, no masks (train.py:405-406,413)."""
    x = layer_norm(x + drop(mha_module(x, x, W, prefix + ".self_attn", nhead, r, drop=drop)), W, prefix + ".norm1")
    x = layer_norm(x + drop(mha_module(x, mem, W, prefix + ".multihead_attn", nhead, r, drop=drop)), W, prefix + ".norm2")
    f = r(drop(torch.relu(linear(x, W, prefix + ".linear1", r))))
    x = layer_norm(x + drop(linear(f, W, prefix + ".linear2", r)), W, prefix + ".norm3")
    return x


# --------------------------------------------------------------------------------------
# A2: LanePolygonEncoder.forward  (train.py:362-383) -- fp32 in both contracts
# --------------------------------------------------------------------------------------
def lane_polygon_encoder(W, cfg, polygon, lens, drop=_ident):
    ident = _rounder("fp32")
    P = polygon.shape[1]
    x = polygon @ W["lane_polygon_encoder.input_proj.weight"].T + W["lane_polygon_encoder.input_proj.bias"]
    x = x + W["lane_polygon_encoder.pos_embedding"][:, :P]
    lens_t = torch.as_tensor(lens, dtype=torch.long)
    key_len = torch.clamp(lens_t, max=P)
    # a sample with no valid point has every key masked in the reference (NaN rows, then
    # discarded by the zero-vector branch, train.py:378-380); keep it finite here
    key_len_safe = torch.where(key_len > 0, key_len, torch.full_like(key_len, P))
    for i in range(cfg.lane_polygon_layers):
        x = encoder_layer(x, W, f"lane_polygon_encoder.encoder.layers.{i}", cfg.lane_polygon_nhead, ident,
                          key_len_safe, drop=drop)
    j = torch.arange(P)
    valid = (j[None, :] < key_len[:, None]).float()
    s = (x * valid[..., None]).sum(1)
    n = key_len.float().clamp(min=1)[:, None]
    return torch.where(key_len[:, None] > 0, s / n, torch.zeros_like(s))


# --------------------------------------------------------------------------------------
# A3: BlipQFormer.forward (train.py:408-414) and q_proj (train.py:521)
# --------------------------------------------------------------------------------------
def qformer(W, cfg, vision, r, drop=_ident):
    """drop: a DropTape in train mode -- sites in module call order: per encoder layer (attention weights, dropout1,
    dropout, dropout2), per decoder layer (self-attention weights, dropout1, cross-attention weights, dropout2, dropout,
    dropout3), nn.Transformer*Layer defaults p = 0.1 (train.py:401-406)."""
    r = _rounder(r).scoped("qf")
    B = vision.shape[0]
    x = linear(vision, W, "mllm.qformer.vision_proj", r)
    for i in range(cfg.q_enc_layers):
        x = encoder_layer(x, W, f"mllm.qformer.encoder.layers.{i}", cfg.q_nhead, r, drop=drop)
    q = W["mllm.qformer.query_tokens"].unsqueeze(0).expand(B, -1, -1)
    for i in range(cfg.q_dec_layers):
        q = decoder_layer(q, x, W, f"mllm.qformer.decoder.layers.{i}", cfg.q_nhead, r, drop=drop)
    return q


# --------------------------------------------------------------------------------------
# F1-F7: Llama decoder stack as HF LlamaModel executes it (modeling_llama.py as installed)
# --------------------------------------------------------------------------------------
def llama3_inv_freq(ll):
    """modeling_rope_utils._compute_llama3_parameters (rope_type 'llama3')."""
    d = ll.head_dim
    inv = 1.0 / (ll.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
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
    """cos/sin [L, head_dim/2] fp32; positions are arange(L) regardless of padding
    (modeling_llama.py:386-389)."""
    inv = llama3_inv_freq(ll)
    ang = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
    return ang.cos(), ang.sin()


def rms_norm(x, w, eps):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def lora_scale(cfg):
    return cfg.lora_alpha / cfg.lora_r


def llama_decoder(W, cfg, embeds, attn_mask, r, collect=None, drop=_ident):
    """embeds [B,L,H] fp32, attn_mask [B,L] (1 = valid, right padded) -> post-final-norm hidden
    states = outputs.hidden_states[-1] (train.py:553).  drop: LoRA dropout on the adapter branch's input (PEFT:
    lora_B(lora_A(dropout(x))) with one lora_dropout module PER adapted Linear: two sites per layer, q_proj then v_proj,
    the order HF's attention forward calls them in, modeling_llama.py:254-256)."""
    ll = cfg.llama
    r = _rounder(r)
    B, L, H = embeds.shape
    nq, nkv, hd = ll.n_q_heads, ll.n_kv_heads, ll.head_dim
    cos, sin = rope_tables(ll, L)
    cos, sin = cos[None, :, None, :], sin[None, :, None, :]
    i = torch.arange(L)
    causal = i[None, :] <= i[:, None]
    allowed = causal[None] & (attn_mask[:, None, :] > 0)  # [B, Lq, Lk]
    ss = r.stream_scale
    res = lambda x: r(x * ss, "res") / ss  # (the rounded image is ss * x; ss a power of two: exact apart from the rounding itself)
    h = res(embeds)
    # RMSNorm (modeling_llama.py:62-67) in the algebraic form the HIP path computes it in (csrc/stack.hip): the gain is
    # folded into the following projection's weights (product in fp32, rounded once), the 16-bit operand is the rounded
    # residual stream itself, and rs = rsqrt(mean(h^2) + eps) scales the fp32 accumulator rows:
    #     (h rs gamma) W^T  ==  rs * (h (W gamma)^T)          -- identical in exact arithmetic, and in the fp32 contract
    def norm_parts(x):
        return r(x * ss, "xn") / ss, torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + ll.rms_eps)

    for layer in range(ll.layers):
        P = f"{LLAMA}layers.{layer}."
        g1, g2 = W[P + "input_layernorm.weight"], W[P + "post_attention_layernorm.weight"]
        hb, rs = norm_parts(h)
        q = hb @ r(W[P + "self_attn.q_proj.weight"] * g1, "w").T
        k = hb @ r(W[P + "self_attn.k_proj.weight"] * g1, "w").T
        v = hb @ r(W[P + "self_attn.v_proj.weight"] * g1, "w").T
        if cfg.use_lora:
            s = lora_scale(cfg)
            xq = hb if drop is _ident else r(drop(hb), "xn")
            xv = hb if drop is _ident else r(drop(hb), "xn")
            # (t is un-normalised -- the row scale multiplies the whole accumulator -- so it has the stream's range and is kept
            #  at the stream's scale: rounded as ss * t, like the main term ss * x W^T of the accumulator it joins)
            tq = r(ss * s * (xq @ r(W[P + "self_attn.q_proj.lora_A.weight"] * g1, "w").T), "t") / ss
            tv = r(ss * s * (xv @ r(W[P + "self_attn.v_proj.lora_A.weight"] * g1, "w").T), "t") / ss
            q = q + tq @ r(W[P + "self_attn.q_proj.lora_B.weight"], "w").T
            v = v + tv @ r(W[P + "self_attn.v_proj.lora_B.weight"], "w").T
        q = (rs * q).view(B, L, nq, hd)
        k = (rs * k).view(B, L, nkv, hd)
        v = (rs * v).view(B, L, nkv, hd)

        def rot(t):
            t1, t2 = t[..., : hd // 2], t[..., hd // 2:]
            return torch.cat([t1 * cos - t2 * sin, t2 * cos + t1 * sin], dim=-1)

        q, k, v = r(rot(q), "qkv"), r(rot(k), "qkv"), r(v, "qkv")
        qh = q.permute(0, 2, 1, 3)
        kh = k.permute(0, 2, 1, 3).repeat_interleave(nq // nkv, dim=1)
        vh = v.permute(0, 2, 1, 3).repeat_interleave(nq // nkv, dim=1)
        s = (qh @ kh.transpose(-1, -2)) / math.sqrt(hd)
        s = s.masked_fill(~allowed[:, None], float("-inf"))
        a = r((r(torch.softmax(s, dim=-1), "p") @ vh).permute(0, 2, 1, 3).reshape(B, L, nq * hd), "att")
        h = res(h + a @ r(W[P + "self_attn.o_proj.weight"], "w").T)
        hb2, rs2 = norm_parts(h)
        g = rs2 * (hb2 @ r(W[P + "mlp.gate_proj.weight"] * g2, "w").T)
        u = rs2 * (hb2 @ r(W[P + "mlp.up_proj.weight"] * g2, "w").T)
        act = r(F.silu(g) * u, "act")
        h = res(h + act @ r(W[P + "mlp.down_proj.weight"], "w").T)
        if collect is not None:
            collect.append(h)
    return rms_norm(h, W[LLAMA + "norm.weight"], ll.rms_eps)  # the final norm is a kernel of its own: fp32 in, fp32 out


# --------------------------------------------------------------------------------------
# A4/A5: LlamaMultiModal.forward, ids branch (train.py:516-554)
# --------------------------------------------------------------------------------------
def lm_head_and_loss(W, final_hidden, fused_labels):
    """The part of the reference's LLM call whose results it discards (train.py:547-554 keeps only
    hidden_states[-1]): lm_head (tied to embed_tokens) + shifted cross-entropy with ignore_index -100
    (HF LlamaForCausalLM.forward with labels).  Only bench.py's reference-faithful CPU timing runs it."""
    logits = final_hidden @ W["mllm.llama_wrapper.llama_model.lm_head.weight"].T
    shift_logits = logits[:, :-1].reshape(-1, logits.shape[-1])
    shift_labels = fused_labels[:, 1:].reshape(-1)
    return F.cross_entropy(shift_logits, shift_labels, ignore_index=-100)


def mllm_forward(W, cfg, vision, input_ids, attention_mask, r, collect=None, labels=None, drop_qf=_ident,
                 drop_lora=_ident):
    """labels given => also run the reference's discarded lm_head + CE (timing fidelity only)."""
    r = _rounder(r)
    if labels is not None:
        final = mllm_forward(W, cfg, vision, input_ids, attention_mask, r, collect, drop_qf=drop_qf, drop_lora=drop_lora)
        fused = torch.cat([torch.full((labels.shape[0], cfg.q_num_query_tokens), -100, dtype=labels.dtype), labels], 1)
        lm_head_and_loss(W, final, fused)
        return final
    img = linear(qformer(W, cfg, vision, r, drop=drop_qf), W, "mllm.q_proj", r.scoped("qf"))
    img = img + W["mllm.vision_modality_embedding"]
    txt = r(W[LLAMA + "embed_tokens.weight"][input_ids], "emb") + W["mllm.text_modality_embedding"]
    fused = torch.cat([img, txt], dim=1)
    mask = torch.cat([torch.ones(img.shape[0], img.shape[1], dtype=attention_mask.dtype), attention_mask], dim=1)
    if collect is not None:
        collect.append(fused)
    return llama_decoder(W, cfg, fused, mask, r, collect, drop=drop_lora)


# --------------------------------------------------------------------------------------
# A6-A9: TransformerLTSF.forward (train.py:836-842) with its sub-blocks
# --------------------------------------------------------------------------------------
def _stack(W, fmt, n, suffix):
    return torch.stack([W[fmt.format(c) + suffix] for c in range(n)], dim=0)


def ltsf_forward(W, cfg, x, poly_emb, final_hidden, r, drop=_ident, xattn_pad=64):
    r = _rounder(r).scoped("xa")
    ident = _rounder("fp32")
    B = x.shape[0]
    C, T, To = cfg.d_model, cfg.seq_len, cfg.out_len
    # token_proj: Conv1d(k=1)  (train.py:837)
    xp = torch.einsum("cf,bft->bct", W["ltsf.token_proj.weight"][:, :, 0], x) + W["ltsf.token_proj.bias"][None, :, None]
    # per-channel N-Linear encoder (train.py:701-716)
    last = xp[:, :, -1:]
    We = _stack(W, "ltsf.nlinear_encoder.encoder_linears.{}", C, ".weight")
    be = _stack(W, "ltsf.nlinear_encoder.encoder_linears.{}", C, ".bias")
    enc = torch.einsum("cst,bct->bcs", We, xp - last) + be[None] + last
    enc = enc + W["ltsf.pos_encoding"][:, :, :T]
    # SelfAttentionBlock (train.py:674-686): residuals start from the NORMED tensors
    tok = enc.permute(0, 2, 1)  # [B,T,C] (batch-first; attention is order-agnostic)
    xn = layer_norm(tok, W, "ltsf.attn_block.norm1")
    res1 = xn + drop(mha_module(xn, xn, W, "ltsf.attn_block.mha", cfg.ltsf_nhead, ident, drop=drop))
    rn = layer_norm(res1, W, "ltsf.attn_block.norm2")
    ffn = linear(drop(torch.relu(linear(rn, W, "ltsf.attn_block.ffn.0", ident))), W, "ltsf.attn_block.ffn.3", ident)
    e = (rn + drop(ffn)).permute(0, 2, 1)  # [B,C,T]
    # LTSF_NLinearDecoder (train.py:767-806)
    last = e[:, :, -1:]
    Wd = _stack(W, "ltsf.decoder.decoder_linears.{}", C, ".weight")
    bd = _stack(W, "ltsf.decoder.decoder_linears.{}", C, ".bias")
    dec = torch.einsum("cst,bct->bcs", Wd, e - last) + bd[None] + last
    dec = dec + linear(poly_emb, W, "ltsf.decoder.lane_fc", ident).view(B, C, To)
    if cfg.use_post_mlp:
        hid = drop(torch.relu(linear(dec.reshape(B, -1), W, "ltsf.decoder.post_mlp.0", ident)))
        dec = linear(hid, W, "ltsf.decoder.post_mlp.3", ident).view(B, C, To)
    dec_t = dec.permute(0, 2, 1)  # [B,To,C]
    proj = r(linear(dec_t, W, "ltsf.decoder.dec_proj", r))
    # cross attention: K = V = final_hidden, NO key padding mask (train.py:795-798)
    H = proj.shape[-1]
    Win, bin_ = W["ltsf.decoder.cross_attn.in_proj_weight"], W["ltsf.decoder.cross_attn.in_proj_bias"]
    fh = r(final_hidden, "fh")
    q = r(proj @ r(Win[:H]).T + bin_[:H])
    k = r(fh @ r(Win[H:2 * H]).T + bin_[H:2 * H])
    v = r(fh @ r(Win[2 * H:]).T + bin_[2 * H:])
    # (the HIP kernel draws this site's mask over score rows padded to a multiple of `xattn_pad` keys)
    Lk = fh.shape[1]
    drop_cols = (Lk + xattn_pad - 1) // xattn_pad * xattn_pad
    if r.mode() == "fp32":
        a = r(mha_core(q, k, v, cfg.cross_nhead, drop=drop, drop_cols=drop_cols))
    else:
        # The 16-bit contracts follow the HIP path's absorbed form (model.TransformerLTSF.forward): with To queries against
        # Lk keys per sample the K / V projections move to the query / output side,
        #     q_h (fh W_k[h]^T + b_k)^T = (q_h W_k[h]) fh^T + const per query (softmax-invariant),
        #     P_h (fh W_v[h]^T + b_v)   = (P_h fh) W_v[h]^T + b_v,
        # identical to the lines above in exact arithmetic; the rounding points are q' and ctx instead of k and v.
        nh = cfg.cross_nhead
        dh = H // nh
        Wk, Wv = r(Win[H:2 * H]).view(nh, dh, H), r(Win[2 * H:]).view(nh, dh, H)
        qh = q.view(B, To, nh, dh)
        s = torch.stack([r(qh[:, :, h] @ Wk[h]) @ fh.transpose(1, 2) for h in range(nh)], dim=1) / math.sqrt(dh)
        p = drop(torch.softmax(s, dim=-1), cols=drop_cols) if drop is not _ident else torch.softmax(s, dim=-1)
        a = r(torch.cat([r(p[:, h] @ fh) @ Wv[h].T + bin_[2 * H + h * dh: 2 * H + (h + 1) * dh] for h in range(nh)], dim=-1))
    cross = r(a @ r(W["ltsf.decoder.cross_attn.out_proj.weight"]).T + W["ltsf.decoder.cross_attn.out_proj.bias"])
    fused = dec_t + linear(cross, W, "ltsf.decoder.dec_unproj", r)
    f = layer_norm(fused, W, "ltsf.decoder.fusion_layer.0")
    f = linear(torch.relu(linear(f, W, "ltsf.decoder.fusion_layer.1", ident)), W, "ltsf.decoder.fusion_layer.3", ident)
    out = linear(f, W, "ltsf.decoder.out_proj", ident)  # [B,To,2]
    return out.permute(0, 2, 1)


# --------------------------------------------------------------------------------------
# A1: MultiModalTrajectoryModel.forward (train.py:914-964)
# --------------------------------------------------------------------------------------
def denorm(t, norm_stat):
    """t [B,2,T] normalised -> pixels, as train.py:946-957 (norm_stat rows: min_x,max_x,min_y,max_y)."""
    ns = torch.as_tensor(norm_stat, dtype=torch.float32)
    out = t.clone()
    out[:, 0, :] = t[:, 0, :] * (ns[:, 1] - ns[:, 0])[:, None] + ns[:, 0][:, None]
    out[:, 1, :] = t[:, 1, :] * (ns[:, 3] - ns[:, 2])[:, None] + ns[:, 2][:, None]
    return out


def dropout_tapes(cfg, seed, p_layers=0.1):
    """The four DropTapes of one train-mode forward, numbered as tcavt_amd.model.DropoutCtx.sub(block) numbers the HIP
    path's sites (block << 16 + 1, ...): 0 lane-polygon encoder, 1 Q-Former (nn.Transformer*Layer default p = 0.1),
    2 LoRA dropout (cfg.lora_dropout), 3 LTSF (cfg.ltsf_dropout)."""
    ps = (p_layers, p_layers, cfg.lora_dropout, cfg.ltsf_dropout)
    return {name: (DropTape(seed, p, first_site=(blk << 16) + 1) if p > 0.0 else _ident)
            for blk, (name, p) in enumerate(zip(("poly", "qf", "lora", "ltsf"), ps))}


def model_forward(W, cfg, x, vision, polygon, polygon_len, input_ids, attention_mask, y=None, norm_stat=None,
                  contract="fp32", extras=None, labels=None, dropout_seed=None):
    """Returns decoded [B,2,To] or (loss, decoded) like the reference.  `extras` (dict) receives
    intermediate tensors (poly_emb, final_hidden) for stage-wise parity checks.  `labels` switches
    on the reference's discarded lm_head + CE work (see lm_head_and_loss).  dropout_seed: a train-mode forward
    (ddp_model.train(), train.py:1152) with the HIP path's Philox masks of that seed; extras["tapes"] then holds the
    DropTapes (their .log = the dropout sites visited, in order)."""
    r = _rounder(contract)
    W = as_torch(W)
    tp = dropout_tapes(cfg, dropout_seed) if dropout_seed is not None else dict.fromkeys(("poly", "qf", "lora", "ltsf"), _ident)
    poly_emb = lane_polygon_encoder(W, cfg, polygon, polygon_len, drop=tp["poly"])
    final_hidden = mllm_forward(W, cfg, vision, input_ids, attention_mask, r, labels=labels, drop_qf=tp["qf"],
                                drop_lora=tp["lora"])
    decoded = ltsf_forward(W, cfg, x, poly_emb, final_hidden, r, drop=tp["ltsf"])
    decoded = decoded + x[:, :, -1:]
    if extras is not None:
        extras["poly_emb"] = poly_emb
        extras["final_hidden"] = final_hidden
        extras["tapes"] = tp
    if y is not None and norm_stat is not None:
        dp, dg = denorm(decoded, norm_stat), denorm(y, norm_stat)
        loss = F.mse_loss(dp[:, 0], dg[:, 0]) + F.mse_loss(dp[:, 1], dg[:, 1])
        return loss, decoded
    return decoded


# --------------------------------------------------------------------------------------
# A10: metrics (train.py:1302-1325; test.py:1342-1382; ablation_study_without_lora.py:1233-1238)
# --------------------------------------------------------------------------------------
def traj_metrics(pred, gt, norm_stat):
    """pred [B,K,2,To] or [B,2,To]; returns dict of batch SUMS of min-over-K ADE/FDE/RMSE and argmins."""
    if pred.dim() == 3:
        pred = pred[:, None]
    B, K = pred.shape[:2]
    dg = denorm(gt, norm_stat)
    dp = torch.stack([denorm(pred[:, k], norm_stat) for k in range(K)], dim=1)
    diff = dp - dg[:, None]
    err = torch.sqrt((diff ** 2).sum(dim=2))          # [B,K,To]
    ade, fde = err.mean(dim=2), err[:, :, -1]
    rmse = torch.sqrt((diff ** 2).mean(dim=(2, 3)))
    return {
        "ade_sum": ade.min(dim=1).values.sum().item(), "fde_sum": fde.min(dim=1).values.sum().item(),
        "rmse_sum": rmse.min(dim=1).values.sum().item(),
        "ade_argmin": ade.argmin(dim=1), "fde_argmin": fde.argmin(dim=1), "rmse_argmin": rmse.argmin(dim=1),
        "ade": ade, "fde": fde, "rmse": rmse,
    }
