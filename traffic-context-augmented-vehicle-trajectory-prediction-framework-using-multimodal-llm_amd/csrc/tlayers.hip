// Stage-level composition of the post-LN nn.TransformerEncoderLayer / nn.TransformerDecoderLayer stacks of the path
// (SURVEY.md 8b "qformer_forward", "polygon_encoder_forward"): the Q-Former's four encoder + four decoder layers
// (scripts/train.py:388-414; 16-bit MFMA contractions) and the lane-polygon encoder's layers (scripts/train.py:352-383; fp32
// end to end) as ONE call each.  Every layer is the launch sequence tcavt_amd.model._TLayerRunner issues from Python --
//   self-attention:  qkv = x W_in^T + b -> tcavt_mha -> y = x + drop(att W_out^T + b) -> LayerNorm
//   cross-attention: q = x1 W_q^T + b, k|v = mem W_kv^T + b -> tcavt_mha -> y2 = x1 + drop(..) -> LayerNorm     (decoder layers)
//   feed-forward:    f = drop(relu(x W_1^T + b)) -> y3 = x + drop(f W_2^T + b) -> LayerNorm
// -- on the same kernels, with the same buffers (all caller-owned; per layer when a backward will read them) and the same
// dropout sites (numbered in call order from first_site), so the result is bit-identical to the Python composition.
#include "common.hpp"

using namespace tcavt;

#define TCAVT_TRY(call)            \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != TCAVT_OK) return rc_; \
  } while (0)

namespace {
struct Ctx {
  const tcavt_tstack_args* a;
  tcavt_stream_t st;
  uint32_t site;
  int M;

  int gemm(const void* A16, const float* A32, const void* W, const float* bias, void* C, int out_dtype, int Mr, int N, int K,
           bool relu, const float* residual, bool drop) {
    const float p = drop ? a->dropout_p : 0.f;
    const uint32_t s = drop && a->dropout_p > 0.f ? site++ : 0u;
    int flags = (bias ? TCAVT_EPI_BIAS : 0) | (relu ? TCAVT_EPI_RELU : 0) | (residual ? TCAVT_EPI_RESIDUAL : 0);
    if (a->dtype16) {
      tcavt_gemm_args g = {};
      g.A = A16; g.lda = K; g.W = W; g.ldw = K; g.C = C; g.ldc = N; g.M = Mr; g.N = N; g.K = K;
      g.out_dtype = out_dtype; g.in_dtype = a->dtype16; g.epilogue = flags; g.bias = bias; g.residual = residual; g.ldr = N;
      g.dropout_p = p; g.dropout_seed = a->dropout_seed; g.dropout_site = s;
      return tcavt_gemm_bf16(&g, st);
    }
    return tcavt_gemm_f32(A32, K, static_cast<const float*>(W), K, bias, residual, N, static_cast<float*>(C), N, Mr, N, K, flags, p,
                          a->dropout_seed, s, st);
  }
  int norm(const float* y, const float* w, const float* b, float* out, void* outb) {
    return tcavt_layernorm(y, nullptr, w, b, 1e-5f, out, a->dtype16 ? outb : nullptr, M, a->E, a->dtype16 ? a->dtype16 : TCAVT_BF16, st);
  }
};
}  // namespace

extern "C" int tcavt_tlayer_stack_forward(const tcavt_tstack_args* a, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(a && a->layers && a->n_layers > 0 && a->x && a->B > 0 && a->L > 0 && a->E > 0 && a->FF > 0 && a->nhead > 0 &&
                      a->E % a->nhead == 0,
                  "tlayer_stack_forward: bad args");
  TCAVT_CHECK_ARG(a->dtype16 == 0 || is16(a->dtype16), "tlayer_stack_forward: dtype16 must be 0 (fp32 layers), TCAVT_BF16 or TCAVT_F16");
  TCAVT_CHECK_ARG(!a->dtype16 || a->xb, "tlayer_stack_forward: 16-bit layers need the 16-bit copy of the input tokens");
  TCAVT_CHECK_ARG(a->dropout_p >= 0.f && a->dropout_p < 1.f, "tlayer_stack_forward: dropout_p must be in [0, 1)");
  const int E = a->E, FF = a->FF, nh = a->nhead, dh = E / nh, M = a->B * a->L;
  const int act = a->dtype16 ? a->dtype16 : TCAVT_F32;
  const float scale = (float)(1.0 / sqrt((double)dh));  // (as the Python caller forms it: double, rounded once)
  Ctx c{a, stream, a->first_site, M};
  auto mha_site = [&]() -> uint32_t { return a->dropout_p > 0.f ? c.site++ : 0u; };
  const float* x = a->x;
  const void* xb = a->xb;
  for (int li = 0; li < a->n_layers; ++li) {
    const tcavt_tlayer& l = a->layers[li];
    const bool dec = l.w_q != nullptr;
    TCAVT_CHECK_ARG(l.w_in && l.b_in && l.w_out && l.b_out && l.w1 && l.b1 && l.w2 && l.b2 && l.n1_w && l.n1_b && l.n2_w && l.n2_b &&
                        l.qkv && l.att && l.y && l.x1 && l.ffh && l.y2 && l.out && (!a->dtype16 || (l.x1b && l.outb)),
                    "tlayer_stack_forward: layer %d: null weight / buffer", li);
    if (dec)
      TCAVT_CHECK_ARG(a->mem && (!a->dtype16 || a->memb) && a->Lk > 0 && l.b_q && l.w_kv && l.b_kv && l.w_co && l.b_co && l.n3_w &&
                          l.n3_b && l.cq && l.ckv && l.catt && l.cy && l.x2 && (!a->dtype16 || l.x2b),
                      "tlayer_stack_forward: layer %d: decoder layer without memory / cross-attention buffers", li);
    // ---- self-attention block
    TCAVT_TRY(c.gemm(xb, x, l.w_in, l.b_in, l.qkv, TCAVT_F32, M, 3 * E, E, false, nullptr, false));
    {
      const uint32_t s = mha_site();
      TCAVT_TRY(tcavt_mha(l.qkv, 3 * E, l.qkv + E, 3 * E, l.qkv + 2 * E, 3 * E, l.att, E, a->key_len, a->B, a->L, a->L, nh, dh, scale,
                          TCAVT_F32, act, a->dropout_p, a->dropout_seed, s, stream));
    }
    TCAVT_TRY(c.gemm(l.att, static_cast<const float*>(l.att), l.w_out, l.b_out, l.y, TCAVT_F32, M, E, E, false, x, true));
    TCAVT_TRY(c.norm(l.y, l.n1_w, l.n1_b, l.x1, l.x1b));
    const float* xin = l.x1;
    const void* xinb = l.x1b;
    // ---- cross-attention block (decoder layers)
    if (dec) {
      const int Mk = a->B * a->Lk;
      TCAVT_TRY(c.gemm(l.x1b, l.x1, l.w_q, l.b_q, l.cq, TCAVT_F32, M, E, E, false, nullptr, false));
      TCAVT_TRY(c.gemm(a->memb, a->mem, l.w_kv, l.b_kv, l.ckv, TCAVT_F32, Mk, 2 * E, E, false, nullptr, false));
      const uint32_t s = mha_site();
      TCAVT_TRY(tcavt_mha(l.cq, E, l.ckv, 2 * E, l.ckv + E, 2 * E, l.catt, E, nullptr, a->B, a->L, a->Lk, nh, dh, scale, TCAVT_F32, act,
                          a->dropout_p, a->dropout_seed, s, stream));
      TCAVT_TRY(c.gemm(l.catt, static_cast<const float*>(l.catt), l.w_co, l.b_co, l.cy, TCAVT_F32, M, E, E, false, l.x1, true));
      TCAVT_TRY(c.norm(l.cy, l.n2_w, l.n2_b, l.x2, l.x2b));
      xin = l.x2;
      xinb = l.x2b;
    }
    // ---- feed-forward block
    TCAVT_TRY(c.gemm(xinb, xin, l.w1, l.b1, l.ffh, act, M, FF, E, true, nullptr, true));
    TCAVT_TRY(c.gemm(l.ffh, static_cast<const float*>(l.ffh), l.w2, l.b2, l.y2, TCAVT_F32, M, E, FF, false, xin, true));
    TCAVT_TRY(c.norm(l.y2, dec ? l.n3_w : l.n2_w, dec ? l.n3_b : l.n2_b, l.out, l.outb));
    x = l.out;
    xb = l.outb;
  }
  return TCAVT_OK;
}

// ---------------------------------------------------------------------------
// The trajectory head's cross-attention over the LLM's final hidden states (nn.MultiheadAttention with query = the To
// decoder tokens of a sample, key = value = its L hidden states; scripts/train.py:795-798) in ABSORBED form, one call
// (SURVEY.md 8b "cross_attn_forward").  Per head h with F = the sample's hidden states:
//     q'_h = q_h W_k[h]                     scores_h = q'_h F^T / sqrt(dh)   (+ a per-query constant from b_k: softmax-invariant)
//     P_h  = dropout(softmax(scores_h))     ctx_h = P_h F                    att_h = ctx_h W_v[h]^T + b_v[h]
// so the K / V projections act on B*To query rows instead of B*L hidden-state rows.  Launches: one
// transpose of F (per sample, keys padded to Lp), nh small GEMMs, one batched GEMM, the row softmax, one batched GEMM,
// nh small GEMMs -- the sequence tcavt_amd.model.TransformerLTSF.forward issued from Python.
// ---------------------------------------------------------------------------
extern "C" int tcavt_cross_attn_forward(const tcavt_cross_attn_args* a, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(a && a->q && a->wk_t && a->w_v && a->b_v && a->fh && a->fh_t && a->qp && a->scores && a->probs && a->ctx && a->att,
                  "cross_attn_forward: null pointer");
  TCAVT_CHECK_ARG(a->B > 0 && a->To > 0 && a->L > 0 && a->Lp >= a->L && a->Lp % 64 == 0 && a->H > 0 && a->nhead > 0 &&
                      a->H % a->nhead == 0 && (a->H / a->nhead) % 64 == 0 && a->H % 64 == 0,
                  "cross_attn_forward: bad shape (Lp %% 64 == 0 >= L, head dim %% 64 == 0)");
  TCAVT_CHECK_ARG(a->dtype16 == TCAVT_F16, "cross_attn_forward: fp16 storage only (the probabilities are carried in fp16)");
  TCAVT_CHECK_ARG(a->dropout_p >= 0.f && a->dropout_p < 1.f, "cross_attn_forward: dropout_p must be in [0, 1)");
  const int B = a->B, To = a->To, L = a->L, Lp = a->Lp, H = a->H, nh = a->nhead, dh = H / nh, M = B * To, dt = a->dtype16;
  const char* q = static_cast<const char*>(a->q);
  const char* wk_t = static_cast<const char*>(a->wk_t);
  const char* w_v = static_cast<const char*>(a->w_v);
  char* qp = static_cast<char*>(a->qp);
  char* ctx = static_cast<char*>(a->ctx);
  char* att = static_cast<char*>(a->att);
  // F^T per sample: [H][B * Lp], keys L..Lp-1 zero
  TCAVT_TRY(tcavt_transpose16(a->fh, H, a->fh_t, (int64_t)B * Lp, L, H, Lp, B, (int64_t)L * H, Lp, 0, stream));
  for (int h = 0; h < nh; ++h) {  // q'_h = q[:, h] W_k[h]   (W operand: W_k[h]^T, [H][dh])
    tcavt_gemm_args g = {};
    g.A = q + (size_t)h * dh * 2; g.lda = H; g.W = wk_t + (size_t)h * H * dh * 2; g.ldw = dh;
    g.C = qp + (size_t)h * M * H * 2; g.ldc = H; g.M = M; g.N = H; g.K = dh; g.out_dtype = dt; g.in_dtype = dt;
    TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
  }
  {  // scores[b, h] = q'_h[b] F[b]^T / sqrt(dh)
    tcavt_gemm_args g = {};
    g.A = a->qp; g.lda = H; g.W = a->fh; g.ldw = H; g.C = a->scores; g.ldc = Lp; g.M = To; g.N = Lp; g.K = H;
    g.out_dtype = TCAVT_F32; g.in_dtype = dt; g.tile = 64; g.acc_scale = (float)(1.0 / sqrt((double)dh));
    g.batch = B * nh; g.batch_inner = nh;
    g.sAo = (int64_t)To * H; g.sAi = (int64_t)M * H; g.sWo = (int64_t)L * H; g.sWi = 0;
    g.sCo = (int64_t)nh * To * Lp; g.sCi = (int64_t)To * Lp;
    TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
  }
  TCAVT_TRY(tcavt_softmax_rows(a->scores, Lp, a->probs, Lp, TCAVT_F16, B * nh * To, L, Lp, a->dropout_p, a->dropout_seed,
                               a->dropout_site, stream));
  {  // ctx[h][b] = P[b, h] F[b]
    tcavt_gemm_args g = {};
    g.A = a->probs; g.lda = Lp; g.W = a->fh_t; g.ldw = (int64_t)B * Lp; g.C = a->ctx; g.ldc = H; g.M = To; g.N = H; g.K = Lp;
    g.out_dtype = dt; g.in_dtype = TCAVT_F16; g.tile = 64;
    g.batch = B * nh; g.batch_inner = nh;
    g.sAo = (int64_t)nh * To * Lp; g.sAi = (int64_t)To * Lp; g.sWo = Lp; g.sWi = 0;
    g.sCo = (int64_t)To * H; g.sCi = (int64_t)M * H;
    TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
  }
  for (int h = 0; h < nh; ++h) {  // att[:, h] = ctx_h W_v[h]^T + b_v[h]
    tcavt_gemm_args g = {};
    g.A = ctx + (size_t)h * M * H * 2; g.lda = H; g.W = w_v + (size_t)h * dh * H * 2; g.ldw = H;
    g.C = att + (size_t)h * dh * 2; g.ldc = H; g.M = M; g.N = dh; g.K = H; g.out_dtype = dt; g.in_dtype = dt;
    g.bias = a->b_v + (size_t)h * dh; g.epilogue = TCAVT_EPI_BIAS;
    TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
  }
  return TCAVT_OK;
}

// ---------------------------------------------------------------------------
// TransformerLTSF.forward (scripts/train.py:808-842) as one call per phase (SURVEY.md 8b "ltsf_forward"):
//   phase 1 (independent of the LLM: the caller runs it on a side stream under the decoder stack): token projection +
//           per-channel N-Linear encoder + positional term (tcavt_ltsf_front) and the SelfAttentionBlock (train.py:674-686:
//           LayerNorm, MHA, residual from the NORMED tensor, LayerNorm, FFN, residual);
//   phase 2 (needs the final hidden states): lane_fc, N-Linear decoder, post-MLP, transpose, dec_proj, q projection,
//           cross-attention (tcavt_cross_attn_forward), out_proj, dec_unproj + residual, fusion layer, output head.
// Same kernels, buffers and dropout sites (first_site + 0..3: self-attention block; + 4: post-MLP; + 5: cross-attention
// weights) as tcavt_amd.model.TransformerLTSF issues from Python; fp32 where the reference's pixel-space arithmetic needs
// it, fp16 contractions around the cross-attention.
// ---------------------------------------------------------------------------
extern "C" int tcavt_ltsf_forward(const tcavt_ltsf_args* a, int phase, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(a && (phase == 1 || phase == 2 || phase == 3), "ltsf_forward: phase must be 1 (front), 2 (head) or 3 (both)");
  TCAVT_CHECK_ARG(a->B > 0 && a->C > 0 && a->T > 0 && a->To > 0 && a->F > 0 && a->nhead_sa > 0 && a->C % a->nhead_sa == 0,
                  "ltsf_forward: bad shape");
  TCAVT_CHECK_ARG(a->dropout_p >= 0.f && a->dropout_p < 1.f, "ltsf_forward: dropout_p must be in [0, 1)");
  const int B = a->B, C = a->C, T = a->T, To = a->To, Mt = B * T, Mo = B * To;
  const float p = a->dropout_p;
  const uint64_t seed = a->dropout_seed;
  const uint32_t s0 = a->first_site;
  auto gf32 = [&](const float* A, const float* W, const float* bias, float* Cc, int M, int N, int K, bool relu, const float* res,
                  int site_off) {
    const int flags = (bias ? TCAVT_EPI_BIAS : 0) | (relu ? TCAVT_EPI_RELU : 0) | (res ? TCAVT_EPI_RESIDUAL : 0);
    const bool drop = site_off >= 0 && p > 0.f;
    return tcavt_gemm_f32(A, K, W, K, bias, res, N, Cc, N, M, N, K, flags, drop ? p : 0.f, seed, drop ? s0 + (uint32_t)site_off : 0u,
                          stream);
  };
  if (phase & 1) {
    TCAVT_CHECK_ARG(a->x && a->conv_w && a->conv_b && a->enc_w && a->enc_b && a->pos && a->tok && a->sa_n1_w && a->sa_n1_b &&
                        a->sa_in_w && a->sa_in_b && a->sa_out_w && a->sa_out_b && a->sa_n2_w && a->sa_n2_b && a->sa_f0_w &&
                        a->sa_f0_b && a->sa_f3_w && a->sa_f3_b && a->sa_xn && a->sa_qkv && a->sa_att && a->sa_res1 && a->sa_rn &&
                        a->sa_f && a->e,
                    "ltsf_forward: phase 1: null pointer");
    TCAVT_TRY(tcavt_ltsf_front(a->x, a->conv_w, a->conv_b, a->enc_w, a->enc_b, a->pos, a->tok, a->xp_tok, B, C, T, stream));
    const int dh = C / a->nhead_sa;
    TCAVT_TRY(tcavt_layernorm(a->tok, nullptr, a->sa_n1_w, a->sa_n1_b, 1e-5f, a->sa_xn, nullptr, Mt, C, TCAVT_BF16, stream));
    TCAVT_TRY(gf32(a->sa_xn, a->sa_in_w, a->sa_in_b, a->sa_qkv, Mt, 3 * C, C, false, nullptr, -1));
    TCAVT_TRY(tcavt_mha(a->sa_qkv, 3 * C, a->sa_qkv + C, 3 * C, a->sa_qkv + 2 * C, 3 * C, a->sa_att, C, nullptr, B, T, T, a->nhead_sa,
                        dh, (float)(1.0 / sqrt((double)dh)), TCAVT_F32, TCAVT_F32, p, seed, p > 0.f ? s0 : 0u, stream));
    TCAVT_TRY(gf32(a->sa_att, a->sa_out_w, a->sa_out_b, a->sa_res1, Mt, C, C, false, a->sa_xn, 1));
    TCAVT_TRY(tcavt_layernorm(a->sa_res1, nullptr, a->sa_n2_w, a->sa_n2_b, 1e-5f, a->sa_rn, nullptr, Mt, C, TCAVT_BF16, stream));
    TCAVT_TRY(gf32(a->sa_rn, a->sa_f0_w, a->sa_f0_b, a->sa_f, Mt, 4 * C, C, true, nullptr, 2));
    TCAVT_TRY(gf32(a->sa_f, a->sa_f3_w, a->sa_f3_b, a->e, Mt, C, 4 * C, false, a->sa_rn, 3));
  }
  if (phase & 2) {
    TCAVT_CHECK_ARG(a->e && a->poly_emb && a->lane_w && a->lane_b && a->dec_w && a->dec_b && a->lane && a->d0 && a->dec_t &&
                        a->dec_tb && a->w_dp && a->b_dp && a->proj && a->w_q && a->b_q && a->xattn.q && a->w_co && a->b_co &&
                        a->cross && a->w_un && a->b_un && a->fused && a->fl_n_w && a->fl_n_b && a->fn && a->fl1_w && a->fl1_b &&
                        a->f1 && a->fl3_w && a->fl3_b && a->f2 && a->out_w && a->out_b && a->x && a->out && a->poly_dim > 0 &&
                        a->H > 0,
                    "ltsf_forward: phase 2: null pointer");
    const int H = a->H, CT = C * To;
    TCAVT_TRY(gf32(a->poly_emb, a->lane_w, a->lane_b, a->lane, B, CT, a->poly_dim, false, nullptr, -1));
    TCAVT_TRY(tcavt_ltsf_decode(a->e, a->dec_w, a->dec_b, a->lane, a->d0, B, C, T, To, stream));
    const float* d1 = a->d0;
    if (a->post_hidden > 0) {
      TCAVT_CHECK_ARG(a->pm0_w && a->pm0_b && a->pm3_w && a->pm3_b && a->hid && a->d1, "ltsf_forward: post-MLP: null pointer");
      TCAVT_TRY(gf32(a->d0, a->pm0_w, a->pm0_b, a->hid, B, a->post_hidden, CT, true, nullptr, 4));
      TCAVT_TRY(gf32(a->hid, a->pm3_w, a->pm3_b, a->d1, B, CT, a->post_hidden, false, nullptr, -1));
      d1 = a->d1;
    }
    TCAVT_TRY(tcavt_transpose_ct(d1, a->dec_t, a->dec_tb, B, C, To, TCAVT_F16, stream));
    auto g16 = [&](const void* A, const void* W, const float* bias, void* Cc, int out_dtype, int M, int N, int K, const float* res) {
      tcavt_gemm_args g = {};
      g.A = A; g.lda = K; g.W = W; g.ldw = K; g.C = Cc; g.ldc = N; g.M = M; g.N = N; g.K = K; g.out_dtype = out_dtype;
      g.in_dtype = TCAVT_F16; g.bias = bias; g.residual = res; g.ldr = N;
      g.epilogue = (bias ? TCAVT_EPI_BIAS : 0) | (res ? TCAVT_EPI_RESIDUAL : 0);
      return tcavt_gemm_bf16(&g, stream);
    };
    TCAVT_TRY(g16(a->dec_tb, a->w_dp, a->b_dp, a->proj, TCAVT_F16, Mo, H, C, nullptr));
    TCAVT_TRY(g16(a->proj, a->w_q, a->b_q, const_cast<void*>(a->xattn.q), TCAVT_F16, Mo, H, H, nullptr));
    TCAVT_TRY(tcavt_cross_attn_forward(&a->xattn, stream));
    TCAVT_TRY(g16(a->xattn.att, a->w_co, a->b_co, a->cross, TCAVT_F16, Mo, H, H, nullptr));
    TCAVT_TRY(g16(a->cross, a->w_un, a->b_un, a->fused, TCAVT_F32, Mo, C, H, a->dec_t));
    TCAVT_TRY(tcavt_layernorm(a->fused, nullptr, a->fl_n_w, a->fl_n_b, 1e-5f, a->fn, nullptr, Mo, C, TCAVT_BF16, stream));
    TCAVT_TRY(gf32(a->fn, a->fl1_w, a->fl1_b, a->f1, Mo, C, C, true, nullptr, -1));
    TCAVT_TRY(gf32(a->f1, a->fl3_w, a->fl3_b, a->f2, Mo, C, C, false, nullptr, -1));
    TCAVT_TRY(tcavt_out_head(a->f2, a->out_w, a->out_b, a->x, a->out, B, To, C, a->F, T, a->add_last, stream));
  }
  return TCAVT_OK;
}

// ---------------------------------------------------------------------------
// Backward of tcavt_cross_attn_forward (SURVEY.md 8b "cross_attn_backward"), the hidden states being a constant (frozen
// MLLM): given g_att = dL/d att,
//     gW_v[h] = g_att_h^T ctx_h      gb_v += colsum(g_att)      g_ctx_h = g_att_h W_v[h]
//     g_P = g_ctx F^T                g_S = softmax'(P, g_P) (with the forward's attention-weight mask)      g_q' = g_S F
//     gW_k[h] = q_h^T g_q'_h         gb_k = 0 (softmax-invariant)                                            g_q_h = g_q'_h W_k[h]^T
// Gradient-side operands are bf16 (forward activations are converted while they are transposed), weight gradients fp32
// straight into the caller's in_proj_weight gradient rows.  Every contraction runs over the B*To query rows.
// ---------------------------------------------------------------------------
extern "C" int tcavt_cross_attn_backward(const tcavt_cross_attn_bwd_args* a, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(a && a->fwd && a->g_att && a->w_in && a->gw_in && a->gb_in && a->g_q && a->fh_tb && a->fh_b && a->ga_t &&
                      a->g_ctx && a->w_t && a->x_t && a->d_p && a->d_s && a->g_qp,
                  "cross_attn_backward: null pointer");
  const tcavt_cross_attn_args& f = *a->fwd;
  const int B = f.B, To = f.To, L = f.L, Lp = f.Lp, H = f.H, nh = f.nhead, dh = H / nh, M = B * To, Mp = (M + 63) / 64 * 64;
  TCAVT_CHECK_ARG(f.dtype16 == TCAVT_F16 && (f.dropout_p == 0.f || a->p_undropped), "cross_attn_backward: fp16 forward; train mode needs p_undropped");
  const float scale = (float)(1.0 / sqrt((double)dh));
  const int BF = TCAVT_BF16;
  char* g_att = static_cast<char*>(const_cast<void*>(a->g_att));
  char* g_ctx = static_cast<char*>(a->g_ctx);
  char* g_qp = static_cast<char*>(a->g_qp);
  char* g_q = static_cast<char*>(a->g_q);
  char* ga_t = static_cast<char*>(a->ga_t);
  const char* ctx = static_cast<const char*>(f.ctx);
  const char* q = static_cast<const char*>(f.q);
  auto gemm = [&](const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int out_dtype, int Mr, int N, int K) {
    tcavt_gemm_args g = {};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = Mr; g.N = N; g.K = K; g.out_dtype = out_dtype; g.in_dtype = BF;
    return tcavt_gemm_bf16(&g, stream);
  };
  // bf16 copies of the hidden states: per-sample transposed [H][B*Lp] (converted while transposing) and, transposed back, row-major
  TCAVT_TRY(tcavt_transpose16(f.fh, H, a->fh_tb, (int64_t)B * Lp, L, H, Lp, B, (int64_t)L * H, Lp, 1, stream));
  TCAVT_TRY(tcavt_transpose16(a->fh_tb, (int64_t)B * Lp, a->fh_b, H, H, B * Lp, H, 1, 0, 0, 0, stream));
  // ---- value side
  TCAVT_TRY(tcavt_colsum(a->g_att, H, BF, a->gb_in + 2 * H, M, H, 1, stream));
  TCAVT_TRY(tcavt_transpose16(a->g_att, H, a->ga_t, Mp, M, H, Mp, 1, 0, 0, 0, stream));
  for (int h = 0; h < nh; ++h) {
    const float* Wv_h = a->w_in + (size_t)(2 * H + h * dh) * H;
    TCAVT_TRY(tcavt_transpose_f32_bf16(Wv_h, H, a->w_t, dh, dh, H, dh, stream));                         // W_v[h]^T  [H][dh]
    TCAVT_TRY(gemm(g_att + (size_t)h * dh * 2, H, a->w_t, dh, g_ctx + (size_t)h * M * H * 2, H, BF, M, H, dh));
    TCAVT_TRY(tcavt_transpose16(ctx + (size_t)h * M * H * 2, H, a->x_t, Mp, M, H, Mp, 1, 0, 0, 1, stream));  // ctx_h^T  [H][Mp]
    TCAVT_TRY(gemm(ga_t + (size_t)h * dh * Mp * 2, Mp, a->x_t, Mp, a->gw_in + (size_t)(2 * H + h * dh) * H, H, TCAVT_F32, dh, H, Mp));
  }
  // ---- scores
  {
    tcavt_gemm_args g = {};
    g.A = a->g_ctx; g.lda = H; g.W = a->fh_b; g.ldw = H; g.C = a->d_p; g.ldc = Lp; g.M = To; g.N = Lp; g.K = H;
    g.out_dtype = TCAVT_F32; g.in_dtype = BF; g.tile = 64; g.batch = B * nh; g.batch_inner = nh;
    g.sAo = (int64_t)To * H; g.sAi = (int64_t)M * H; g.sWo = (int64_t)Lp * H; g.sWi = 0;
    g.sCo = (int64_t)nh * To * Lp; g.sCi = (int64_t)To * Lp;
    TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
  }
  const void* p_used = f.probs;
  if (f.dropout_p > 0.f) {  // the softmax backward needs the un-dropped probabilities and the masked dP
    TCAVT_TRY(tcavt_dropout(a->d_p, a->d_p, (int64_t)B * nh * To * Lp, TCAVT_F32, f.dropout_p, f.dropout_seed, f.dropout_site, nullptr, stream));
    TCAVT_TRY(tcavt_softmax_rows(f.scores, Lp, a->p_undropped, Lp, TCAVT_F16, B * nh * To, L, Lp, 0.f, 0, 0, stream));
    p_used = a->p_undropped;
  }
  TCAVT_TRY(tcavt_softmax_bwd_rows(p_used, Lp, a->d_p, Lp, a->d_s, Lp, scale, B * nh * To, L, Lp, stream));
  // ---- query side
  {
    tcavt_gemm_args g = {};
    g.A = a->d_s; g.lda = Lp; g.W = a->fh_tb; g.ldw = (int64_t)B * Lp; g.C = a->g_qp; g.ldc = H; g.M = To; g.N = H; g.K = Lp;
    g.out_dtype = BF; g.in_dtype = BF; g.tile = 64; g.batch = B * nh; g.batch_inner = nh;
    g.sAo = (int64_t)nh * To * Lp; g.sAi = (int64_t)To * Lp; g.sWo = Lp; g.sWi = 0;
    g.sCo = (int64_t)To * H; g.sCi = (int64_t)M * H;
    TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
  }
  for (int h = 0; h < nh; ++h) {
    const float* Wk_h = a->w_in + (size_t)(H + h * dh) * H;
    TCAVT_TRY(tcavt_cast_f32_16(Wk_h, a->w_t, (int64_t)dh * H, BF, stream));                              // W_k[h] bf16 [dh][H]
    TCAVT_TRY(gemm(g_qp + (size_t)h * M * H * 2, H, a->w_t, H, g_q + (size_t)h * dh * 2, H, BF, M, dh, H));
    TCAVT_TRY(tcavt_transpose16(q + (size_t)h * dh * 2, H, a->ga_t, Mp, M, dh, Mp, 1, 0, 0, 1, stream));     // q_h^T [dh][Mp] (ga_t is free now)
    TCAVT_TRY(tcavt_transpose16(g_qp + (size_t)h * M * H * 2, H, a->x_t, Mp, M, H, Mp, 1, 0, 0, 0, stream)); // g_q'_h^T [H][Mp]
    TCAVT_TRY(gemm(a->ga_t, Mp, a->x_t, Mp, a->gw_in + (size_t)(H + h * dh) * H, H, TCAVT_F32, dh, H, Mp));
  }
  return TCAVT_OK;
}
