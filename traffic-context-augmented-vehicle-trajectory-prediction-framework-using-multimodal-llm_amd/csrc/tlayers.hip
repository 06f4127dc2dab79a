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
// Backward of the fp32 encoder-layer stack (the lane-polygon encoder's layers, scripts/train.py:352-383, in the training
// step): the launch sequence of tcavt_amd.backward.Backward.polygon's layer loop on one stream.  Per layer, last first:
//   out = LN2(y2), y2 = x1 + drop(linear2(drop(relu(linear1(x1)))));  x1 = LN1(y), y = x + drop(out_proj(MHA(in_proj(x))))
// Dropout sites as the forward numbered them: first_site + 4 * layer + {0: attention weights, 1: after out_proj, 2: after
// the ReLU, 3: after linear2}.
// ---------------------------------------------------------------------------
extern "C" int tcavt_tlayer_stack_backward(const tcavt_tstack_bwd_args* a, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(a && a->fwd && a->grads && a->g_out && a->g_x && a->g_tmp && a->g_y2 && a->g_x1 && a->g_y && a->g_att && a->g_f &&
                      a->g_qkv,
                  "tlayer_stack_backward: null pointer");
  const tcavt_tstack_args& f = *a->fwd;
  TCAVT_CHECK_ARG(f.layers && f.n_layers > 0 && f.x && f.B > 0 && f.L > 0 && f.E > 0 && f.FF > 0 && f.nhead > 0 && f.E % f.nhead == 0,
                  "tlayer_stack_backward: bad forward args");
  TCAVT_CHECK_ARG(f.dtype16 == 0, "tlayer_stack_backward: fp32 layers only (the 16-bit stacks' backward is composed by the caller)");
  TCAVT_CHECK_ARG(f.dropout_p >= 0.f && f.dropout_p < 1.f && (f.dropout_p == 0.f || (a->g_y2d && a->g_yd)),
                  "tlayer_stack_backward: dropout_p in [0, 1); train mode needs g_y2d / g_yd");
  const int E = f.E, FF = f.FF, nh = f.nhead, dh = E / nh, M = f.B * f.L;
  const float p = f.dropout_p;
  const float scale = (float)(1.0 / sqrt((double)dh));
  auto lin32 = [&](const float* x, const float* W, const float* gy, float* gW, float* gb, float* gx, int Mr, int N, int K) -> int {
    TCAVT_TRY(tcavt_gemm_f32_strided(gy, 1, N, x, 1, K, nullptr, nullptr, 0, gW, K, N, K, Mr, TCAVT_EPI_ACCUM, stream));
    TCAVT_TRY(tcavt_colsum(gy, N, TCAVT_F32, gb, Mr, N, 1, stream));
    return tcavt_gemm_f32_strided(gy, N, 1, W, 1, K, nullptr, nullptr, 0, gx, K, Mr, K, N, 0, stream);
  };
  const float* g_in = a->g_out;
  for (int li = f.n_layers - 1; li >= 0; --li) {
    const tcavt_tlayer& l = f.layers[li];
    const tcavt_tlayer_grads& g = a->grads[li];
    TCAVT_CHECK_ARG(!l.w_q, "tlayer_stack_backward: layer %d is a decoder layer (encoder layers only)", li);
    TCAVT_CHECK_ARG(l.w_in && l.w_out && l.w1 && l.w2 && l.n1_w && l.n2_w && l.qkv && l.att && l.y && l.x1 && l.ffh && l.y2 &&
                        (li == 0 || f.layers[li - 1].out) && g.g_w_in && g.g_b_in && g.g_w_out && g.g_b_out && g.g_w1 && g.g_b1 &&
                        g.g_w2 && g.g_b2 && g.g_n1_w && g.g_n1_b && g.g_n2_w && g.g_n2_b,
                    "tlayer_stack_backward: layer %d: null weight / activation / gradient", li);
    const float* x_in = li == 0 ? f.x : f.layers[li - 1].out;
    const uint32_t s = f.first_site + 4u * (uint32_t)li;
    float* g_xin = (li & 1) ? a->g_tmp : a->g_x;  // alternate, so that layer 0 writes the caller's g_x
    auto drop = [&](const float* gi, float* go, int64_t n, uint32_t site) {
      return tcavt_dropout(gi, go, n, TCAVT_F32, p, f.dropout_seed, site, nullptr, stream);
    };
    // feed-forward block
    TCAVT_TRY(tcavt_layernorm_bwd(l.y2, l.n2_w, g_in, 1e-5f, a->g_y2, g.g_n2_w, g.g_n2_b, M, E, stream));
    const float* g_y2d = a->g_y2;
    if (p > 0.f) {
      TCAVT_TRY(drop(a->g_y2, a->g_y2d, (int64_t)M * E, s + 3));
      g_y2d = a->g_y2d;
    }
    const float* ffh = static_cast<const float*>(l.ffh);
    TCAVT_TRY(lin32(ffh, static_cast<const float*>(l.w2), g_y2d, g.g_w2, g.g_b2, a->g_f, M, E, FF));
    if (p > 0.f) TCAVT_TRY(drop(a->g_f, a->g_f, (int64_t)M * FF, s + 2));
    TCAVT_TRY(tcavt_relu_bwd(a->g_f, ffh, TCAVT_F32, (int64_t)M * FF, stream));
    TCAVT_TRY(lin32(l.x1, static_cast<const float*>(l.w1), a->g_f, g.g_w1, g.g_b1, a->g_x1, M, FF, E));
    TCAVT_TRY(tcavt_add_inplace(a->g_x1, a->g_y2, (int64_t)M * E, stream));
    // self-attention block
    TCAVT_TRY(tcavt_layernorm_bwd(l.y, l.n1_w, a->g_x1, 1e-5f, a->g_y, g.g_n1_w, g.g_n1_b, M, E, stream));
    const float* g_yd = a->g_y;
    if (p > 0.f) {
      TCAVT_TRY(drop(a->g_y, a->g_yd, (int64_t)M * E, s + 1));
      g_yd = a->g_yd;
    }
    TCAVT_TRY(lin32(static_cast<const float*>(l.att), static_cast<const float*>(l.w_out), g_yd, g.g_w_out, g.g_b_out, a->g_att, M, E, E));
    TCAVT_TRY(tcavt_mha_bwd(l.qkv, 3 * E, l.qkv + E, 3 * E, l.qkv + 2 * E, 3 * E, a->g_att, E, a->g_qkv, a->g_qkv + E, a->g_qkv + 2 * E,
                            3 * E, f.key_len, f.B, f.L, f.L, nh, dh, scale, p, f.dropout_seed, p > 0.f ? s : 0u, stream));
    TCAVT_TRY(lin32(x_in, static_cast<const float*>(l.w_in), a->g_qkv, g.g_w_in, g.g_b_in, g_xin, M, 3 * E, E));
    TCAVT_TRY(tcavt_add_inplace(g_xin, a->g_y, (int64_t)M * E, stream));
    g_in = g_xin;
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

// ---------------------------------------------------------------------------
// Backward of tcavt_ltsf_forward (SURVEY.md 8b "ltsf_backward"): what autograd does for TransformerLTSF in the training step
// (scripts/train.py:1168-1183; the module: :808-842), as the launch sequence tcavt_amd.backward.Backward.ltsf issues from
// Python -- same kernels, same operands, so the gradients are bit-identical; here everything runs on the caller's one
// stream, the Python composition moves the weight-gradient leaves to side streams.
//   fp32 nn.Linear y = x W^T + b:   gW += gy^T x   gb += colsum(gy)   gx = gy W                       (strided fp32 GEMMs)
//   16-bit projection (MFMA):       gW = gy16^T x16 over transposed bf16 copies, gb += colsum(gy), gx = gy16 W16
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, long ld_src, float* __restrict__ dst,
                                                        long ld_dst, int width) {
  const float* s = src + (long)blockIdx.x * ld_src;
  float* d = dst + (long)blockIdx.x * ld_dst;
  for (int i = threadIdx.x; i < width; i += 256) d[i] = s[i];
}

int copy_rows(const float* src, long ld_src, float* dst, long ld_dst, int rows, int width, tcavt_stream_t stream) {
  hipLaunchKernelGGL(copy_rows_kernel, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream), src, ld_src, dst, ld_dst, width);
  TCAVT_CHECK_LAUNCH("ltsf_backward: copy_rows");
  return TCAVT_OK;
}
}  // namespace

extern "C" int tcavt_ltsf_backward(const tcavt_ltsf_bwd_args* a, int phase, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(a && a->fwd && (phase == 1 || phase == 2 || phase == 3), "ltsf_backward: phase must be 1 (head), 2 (front) or 3 (both)");
  const tcavt_ltsf_args& f = *a->fwd;
  TCAVT_CHECK_ARG(f.B > 0 && f.C > 0 && f.T > 0 && f.To > 0 && f.F > 0 && f.H > 0 && f.nhead_sa > 0 && f.C % f.nhead_sa == 0 &&
                      f.poly_dim > 0,
                  "ltsf_backward: bad shape");
  TCAVT_CHECK_ARG(f.dropout_p >= 0.f && f.dropout_p < 1.f, "ltsf_backward: dropout_p must be in [0, 1)");
  const int B = f.B, C = f.C, T = f.T, To = f.To, H = f.H, Mo = B * To, Mt = B * T, CT = C * To, Mp = (Mo + 63) / 64 * 64;
  const float p = f.dropout_p;
  const uint64_t seed = f.dropout_seed;
  const uint32_t s0 = f.first_site;
  const int BF = TCAVT_BF16;
  // fp32 nn.Linear backward: x [M][K], W [N][K], gy [M][N]
  auto lin32 = [&](const float* x, const float* W, const float* gy, float* gW, float* gb, float* gx, int M, int N, int K) -> int {
    TCAVT_TRY(tcavt_gemm_f32_strided(gy, 1, N, x, 1, K, nullptr, nullptr, 0, gW, K, N, K, M, TCAVT_EPI_ACCUM, stream));
    TCAVT_TRY(tcavt_colsum(gy, N, TCAVT_F32, gb, M, N, 1, stream));
    if (gx) TCAVT_TRY(tcavt_gemm_f32_strided(gy, N, 1, W, 1, K, nullptr, nullptr, 0, gx, K, M, K, N, 0, stream));
    return TCAVT_OK;
  };
  // 16-bit projection backward over the Mo decoder-token rows: x16 fp16 [Mo][K] (forward activation), W fp32 [N][K],
  // gy [Mo][N] fp32 or bf16, gx [Mo][K] of gx_dtype
  auto lin16 = [&](const void* x16, const float* W, const void* gy, int gy_dtype, float* gW, float* gb, void* gx, int gx_dtype,
                   int N, int K) -> int {
    const void* gyb = gy;
    if (gy_dtype == TCAVT_F32) {
      TCAVT_TRY(tcavt_cast_f32_16(static_cast<const float*>(gy), a->s_gyb, (int64_t)Mo * N, BF, stream));
      gyb = a->s_gyb;
    }
    TCAVT_TRY(tcavt_transpose16(gyb, N, a->s_gyt, Mp, Mo, N, Mp, 1, 0, 0, 0, stream));
    TCAVT_TRY(tcavt_transpose16(x16, K, a->s_xt, Mp, Mo, K, Mp, 1, 0, 0, 1, stream));
    tcavt_gemm_args g = {};
    g.A = a->s_gyt; g.lda = Mp; g.W = a->s_xt; g.ldw = Mp; g.C = gW; g.ldc = K; g.M = N; g.N = K; g.K = Mp;
    g.out_dtype = TCAVT_F32; g.in_dtype = BF;
    TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
    TCAVT_TRY(tcavt_colsum(gy, N, gy_dtype, gb, Mo, N, 1, stream));
    TCAVT_TRY(tcavt_transpose_f32_bf16(W, K, a->s_wt, N, N, K, N, stream));
    tcavt_gemm_args d = {};
    d.A = gyb; d.lda = N; d.W = a->s_wt; d.ldw = N; d.C = gx; d.ldc = K; d.M = Mo; d.N = K; d.K = N;
    d.out_dtype = gx_dtype; d.in_dtype = BF;
    return tcavt_gemm_bf16(&d, stream);
  };
  // gradient through a dropout site of the forward
  auto drop = [&](const float* g, float* out, int64_t n, int site_off) -> int {
    return tcavt_dropout(g, out, n, TCAVT_F32, p, seed, s0 + (uint32_t)site_off, nullptr, stream);
  };

  if (phase & 1) {
    TCAVT_CHECK_ARG(a->g_out && f.f2 && f.f1 && f.fn && f.fused && f.cross && f.xattn.att && f.xattn.q && f.proj && f.dec_tb && f.d0 &&
                        f.poly_emb && f.out_w && f.fl3_w && f.fl1_w && f.fl_n_w && f.lane_w && a->w_un && a->w_co && a->w_dp &&
                        a->xattn.w_in && a->xattn.gw_in && a->xattn.gb_in && a->xattn.g_att && a->xattn.g_q,
                    "ltsf_backward: phase 1: null forward pointer");
    TCAVT_CHECK_ARG(a->g_out_w && a->g_out_b && a->g_fl3_w && a->g_fl3_b && a->g_fl1_w && a->g_fl1_b && a->g_fl_n_w && a->g_fl_n_b &&
                        a->g_un_w && a->g_un_b && a->g_co_w && a->g_co_b && a->g_dp_w && a->g_dp_b && a->g_lane_w && a->g_lane_b &&
                        a->g_poly && a->g_f2 && a->g_f1 && a->g_fn && a->g_dec_t && a->g_dt2 && a->g_cross && a->g_proj && a->g_d1 &&
                        a->s_gyb && a->s_gyt && a->s_xt && a->s_wt,
                    "ltsf_backward: phase 1: null gradient pointer / workspace");
    TCAVT_CHECK_ARG(a->xattn.fwd == &f.xattn, "ltsf_backward: xattn.fwd must point at fwd->xattn");
    // output head: out = f2 W_out^T + b (+ last position: no gradient needed)
    TCAVT_TRY(tcavt_out_head_bwd(a->g_out, f.f2, f.out_w, a->g_f2, a->g_out_w, a->g_out_b, B, To, C, f.F, stream));
    // fusion layer: LayerNorm -> Linear -> ReLU -> Linear
    TCAVT_TRY(lin32(f.f1, f.fl3_w, a->g_f2, a->g_fl3_w, a->g_fl3_b, a->g_f1, Mo, C, C));
    TCAVT_TRY(tcavt_relu_bwd(a->g_f1, f.f1, TCAVT_F32, (int64_t)Mo * C, stream));
    TCAVT_TRY(lin32(f.fn, f.fl1_w, a->g_f1, a->g_fl1_w, a->g_fl1_b, a->g_fn, Mo, C, C));
    TCAVT_TRY(tcavt_layernorm_bwd(f.fused, f.fl_n_w, a->g_fn, 1e-5f, a->g_dec_t, a->g_fl_n_w, a->g_fl_n_b, Mo, C, stream));
    // fused = cross W_un^T + b_un + dec_t;  cross = att W_co^T + b_co
    TCAVT_TRY(lin16(f.cross, a->w_un, a->g_dec_t, TCAVT_F32, a->g_un_w, a->g_un_b, a->g_cross, BF, C, H));
    TCAVT_TRY(lin16(f.xattn.att, a->w_co, a->g_cross, BF, a->g_co_w, a->g_co_b, const_cast<void*>(a->xattn.g_att), BF, H, H));
    // attention core (absorbed K / V projections), then the q projection (rows 0..H of the packed in_proj)
    TCAVT_TRY(tcavt_cross_attn_backward(&a->xattn, stream));
    TCAVT_TRY(lin16(f.proj, a->xattn.w_in, a->xattn.g_q, BF, a->xattn.gw_in, a->xattn.gb_in, a->g_proj, BF, H, H));
    // proj = dec_t W_dp^T + b_dp; total gradient of dec_t; dec_t [B][To][C] -> d1 [B][C][To]
    TCAVT_TRY(lin16(f.dec_tb, a->w_dp, a->g_proj, BF, a->g_dp_w, a->g_dp_b, a->g_dt2, TCAVT_F32, H, C));
    TCAVT_TRY(tcavt_add_inplace(a->g_dt2, a->g_dec_t, (int64_t)Mo * C, stream));
    TCAVT_TRY(tcavt_transpose_ct(a->g_dt2, a->g_d1, nullptr, B, To, C, BF, stream));
    // post-MLP: d1 = drop(relu(d0 W0^T + b0)) W3^T + b3
    const float* g_d0 = a->g_d1;
    if (f.post_hidden > 0) {
      TCAVT_CHECK_ARG(f.hid && f.pm0_w && f.pm3_w && a->g_hid && a->g_d0 && a->g_pm0_w && a->g_pm0_b && a->g_pm3_w && a->g_pm3_b,
                      "ltsf_backward: post-MLP: null pointer");
      TCAVT_TRY(lin32(f.hid, f.pm3_w, a->g_d1, a->g_pm3_w, a->g_pm3_b, a->g_hid, B, CT, f.post_hidden));
      if (p > 0.f) TCAVT_TRY(drop(a->g_hid, a->g_hid, (int64_t)B * f.post_hidden, 4));
      TCAVT_TRY(tcavt_relu_bwd(a->g_hid, f.hid, TCAVT_F32, (int64_t)B * f.post_hidden, stream));
      TCAVT_TRY(lin32(f.d0, f.pm0_w, a->g_hid, a->g_pm0_w, a->g_pm0_b, a->g_d0, B, f.post_hidden, CT));
      g_d0 = a->g_d0;
    }
    // d0 = NLinear_dec(e) + lane_fc(poly_emb)
    TCAVT_TRY(lin32(f.poly_emb, f.lane_w, g_d0, a->g_lane_w, a->g_lane_b, a->g_poly, B, CT, f.poly_dim));
  }
  if (phase & 2) {
    TCAVT_CHECK_ARG(f.e && f.dec_w && f.enc_w && f.tok && f.xp_tok && f.x && f.sa_xn && f.sa_qkv && f.sa_att && f.sa_res1 && f.sa_rn &&
                        f.sa_f && f.sa_n1_w && f.sa_in_w && f.sa_out_w && f.sa_n2_w && f.sa_f0_w && f.sa_f3_w,
                    "ltsf_backward: phase 2: null forward pointer (the forward must keep xp_tok)");
    TCAVT_CHECK_ARG(a->g_dec_w && a->g_dec_b && a->g_enc_w && a->g_enc_b && a->g_pos && a->pos_ld >= T && a->g_conv_w && a->g_conv_b &&
                        a->g_sa_n1_w && a->g_sa_n1_b && a->g_sa_in_w && a->g_sa_in_b && a->g_sa_out_w && a->g_sa_out_b && a->g_sa_n2_w &&
                        a->g_sa_n2_b && a->g_sa_f0_w && a->g_sa_f0_b && a->g_sa_f3_w && a->g_sa_f3_b && a->g_dw && a->g_db && a->g_e &&
                        a->g_ff && a->g_rn && a->g_res1 && a->g_att_sa && a->g_qkv && a->g_xn && a->g_tok && a->g_ew && a->g_eb &&
                        a->g_xp && (p == 0.f || (a->g_e_d && a->g_res1_d)) && (f.post_hidden > 0 ? a->g_d0 != nullptr : a->g_d1 != nullptr),
                    "ltsf_backward: phase 2: null gradient pointer / workspace");
    const float* g_d0 = f.post_hidden > 0 ? a->g_d0 : a->g_d1;
    // N-Linear decoder: stacked gradients, then one strided copy into the per-channel nn.Linear gradient views
    TCAVT_TRY(tcavt_nlinear_bwd(f.e, f.dec_w, g_d0, CT, To, 1, a->g_dw, a->g_db, a->g_e, B, C, T, To, stream));
    TCAVT_TRY(copy_rows(a->g_dw, (long)To * T, a->g_dec_w, a->dec_stride, C, To * T, stream));
    TCAVT_TRY(copy_rows(a->g_db, To, a->g_dec_b, a->dec_stride, C, To, stream));
    // SelfAttentionBlock (train.py:674-686): e = drop(ff W3^T + b3) + rn, ff = drop(relu(rn W0^T + b0)), rn = LN2(res1),
    // res1 = drop(att Wo^T + bo) + xn, att = MHA(xn), xn = LN1(tok)
    const float* g_e_d = a->g_e;
    if (p > 0.f) {
      TCAVT_TRY(drop(a->g_e, a->g_e_d, (int64_t)Mt * C, 3));
      g_e_d = a->g_e_d;
    }
    TCAVT_TRY(lin32(f.sa_f, f.sa_f3_w, g_e_d, a->g_sa_f3_w, a->g_sa_f3_b, a->g_ff, Mt, C, 4 * C));
    if (p > 0.f) TCAVT_TRY(drop(a->g_ff, a->g_ff, (int64_t)Mt * 4 * C, 2));
    TCAVT_TRY(tcavt_relu_bwd(a->g_ff, f.sa_f, TCAVT_F32, (int64_t)Mt * 4 * C, stream));
    TCAVT_TRY(lin32(f.sa_rn, f.sa_f0_w, a->g_ff, a->g_sa_f0_w, a->g_sa_f0_b, a->g_rn, Mt, 4 * C, C));
    TCAVT_TRY(tcavt_add_inplace(a->g_rn, a->g_e, (int64_t)Mt * C, stream));
    TCAVT_TRY(tcavt_layernorm_bwd(f.sa_res1, f.sa_n2_w, a->g_rn, 1e-5f, a->g_res1, a->g_sa_n2_w, a->g_sa_n2_b, Mt, C, stream));
    const float* g_res1_d = a->g_res1;
    if (p > 0.f) {
      TCAVT_TRY(drop(a->g_res1, a->g_res1_d, (int64_t)Mt * C, 1));
      g_res1_d = a->g_res1_d;
    }
    TCAVT_TRY(lin32(f.sa_att, f.sa_out_w, g_res1_d, a->g_sa_out_w, a->g_sa_out_b, a->g_att_sa, Mt, C, C));
    const int dh = C / f.nhead_sa;
    TCAVT_TRY(tcavt_mha_bwd(f.sa_qkv, 3 * C, f.sa_qkv + C, 3 * C, f.sa_qkv + 2 * C, 3 * C, a->g_att_sa, C, a->g_qkv, a->g_qkv + C,
                            a->g_qkv + 2 * C, 3 * C, nullptr, B, T, T, f.nhead_sa, dh, (float)(1.0 / sqrt((double)dh)), p, seed,
                            p > 0.f ? s0 : 0u, stream));
    TCAVT_TRY(lin32(f.sa_xn, f.sa_in_w, a->g_qkv, a->g_sa_in_w, a->g_sa_in_b, a->g_xn, Mt, 3 * C, C));
    TCAVT_TRY(tcavt_add_inplace(a->g_xn, a->g_res1, (int64_t)Mt * C, stream));
    TCAVT_TRY(tcavt_layernorm_bwd(f.tok, f.sa_n1_w, a->g_xn, 1e-5f, a->g_tok, a->g_sa_n1_w, a->g_sa_n1_b, Mt, C, stream));
    // front: tok = NLinear_enc(conv(x)) + pos
    TCAVT_TRY(tcavt_nlinear_bwd(f.xp_tok, f.enc_w, a->g_tok, (int64_t)T * C, 1, C, a->g_ew, a->g_eb, a->g_xp, B, C, T, T, stream));
    TCAVT_TRY(copy_rows(a->g_ew, (long)T * T, a->g_enc_w, a->enc_stride, C, T * T, stream));
    TCAVT_TRY(copy_rows(a->g_eb, T, a->g_enc_b, a->enc_stride, C, T, stream));
    TCAVT_TRY(copy_rows(a->g_eb, T, a->g_pos, a->pos_ld, C, T, stream));  // pos_encoding: the reduction of the encoder bias
    TCAVT_TRY(tcavt_conv1x1_bwd(a->g_xp, f.x, a->g_conv_w, a->g_conv_b, B, C, T, f.F, stream));
  }
  return TCAVT_OK;
}
