// Stage-level entry points: the Llama decoder stack composed in C++ over the kernels of this library.
//
// What the reference does in Python -- HF LlamaModel.forward looping over LlamaDecoderLayer (modeling_llama.py:376-424,
// 296-326 as called from scripts/train.py:446-452) -- is one C call here: every launch of every layer is enqueued on
// the caller's stream without returning to the interpreter (6 launches per layer in eval, 8 with LoRA dropout), so the
// host side of a 16-layer pass costs ~0.3 ms instead of ~6 ms of ctypes calls, and a hipGraph capture of the pass is a
// straight line.
//
// RMSNorm is not a kernel of its own inside the stack ("fused RMSNorm+RoPE+QKV-proj"):
//     (x * rsqrt(mean(x^2) + eps) * gamma) . W^T  ==  rsqrt(mean(x^2) + eps) * ( x16 . (W * gamma)^T )
// gamma is folded into the packed weights by the caller; the residual epilogues of o_proj / down_proj (and the embedding
// kernel for layer 0) leave the 16-bit copy x16 of the stream and, per row, H / 64 partial sums of squares; the q|k|v
// and gate|up epilogues add the partials in index order and apply the row scale (TCAVT_EPI_NORM_OUT / _ROWSCALE).
// Only the final norm, whose fp32 result is an output, is a launch of tcavt_rmsnorm.
#include "common.hpp"

namespace tcavt {

// rotated k / v of one layer -> the generation cache [B][lmax][nkv * 64]
__global__ __launch_bounds__(256) void kv_store_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ kc,
                                                       bf16_t* __restrict__ vc, int L, int lmax, int nq, int nkv) {
  const int ld = (nq + 2 * nkv) * 64, w = nkv * 64;  // elements per row: source / cache
  const long row = blockIdx.x;                       // b * L + l
  const int b = (int)(row / L), l = (int)(row % L);
  const bf16_t* src = qkv + row * ld + nq * 64;
  bf16_t* kd = kc + ((long)b * lmax + l) * w;
  bf16_t* vd = vc + ((long)b * lmax + l) * w;
  for (int c = threadIdx.x * 8; c < w; c += 256 * 8) {
    *reinterpret_cast<u32x4*>(kd + c) = *reinterpret_cast<const u32x4*>(src + c);
    *reinterpret_cast<u32x4*>(vd + c) = *reinterpret_cast<const u32x4*>(src + w + c);
  }
}

}  // namespace tcavt

using namespace tcavt;

#define TCAVT_TRY(call)            \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != TCAVT_OK) return rc_; \
  } while (0)

namespace {
struct Ev {
  void* const* ev;
  hipStream_t st;
  int layer;
  void rec(int slot) const {  // slot: 2 * stage + (0 start | 1 stop), stage = qkv, attn, o, gateup, down
    if (ev) (void)hipEventRecord(static_cast<hipEvent_t>(ev[layer * 10 + slot]), st);
  }
};
}  // namespace

extern "C" int tcavt_llama_stack_forward(const tcavt_llama_stack_args* a, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(a && a->layers && a->gamma_final && a->rope_cos && a->rope_sin && a->h && a->h16 && a->part && a->kv_len,
                  "llama_stack_forward: null pointer");
  TCAVT_CHECK_ARG(a->n_layers > 0 && a->B > 0 && a->L > 0 && a->nq > 0 && a->nkv > 0 && a->I > 0, "llama_stack_forward: bad shape");
  TCAVT_CHECK_ARG(a->H % 256 == 0, "llama_stack_forward: H = %d must be a multiple of 256 (H / 64 partial sums, added four at a time)", a->H);
  TCAVT_CHECK_ARG(is16(a->dtype16), "llama_stack_forward: dtype16 must be TCAVT_BF16 or TCAVT_F16");
  TCAVT_CHECK_ARG(a->att && a->act && (a->out_f32 || a->out16), "llama_stack_forward: null workspace / no output");
  TCAVT_CHECK_ARG(a->lora_dropout_p >= 0.f && a->lora_dropout_p < 1.f, "llama_stack_forward: lora_dropout_p must be in [0, 1)");
  TCAVT_CHECK_ARG((a->k_cache == nullptr) == (a->v_cache == nullptr) && (!a->k_cache || a->kv_lmax >= a->L),
                  "llama_stack_forward: k_cache / v_cache go together, kv_lmax >= L");
  const int M = a->B * a->L, H = a->H, I = a->I, nq = a->nq, nkv = a->nkv;
  const int nqkv = (nq + 2 * nkv) * 64, dt = a->dtype16;
  // partial sums of squares per row: what the producer of each fused norm's input writes (embedding kernel and down_proj
  // for the input norm, o_proj for the post-attention norm); tiny shapes (M <= 32) run the skinny GEMM form, which writes
  // one per 16 columns instead of one per 64
  const int np_in = norm_out_npart(M, H, I), np_post = norm_out_npart(M, H, nq * 64);
  TCAVT_CHECK_ARG(a->npart_in == np_in, "llama_stack_forward: npart_in = %d, but h16 / part must carry tcavt_norm_npart(M, H, I) = %d partials per row", a->npart_in, np_in);
  const float scale = 0.125f;  // 1 / sqrt(head_dim 64)
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* h = a->h;
  for (int li = 0; li < a->n_layers; ++li) {
    const tcavt_llama_layer& w = a->layers[li];
    TCAVT_CHECK_ARG(w.w_qkv && w.w_o && w.w_gu && w.w_d && (w.a_cat == nullptr) == (w.b_ext == nullptr),
                    "llama_stack_forward: layer %d: null weight", li);
    const bool tape = w.tape_h_mid != nullptr;
    if (tape) TCAVT_CHECK_ARG(w.tape_h_out && w.tape_qkv && w.tape_gu && (!w.a_cat || w.tape_t), "llama_stack_forward: layer %d: incomplete tape", li);
    void* qkv = tape ? w.tape_qkv : a->qkv;
    void* t = tape && w.a_cat ? w.tape_t : a->t;
    float* h_mid = tape ? w.tape_h_mid : h;
    float* h_out = tape ? w.tape_h_out : h;
    TCAVT_CHECK_ARG(qkv && (!w.a_cat || t), "llama_stack_forward: qkv / t workspace missing");
    const Ev ev{a->events, st, li};
    // ---- LoRA down-projection: t = (alpha / r) * dropout(x16) . (A * gamma)^T, un-normalised (the row scale is applied
    // to the whole q|k|v accumulator, the adapter update included)
    if (w.a_cat) {
      tcavt_gemm_args g = {};
      g.W = w.a_cat; g.ldw = H; g.C = t; g.ldc = 64; g.M = M; g.K = H;
      g.out_dtype = dt; g.in_dtype = dt; g.tile = 128; g.acc_scale = a->lora_scale;
      if (a->lora_dropout_p > 0.f) {  // PEFT: one lora_dropout module per adapted Linear -> two masks, q_proj then v_proj
        TCAVT_CHECK_ARG(a->xq && a->xv, "llama_stack_forward: xq / xv workspaces are needed with lora_dropout_p > 0");
        const uint32_t site = a->lora_first_site + 2u * (uint32_t)li;
        TCAVT_TRY(tcavt_dropout(a->h16, a->xq, (int64_t)M * H, dt, a->lora_dropout_p, a->dropout_seed, site, nullptr, stream));
        TCAVT_TRY(tcavt_dropout(a->h16, a->xv, (int64_t)M * H, dt, a->lora_dropout_p, a->dropout_seed, site + 1, nullptr, stream));
        g.N = 16;
        g.A = a->xq; g.lda = H;
        TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
        g.A = a->xv;
        g.W = static_cast<const bf16_t*>(w.a_cat) + (size_t)16 * H;
        g.C = static_cast<bf16_t*>(t) + 16;
        TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
      } else {
        g.A = a->h16; g.lda = H; g.N = 64;
        TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
      }
    }
    // ---- q|k|v = rs * (x16 . (W_qkv * gamma1)^T + t . B_ext^T), RoPE on q and k
    {
      tcavt_gemm_args g = {};
      g.A = a->h16; g.lda = H; g.W = w.w_qkv; g.ldw = H; g.C = qkv; g.ldc = nqkv;
      g.M = M; g.N = nqkv; g.K = H; g.out_dtype = dt; g.in_dtype = dt; g.tile = a->gemm_tile;
      if (w.a_cat) { g.A2 = t; g.lda2 = 64; g.W2 = w.b_ext; g.ldw2 = 64; g.K2 = 64; }
      g.epilogue = TCAVT_EPI_ROPE | TCAVT_EPI_ROWSCALE;
      g.rope_cos = a->rope_cos; g.rope_sin = a->rope_sin; g.rope_L = a->L; g.rope_cols = (nq + nkv) * 64;
      g.rowscale_part = a->part; g.rowscale_npart = np_in; g.rowscale_h = H; g.rowscale_eps = a->rms_eps;
      ev.rec(0);
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
      ev.rec(1);
    }
    if (a->k_cache) {
      const size_t per_layer = (size_t)a->B * a->kv_lmax * nkv * 64;
      hipLaunchKernelGGL(kv_store_kernel, dim3((unsigned)M), dim3(256), 0, st, static_cast<const bf16_t*>(qkv),
                         static_cast<bf16_t*>(a->k_cache) + li * per_layer, static_cast<bf16_t*>(a->v_cache) + li * per_layer,
                         a->L, a->kv_lmax, nq, nkv);
      TCAVT_CHECK_LAUNCH("kv_store");
    }
    // ---- causal grouped-query attention
    ev.rec(2);
    TCAVT_TRY(tcavt_attn_causal_gqa(qkv, a->att, a->kv_len, a->B, a->L, nq, nkv, scale, dt, stream));
    ev.rec(3);
    // ---- h_mid = h + att . W_o^T; 16-bit copy + partial sums of squares for the post-attention norm
    {
      tcavt_gemm_args g = {};
      g.A = a->att; g.lda = nq * 64; g.W = w.w_o; g.ldw = nq * 64; g.C = h_mid; g.ldc = H;
      g.M = M; g.N = H; g.K = nq * 64; g.out_dtype = TCAVT_F32; g.in_dtype = dt; g.tile = a->gemm_tile;
      g.residual = h; g.ldr = H; g.epilogue = TCAVT_EPI_RESIDUAL | TCAVT_EPI_NORM_OUT;
      g.norm_h16 = a->h16; g.norm_part = a->part;
      ev.rec(4);
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
      ev.rec(5);
    }
    // ---- act = silu(rs * gate) * (rs * up),  gate|up = x16 . (W_gu * gamma2)^T
    {
      tcavt_gemm_args g = {};
      g.A = a->h16; g.lda = H; g.W = w.w_gu; g.ldw = H; g.C = a->act; g.ldc = I;
      g.M = M; g.N = 2 * I; g.K = H; g.out_dtype = dt; g.in_dtype = dt; g.tile = a->gemm_tile;
      g.epilogue = TCAVT_EPI_SILU_MUL | TCAVT_EPI_ROWSCALE;
      g.rowscale_part = a->part; g.rowscale_npart = np_post; g.rowscale_h = H; g.rowscale_eps = a->rms_eps;
      if (tape) { g.silu_preact = w.tape_gu; g.ld_preact = 2 * I; }
      ev.rec(6);
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
      ev.rec(7);
    }
    // ---- h_out = h_mid + act . W_d^T; again the next norm's inputs
    {
      tcavt_gemm_args g = {};
      g.A = a->act; g.lda = I; g.W = w.w_d; g.ldw = I; g.C = h_out; g.ldc = H;
      g.M = M; g.N = H; g.K = I; g.out_dtype = TCAVT_F32; g.in_dtype = dt; g.tile = a->gemm_tile;
      g.residual = h_mid; g.ldr = H; g.epilogue = TCAVT_EPI_RESIDUAL | TCAVT_EPI_NORM_OUT;
      g.norm_h16 = a->h16; g.norm_part = a->part;
      ev.rec(8);
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
      ev.rec(9);
    }
    h = h_out;
  }
  // final RMSNorm: its fp32 result is hidden_states[-1] (scripts/train.py:553), the 16-bit copy feeds the head's K / V projections
  return tcavt_rmsnorm(h, a->gamma_final, a->rms_eps, a->out16, a->out_f32, M, H, nullptr, 0.f, 0, 0, dt, stream);
}

extern "C" int tcavt_norm_npart(int M, int N, int K) { return norm_out_npart(M, N, K); }

extern "C" int tcavt_events_create(void** events, int n) {
  TCAVT_CHECK_ARG(events && n > 0, "events_create: bad args");
  for (int i = 0; i < n; ++i) {
    hipEvent_t e;
    const hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) {
      set_error("events_create: hipEventCreate failed: %s", hipGetErrorString(rc));
      return TCAVT_ERR_HIP;
    }
    events[i] = e;
  }
  return TCAVT_OK;
}

extern "C" int tcavt_events_destroy(void** events, int n) {
  TCAVT_CHECK_ARG(events && n > 0, "events_destroy: bad args");
  for (int i = 0; i < n; ++i)
    if (events[i]) (void)hipEventDestroy(static_cast<hipEvent_t>(events[i]));
  return TCAVT_OK;
}

extern "C" int tcavt_event_elapsed_ms(void* start, void* stop, float* ms) {
  TCAVT_CHECK_ARG(start && stop && ms, "event_elapsed_ms: null pointer");
  const hipError_t rc = hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop));
  if (rc != hipSuccess) {
    set_error("event_elapsed_ms: %s", hipGetErrorString(rc));
    return TCAVT_ERR_HIP;
  }
  return TCAVT_OK;
}
