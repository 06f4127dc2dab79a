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
#include "philox.hpp"

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

// ---------------------------------------------------------------------------
// LoRA down-projection of both adapters in one pass over the 16-bit residual-stream copy:
//     t[m][0:16]  = s * (mask_q o x[m]) . A_q'^T        t[m][16:32] = s * (mask_v o x[m]) . A_v'^T
// (A' = the packed a_cat rows, gain folded in; PEFT: lora_B(lora_A(lora_dropout(x))) with one dropout module per adapted
// Linear -> two independent Philox sites in train mode, no mask in eval).  x is read ONCE (32 MB at B*L = 8192): the two
// dropped copies that tcavt_dropout would write and two skinny GEMMs would read back never exist.  The masked operand is
// rounded exactly as tcavt_dropout rounds it (to16(x * keep / (1 - p))).  One wave per 16 tokens, K in 32-deep MFMA steps.
// ---------------------------------------------------------------------------
template <bool F16>
__device__ __forceinline__ f32x4 mfma16s(const u32x4& a, const u32x4& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool F16>
__device__ __forceinline__ u32x4 masked8(const u32x4& x, const DropoutP& d, unsigned long long quad) {
  // eight consecutive elements starting at an even quad = one octet = one generator call
  float s[8];
  dropout_oct(d, quad >> 1, s);
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = pack16x2<F16>(from16_lo<F16>(x[i]) * s[2 * i], from16_hi<F16>(x[i]) * s[2 * i + 1]);
  return o;
}

template <bool F16>
__global__ __launch_bounds__(256) void lora_down_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ a_cat,
                                                        bf16_t* __restrict__ t, int M, int H, float scale, DropoutP dq,
                                                        DropoutP dv) {
  // one workgroup per 16 tokens; its four waves split K (the mask generation is VALU work -- ~120 instructions per four
  // elements and site -- so it is spread over as many waves as possible: M / 16 x 4), partial sums meet in LDS in wave order
  __shared__ f32x4 red[4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 16;
  const int r16 = lane & 15, kq = lane >> 4;
  const long m = min(m0 + r16, M - 1);
  const int kper = H / 4;  // H % 256 == 0: whole 64-deep steps per wave
  const bf16_t* xp = x + m * H + wave * kper + kq * 8;
  const bf16_t* aq = a_cat + (long)r16 * H + wave * kper + kq * 8;
  const bf16_t* av = a_cat + (long)(16 + r16) * H + wave * kper + kq * 8;
  f32x4 cq = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
  const bool drop = dq.p > 0.f;
  for (int k = 0; k < kper; k += 64) {  // two MFMA steps per iteration: 2 x 3 loads in flight
    const u32x4 x0 = *reinterpret_cast<const u32x4*>(xp + k), x1 = *reinterpret_cast<const u32x4*>(xp + k + 32);
    const u32x4 q0 = *reinterpret_cast<const u32x4*>(aq + k), q1 = *reinterpret_cast<const u32x4*>(aq + k + 32);
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(av + k), v1 = *reinterpret_cast<const u32x4*>(av + k + 32);
    if (drop) {
      const unsigned long long e0 =
          ((unsigned long long)m * (unsigned long long)H + (unsigned long long)(wave * kper + k + kq * 8)) >> 2;
      cq = mfma16s<F16>(q0, masked8<F16>(x0, dq, e0), cq);
      cv = mfma16s<F16>(v0, masked8<F16>(x0, dv, e0), cv);
      cq = mfma16s<F16>(q1, masked8<F16>(x1, dq, e0 + 8), cq);
      cv = mfma16s<F16>(v1, masked8<F16>(x1, dv, e0 + 8), cv);
    } else {
      cq = mfma16s<F16>(q0, x0, cq);
      cv = mfma16s<F16>(v0, x0, cv);
      cq = mfma16s<F16>(q1, x1, cq);
      cv = mfma16s<F16>(v1, x1, cv);
    }
  }
  red[wave][0][lane] = cq;
  red[wave][1][lane] = cv;
  __syncthreads();
  if (wave != 0) return;
  cq = (red[0][0][lane] + red[1][0][lane]) + (red[2][0][lane] + red[3][0][lane]);
  cv = (red[0][1][lane] + red[1][1][lane]) + (red[2][1][lane] + red[3][1][lane]);
  // lane: features 4 kq .. + 3 of token m0 + r16 in each 16-column group
  if (m0 + r16 < M) {
    bf16_t* row = t + (long)(m0 + r16) * 64 + 4 * kq;
    *reinterpret_cast<u32x2*>(row) = u32x2{pack16x2<F16>(cq[0] * scale, cq[1] * scale), pack16x2<F16>(cq[2] * scale, cq[3] * scale)};
    *reinterpret_cast<u32x2*>(row + 16) = u32x2{pack16x2<F16>(cv[0] * scale, cv[1] * scale), pack16x2<F16>(cv[2] * scale, cv[3] * scale)};
  }
}

// ---------------------------------------------------------------------------
// Backward of the adapters' down-projections w.r.t. their (dropped) input, both adapters in one pass:
//     g_x[m][n] = mask_q[m][n] * (g_t[m][0:16] . A_q[:, n]) + mask_v[m][n] * (g_t[m][16:32] . A_v[:, n])
// (g_t = dL/dt of tcavt_lora_down's output; one dropout site per adapted Linear, as in the forward).  Replaces two skinny
// GEMMs, two mask kernels and their 32 MB round trips per layer.  A wave owns 16 tokens and walks the features in pairs of
// 16-column tiles; per tile and adapter one MFMA (A operand = rows of A_q^T / A_v^T [H][64] with the other adapter's columns
// zero, B operand = the tokens' g_t rows, K = 32), issued "swapped" so that a lane ends with four consecutive features of
// its token; v_permlane16_swap pairs two tiles into eight consecutive features per lane = one mask octet = one generator
// call per site, and one 16-byte store.
// ---------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void lora_dgrad_kernel(const bf16_t* __restrict__ g_t, const bf16_t* __restrict__ aqT,
                                                         const bf16_t* __restrict__ avT, bf16_t* __restrict__ out, int M, int H,
                                                         DropoutP dq, DropoutP dv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 16;
  const int r16 = lane & 15, kq = lane >> 4;
  const long m = min(m0 + r16, M - 1);
  const u32x4 gt = *reinterpret_cast<const u32x4*>(g_t + m * 64 + kq * 8);
  const bool drop = dq.p > 0.f;
  const int foff = (kq & 1) ? 16 + 4 * (kq - 1) : 4 * kq;  // this lane's eight features inside a pair of tiles (after the swap)
  for (int n0 = wave * 32; n0 < H; n0 += 128) {
    f32x4 cq[2], cv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long arow = (long)(n0 + 16 * h + r16) * 64 + kq * 8;
      const u32x4 aq = *reinterpret_cast<const u32x4*>(aqT + arow), av = *reinterpret_cast<const u32x4*>(avT + arow);
      cq[h] = mfma16s<F16>(aq, gt, f32x4{0.f, 0.f, 0.f, 0.f});
      cv[h] = mfma16s<F16>(av, gt, f32x4{0.f, 0.f, 0.f, 0.f});
    }
    float vq[8], vv[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // (cq[0][e], cq[1][e]) -> elements e and 4 + e of the lane's eight consecutive features
      unsigned int a = __float_as_uint(cq[0][e]), b = __float_as_uint(cq[1][e]);
      auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
      vq[e] = __uint_as_float(r[0]);
      vq[4 + e] = __uint_as_float(r[1]);
      a = __float_as_uint(cv[0][e]);
      b = __float_as_uint(cv[1][e]);
      r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
      vv[e] = __uint_as_float(r[0]);
      vv[4 + e] = __uint_as_float(r[1]);
    }
    const int f0 = n0 + foff;
    float o[8];
    if (drop) {
      const unsigned long long oct = ((unsigned long long)m * (unsigned long long)H + (unsigned long long)f0) >> 3;
      float sq[8], sv[8];
      dropout_oct(dq, oct, sq);
      dropout_oct(dv, oct, sv);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = vq[k] * sq[k] + vv[k] * sv[k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = vq[k] + vv[k];
    }
    if (m0 + r16 < M)
      *reinterpret_cast<u32x4*>(out + m * H + f0) = u32x4{pack16x2<F16>(o[0], o[1]), pack16x2<F16>(o[2], o[3]),
                                                          pack16x2<F16>(o[4], o[5]), pack16x2<F16>(o[6], o[7])};
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
  TCAVT_CHECK_ARG(a && a->layers && a->gamma_final && a->rope_cos && a->rope_sin && a->h16 && a->part && a->kv_len,
                  "llama_stack_forward: null pointer");
  const bool stream16 = a->h == nullptr;  // the residual stream is the 16-bit h16 itself (no fp32 copy anywhere)
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
  // scaled 16-bit image of the residual stream (stream_scale: include/tcavt.h): h16 / part arrive at that scale, the residual
  // epilogues keep it, the fused norms see eps * s^2 (RMSNorm of s x with that eps IS RMSNorm of x); the adapters' t = lora_scale *
  // (s x) . A^T is at the stream's scale as well, so the whole q|k|v accumulator is s (x W^T + t B^T) and the row scale undoes s
  TCAVT_CHECK_ARG(a->stream_scale >= 0.f && a->stream_scale <= 1.f, "llama_stack_forward: stream_scale must be in (0, 1] (0 means 1)");
  const float ss_ = a->stream_scale == 0.f ? 1.f : a->stream_scale;
  const float eps_s = a->rms_eps * ss_ * ss_;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* h = a->h;
  const void* x16 = a->h16;  // the 16-bit stream (or 16-bit copy of the fp32 stream) the current layer reads
  for (int li = 0; li < a->n_layers; ++li) {
    const tcavt_llama_layer& w = a->layers[li];
    TCAVT_CHECK_ARG(w.w_qkv && w.w_o && w.w_gu && w.w_d && (w.a_cat == nullptr) == (w.b_ext == nullptr),
                    "llama_stack_forward: layer %d: null weight", li);
    const bool tape = w.tape_h_mid != nullptr;
    if (tape) TCAVT_CHECK_ARG(ss_ == 1.f, "llama_stack_forward: a tape keeps the streams at scale 1 (stream_scale must be 0 or 1)");
    if (tape) TCAVT_CHECK_ARG(w.tape_h_out && w.tape_qkv && w.tape_gu && (!w.a_cat || w.tape_t), "llama_stack_forward: layer %d: incomplete tape", li);
    void* qkv = tape ? w.tape_qkv : a->qkv;
    void* t = tape && w.a_cat ? w.tape_t : a->t;
    float* h_mid = tape ? w.tape_h_mid : h;   // (stream16 + tape: these two name 16-bit buffers, see x_mid / x_out)
    float* h_out = tape ? w.tape_h_out : h;
    // 16-bit residual stream: where this layer reads it (x16) and writes it -- in place without a tape, into the layer's own
    // buffers with one (the backward then finds every layer's stream as the forward computed it)
    void* x_mid = (stream16 && tape) ? static_cast<void*>(w.tape_h_mid) : a->h16;
    void* x_out = (stream16 && tape) ? static_cast<void*>(w.tape_h_out) : a->h16;
    TCAVT_CHECK_ARG(qkv && (!w.a_cat || t), "llama_stack_forward: qkv / t workspace missing");
    const Ev ev{a->events, st, li};
    if (tape && w.tape_part) {  // the input stream's partial sums of squares, before the o_proj epilogue reuses `part`
      const hipError_t rc = hipMemcpyAsync(w.tape_part, a->part, (size_t)M * np_in * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (rc != hipSuccess) {
        set_error("llama_stack_forward: layer %d: copy of the partial sums failed: %s", li, hipGetErrorString(rc));
        return TCAVT_ERR_HIP;
      }
    }
    // ---- LoRA down-projection: t = (alpha / r) * dropout(x16) . (A * gamma)^T, un-normalised (the row scale is applied
    // to the whole q|k|v accumulator, the adapter update included); one fused kernel for both adapters and their masks
    if (w.a_cat) {
      const uint32_t site = a->lora_first_site + 2u * (uint32_t)li;
      TCAVT_TRY(tcavt_lora_down(x16, w.a_cat, t, M, H, a->lora_scale, a->lora_dropout_p, a->dropout_seed, site, site + 1, dt,
                                stream));
    }
    // ---- q|k|v = rs * (x16 . (W_qkv * gamma1)^T + t . B_ext^T), RoPE on q and k
    {
      tcavt_gemm_args g = {};
      g.A = x16; g.lda = H; g.W = w.w_qkv; g.ldw = H; g.C = qkv; g.ldc = nqkv;
      g.M = M; g.N = nqkv; g.K = H; g.out_dtype = dt; g.in_dtype = dt; g.tile = a->gemm_tile;
      if (w.a_cat) { g.A2 = t; g.lda2 = 64; g.W2 = w.b_ext; g.ldw2 = 64; g.K2 = 64; }
      g.epilogue = TCAVT_EPI_ROPE | TCAVT_EPI_ROWSCALE;
      g.rope_cos = a->rope_cos; g.rope_sin = a->rope_sin; g.rope_L = a->L; g.rope_cols = (nq + nkv) * 64;
      g.rowscale_part = a->part; g.rowscale_npart = np_in; g.rowscale_h = H; g.rowscale_eps = eps_s;
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
    TCAVT_CHECK_ARG((w.tape_att != nullptr) == (w.tape_lse != nullptr), "llama_stack_forward: layer %d: tape_att and tape_lse come together", li);
    void* att = tape && w.tape_att ? w.tape_att : a->att;
    TCAVT_TRY(tcavt_attn_causal_gqa_lse(qkv, att, tape ? w.tape_lse : nullptr, a->kv_len, a->B, a->L, nq, nkv, scale, dt, stream));
    ev.rec(3);
    // ---- h_mid = h + att . W_o^T; 16-bit copy + partial sums of squares for the post-attention norm
    {
      tcavt_gemm_args g = {};
      g.A = att; g.lda = nq * 64; g.W = w.w_o; g.ldw = nq * 64; g.C = h_mid; g.ldc = H;
      g.M = M; g.N = H; g.K = nq * 64; g.out_dtype = TCAVT_F32; g.in_dtype = dt; g.tile = a->gemm_tile;
      g.residual = h; g.ldr = H; g.epilogue = TCAVT_EPI_RESIDUAL | TCAVT_EPI_NORM_OUT;  // (stream16: C = residual = NULL)
      if (stream16) { g.C = nullptr; g.residual = nullptr; g.norm_h16 = x_mid; g.norm_res16 = x16; }
      else g.norm_h16 = a->h16;
      g.norm_part = a->part; g.norm_scale = ss_;
      g.splitk_ws = a->splitk_ws; g.splitk_ws_bytes = a->splitk_ws_bytes;
      g.nonfinite_flag = a->nonfinite_flag; g.nonfinite_tag = 1 + 2 * li;
      ev.rec(4);
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
      ev.rec(5);
    }
    // ---- act = silu(rs * gate) * (rs * up),  gate|up = x16 . (W_gu * gamma2)^T
    {
      tcavt_gemm_args g = {};
      g.A = stream16 ? x_mid : a->h16; g.lda = H; g.W = w.w_gu; g.ldw = H; g.C = a->act; g.ldc = I;
      g.M = M; g.N = 2 * I; g.K = H; g.out_dtype = dt; g.in_dtype = dt; g.tile = a->gemm_tile;
      g.epilogue = TCAVT_EPI_SILU_MUL | TCAVT_EPI_ROWSCALE;
      g.rowscale_part = a->part; g.rowscale_npart = np_post; g.rowscale_h = H; g.rowscale_eps = eps_s;
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
      if (stream16) { g.C = nullptr; g.residual = nullptr; g.norm_h16 = x_out; g.norm_res16 = x_mid; }
      else g.norm_h16 = a->h16;
      g.norm_part = a->part; g.norm_scale = ss_;
      g.splitk_ws = a->splitk_ws; g.splitk_ws_bytes = a->splitk_ws_bytes;
      g.nonfinite_flag = a->nonfinite_flag; g.nonfinite_tag = 2 + 2 * li;
      ev.rec(8);
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
      ev.rec(9);
    }
    h = h_out;
    x16 = x_out;
  }
  // final RMSNorm: its fp32 result is hidden_states[-1] (scripts/train.py:553), the 16-bit copy feeds the head's K / V projections
  if (stream16) return tcavt_rmsnorm16(x16, a->gamma_final, eps_s, a->out16, a->out_f32, M, H, dt, stream);  // (s x, eps s^2: the norm of x)
  return tcavt_rmsnorm(h, a->gamma_final, a->rms_eps, a->out16, a->out_f32, M, H, nullptr, 0.f, 0, 0, dt, stream);
}

// RMSNorm of 16-bit rows (the 16-bit residual stream's final norm): one wave per row, the row stays in registers when
// H <= 4096 (u32x2 = 4 elements per lane and step), fp32 arithmetic as tcavt_rmsnorm.
namespace {
template <bool F16>
__global__ __launch_bounds__(256) void rmsnorm16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma, float eps,
                                                        bf16_t* __restrict__ out16, float* __restrict__ out_f32, int M, int H, int frag16) {
  // one wave per row, 16-byte accesses (8 elements per lane and step: 512 columns per wave-instruction), the row in registers when
  // H <= 4096, every load of the row requested before the first is used (the first version moved 8 bytes per lane and step and ran
  // at 0.9 TB/s: 71 us for the decoder's final norm at M = 8192, on the decoder stream's critical path)
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  // (frag16: the <= 32 rows of a decode step in the skinny GEMM's operand order, in and out: same lanes, same columns, same sums)
  auto xat = [&](int c) { return frag16 ? x + frag_off((int)row, c, H, frag16) : x + row * H + c; };
  auto oat = [&](int c) { return frag16 ? out16 + frag_off((int)row, c, H, frag16) : out16 + row * H + c; };
  constexpr int NV = 8;  // 8 x 512 = 4096 columns in registers
  u32x4 w[NV];
  float ss = 0.f;
  const int nv = H >> 9, tail = H & 511;  // whole 512-column steps, then one 256-column step (H % 256 == 0)
  auto sq4 = [&](const u32x4& v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = from16_lo<F16>(v[e]), b = from16_hi<F16>(v[e]);
      ss += a * a + b * b;
    }
  };
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i < nv) w[i] = *reinterpret_cast<const u32x4*>(xat(i * 512 + lane * 8));
  u32x2 wt = {0u, 0u};
  if (tail) wt = *reinterpret_cast<const u32x2*>(xat(nv * 512 + lane * 4));
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i < nv) sq4(w[i]);
  for (int i = NV; i < nv; ++i) sq4(*reinterpret_cast<const u32x4*>(xat(i * 512 + lane * 8)));
  if (tail) {
    const float a = from16_lo<F16>(wt[0]), b = from16_hi<F16>(wt[0]), c = from16_lo<F16>(wt[1]), d = from16_hi<F16>(wt[1]);
    ss += a * a + b * b + c * c + d * d;
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)H + eps);
  auto emit8 = [&](int c, const u32x4& v) {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4);
    float y[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y[2 * e] = from16_lo<F16>(v[e]) * rs * (e < 2 ? g0[2 * e] : g1[2 * e - 4]);
      y[2 * e + 1] = from16_hi<F16>(v[e]) * rs * (e < 2 ? g0[2 * e + 1] : g1[2 * e - 3]);
    }
    if (out16)
      *reinterpret_cast<u32x4*>(oat(c)) = u32x4{pack16x2<F16>(y[0], y[1]), pack16x2<F16>(y[2], y[3]),
                                                             pack16x2<F16>(y[4], y[5]), pack16x2<F16>(y[6], y[7])};
    if (out_f32) {
      *reinterpret_cast<f32x4*>(out_f32 + row * H + c) = f32x4{y[0], y[1], y[2], y[3]};
      *reinterpret_cast<f32x4*>(out_f32 + row * H + c + 4) = f32x4{y[4], y[5], y[6], y[7]};
    }
  };
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i < nv) emit8(i * 512 + lane * 8, w[i]);
  for (int i = NV; i < nv; ++i) emit8(i * 512 + lane * 8, *reinterpret_cast<const u32x4*>(xat(i * 512 + lane * 8)));
  if (tail) {
    const int c = nv * 512 + lane * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
    f32x4 y;
    y[0] = from16_lo<F16>(wt[0]) * rs * g[0];
    y[1] = from16_hi<F16>(wt[0]) * rs * g[1];
    y[2] = from16_lo<F16>(wt[1]) * rs * g[2];
    y[3] = from16_hi<F16>(wt[1]) * rs * g[3];
    if (out16) *reinterpret_cast<u32x2*>(oat(c)) = u32x2{pack16x2<F16>(y[0], y[1]), pack16x2<F16>(y[2], y[3])};
    if (out_f32) *reinterpret_cast<f32x4*>(out_f32 + row * H + c) = y;
  }
}
}  // namespace

extern "C" int tcavt_rmsnorm16(const void* x16, const float* gamma, float eps, void* out16, float* out_f32, int M, int H,
                               int dtype16, tcavt_stream_t stream) {
  return tcavt::rmsnorm16_impl(x16, gamma, eps, out16, out_f32, M, H, dtype16, 0, stream);
}

int tcavt::rmsnorm16_impl(const void* x16, const float* gamma, float eps, void* out16, float* out_f32, int M, int H, int dtype16,
                          int frag16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(!frag16 || ((frag16 == 1 || frag16 == 2) && M <= (frag16 == 2 ? 8 : 32) && !out_f32),
                  "rmsnorm16: fragment-major rows: mode 1 (at most 32 rows) or 2 (at most 8), 16-bit output only");
  TCAVT_CHECK_ARG(x16 && gamma && (out16 || out_f32) && M > 0 && H > 0 && H % 256 == 0 && is16(dtype16),
                  "rmsnorm16: bad args (H %% 256 == 0)");
  TCAVT_CHECK_ARG(aligned16(x16) && aligned16(gamma), "rmsnorm16: unaligned input");
  const dim3 grid((unsigned)((M + 3) / 4)), block(256);
  if (dtype16 == TCAVT_F16)
    hipLaunchKernelGGL(rmsnorm16_kernel<true>, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(x16), gamma,
                       eps, static_cast<bf16_t*>(out16), out_f32, M, H, frag16);
  else
    hipLaunchKernelGGL(rmsnorm16_kernel<false>, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(x16), gamma,
                       eps, static_cast<bf16_t*>(out16), out_f32, M, H, frag16);
  TCAVT_CHECK_LAUNCH("rmsnorm16");
  return TCAVT_OK;
}

extern "C" int tcavt_norm_npart(int M, int N, int K) { return norm_out_npart(M, N, K); }

extern "C" int tcavt_lora_down(const void* x16, const void* a_cat, void* t, int M, int H, float scale, float dropout_p,
                               uint64_t dropout_seed, uint32_t site_q, uint32_t site_v, int dtype16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(x16 && a_cat && t && M > 0 && H > 0 && H % 256 == 0 && is16(dtype16) && dropout_p >= 0.f && dropout_p < 1.f,
                  "lora_down: bad args (H %% 256 == 0, 0 <= dropout_p < 1)");
  TCAVT_CHECK_ARG(aligned16(x16) && aligned16(a_cat) && aligned16(t), "lora_down: 16-byte alignment required");
  const DropoutP dq = make_dropout(dropout_p, dropout_seed, site_q), dv = make_dropout(dropout_p, dropout_seed, site_v);
  const dim3 grid((unsigned)((M + 15) / 16)), block(256);
  if (dtype16 == TCAVT_F16)
    hipLaunchKernelGGL(lora_down_kernel<true>, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(x16),
                       static_cast<const bf16_t*>(a_cat), static_cast<bf16_t*>(t), M, H, scale, dq, dv);
  else
    hipLaunchKernelGGL(lora_down_kernel<false>, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(x16),
                       static_cast<const bf16_t*>(a_cat), static_cast<bf16_t*>(t), M, H, scale, dq, dv);
  TCAVT_CHECK_LAUNCH("lora_down");
  return TCAVT_OK;
}

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

extern "C" int tcavt_lora_dgrad(const void* g_t, const void* aqT, const void* avT, void* out, int M, int H, float dropout_p,
                                uint64_t dropout_seed, uint32_t site_q, uint32_t site_v, int dtype16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(g_t && aqT && avT && out && M > 0 && H > 0 && H % 128 == 0 && is16(dtype16) && dropout_p >= 0.f && dropout_p < 1.f,
                  "lora_dgrad: bad args (H %% 128 == 0, 0 <= dropout_p < 1)");
  TCAVT_CHECK_ARG(aligned16(g_t) && aligned16(aqT) && aligned16(avT) && aligned16(out), "lora_dgrad: 16-byte alignment required");
  const DropoutP dq = make_dropout(dropout_p, dropout_seed, site_q), dv = make_dropout(dropout_p, dropout_seed, site_v);
  const dim3 grid((unsigned)((M + 15) / 16)), block(256);
  auto kfn = dtype16 == TCAVT_F16 ? lora_dgrad_kernel<true> : lora_dgrad_kernel<false>;
  hipLaunchKernelGGL(kfn, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(g_t),
                     static_cast<const bf16_t*>(aqT), static_cast<const bf16_t*>(avT), static_cast<bf16_t*>(out), M, H, dq, dv);
  TCAVT_CHECK_LAUNCH("lora_dgrad");
  return TCAVT_OK;
}
