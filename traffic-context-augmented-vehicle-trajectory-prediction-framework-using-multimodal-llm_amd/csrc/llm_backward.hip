// Backward of the decoder layers for the LoRA-trainable variant (SURVEY.md 8f.1;
// modify_scripts/modify_train.py:512-528 trains lora_A / lora_B of q_proj, v_proj while every base weight stays frozen):
// the activation gradient walks back through all layers -- dgrad GEMMs against transposed copies of the frozen
// weights (gemm_bf16.hip), and the pointwise / row / attention pieces here.
// Attention backward: attn_bwd_scores_kernel (query-major: row statistics, dQ) + attn_bwd_dkv_kernel (key-major: dK, dV),
// both on the matrix cores, nothing of S / P / dS stored; the GEMM-composed forms (llm_backward.attn_bwd_composed) and the
// scalar-FMA attn_causal_gqa_bwd_kernel are cross-checks, not product paths.
#include "common.hpp"
#include "philox.hpp"
#include <stdlib.h>

namespace tcavt {

__device__ __forceinline__ float sigmoid_f(float g) { return __builtin_amdgcn_rcpf(1.f + __expf(-g)); }

// one MFMA step on raw 16-byte fragments of the 16-bit type the backward runs in (F16: IEEE half = the forward's storage
// contract, gradients carried under a power-of-two scale, tcavt_grad_scale_pick; otherwise bf16, round 1's contract)
template <bool F16>
__device__ __forceinline__ f32x4 mfma16b(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ float half16(unsigned int w, int k) { return k ? from16_hi<F16>(w) : from16_lo<F16>(w); }

// gu: [M, 2I] bf16 in the interleaved layout of TCAVT_EPI_SILU_MUL (blocks of 16 gate columns, then the 16 up columns
// of the same features); g_act: [M, I] bf16 = dL/d(silu(gate) * up).  g_gu gets dL/dgate, dL/dup in the same layout.
template <bool F16>
__global__ __launch_bounds__(256) void silu_mul_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ g_act,
                                                           bf16_t* __restrict__ g_gu, long M, int I) {
  // one thread per block of 16 features: 32 B of gate, 32 B of up (adjacent), 32 B of g_act
  const long blk = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nblk = M * (I >> 4);
  if (blk >= nblk) return;
  const u32x4* src = reinterpret_cast<const u32x4*>(gu + blk * 32);
  const u32x4* dsrc = reinterpret_cast<const u32x4*>(g_act + blk * 16);
  u32x4 gv[2] = {src[0], src[1]}, uv[2] = {src[2], src[3]}, dv[2] = {dsrc[0], dsrc[1]};
  u32x4 og[2], ou[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float r[2][2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float g = half16<F16>(gv[h][e], k), u = half16<F16>(uv[h][e], k), d = half16<F16>(dv[h][e], k);
        const float sg = sigmoid_f(g);
        r[0][k] = d * u * sg * (1.f + g * (1.f - sg));
        r[1][k] = d * g * sg;
      }
      og[h][e] = pack16x2<F16>(r[0][0], r[0][1]);
      ou[h][e] = pack16x2<F16>(r[1][0], r[1][1]);
    }
  u32x4* dst = reinterpret_cast<u32x4*>(g_gu + blk * 32);
  dst[0] = og[0]; dst[1] = og[1]; dst[2] = ou[0]; dst[3] = ou[1];
}

// RMSNorm backward, one wave per row:  y = x * r * gamma,  r = rsqrt(mean(x^2) + eps)
//   gx = r * (gy*gamma) - x * r^3 * mean(gy*gamma*x)
// gy = gy_a (+ gy_b): the LoRA branch hands in its own gradient of the same normed row.  accumulate: gx += .
// GF16 / OF16: 16-bit type of the incoming gradients / of the outgoing 16-bit copy.  gy_scale (optional, device scalar):
// the incoming gradients are multiplied by it -- where the backward enters its power-of-two scale (tcavt_grad_scale_pick).
// XT: type of x -- 0 fp32, 1 fp16, 2 bf16 (the forward's 16-bit residual stream, kept per layer by the tape)
template <int XT>
__device__ __forceinline__ void load_x8(const void* xr, int c, f32x4& x0, f32x4& x1) {
  if constexpr (XT == 0) {
    const float* f = static_cast<const float*>(xr);
    x0 = *reinterpret_cast<const f32x4*>(f + c);
    x1 = *reinterpret_cast<const f32x4*>(f + c + 4);
  } else {
    const u32x4 v = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(xr) + c);
    x0 = f32x4{from16_lo<XT == 1>(v[0]), from16_hi<XT == 1>(v[0]), from16_lo<XT == 1>(v[1]), from16_hi<XT == 1>(v[1])};
    x1 = f32x4{from16_lo<XT == 1>(v[2]), from16_hi<XT == 1>(v[2]), from16_lo<XT == 1>(v[3]), from16_hi<XT == 1>(v[3])};
  }
}
template <bool GF16, bool OF16, int XT>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                          const bf16_t* __restrict__ gy_a, const bf16_t* __restrict__ gy_b,
                                                          float eps, float* __restrict__ gx, bf16_t* __restrict__ gx_bf16,
                                                          int accumulate, int M, int H, const float* __restrict__ gy_scale) {
  const float gsc = gy_scale ? *gy_scale : 1.f;
  // H % 8 == 0: a lane handles 8 consecutive columns per step (2 x 16 B of x, 16 B of each gradient)
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const void* xr = XT == 0 ? static_cast<const void*>(static_cast<const float*>(x) + (long)row * H)
                           : static_cast<const void*>(static_cast<const bf16_t*>(x) + (long)row * H);
  const bf16_t* ga = gy_a + (long)row * H;
  const bf16_t* gb = gy_b ? gy_b + (long)row * H : nullptr;
  auto load_g = [&](int c, float (&g)[8]) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(ga + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[2 * e] = from16_lo<GF16>(a[e]);
      g[2 * e + 1] = from16_hi<GF16>(a[e]);
    }
    if (gb) {
      const u32x4 b = *reinterpret_cast<const u32x4*>(gb + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        g[2 * e] += from16_lo<GF16>(b[e]);
        g[2 * e + 1] += from16_hi<GF16>(b[e]);
      }
    }
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(gamma + c) * gsc, w1 = *reinterpret_cast<const f32x4*>(gamma + c + 4) * gsc;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[e] *= w0[e];
      g[4 + e] *= w1[e];
    }
  };
  float ss = 0.f, dot = 0.f;
  for (int c = lane * 8; c < H; c += 512) {
    float g[8];
    load_g(c, g);
    f32x4 x0, x1;
    load_x8<XT>(xr, c, x0, x1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ss = fmaf(x0[e], x0[e], ss);
      ss = fmaf(x1[e], x1[e], ss);
      dot = fmaf(g[e], x0[e], dot);
      dot = fmaf(g[4 + e], x1[e], dot);
    }
  }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float r = rsqrtf(ss / (float)H + eps);
  const float k = dot / (float)H * r * r * r;
  float* o = gx + (long)row * H;
  for (int c = lane * 8; c < H; c += 512) {
    float g[8];
    load_g(c, g);
    f32x4 x0, x1;
    load_x8<XT>(xr, c, x0, x1);
    f32x4 v0, v1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v0[e] = r * g[e] - x0[e] * k;
      v1[e] = r * g[4 + e] - x1[e] * k;
    }
    if (accumulate) {
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(o + c), p1 = *reinterpret_cast<const f32x4*>(o + c + 4);
      v0 += p0;
      v1 += p1;
    }
    *reinterpret_cast<f32x4*>(o + c) = v0;
    *reinterpret_cast<f32x4*>(o + c + 4) = v1;
    if (gx_bf16) {
      u32x4 b = {pack16x2<OF16>(v0[0], v0[1]), pack16x2<OF16>(v0[2], v0[3]), pack16x2<OF16>(v1[0], v1[1]), pack16x2<OF16>(v1[2], v1[3])};
      *reinterpret_cast<u32x4*>(gx_bf16 + (long)row * H + c) = b;
    }
  }
}

// fp32 [M, ncols] gradient of the rotated q | k | v  ->  bf16 gradient of the projections' outputs: the first
// rope_cols columns (q and k heads, head_dim 64 each) get the transposed rotation
//   g_t1 = g_o1 * cos + g_o2 * sin,   g_t2 = g_o2 * cos - g_o1 * sin      (forward: o1 = t1 cos - t2 sin, o2 = t2 cos + t1 sin)
template <bool F16>
__global__ __launch_bounds__(256) void rope_bwd_pack_kernel(const float* __restrict__ g32, bf16_t* __restrict__ out,
                                                            const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                            long M, int ncols, int rope_cols, int L) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (row, column pair c / c + 32 of a head)
  const int half = ncols / 2;
  if (idx >= M * half) return;
  const long row = idx / half;
  const int p = (int)(idx - row * half);
  const int head = p >> 5, w = p & 31;
  const int c1 = head * 64 + w, c2 = c1 + 32;
  const float a = g32[row * ncols + c1], b = g32[row * ncols + c2];
  float o1 = a, o2 = b;
  if (c1 < rope_cols) {
    const int pos = (int)(row % L);
    const float cs = cosT[pos * 32 + w], sn = sinT[pos * 32 + w];
    o1 = a * cs + b * sn;
    o2 = b * cs - a * sn;
  }
  out[row * ncols + c1] = to16<F16>(o1);
  out[row * ncols + c2] = to16<F16>(o2);
}

// ---------------------------------------------------------------------------
// (cross-check kernel) Causal GQA attention backward, head_dim 64.  One workgroup per (sample, query head, block of QB = 32 query rows).
//   P = softmax(scale * q K^T) over keys c < min(i + 1, kv_len[b]);   dP = dO V^T;   dS = scale * P * (dP - rowsum(P dP))
//   dQ_i = sum_c dS_ic K_c  (written);   dK_c += sum_i dS_ic q_i,  dV_c += sum_i P_ic dO_i  (float atomics: the four
//   query heads of a group and the query blocks all add into the same key rows).
// q, k, v are read from the forward's rotated q|k|v buffer [B*T, (nq + 2 nkv) * 64] (bf16), dO from [B*T, nq*64] (bf16);
// g32 is the fp32 gradient in the q|k|v layout, zeroed by the caller.
// ---------------------------------------------------------------------------
constexpr int AB_QB = 32, AB_HD = 64, AB_LDK = 66;  // K/V rows padded to 33 dwords (conflict-free down a column)

__global__ __launch_bounds__(256) void attn_causal_gqa_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                                  float* __restrict__ g32, const int* __restrict__ kv_len,
                                                                  int T, int nq, int nkv, float scale, int nqb) {
  extern __shared__ float sm[];
  const int blk = blockIdx.x;
  const int qb = blk % nqb, h = (blk / nqb) % nq, b = blk / (nqb * nq);
  const int j = h / (nq / nkv);
  const int q0 = qb * AB_QB;
  const int klen = min(kv_len[b], T);
  const int nk = min(min(q0 + AB_QB, T), klen);  // keys any row of this block can see
  const int nqkv = (nq + 2 * nkv) * AB_HD;
  float* S = sm;                                   // [QB][nk]  scores -> P
  float* D = S + AB_QB * nk;                       // [QB][nk]  dP -> dS
  float* qs = D + AB_QB * nk;                      // [QB][65]
  float* gs = qs + AB_QB * (AB_HD + 1);            // [QB][65]
  bf16_t* ks = reinterpret_cast<bf16_t*>(gs + AB_QB * (AB_HD + 1));  // [nk][66]
  bf16_t* vs = ks + (long)nk * AB_LDK;                                // [nk][66]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)b * T;
  for (int id = tid; id < AB_QB * AB_HD; id += 256) {
    const int i = id >> 6, e = id & 63;
    const bool ok = q0 + i < T;
    qs[i * 65 + e] = ok ? bf16_to_f32(qkv[(row0 + q0 + i) * nqkv + h * AB_HD + e]) : 0.f;
    gs[i * 65 + e] = ok ? bf16_to_f32(dO[(row0 + q0 + i) * (long)(nq * AB_HD) + h * AB_HD + e]) : 0.f;
  }
  for (int id = tid; id < nk * AB_HD; id += 256) {
    const int c = id >> 6, e = id & 63;
    ks[c * AB_LDK + e] = qkv[(row0 + c) * nqkv + (nq + j) * AB_HD + e];
    vs[c * AB_LDK + e] = qkv[(row0 + c) * nqkv + (nq + nkv + j) * AB_HD + e];
  }
  __syncthreads();
  for (int ij = tid; ij < AB_QB * nk; ij += 256) {
    const int i = ij / nk, c = ij - i * nk;
    float s = -1e30f, d = 0.f;
    if (q0 + i < T && c < min(q0 + i + 1, klen)) {
      s = 0.f;
      const float* qr = qs + i * 65;
      const float* gr = gs + i * 65;
      const unsigned int* kr = reinterpret_cast<const unsigned int*>(ks + c * AB_LDK);
      const unsigned int* vr = reinterpret_cast<const unsigned int*>(vs + c * AB_LDK);
#pragma unroll 8
      for (int e2 = 0; e2 < AB_HD / 2; ++e2) {
        const unsigned int kk = kr[e2], vv = vr[e2];
        s = fmaf(qr[2 * e2], __uint_as_float(kk << 16), s);
        s = fmaf(qr[2 * e2 + 1], __uint_as_float(kk & 0xffff0000u), s);
        d = fmaf(gr[2 * e2], __uint_as_float(vv << 16), d);
        d = fmaf(gr[2 * e2 + 1], __uint_as_float(vv & 0xffff0000u), d);
      }
      s *= scale;
    }
    S[ij] = s;
    D[ij] = d;
  }
  __syncthreads();
  for (int i = wave; i < AB_QB; i += 4) {
    float* row = S + i * nk;
    float* drow = D + i * nk;
    float m = -1e30f;
    for (int c = lane; c < nk; c += 64) m = fmaxf(m, row[c]);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < nk; c += 64) {
      const float e = row[c] > -1e29f ? __expf(row[c] - m) : 0.f;
      row[c] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    float dot = 0.f;
    for (int c = lane; c < nk; c += 64) {
      row[c] *= inv;
      dot = fmaf(row[c], drow[c], dot);
    }
    dot = wave_sum(dot);
    for (int c = lane; c < nk; c += 64) drow[c] = row[c] * (drow[c] - dot) * scale;
  }
  __syncthreads();
  for (int id = tid; id < AB_QB * AB_HD; id += 256) {  // dQ
    const int i = id >> 6, e = id & 63;
    if (q0 + i >= T) continue;
    const float* dr = D + i * nk;
    float a = 0.f;
#pragma unroll 8
    for (int c = 0; c < nk; ++c) a = fmaf(dr[c], bf16_to_f32(ks[c * AB_LDK + e]), a);
    g32[(row0 + q0 + i) * nqkv + h * AB_HD + e] = a;
  }
  for (int id = tid; id < nk * AB_HD; id += 256) {  // dK, dV
    const int c = id >> 6, e = id & 63;
    float a = 0.f, v = 0.f;
#pragma unroll 8
    for (int i = 0; i < AB_QB; ++i) {
      a = fmaf(D[i * nk + c], qs[i * 65 + e], a);
      v = fmaf(S[i * nk + c], gs[i * 65 + e], v);
    }
    atomicAdd(g32 + (row0 + c) * nqkv + (nq + j) * AB_HD + e, a);
    atomicAdd(g32 + (row0 + c) * nqkv + (nq + nkv + j) * AB_HD + e, v);
  }
}


// ---------------------------------------------------------------------------
// Composed attention backward (the production path; the scalar kernel above is its cross-check): the five products run
// on the batched MFMA GEMM, and this kernel is the row-wise middle.  Row r = (b * nq + h) * T + i of S (already scaled
// scores, fp32 [.., Tp]) and dP = dO V^T (fp32):
//   P = softmax(S[:nv]), nv = min(i + 1, kv_len[b]);   dS = scale * P * (dP - sum P dP)
// both written as bf16 rows of Tp columns, zero beyond nv (the GEMMs that consume them contract over all Tp columns).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void causal_softmax_bwd_rows_kernel(const float* __restrict__ S, const float* __restrict__ dP,
                                                                      bf16_t* __restrict__ P, bf16_t* __restrict__ dS,
                                                                      const int* __restrict__ kv_len, int T, int Tp, int nq,
                                                                      float scale, long rows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int i = (int)(row % T);
  const int b = (int)(row / ((long)T * nq));
  const int nv = min(i + 1, min(kv_len[b], T));
  const f32x4* s4 = reinterpret_cast<const f32x4*>(S + row * Tp);
  const f32x4* d4 = reinterpret_cast<const f32x4*>(dP + row * Tp);
  const int nv4 = (nv + 3) >> 2, n4 = Tp >> 2;
  float m = -1e30f;
  for (int q = lane; q < nv4; q += 64) {
    const f32x4 v = s4[q];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * q + e < nv) m = fmaxf(m, v[e]);
  }
  m = wave_max(m);
  float sum = 0.f, dot = 0.f;
  for (int q = lane; q < nv4; q += 64) {
    const f32x4 v = s4[q], dd = d4[q];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * q + e < nv) {
        const float ex = __expf(v[e] - m);
        sum += ex;
        dot = fmaf(ex, dd[e], dot);
      }
  }
  sum = wave_sum(sum);
  dot = wave_sum(dot);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  dot *= inv;
  u32x2* po = reinterpret_cast<u32x2*>(P + row * Tp);
  u32x2* so = reinterpret_cast<u32x2*>(dS + row * Tp);
  for (int q = lane; q < n4; q += 64) {
    float pv[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
    if (q < nv4) {
      const f32x4 v = s4[q], dd = d4[q];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < nv) {
          pv[e] = __expf(v[e] - m) * inv;
          dv[e] = scale * pv[e] * (dd[e] - dot);
        }
    }
    const u32x2 pb = {pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])};
    const u32x2 db = {pack_bf16x2(dv[0], dv[1]), pack_bf16x2(dv[2], dv[3])};
    po[q] = pb;
    so[q] = db;
  }
}


// The same row arithmetic, tiled: one workgroup per (sample, query head, block of 64 queries) writes dS (row-major, for
// dQ = dS K) and the TRANSPOSED tiles P^T, dS^T (for dV = P^T dO, dK = dS^T q) through LDS, so that no separate
// transposition pass (and no row-major P) is needed.  Key blocks above the causal diagonal are never touched: the
// three outputs must be zero-initialised once by the caller and stay zero there whatever kv_len is.
__global__ __launch_bounds__(256) void causal_softmax_bwd_tiles_kernel(const float* __restrict__ S, const float* __restrict__ dP,
                                                                       bf16_t* __restrict__ dS, bf16_t* __restrict__ PT,
                                                                       bf16_t* __restrict__ dST, const int* __restrict__ kv_len,
                                                                       int T, int Tp, int nq, float scale) {
  __shared__ float st_m[64], st_inv[64], st_dot[64];
  __shared__ int st_nv[64];
  __shared__ bf16_t tP[64][66], tD[64][66];
  const int nqb = Tp >> 6;
  const int qb = blockIdx.x % nqb;
  const long bh = blockIdx.x / nqb;
  const int b = (int)(bh / nq);
  const int klen = min(kv_len[b], T);
  const int q0 = qb * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int rr = 0; rr < 16; ++rr) {  // row statistics: a wave per row
    const int r = wave * 16 + rr, i = q0 + r;
    const int nv = i < T ? min(i + 1, klen) : 0;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(S + (bh * T + i) * Tp);
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dP + (bh * T + i) * Tp);
    const int nv4 = (nv + 3) >> 2;
    float m = -1e30f;
    for (int q = lane; q < nv4; q += 64) {
      const f32x4 v = s4[q];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < nv) m = fmaxf(m, v[e]);
    }
    m = wave_max(m);
    float sum = 0.f, dot = 0.f;
    for (int q = lane; q < nv4; q += 64) {
      const f32x4 v = s4[q], dd = d4[q];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < nv) {
          const float ex = __expf(v[e] - m);
          sum += ex;
          dot = fmaf(ex, dd[e], dot);
        }
    }
    sum = wave_sum(sum);
    dot = wave_sum(dot);
    if (lane == 0) {
      const float inv = sum > 0.f ? 1.f / sum : 0.f;
      st_m[r] = m;
      st_inv[r] = inv;
      st_dot[r] = dot * inv;
      st_nv[r] = nv;
    }
  }
  __syncthreads();
  for (int kb = 0; kb <= qb; ++kb) {
    const int c = kb * 64 + lane;
    for (int rr = 0; rr < 16; ++rr) {
      const int r = wave * 16 + rr, i = q0 + r;
      float pv = 0.f, dv = 0.f;
      if (c < st_nv[r]) {
        const long o = (bh * T + i) * Tp + c;
        pv = __expf(S[o] - st_m[r]) * st_inv[r];
        dv = scale * pv * (dP[o] - st_dot[r]);
      }
      const bf16_t db = f32_to_bf16(dv);
      if (i < T) dS[(bh * T + i) * Tp + c] = db;
      tP[r][lane] = f32_to_bf16(pv);
      tD[r][lane] = db;
    }
    __syncthreads();
    for (int cc = wave * 16; cc < wave * 16 + 16; ++cc) {  // transposed tiles: key row kb*64 + cc, 64 queries across the lanes
      const long o = (bh * Tp + kb * 64 + cc) * Tp + q0 + lane;
      PT[o] = tP[lane][cc];
      dST[o] = tD[lane][cc];
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------
// Scores on the matrix cores, fused with the softmax backward (the production form of the middle of the attention
// backward): one workgroup per (sample, query head, block of 64 queries); wave w owns 16 query rows and computes its
// 16 x 64 tiles of  S = q K^T  and  dP = dO V^T  with v_mfma_f32_16x16x32_bf16 (q, dO rows are the A fragments, held in
// registers; K / V blocks are staged through LDS one block ahead and give the B fragments), in two sweeps over the key blocks at
// or below the diagonal: row maxima, row sums and sum(P dP) (online softmax); then P and dS, which leave through LDS tiles as dS row-major
// and P^T, dS^T (as causal_softmax_bwd_tiles_kernel).  S and dP never exist in memory.
// D[m][n] of the MFMA sits in lane l as m = 4 (l >> 4) + e, n = l & 15.
// ---------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256, 3) void attn_bwd_scores_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                              bf16_t* __restrict__ dS, bf16_t* __restrict__ PT,
                                                              bf16_t* __restrict__ dST, float* __restrict__ dQ, long ld_dq,
                                                              float* __restrict__ stats, const int* __restrict__ kv_len, int T,
                                                              int Tp, int nq, int nkv, float scale,
                                                              const float* __restrict__ lse, const bf16_t* __restrict__ att) {
  // lse + att (optional, both or neither): the forward's log-sum-exp per query row [B*nq*T] (tcavt_attn_causal_gqa_lse) and
  // its output [B*T, nq*64].  With them the row statistics are known before any key is read -- P = exp(s - lse), and
  // sum(P dP) = dO . O (the row of the output times the row of its gradient, as in FlashAttention-2's backward) -- so the
  // first of the two sweeps over the key blocks (S and dP on the matrix cores, for the maximum, the sum and sum(P dP)) is
  // not run: 3 instead of 5 block products per key block.
  // stats (optional, fp32 [B*nq*T, 4]): row maximum of the scaled scores, 1 / row sum, sum(P dP) -- what
  // attn_bwd_dkv_kernel needs to rebuild P and dS for its key block.  PT / dST (optional): only for the GEMM form of dK, dV.
  // dQ (optional, fp32 [B*T, ld_dq], head h at columns 64 h): dQ = dS K accumulated over the key blocks inside the last
  // sweep (A = the dS tile in LDS, B = a transposed copy of the K block); dS (optional): the row-major copy for an
  // external dQ product.
  __shared__ bf16_t tP[64][66], tD[64][66];
  __shared__ __attribute__((aligned(16))) bf16_t ksT[64][72];  // K block transposed [d][key] (last sweep, dQ only)
  const int nqb = Tp >> 6;
  const int qb = blockIdx.x % nqb;
  const long bh = blockIdx.x / nqb;
  const int b = (int)(bh / nq), h = (int)(bh % nq);
  const int j = h / (nq / nkv);
  const int klen = min(kv_len[b], T);
  const int q0 = qb * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const long nqkv = (long)(nq + 2 * nkv) * 64;
  const long row0 = (long)b * T;
  bf16x8 qf[2], gf[2];
  {
    const long ar = row0 + min(q0 + wave * 16 + l15, T - 1);  // rows >= T: clamped load, masked below (nv = 0)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qf[kk] = *reinterpret_cast<const bf16x8*>(qkv + ar * nqkv + h * 64 + kk * 32 + l4 * 8);
      gf[kk] = *reinterpret_cast<const bf16x8*>(dO + ar * (long)(nq * 64) + h * 64 + kk * 32 + l4 * 8);
    }
  }
  int nv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = q0 + wave * 16 + l4 * 4 + e;
    nv[e] = i < T ? min(i + 1, klen) : 0;
  }
  // K / V key blocks go global -> registers -> LDS with whole 128-byte rows per 8 lanes (coalesced), one block ahead of the
  // compute; the B fragments are then 16-byte LDS reads (rows padded to 144 bytes).  Loading the fragments straight from
  // global memory put every lane on its own cache line: 64 address cycles per load, 372 us per launch.
  __shared__ __attribute__((aligned(16))) bf16_t ks[64][72], vs[64][72];
  const int srow = threadIdx.x >> 2, sch = (threadIdx.x & 3) * 16;
  // transposed staging: the four threads of a source row write rows 16 apart, i.e. the same bank; the 16-byte column
  // chunk is therefore XOR-ed with the row's 16-block (= threadIdx.x & 3 for the writer, dt for the reader)
  const int tcol = (((srow >> 3) ^ (threadIdx.x & 3)) << 3) | (srow & 7);
  const bf16_t* kbase = qkv + (nq + j) * 64 + sch;
  const bf16_t* vbase = qkv + (nq + nkv + j) * 64 + sch;
  u32x4 kreg[2], vreg[2];
  auto fetch = [&](int kb, bool want_v) {
    const long kr = row0 + min(kb * 64 + srow, T - 1);  // keys >= T: clamped load, masked (>= nv)
    kreg[0] = *reinterpret_cast<const u32x4*>(kbase + kr * nqkv);
    kreg[1] = *reinterpret_cast<const u32x4*>(kbase + kr * nqkv + 8);
    if (want_v) {
      vreg[0] = *reinterpret_cast<const u32x4*>(vbase + kr * nqkv);
      vreg[1] = *reinterpret_cast<const u32x4*>(vbase + kr * nqkv + 8);
    }
  };
  bool want_t = false;
  auto stage = [&](bool want_v) {
    *reinterpret_cast<u32x4*>(&ks[srow][sch]) = kreg[0];
    *reinterpret_cast<u32x4*>(&ks[srow][sch + 8]) = kreg[1];
    if (want_t) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ksT[sch + c * 8 + 2 * e][tcol] = (bf16_t)(kreg[c][e] & 0xffffu);
          ksT[sch + c * 8 + 2 * e + 1][tcol] = (bf16_t)(kreg[c][e] >> 16);
        }
    }
    if (want_v) {
      *reinterpret_cast<u32x4*>(&vs[srow][sch]) = vreg[0];
      *reinterpret_cast<u32x4*>(&vs[srow][sch + 8]) = vreg[1];
    }
  };
  auto scores = [&](f32x4 (&sa)[4], f32x4 (&da)[4], bool want_d) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&ks[t * 16 + l15][kk * 32 + l4 * 8]);
        sacc = mfma16b<F16>(qf[kk], kf, sacc);
        if (want_d) {
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&vs[t * 16 + l15][kk * 32 + l4 * 8]);
          dacc = mfma16b<F16>(gf[kk], vf, dacc);
        }
      }
      sa[t] = sacc * scale;
      da[t] = dacc;
    }
  };
  // one sweep over the key blocks 0..qb: body(kb, S tiles, dP tiles) runs with block kb staged and block kb + 1 in flight
  auto sweep = [&](bool want_v, auto&& body) {
    fetch(0, want_v);
    for (int kb = 0; kb <= qb; ++kb) {
      __syncthreads();  // everyone is done with the previous block (and with the output tiles of the previous body)
      stage(want_v);
      __syncthreads();
      if (kb < qb) fetch(kb + 1, want_v);
      f32x4 sa[4], da[4];
      scores(sa, da, want_v);
      body(kb, sa, da);
    }
  };
  auto group16 = [&](float v, bool is_max) {  // reduce over the 16 lanes that share l >> 4 (one query row each e)
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const float w = __shfl_xor(v, o, 64);
      v = is_max ? fmaxf(v, w) : v + w;
    }
    return v;
  };
  // first sweep: row maximum, row sum and sum(P dP) in ONE pass -- every lane runs an online softmax over the keys it sees
  // (its own running maximum; sums rescaled when it moves), the 16 lanes of a row are merged afterwards
  float m[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
  float sum[4] = {0.f, 0.f, 0.f, 0.f}, dot[4] = {0.f, 0.f, 0.f, 0.f};
  float inv[4];
  if (lse) {  // (uniform) statistics from the forward: no first sweep
    // dO . O of query row wave * 16 + l15: this lane's 16 elements of the A-fragment rows, then the four lane groups
    float dsum = 0.f;
    {
      const long ar = row0 + min(q0 + wave * 16 + l15, T - 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const u32x4 ov = *reinterpret_cast<const u32x4*>(att + ar * (long)(nq * 64) + h * 64 + kk * 32 + l4 * 8);
        const u32x4 gv = __builtin_bit_cast(u32x4, gf[kk]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dsum = fmaf(from16_lo<F16>(ov[e]), from16_lo<F16>(gv[e]), dsum);
          dsum = fmaf(from16_hi<F16>(ov[e]), from16_hi<F16>(gv[e]), dsum);
        }
      }
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = q0 + wave * 16 + l4 * 4 + e;
      m[e] = lse[bh * T + min(i, T - 1)];
      inv[e] = 1.f;
      dot[e] = __shfl(dsum, l4 * 4 + e, 64);  // (lane l4 * 4 + e holds row l4 * 4 + e: its l15)
      if (stats && l15 == 0 && i < T) *reinterpret_cast<f32x4*>(stats + (bh * T + i) * 4) = f32x4{m[e], 1.f, dot[e], 0.f};
    }
  } else {
  sweep(true, [&](int kb, f32x4 (&sa)[4], f32x4 (&da)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float tm = m[e];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (kb * 64 + t * 16 + l15 < nv[e]) tm = fmaxf(tm, sa[t][e]);
      const float resc = __expf(m[e] - tm);  // 1 when the maximum did not move; 0 * 0 on the first valid key
      sum[e] *= resc;
      dot[e] *= resc;
      m[e] = tm;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (kb * 64 + t * 16 + l15 < nv[e]) {
          const float ex = __expf(sa[t][e] - tm);
          sum[e] += ex;
          dot[e] = fmaf(ex, da[t][e], dot[e]);
        }
    }
  });
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float mr = group16(m[e], true);
    const float resc = __expf(m[e] - mr);
    sum[e] *= resc;
    dot[e] *= resc;
    m[e] = mr;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sum[e] = group16(sum[e], false);
    dot[e] = group16(dot[e], false);
    inv[e] = sum[e] > 0.f ? 1.f / sum[e] : 0.f;
    dot[e] *= inv[e];
    const int i = q0 + wave * 16 + l4 * 4 + e;
    if (stats && l15 == 0 && i < T) *reinterpret_cast<f32x4*>(stats + (bh * T + i) * 4) = f32x4{m[e], inv[e], dot[e], 0.f};
  }
  }  // (two-sweep form)
  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  want_t = dQ != nullptr;
  sweep(true, [&](int kb, f32x4 (&sa)[4], f32x4 (&da)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int cl = t * 16 + l15;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pv = 0.f, dv = 0.f;
        if (kb * 64 + cl < nv[e]) {
          pv = __expf(sa[t][e] - m[e]) * inv[e];
          dv = scale * pv * (da[t][e] - dot[e]);
        }
        const int r = wave * 16 + l4 * 4 + e;
        tP[r][cl] = to16<F16>(pv);
        tD[r][cl] = to16<F16>(dv);
      }
    }
    __syncthreads();
    // 8-byte stores: one instruction covers 4 rows x 16 lanes x 4 elements (2-byte stores, one row per instruction, cost
    // 48 store instructions per wave and block; these are 12)
#pragma unroll
    for (int it = 0; it < 4; ++it) {  // dS row-major: query row r, key quad l15
      const int r = wave * 16 + it * 4 + l4, i = q0 + r;
      const unsigned int* src = reinterpret_cast<const unsigned int*>(&tD[r][l15 * 4]);
      const u32x2 v = {src[0], src[1]};
      if (dS && i < T) *reinterpret_cast<u32x2*>(dS + (bh * T + i) * Tp + kb * 64 + l15 * 4) = v;
    }
    if (dQ) {  // dQ[16 rows of this wave][64] += dS tile [16 x 64 keys] . K block [64 keys x 64]
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const unsigned int* ap = reinterpret_cast<const unsigned int*>(&tD[wave * 16 + l15][kk * 32 + l4 * 8]);
        const u32x4 av = {ap[0], ap[1], ap[2], ap[3]};
        const bf16x8 af = __builtin_bit_cast(bf16x8, av);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&ksT[dt * 16 + l15][((kk * 4 + l4) ^ dt) * 8]);
          dq[dt] = mfma16b<F16>(af, bfr, dq[dt]);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < 4 && PT != nullptr; ++it) {  // transposed tiles: key row cc, query quad l15
      const int cc = wave * 16 + it * 4 + l4, qs = l15 * 4;
      const u32x2 pv = {(unsigned int)tP[qs][cc] | ((unsigned int)tP[qs + 1][cc] << 16),
                        (unsigned int)tP[qs + 2][cc] | ((unsigned int)tP[qs + 3][cc] << 16)};
      const u32x2 dv = {(unsigned int)tD[qs][cc] | ((unsigned int)tD[qs + 1][cc] << 16),
                        (unsigned int)tD[qs + 2][cc] | ((unsigned int)tD[qs + 3][cc] << 16)};
      const long o = (bh * Tp + kb * 64 + cc) * Tp + q0 + qs;
      *reinterpret_cast<u32x2*>(PT + o) = pv;
      *reinterpret_cast<u32x2*>(dST + o) = dv;
    }
    // (the sweep's barrier before the next stage() also covers these tile reads)
  });
  if (dQ) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = q0 + wave * 16 + l4 * 4 + e;
        if (i < T) dQ[(row0 + i) * ld_dq + h * 64 + dt * 16 + l15] = dq[dt][e];
      }
  }
}


// ---------------------------------------------------------------------------
// dQ of the attention backward from the forward's row statistics (the product form of the query-major half when the tape
// holds lse and the attention's output: T <= 256, 16 % (nq / nkv) == 0).  One workgroup per (sample, key/value head) as in
// the forward: K, V (row-major) and K^T of the head are staged in LDS ONCE -- 110 KB -- and sixteen waves walk
// (query head, strip of 16 queries) units with no barrier after the staging one; the strips are dealt in (short, long)
// pairs, so every wave sees the same number of keys.  attn_bwd_scores_kernel re-staged the K / V block for every query
// head and every block of 64 queries (two barriers each) and passed P and dS through LDS tiles with two-byte stores.
//   S^T = K q^T and dP^T = V dO^T  (keys x queries: A = K / V rows from LDS, B = the strip's q / dO rows in registers)
//   P = exp(s - lse),  dS = scale P (dP - dO . O)   per element; D[m][n] sits in lane l as key 4 (l >> 4) + e, query l & 15
//   dQ += dS K: the lane's eight values of two key tiles ARE its A fragment (contraction index = keys, taken in the order
//   tile 0 keys 4 g .. 4 g + 3, tile 1 keys 4 g .. 4 g + 3), the B fragment reads K^T with the same key order.
// Nothing is written but dQ (fp32, head h at columns 64 h) and stats = (lse, 1, dO . O, 0) for attn_bwd_dkv_kernel.
// ---------------------------------------------------------------------------
constexpr int ABQ_WAVES = 16;

// The gradient of 16 rows x 64 dimensions of one head in MFMA accumulator layout (acc[dt][e]: row r0 + 4 (l >> 4) + e,
// dimension 16 dt + (l & 15)) -> the 16-bit gradient of the projections' outputs: dimensions d and d + 32 of a row sit in
// the same lane (dt and dt + 2), so the transposed rotation of rope_bwd_pack_kernel needs no exchange:
//   g_t1 = g_o1 cos + g_o2 sin,   g_t2 = g_o2 cos - g_o1 sin     (position = row inside the sample)
template <bool F16>
__device__ __forceinline__ void store_grad16(const f32x4 (&acc)[4], bf16_t* __restrict__ out, long ld, long row0, int r0, int T,
                                             int col0, bool rotate, const float* __restrict__ cosT,
                                             const float* __restrict__ sinT, int lane) {
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = r0 + 4 * l4 + e;
    if (i >= T) continue;
    bf16_t* o = out + (row0 + i) * ld + col0 + l15;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      float a = acc[w][e], b2 = acc[w + 2][e];
      if (rotate) {
        const float cs = cosT[i * 32 + w * 16 + l15], sn = sinT[i * 32 + w * 16 + l15];
        const float t1 = a * cs + b2 * sn, t2 = b2 * cs - a * sn;
        a = t1;
        b2 = t2;
      }
      o[w * 16] = to16<F16>(a);
      o[w * 16 + 32] = to16<F16>(b2);
    }
  }
}

template <bool F16>
__global__ __launch_bounds__(ABQ_WAVES * 64) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                                     const bf16_t* __restrict__ att, const float* __restrict__ lse,
                                                                     float* __restrict__ dQ, long ld_dq, float* __restrict__ stats,
                                                                     const int* __restrict__ kv_len, int T, int Tp, int nq, int nkv,
                                                                     float scale, bf16_t* __restrict__ g16,
                                                                     const float* __restrict__ cosT, const float* __restrict__ sinT) {
  // g16 (optional, with cosT / sinT [T][32]): the 16-bit gradient of the q|k|v projection's output [B*T][(nq + 2 nkv) * 64];
  // the q columns are written here directly, transposed rotation applied (then dQ is not written: no fp32 round trip and
  // no rope_bwd_pack launch)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);                 // [Tp][72]
  bf16_t* Vs = Ks + Tp * 72;                                    // [Tp][72]
  bf16_t* KT = Vs + Tp * 72;                                    // [64][Tp + 4]
  const int kts = Tp + 4;
  const int group = nq / nkv;
  const int b = blockIdx.x / nkv, j = blockIdx.x % nkv;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const long nqkv = (long)(nq + 2 * nkv) * 64;
  const long row0 = (long)b * T;
  const int klen = min(kv_len[b], T);
  // ---- stage K, V, K^T: thread (key pair kp, 16-byte chunk c): rows 2 kp, 2 kp + 1; all loads first, rows >= T as zeros
  {
    const bf16_t* kbase = qkv + (nq + j) * 64;
    const bf16_t* vbase = qkv + (nq + nkv + j) * 64;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    for (int idx = threadIdx.x; idx < (Tp >> 1) * 8; idx += ABQ_WAVES * 64) {
      const int kp = idx >> 3, c = idx & 7;
      const int r0 = 2 * kp, r1 = r0 + 1;
      const long g0 = (row0 + min(r0, T - 1)) * nqkv + c * 8, g1 = (row0 + min(r1, T - 1)) * nqkv + c * 8;
      u32x4 ka = *reinterpret_cast<const u32x4*>(kbase + g0), kb = *reinterpret_cast<const u32x4*>(kbase + g1);
      u32x4 va = *reinterpret_cast<const u32x4*>(vbase + g0), vb = *reinterpret_cast<const u32x4*>(vbase + g1);
      if (r0 >= T) { ka = zero4; va = zero4; }
      if (r1 >= T) { kb = zero4; vb = zero4; }
      *reinterpret_cast<u32x4*>(Ks + r0 * 72 + c * 8) = ka;
      *reinterpret_cast<u32x4*>(Ks + r1 * 72 + c * 8) = kb;
      *reinterpret_cast<u32x4*>(Vs + r0 * 72 + c * 8) = va;
      *reinterpret_cast<u32x4*>(Vs + r1 * 72 + c * 8) = vb;
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // two keys of one dimension per dword
        *reinterpret_cast<unsigned int*>(KT + (c * 8 + 2 * e) * kts + r0) = (ka[e] & 0xffffu) | (kb[e] << 16);
        *reinterpret_cast<unsigned int*>(KT + (c * 8 + 2 * e + 1) * kts + r0) = (ka[e] >> 16) | (kb[e] & 0xffff0000u);
      }
    }
  }
  __syncthreads();
  // ---- units: wave w serves query head w % group; its strips come in pairs (k, nstrips - 1 - k), k = w / group + NG * i
  const int h = j * group + wave % group;
  const int g0w = wave / group, NG = ABQ_WAVES / group;
  const int nstrips = (T + 15) >> 4;
  const int npair = (nstrips + 1) >> 1;
  const float c2 = scale * 1.4426950408889634f;
  for (int u = 0;; ++u) {
    const int pr = g0w + NG * (u >> 1);
    if (pr >= npair) break;  // (uniform)
    const int strip = (u & 1) ? nstrips - 1 - pr : pr;
    if ((u & 1) && strip == pr) continue;  // odd count: the middle strip is its own pair
    const int qi = strip * 16 + l15;       // this lane's query (B-fragment column / D column)
    const long ar = row0 + min(qi, T - 1);
    bf16x8 qf[2], gf[2];
    float delta = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qf[kk] = *reinterpret_cast<const bf16x8*>(qkv + ar * nqkv + h * 64 + kk * 32 + l4 * 8);
      gf[kk] = *reinterpret_cast<const bf16x8*>(dO + ar * (long)(nq * 64) + h * 64 + kk * 32 + l4 * 8);
      const u32x4 ov = *reinterpret_cast<const u32x4*>(att + ar * (long)(nq * 64) + h * 64 + kk * 32 + l4 * 8);
      const u32x4 gv = __builtin_bit_cast(u32x4, gf[kk]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        delta = fmaf(from16_lo<F16>(ov[e]), from16_lo<F16>(gv[e]), delta);
        delta = fmaf(from16_hi<F16>(ov[e]), from16_hi<F16>(gv[e]), delta);
      }
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    const long srow = ((long)b * nq + h) * T + min(qi, T - 1);
    const float l_nat = lse[srow];
    const float l2 = l_nat * 1.4426950408889634f;
    const int nv = qi < T ? min(qi + 1, klen) : 0;
    if (stats && l4 == 0 && qi < T) *reinterpret_cast<f32x4*>(stats + srow * 4) = f32x4{l_nat, 1.f, delta, 0.f};
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkeys = min(strip * 16 + 16, klen);  // keys any query of the strip attends
    for (int k0 = 0; k0 < nkeys; k0 += 32) {
      f32x4 sa[2], da[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
        const int krow = k0 + t * 16 + l15;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + krow * 72 + kk * 32 + l4 * 8);
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + krow * 72 + kk * 32 + l4 * 8);
          sacc = mfma16b<F16>(kf, qf[kk], sacc);
          dacc = mfma16b<F16>(vf, gf[kk], dacc);
        }
        sa[t] = sacc;
        da[t] = dacc;
      }
      u32x4 af;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = k0 + t * 16 + 4 * l4 + e;
          const float pv = key < nv ? __builtin_amdgcn_exp2f(fmaf(sa[t][e], c2, -l2)) : 0.f;
          ds[e] = scale * pv * (da[t][e] - delta);
        }
        af[2 * t] = pack16x2<F16>(ds[0], ds[1]);
        af[2 * t + 1] = pack16x2<F16>(ds[2], ds[3]);
      }
      const bf16x8 afr = __builtin_bit_cast(bf16x8, af);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16_t* kt = KT + (dt * 16 + l15) * kts + k0 + 4 * l4;
        const u32x2 b0 = *reinterpret_cast<const u32x2*>(kt), b1 = *reinterpret_cast<const u32x2*>(kt + 16);
        const u32x4 bv = {b0[0], b0[1], b1[0], b1[1]};
        dq[dt] = mfma16b<F16>(afr, __builtin_bit_cast(bf16x8, bv), dq[dt]);
      }
    }
    if (g16) {
      store_grad16<F16>(dq, g16, nqkv, row0, strip * 16, T, h * 64, true, cosT, sinT, lane);
      continue;
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = strip * 16 + l4 * 4 + e;
        if (i < T) dQ[(row0 + i) * ld_dq + h * 64 + dt * 16 + l15] = dq[dt][e];
      }
  }
}

// ---------------------------------------------------------------------------
// dK, dV of the attention backward, key-major: one workgroup per (sample, key/value head, block of 64 keys); wave w owns
// 16 keys, whose K and V rows are its A fragments for the whole kernel.  For every query head of the group and every
// query block at or below the diagonal, the Q and dO blocks are staged in LDS (row-major for the score products,
// transposed for the gradient products), the tiles  S^T = K q^T  and  dP^T = V dO^T  are rebuilt on the matrix cores,
// P^T and dS^T follow from the per-query statistics attn_bwd_scores_kernel left behind, pass through a wave-private LDS
// tile (accumulator layout -> A-fragment layout), and  dV += P^T dO,  dK += dS^T q  accumulate in registers over all of
// it -- the group sum included.  Written once, fp32, into the k / v columns of the q|k|v-layout gradient.
// ---------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256, 3) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                           const float* __restrict__ stats, float* __restrict__ g32,
                                                           const int* __restrict__ kv_len, int T, int Tp, int nq, int nkv,
                                                           float scale) {
  // (tP / tD, the P^T / dS^T tiles, reuse the storage of the row-major Q / dO blocks: one more barrier per iteration, 37 KB
  //  instead of 55 KB of LDS, three workgroups per CU instead of two)
  __shared__ __attribute__((aligned(16))) bf16_t qs[64][72], gs[64][72], qsT[64][72], gsT[64][72];
  bf16_t (*tP)[72] = qs;
  bf16_t (*tD)[72] = gs;
  __shared__ float st_m[64], st_inv[64], st_dot[64];
  __shared__ int st_nv[64];
  const int nkb = Tp >> 6;
  const int kb = blockIdx.x % nkb;
  const int bj = blockIdx.x / nkb;
  const int b = bj / nkv, j = bj % nkv;
  const int grp = nq / nkv;
  const int klen = min(kv_len[b], T);
  const int k0 = kb * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const long nqkv = (long)(nq + 2 * nkv) * 64;
  const long row0 = (long)b * T;
  bf16x8 kf[2], vf[2];
  {
    const long ar = row0 + min(k0 + wave * 16 + l15, T - 1);  // keys >= T: clamped load; they are >= nv of every query
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      kf[kk] = *reinterpret_cast<const bf16x8*>(qkv + ar * nqkv + (nq + j) * 64 + kk * 32 + l4 * 8);
      vf[kk] = *reinterpret_cast<const bf16x8*>(qkv + ar * nqkv + (nq + nkv + j) * 64 + kk * 32 + l4 * 8);
    }
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int srow = threadIdx.x >> 2, sch = (threadIdx.x & 3) * 16;
  const int tcol = (((srow >> 3) ^ (threadIdx.x & 3)) << 3) | (srow & 7);  // XOR-swizzled column of the transposed copies
  const int nit = grp * (nkb - kb);
  u32x4 qreg[2], greg[2];
  auto fetch = [&](int it) {
    const int h = j * grp + it / (nkb - kb), qb = kb + it % (nkb - kb);
    const long qr = row0 + min(qb * 64 + srow, T - 1);
    qreg[0] = *reinterpret_cast<const u32x4*>(qkv + qr * nqkv + h * 64 + sch);
    qreg[1] = *reinterpret_cast<const u32x4*>(qkv + qr * nqkv + h * 64 + sch + 8);
    greg[0] = *reinterpret_cast<const u32x4*>(dO + qr * (long)(nq * 64) + h * 64 + sch);
    greg[1] = *reinterpret_cast<const u32x4*>(dO + qr * (long)(nq * 64) + h * 64 + sch + 8);
  };
  fetch(0);
  for (int it = 0; it < nit; ++it) {
    const int h = j * grp + it / (nkb - kb), qb = kb + it % (nkb - kb);
    const int q0 = qb * 64;
    __syncthreads();  // the previous iteration is done with the staged blocks and the tiles
    *reinterpret_cast<u32x4*>(&qs[srow][sch]) = qreg[0];
    *reinterpret_cast<u32x4*>(&qs[srow][sch + 8]) = qreg[1];
    *reinterpret_cast<u32x4*>(&gs[srow][sch]) = greg[0];
    *reinterpret_cast<u32x4*>(&gs[srow][sch + 8]) = greg[1];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qsT[sch + c * 8 + 2 * e][tcol] = (bf16_t)(qreg[c][e] & 0xffffu);
        qsT[sch + c * 8 + 2 * e + 1][tcol] = (bf16_t)(qreg[c][e] >> 16);
        gsT[sch + c * 8 + 2 * e][tcol] = (bf16_t)(greg[c][e] & 0xffffu);
        gsT[sch + c * 8 + 2 * e + 1][tcol] = (bf16_t)(greg[c][e] >> 16);
      }
    if (threadIdx.x < 64) {
      const int i = q0 + threadIdx.x;
      float m = 0.f, inv = 0.f, dot = 0.f;
      int nv = 0;
      if (i < T) {
        const f32x4 st = *reinterpret_cast<const f32x4*>(stats + (((long)b * nq + h) * T + i) * 4);
        m = st[0]; inv = st[1]; dot = st[2];
        nv = min(i + 1, klen);
      }
      st_m[threadIdx.x] = m; st_inv[threadIdx.x] = inv; st_dot[threadIdx.x] = dot; st_nv[threadIdx.x] = nv;
    }
    __syncthreads();
    if (it + 1 < nit) fetch(it + 1);
    f32x4 sT[4], dT[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {  // 16 keys of this wave x 16 queries
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(&qs[t * 16 + l15][kk * 32 + l4 * 8]);
        const bf16x8 gf = *reinterpret_cast<const bf16x8*>(&gs[t * 16 + l15][kk * 32 + l4 * 8]);
        sacc = mfma16b<F16>(kf[kk], qf, sacc);
        dacc = mfma16b<F16>(vf[kk], gf, dacc);
      }
      sT[t] = sacc;
      dT[t] = dacc;
    }
    __syncthreads();  // every wave has read the row-major Q / dO blocks: their storage becomes the P^T / dS^T tiles
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 sacc = sT[t], dacc = dT[t];
      const int qi = t * 16 + l15;  // D[m][n]: m = key 4 l4 + e of the wave's 16, n = query qi
      const float qm = st_m[qi], qinv = st_inv[qi], qdot = st_dot[qi];
      const int qnv = st_nv[qi];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pv = 0.f, dsv = 0.f;
        if (k0 + wave * 16 + l4 * 4 + e < qnv) {
          pv = __expf(sacc[e] * scale - qm) * qinv;
          dsv = scale * pv * (dacc[e] - qdot);
        }
        tP[wave * 16 + l4 * 4 + e][qi] = to16<F16>(pv);
        tD[wave * 16 + l4 * 4 + e][qi] = to16<F16>(dsv);
      }
    }
    __syncthreads();  // (the tile rows of a wave are private to it; the barrier only orders its own writes and reads)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {  // contraction over the 64 queries of the block
      const bf16x8 pf = *reinterpret_cast<const bf16x8*>(&tP[wave * 16 + l15][kk * 32 + l4 * 8]);
      const bf16x8 df = *reinterpret_cast<const bf16x8*>(&tD[wave * 16 + l15][kk * 32 + l4 * 8]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 gT = *reinterpret_cast<const bf16x8*>(&gsT[dt * 16 + l15][((kk * 4 + l4) ^ dt) * 8]);
        const bf16x8 qT = *reinterpret_cast<const bf16x8*>(&qsT[dt * 16 + l15][((kk * 4 + l4) ^ dt) * 8]);
        dv[dt] = mfma16b<F16>(pf, gT, dv[dt]);
        dk[dt] = mfma16b<F16>(df, qT, dk[dt]);
      }
    }
  }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = k0 + wave * 16 + l4 * 4 + e;
      if (key < T) {
        g32[(row0 + key) * nqkv + (nq + j) * 64 + dt * 16 + l15] = dk[dt][e];
        g32[(row0 + key) * nqkv + (nq + nkv + j) * 64 + dt * 16 + l15] = dv[dt][e];
      }
    }
}

// ---------------------------------------------------------------------------
// dK, dV with the query side resident (T <= 256): one workgroup per (sample, key/value head), sixteen waves of 16 keys each
// (their K / V rows are B fragments held in registers for the whole kernel).  Per query head of the group, q and dO of ALL
// queries are staged once -- row-major for the score products, transposed (two queries per dword) for the gradient
// products: 140 KB -- and every wave walks the queries at or below its keys' diagonal, 32 at a time, without a barrier:
//   S = q K^T, dP = dO V^T as [queries x keys] tiles (A = q / dO rows from LDS): D[m][n] = query 4 (l >> 4) + e, key l & 15
//   P = exp(s - m) inv,  dS = scale P (dP - dot)         (statistics of the query, from LDS)
//   dV += P^T dO, dK += dS^T q: the lane's eight values of two query tiles ARE its A fragment (rows = keys, contraction
//   over queries in the order tile 0 queries 4 g .. 4 g + 3, tile 1 the same), B = dO^T / q^T read with the same order.
// attn_bwd_dkv_kernel re-staged a 64-query
// block per iteration behind four barriers and moved P^T / dS^T through LDS tiles with two-byte stores.
// ---------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(1024) void attn_bwd_dkv_res_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                                const float* __restrict__ stats, float* __restrict__ g32,
                                                                const int* __restrict__ kv_len, int T, int Tp, int nq, int nkv,
                                                                float scale, bf16_t* __restrict__ g16,
                                                                const float* __restrict__ cosT, const float* __restrict__ sinT) {
  // g16 (optional): as in attn_bwd_dq_kernel -- the k (rotation undone) and v columns go out as 16 bits, g32 is not written
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tts = Tp + 4;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);  // [Tp][72]
  bf16_t* Gs = Qs + Tp * 72;                     // [Tp][72]
  bf16_t* QT = Gs + Tp * 72;                     // [64][Tp + 4]
  bf16_t* GT = QT + 64 * tts;                    // [64][Tp + 4]
  float* st = reinterpret_cast<float*>(GT + 64 * tts);  // [3][Tp]: m * log2(e), 1 / sum, sum(P dP)
  const int grp = nq / nkv;
  const int b = blockIdx.x / nkv, j = blockIdx.x % nkv;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const long nqkv = (long)(nq + 2 * nkv) * 64;
  const long row0 = (long)b * T;
  const int klen = min(kv_len[b], T);
  // key strip of this wave: (s, 7 - s, 8 + s, 15 - s) for the four waves of SIMD s -- equal numbers of query tiles per SIMD
  const int sd = wave & 3, rr = wave >> 2;
  const int strip = rr == 0 ? sd : rr == 1 ? 7 - sd : rr == 2 ? 8 + sd : 15 - sd;
  const int key0 = strip * 16;
  const bool active = key0 < Tp;  // (uniform)
  bf16x8 kf[2], vf[2];
  {
    const long ar = row0 + min(key0 + l15, T - 1);  // keys >= T: clamped load; masked (>= klen)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      kf[kk] = *reinterpret_cast<const bf16x8*>(qkv + ar * nqkv + (nq + j) * 64 + kk * 32 + l4 * 8);
      vf[kk] = *reinterpret_cast<const bf16x8*>(qkv + ar * nqkv + (nq + nkv + j) * 64 + kk * 32 + l4 * 8);
    }
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging: thread (query pair qp, 16-byte chunk c) -> rows 2 qp, 2 qp + 1 of q and dO (Tp / 2 * 8 <= 1024 pair-chunks)
  const int qp = threadIdx.x >> 3, ch = threadIdx.x & 7;
  const bool stager = threadIdx.x < (Tp >> 1) * 8;
  u32x4 qa, qb, ga, gb;
  auto fetch = [&](int hh) {
    const int h = j * grp + hh;
    const long g0 = row0 + min(2 * qp, T - 1), g1 = row0 + min(2 * qp + 1, T - 1);
    qa = *reinterpret_cast<const u32x4*>(qkv + g0 * nqkv + h * 64 + ch * 8);
    qb = *reinterpret_cast<const u32x4*>(qkv + g1 * nqkv + h * 64 + ch * 8);
    ga = *reinterpret_cast<const u32x4*>(dO + g0 * (long)(nq * 64) + h * 64 + ch * 8);
    gb = *reinterpret_cast<const u32x4*>(dO + g1 * (long)(nq * 64) + h * 64 + ch * 8);
  };
  const float c2 = scale * 1.4426950408889634f;
  for (int hh = 0; hh < grp; ++hh) {
    const int h = j * grp + hh;
    if (hh > 0) __syncthreads();  // everyone is done with the previous head's blocks
    // (keeping the next head's rows in flight in registers over this head's arithmetic pushed the kernel past its 128
    //  registers: measured slower, 40.5 vs 40.2 ms per step)
    if (stager) fetch(hh);
    if (stager) {
      const int r0 = 2 * qp, r1 = r0 + 1;
      const u32x4 zero4 = {0u, 0u, 0u, 0u};
      if (r0 >= T) { qa = zero4; ga = zero4; }
      if (r1 >= T) { qb = zero4; gb = zero4; }
      *reinterpret_cast<u32x4*>(Qs + r0 * 72 + ch * 8) = qa;
      *reinterpret_cast<u32x4*>(Qs + r1 * 72 + ch * 8) = qb;
      *reinterpret_cast<u32x4*>(Gs + r0 * 72 + ch * 8) = ga;
      *reinterpret_cast<u32x4*>(Gs + r1 * 72 + ch * 8) = gb;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        *reinterpret_cast<unsigned int*>(QT + (ch * 8 + 2 * e) * tts + r0) = (qa[e] & 0xffffu) | (qb[e] << 16);
        *reinterpret_cast<unsigned int*>(QT + (ch * 8 + 2 * e + 1) * tts + r0) = (qa[e] >> 16) | (qb[e] & 0xffff0000u);
        *reinterpret_cast<unsigned int*>(GT + (ch * 8 + 2 * e) * tts + r0) = (ga[e] & 0xffffu) | (gb[e] << 16);
        *reinterpret_cast<unsigned int*>(GT + (ch * 8 + 2 * e + 1) * tts + r0) = (ga[e] >> 16) | (gb[e] & 0xffff0000u);
      }
    }
    if ((int)threadIdx.x < Tp) {
      const int i = threadIdx.x;
      f32x4 sv = {0.f, 0.f, 0.f, 0.f};
      if (i < T) sv = *reinterpret_cast<const f32x4*>(stats + (((long)b * nq + h) * T + i) * 4);
      st[i] = sv[0] * 1.4426950408889634f;
      st[Tp + i] = i < T ? sv[1] : 0.f;  // (queries >= T: P = 0)
      st[2 * Tp + i] = sv[2];
    }
    __syncthreads();
    if (active) {
      for (int q0 = key0 & ~31; q0 < Tp; q0 += 32) {  // queries at or below the diagonal of this wave's keys
        f32x4 sa[2], da[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
          const int qrow = q0 + t * 16 + l15;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + qrow * 72 + kk * 32 + l4 * 8);
            const bf16x8 gf = *reinterpret_cast<const bf16x8*>(Gs + qrow * 72 + kk * 32 + l4 * 8);
            sacc = mfma16b<F16>(qf, kf[kk], sacc);
            dacc = mfma16b<F16>(gf, vf[kk], dacc);
          }
          sa[t] = sacc;
          da[t] = dacc;
        }
        u32x4 pfr, dfr;
        const int key = key0 + l15;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int qi = q0 + t * 16 + 4 * l4;
          const f32x4 m2 = *reinterpret_cast<const f32x4*>(st + qi), iv = *reinterpret_cast<const f32x4*>(st + Tp + qi),
                      dt_ = *reinterpret_cast<const f32x4*>(st + 2 * Tp + qi);
          float pv[4], ds[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = key <= qi + e && key < klen;  // (queries >= T have 1 / sum = 0)
            pv[e] = ok ? __builtin_amdgcn_exp2f(fmaf(sa[t][e], c2, -m2[e])) * iv[e] : 0.f;
            ds[e] = scale * pv[e] * (da[t][e] - dt_[e]);
          }
          pfr[2 * t] = pack16x2<F16>(pv[0], pv[1]);
          pfr[2 * t + 1] = pack16x2<F16>(pv[2], pv[3]);
          dfr[2 * t] = pack16x2<F16>(ds[0], ds[1]);
          dfr[2 * t + 1] = pack16x2<F16>(ds[2], ds[3]);
        }
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pfr), df = __builtin_bit_cast(bf16x8, dfr);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16_t* gt = GT + (dt * 16 + l15) * tts + q0 + 4 * l4;
          const bf16_t* qt = QT + (dt * 16 + l15) * tts + q0 + 4 * l4;
          const u32x2 g0 = *reinterpret_cast<const u32x2*>(gt), g1 = *reinterpret_cast<const u32x2*>(gt + 16);
          const u32x2 q0v = *reinterpret_cast<const u32x2*>(qt), q1v = *reinterpret_cast<const u32x2*>(qt + 16);
          const u32x4 gv = {g0[0], g0[1], g1[0], g1[1]}, qv = {q0v[0], q0v[1], q1v[0], q1v[1]};
          dv[dt] = mfma16b<F16>(pf, __builtin_bit_cast(bf16x8, gv), dv[dt]);
          dk[dt] = mfma16b<F16>(df, __builtin_bit_cast(bf16x8, qv), dk[dt]);
        }
      }
    }
  }
  if (!active) return;
  if (g16) {
    store_grad16<F16>(dk, g16, nqkv, row0, key0, T, (nq + j) * 64, true, cosT, sinT, lane);
    store_grad16<F16>(dv, g16, nqkv, row0, key0, T, (nq + nkv + j) * 64, false, cosT, sinT, lane);
    return;
  }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = key0 + l4 * 4 + e;
      if (key < T) {
        g32[(row0 + key) * nqkv + (nq + j) * 64 + dt * 16 + l15] = dk[dt][e];
        g32[(row0 + key) * nqkv + (nq + nkv + j) * 64 + dt * 16 + l15] = dv[dt][e];
      }
    }
}

// G3 fp32 [M, 3 * nq * 64] = dQ | dK per QUERY head | dV per QUERY head  ->  bf16 [M, (nq + 2 nkv) * 64]: the query heads of
// a group are summed into their key / value head, q and k get the transposed RoPE rotation (rope_bwd_pack_kernel).
__global__ __launch_bounds__(256) void gqa_rope_bwd_pack_kernel(const float* __restrict__ G3, bf16_t* __restrict__ out,
                                                                const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                                long M, int nq, int nkv, int L) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int heads = nq + 2 * nkv;
  if (idx >= M * heads * 32) return;
  const int w = (int)(idx & 31);
  const int oh = (int)((idx >> 5) % heads);
  const long row = idx / ((long)heads * 32);
  const int grp = nq / nkv;
  const float* g = G3 + row * (3L * nq * 64);
  float a = 0.f, b = 0.f;
  bool rot = true;
  if (oh < nq) {
    a = g[oh * 64 + w];
    b = g[oh * 64 + w + 32];
  } else {
    const bool isv = oh >= nq + nkv;
    const int j = oh - nq - (isv ? nkv : 0);
    const float* src = g + (isv ? 2 : 1) * nq * 64 + j * grp * 64;
    for (int q = 0; q < grp; ++q) {
      a += src[q * 64 + w];
      b += src[q * 64 + w + 32];
    }
    rot = !isv;
  }
  float o1 = a, o2 = b;
  if (rot) {
    const int pos = (int)(row % L);
    const float cs = cosT[pos * 32 + w], sn = sinT[pos * 32 + w];
    o1 = a * cs + b * sn;
    o2 = b * cs - a * sn;
  }
  out[row * (heads * 64L) + oh * 64 + w] = f32_to_bf16(o1);
  out[row * (heads * 64L) + oh * 64 + w + 32] = f32_to_bf16(o2);
}


// ---------------------------------------------------------------------------
// torch.nn.utils.clip_grad_norm_(params, max_norm) on the flat gradient vector (modify_scripts/modify_train.py:1192), without
// a host round trip and with a fixed summation order: 1024 block partials of sum(g^2), one block adds them in index order
// and leaves  grad_scale * min(1, max_norm / (norm + 1e-6))  in scratch[1024] (norm = grad_scale * ||g|| in scratch[1025]),
// a third launch scales g.  grad_scale = 1 / world turns the SUM-all-reduced gradient of a data-parallel step into the
// DDP-averaged one FIRST, so that the threshold applies to the gradient the reference clips (modify_train.py:1192 clips
// after DDP's averaging all-reduce).
// ---------------------------------------------------------------------------
constexpr int CLIP_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
  __shared__ float sh[4];
  const long per = (n + CLIP_BLOCKS - 1) / CLIP_BLOCKS;
  const long lo = (long)blockIdx.x * per, hi = min(lo + per, n);
  float a = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) a = fmaf(g[i], g[i], a);
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(64) void clip_scale_kernel(float* __restrict__ part, float max_norm, float grad_scale) {
  float a = 0.f;
  for (int i = threadIdx.x; i < CLIP_BLOCKS; i += 64) a += part[i];  // lane l: partials l, l + 64, ... in order
  a = wave_sum(a);
  if (threadIdx.x == 0) {
    const float norm = grad_scale * sqrtf(a);
    part[CLIP_BLOCKS] = grad_scale * fminf(1.f, max_norm / (norm + 1e-6f));
    part[CLIP_BLOCKS + 1] = norm;
  }
}
__global__ __launch_bounds__(256) void scale_by_kernel(float* __restrict__ g, long n, const float* __restrict__ scale) {
  const float sc = *scale;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= sc;
}

static size_t attn_bwd_lds(int T) {
  return (size_t)2 * AB_QB * T * 4 + (size_t)2 * AB_QB * (AB_HD + 1) * 4 + (size_t)2 * T * AB_LDK * 2;
}

// 1 / rms of a token from the fused RMSNorm's partial sums of squares, added in index order (as row_rscale of the GEMMs)
__device__ __forceinline__ float part_rscale(const float* __restrict__ q, int npart, float inv_h, float eps) {
  float ss = 0.f;
  for (int i = 0; i < npart; i += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(q + i);
    ss += v[0];
    ss += v[1];
    ss += v[2];
    ss += v[3];
  }
  return rsqrtf(ss * inv_h + eps);
}

// ---------------------------------------------------------------------------
// dA of both adapters in ONE pass over the taped residual stream (LoRA-trainable variant):
//     dA_q[r][n] = gamma[n] * sum_m g_t[m][r] * drop_q(rs[m] h[m][n]),   dA_v likewise with g_t[m][16 + r] and drop_v
// h = the layer's 16-bit input stream, rs = 1 / rms of its rows (from the forward's partial sums), gamma = the input
// norm's gain, drop = the forward's masks regenerated (Philox, one call per octet and site) -- i.e. g_t^T (mask * rmsnorm(h))
// without ever writing rmsnorm(h), its two dropped copies or reading them back (round 3 before this: one norm kernel, two mask
// kernels and two wgrad_tn launches per layer, ~230 MB of traffic instead of 33).  Layout of the work as wgrad_tn_kernel:
// a workgroup owns 256 columns and a range of tokens, stages 32 tokens per step transposed in LDS (two tokens per dword),
// partial sums meet in dA through fp32 atomics.
// ---------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void lora_wgrad_a_kernel(const bf16_t* __restrict__ X, const float* __restrict__ part, int npart,
                                                           float inv_h, float eps, const float* __restrict__ gamma,
                                                           const bf16_t* __restrict__ Gt, float* __restrict__ dA, long ldc, int M,
                                                           int H, int m_per_wg, DropoutP dq, DropoutP dv) {
  constexpr int XS = 40;
  __shared__ __attribute__((aligned(16))) bf16_t xq[256 * XS];
  __shared__ __attribute__((aligned(16))) bf16_t xv[256 * XS];
  __shared__ __attribute__((aligned(16))) bf16_t gt[32 * XS];
  // 1 / rms of the 32 tokens of a step.  It goes to the STREAM operand (rs h: rms 1 whatever the layer), not to g_t: the
  // gradient operand sits under the backward's power-of-two scale near the top of the half range, and 1 / rms of the
  // embedding rows layer 0 reads is ~50 -- g_t rs left the range there (inf, then NaN in the product)
  __shared__ float rs_s[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h0 = blockIdx.x * 256;
  const int mbeg = blockIdx.y * m_per_wg, mend = min(mbeg + m_per_wg, M);
  const int r16 = lane & 15, kq = lane >> 4;
  const bool drop = dq.p > 0.f;
  const bf16_t* xvp = drop ? xv : xq;
  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tp = tid >> 4, cc = (tid & 15) * 16;  // this thread stages tokens 2 tp, 2 tp + 1, columns cc .. cc + 15
  const int grow = tid >> 2, gc0 = (tid & 3) * 8;  // ... and (tid < 128) token grow, g_t columns gc0 .. gc0 + 7
  for (int m0 = mbeg; m0 < mend; m0 += 32) {
    {
      u32x4 v[2][2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int m = m0 + 2 * tp + r;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          v[r][o] = u32x4{0u, 0u, 0u, 0u};
          if (m < mend && h0 + cc + 8 * o < H) v[r][o] = *reinterpret_cast<const u32x4*>(X + (long)m * H + h0 + cc + 8 * o);
        }
      }
      u32x4 gv = {0u, 0u, 0u, 0u};
      if (tid < 128 && m0 + grow < mend) gv = *reinterpret_cast<const u32x4*>(Gt + (long)(m0 + grow) * 64 + gc0);
      if (tid < 32) rs_s[tid] = m0 + tid < mend ? part_rscale(part + (long)(m0 + tid) * npart, npart, inv_h, eps) : 0.f;
      __syncthreads();
      const float rs2[2] = {rs_s[2 * tp], rs_s[2 * tp + 1]};
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        u32x4 wq[2], wv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float sq[8], sv[8];
          if (drop) {  // (uniform) the forward's masks: element (m, n) of site q / v, one generator call per octet
            const unsigned long long oct =
                ((unsigned long long)(m0 + 2 * tp + r) * (unsigned long long)H + (unsigned long long)(h0 + cc + 8 * o)) >> 3;
            dropout_oct(dq, oct, sq);
            dropout_oct(dv, oct, sv);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = from16_lo<F16>(v[r][o][e]) * rs2[r], hi = from16_hi<F16>(v[r][o][e]) * rs2[r];
            wq[r][e] = drop ? pack16x2<F16>(lo * sq[2 * e], hi * sq[2 * e + 1]) : pack16x2<F16>(lo, hi);
            if (drop) wv[r][e] = pack16x2<F16>(lo * sv[2 * e], hi * sv[2 * e + 1]);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // columns cc + 8 o + 2 e (+ 1): tokens 2 tp | 2 tp + 1 in one dword
          const int c = cc + 8 * o + 2 * e;
          *reinterpret_cast<unsigned int*>(&xq[c * XS + 2 * tp]) = (wq[0][e] & 0xffffu) | (wq[1][e] << 16);
          *reinterpret_cast<unsigned int*>(&xq[(c + 1) * XS + 2 * tp]) = (wq[0][e] >> 16) | (wq[1][e] & 0xffff0000u);
          if (drop) {
            *reinterpret_cast<unsigned int*>(&xv[c * XS + 2 * tp]) = (wv[0][e] & 0xffffu) | (wv[1][e] << 16);
            *reinterpret_cast<unsigned int*>(&xv[(c + 1) * XS + 2 * tp]) = (wv[0][e] >> 16) | (wv[1][e] & 0xffff0000u);
          }
        }
      }
      if (tid < 128) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          gt[(gc0 + 2 * e) * XS + grow] = static_cast<bf16_t>(gv[e] & 0xffffu);
          gt[(gc0 + 2 * e + 1) * XS + grow] = static_cast<bf16_t>(gv[e] >> 16);
        }
      }
    }
    __syncthreads();
    // wave: columns h0 + 64 wave .. + 63; A = g_t^T rows (q adapter: 0..15, v adapter: 16..31), B = masked (rs h)^T rows, K = 32 tokens
    const u32x4 aq = *reinterpret_cast<const u32x4*>(&gt[r16 * XS + kq * 8]);
    const u32x4 av = *reinterpret_cast<const u32x4*>(&gt[(16 + r16) * XS + kq * 8]);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int row = (wave * 64 + b * 16 + r16) * XS + kq * 8;
      const u32x4 bq = *reinterpret_cast<const u32x4*>(&xq[row]);
      const u32x4 bv = *reinterpret_cast<const u32x4*>(&xvp[row]);
      acc[0][b] = mfma16b<F16>(__builtin_bit_cast(bf16x8, aq), __builtin_bit_cast(bf16x8, bq), acc[0][b]);
      acc[1][b] = mfma16b<F16>(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[1][b]);
    }
    __syncthreads();
  }
  // acc[a][b]: adapter a, rank row 4 kq + e, column h0 + 64 wave + 16 b + r16
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int h = h0 + wave * 64 + b * 16 + r16;
      if (h >= H) continue;
      const float gm = gamma[h];
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(dA + (long)(a * 16 + 4 * kq + e) * ldc + h, acc[a][b][e] * gm);
    }
}

// ---------------------------------------------------------------------------
// Weight gradient with a SKINNY output, without physical transposes:
//     C[n][h] (+)= sum_m G[m][g0 + n] * X[m][h]          n < 16 * NB (NB <= 4),  h < H,  contraction over the M tokens
// (LoRA: dA = g_t^T x with n = 16 per adapter, dB^T = t^T g_qkv with n = 32).  Both operands are token-major, so the
// MFMA's K index (the token) runs DOWN their rows; round 1 transposed both with tcavt_transpose16 first (a 32 MB read +
// write per operand and layer).  Here a workgroup stages 32 tokens x 256 columns of X and 32 x 16 NB of G per step and
// writes them into LDS already transposed ([column][token], 16-bit scattered stores), so the fragments are plain
// 16-byte row reads; X is read exactly once.  The token range is split over gridDim.y workgroups, whose partial sums
// meet in C through fp32 atomics (C zeroed or holding an earlier contribution; same reproducibility class as the other
// weight gradients: up to the float summation order).  trans_out: store C^T ([h][ldc] layout) instead.
// ---------------------------------------------------------------------------
// GF16: G (and then X as well) is IEEE half and the products run on the f16 MFMA; otherwise G is bf16 and an fp16 X
// (XF16: a forward activation of the fp16 storage contract) is converted while it is staged
template <bool XF16, bool GF16 = false>
__global__ __launch_bounds__(256) void wgrad_tn_kernel(const bf16_t* __restrict__ G, long ldg, int g0, int NB,
                                                       const bf16_t* __restrict__ X, long ldx, float* __restrict__ C,
                                                       long ldc, int M, int H, int m_per_wg, int trans_out,
                                                       const float* __restrict__ rs_part, int rs_npart, float rs_inv_h,
                                                       float rs_eps) {
  // rs_part (optional): G's rows are multiplied by 1 / rms of their token while they are staged -- the fused RMSNorm's partial
  // sums of squares [M][rs_npart], added in index order as the forward's GEMMs do (dB of the adapters: the tape holds the
  // un-normalised t, the graph's t is rs * t)
  constexpr int XS = 40;  // LDS row stride in 16-bit elements (32 tokens + pad; 80 bytes: 16-byte aligned rows)
  __shared__ __attribute__((aligned(16))) bf16_t xt[256 * XS];
  __shared__ __attribute__((aligned(16))) bf16_t gt[64 * XS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h0 = blockIdx.x * 256;
  const int mbeg = blockIdx.y * m_per_wg, mend = min(mbeg + m_per_wg, M);
  const int r16 = lane & 15, kq = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int xrow = tid >> 3, xc0 = (tid & 7) * 32;  // this thread stages token xrow, columns xc0 .. xc0 + 31
  const int grow = tid >> 3, gc0 = (tid & 7) * 8;   // ... and token grow, G columns gc0 .. gc0 + 7 (when < 16 NB)
  for (int m0 = mbeg; m0 < mend; m0 += 32) {
    {
      const int m = m0 + xrow;
      const bool ok = m < mend;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        u32x4 v = {0u, 0u, 0u, 0u};
        const int col = h0 + xc0 + c * 8;
        if (ok && col < H) v = *reinterpret_cast<const u32x4*>(X + (long)m * ldx + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bf16_t lo = static_cast<bf16_t>(v[e] & 0xffffu), hi = static_cast<bf16_t>(v[e] >> 16);
          if constexpr (XF16 && !GF16) {  // forward activation in fp16, gradient operand in bf16: one type for the MFMA
            lo = f32_to_bf16(f16_to_f32(lo));
            hi = f32_to_bf16(f16_to_f32(hi));
          }
          xt[(xc0 + c * 8 + 2 * e) * XS + xrow] = lo;
          xt[(xc0 + c * 8 + 2 * e + 1) * XS + xrow] = hi;
        }
      }
      if (gc0 < 16 * NB) {
        u32x4 v = {0u, 0u, 0u, 0u};
        const int mg = m0 + grow;
        if (mg < mend) v = *reinterpret_cast<const u32x4*>(G + (long)mg * ldg + g0 + gc0);
        if (rs_part && mg < mend) {
          const float rs = part_rscale(rs_part + (long)mg * rs_npart, rs_npart, rs_inv_h, rs_eps);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pack16x2<GF16>(from16_lo<GF16>(v[e]) * rs, from16_hi<GF16>(v[e]) * rs);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          gt[(gc0 + 2 * e) * XS + grow] = static_cast<bf16_t>(v[e] & 0xffffu);
          gt[(gc0 + 2 * e + 1) * XS + grow] = static_cast<bf16_t>(v[e] >> 16);
        }
      }
    }
    __syncthreads();
    // wave: columns h0 + 64 wave .. + 63 (four 16-column blocks); A = G^T rows (n), B = X^T rows (h), K = 32 tokens
    u32x4 bf[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) bf[b] = *reinterpret_cast<const u32x4*>(&xt[(wave * 64 + b * 16 + r16) * XS + kq * 8]);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (a < NB) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(&gt[(a * 16 + r16) * XS + kq * 8]);
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = mfma16b<GF16>(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf[b]), acc[a][b]);
      }
    }
    __syncthreads();
  }
  // acc[a][b]: rows n = 16 a + 4 kq .. + 3, column h = h0 + 64 wave + 16 b + r16
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (a >= NB) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int h = h0 + wave * 64 + b * 16 + r16;
      if (h >= H) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = a * 16 + 4 * kq + e;
        float* dst = trans_out ? C + (long)h * ldc + n : C + (long)n * ldc + h;
        atomicAdd(dst, acc[a][b][e]);
      }
    }
  }
}

// The four adapter gradients of one layer leave the two scratch matrices (dA [64][H]: rows 0.. = A_q, 16.. = A_v; dB [nqkv][64]:
// q rows x columns 0.., v rows x columns 16..) for the caller's tensors, times *scale, in ONE launch that also zeroes the
// scratch for the next layer's atomically accumulated sums (was: two memsets + four scale-and-copy launches per layer on the
// leaf queue, 96 launches per step).
__global__ __launch_bounds__(256) void adapter_grads_out_kernel(float* __restrict__ dA, float* __restrict__ dB, float* __restrict__ gAq,
                                                                float* __restrict__ gAv, float* __restrict__ gBq, float* __restrict__ gBv,
                                                                int H, int nqkv, int q_rows, int v_row0, int r,
                                                                const float* __restrict__ scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nA = (long)64 * H;
  const float sc = scale ? *scale : 1.f;
  if (idx < nA) {
    const int i = (int)(idx / H), j = (int)(idx - (long)i * H);
    const float v = dA[idx];
    dA[idx] = 0.f;
    if (i < r) gAq[(long)i * H + j] = v * sc;
    else if (i >= 16 && i < 16 + r) gAv[(long)(i - 16) * H + j] = v * sc;
    return;
  }
  const long k = idx - nA;
  if (k >= (long)nqkv * 64) return;
  const int i = (int)(k >> 6), j = (int)(k & 63);
  const float v = dB[k];
  dB[k] = 0.f;
  if (i < q_rows && j < r) gBq[(long)i * r + j] = v * sc;
  else if (i >= v_row0 && j >= 16 && j < 16 + r) gBv[(long)(i - v_row0) * r + (j - 16)] = v * sc;
}

// ---------------------------------------------------------------------------
// Power-of-two scale of the fp16 backward: S = 2^k with  max|g| * S  in [target / 2, target]  (g = the 16-bit gradient(s) the
// backward starts from; bf16, so they cannot overflow themselves).  The backward is linear in g: every 16-bit gradient
// tensor downstream carries the factor S, the fp32 weight gradients are multiplied by 1 / S at the end.  Decided on the
// device (no host synchronisation): scale[0] = S, scale[1] = 1 / S.  All-zero or non-finite g: S = 1 (a non-finite
// gradient then reaches the gated optimizer as it did before).
// ---------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void grad_amax_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, long n8,
                                                        unsigned int* __restrict__ amax_bits) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const u32x4 v = reinterpret_cast<const u32x4*>(a)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(from16_lo<F16>(v[e])), fabsf(from16_hi<F16>(v[e]))));
    if (b) {
      const u32x4 w = reinterpret_cast<const u32x4*>(b)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(from16_lo<F16>(w[e])), fabsf(from16_hi<F16>(w[e]))));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax_bits, __float_as_uint(m));  // (non-negative floats order as their bits; NaN is dropped by fmaxf)
}
__global__ void grad_scale_set_kernel(unsigned int* __restrict__ amax_bits, float target, float* __restrict__ scale,
                                      const int* __restrict__ backoff) {
  const float m = __uint_as_float(*amax_bits);
  float s = 1.f;
  if (m > 0.f && m < 3.0e38f) {
    int e;
    (void)frexpf(target / m, &e);       // target / m = f * 2^e, f in [0.5, 1)
    // backoff (tcavt_adamw_gated's ctl[6]): binary orders of extra headroom, raised by four whenever an update was skipped for
    // a non-finite gradient norm -- the walk grew the gradient by more than the 2^8 the target leaves -- and given back slowly
    e = max(-60, min(60, e - 1 - (backoff ? *backoff : 0)));       // 2^(e-1) <= target / m
    s = ldexpf(1.f, e);
  }
  scale[0] = s;
  scale[1] = 1.f / s;
  *amax_bits = 0u;  // (re-armed for the next call)
}

}  // namespace tcavt

using namespace tcavt;

extern "C" int tcavt_silu_mul_bwd(const void* gu_bf16, const void* g_act_bf16, void* g_gu_bf16, int64_t M, int I, int dtype16,
                                  tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(is16(dtype16), "silu_mul_bwd: dtype16 must be TCAVT_BF16 or TCAVT_F16");
  TCAVT_CHECK_ARG(gu_bf16 && g_act_bf16 && g_gu_bf16 && M > 0 && I > 0 && I % 16 == 0, "silu_mul_bwd: bad args (I %% 16 == 0)");
  TCAVT_CHECK_ARG(aligned16(gu_bf16) && aligned16(g_act_bf16) && aligned16(g_gu_bf16), "silu_mul_bwd: 16-byte alignment required");
  const long n = (long)M * (I / 16);
  auto kfn = dtype16 == TCAVT_F16 ? silu_mul_bwd_kernel<true> : silu_mul_bwd_kernel<false>;
  hipLaunchKernelGGL(kfn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const bf16_t*>(gu_bf16), static_cast<const bf16_t*>(g_act_bf16),
                     static_cast<bf16_t*>(g_gu_bf16), (long)M, I);
  TCAVT_CHECK_LAUNCH("silu_mul_bwd");
  return TCAVT_OK;
}

extern "C" int tcavt_rmsnorm_bwd(const void* x, const float* gamma, const void* gy_bf16, const void* gy2_bf16, float eps,
                                 float* gx, void* gx_bf16, int accumulate, int M, int H, int gy_dtype, int out_dtype,
                                 const float* gy_scale, int x_dtype, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(is16(gy_dtype) && (gx_bf16 == nullptr || is16(out_dtype)), "rmsnorm_bwd: gy_dtype / out_dtype must be TCAVT_BF16 or TCAVT_F16");
  TCAVT_CHECK_ARG(x_dtype == TCAVT_F32 || is16(x_dtype), "rmsnorm_bwd: x_dtype must be TCAVT_F32, TCAVT_F16 or TCAVT_BF16");
  TCAVT_CHECK_ARG(x && gamma && gy_bf16 && gx && M > 0 && H > 0 && H % 8 == 0, "rmsnorm_bwd: bad args (H %% 8 == 0)");
  TCAVT_CHECK_ARG(aligned16(x) && aligned16(gamma) && aligned16(gy_bf16) && aligned16(gy2_bf16) && aligned16(gx) &&
                      aligned16(gx_bf16), "rmsnorm_bwd: 16-byte alignment required");
  const bool gf = gy_dtype == TCAVT_F16, of = out_dtype == TCAVT_F16;
  decltype(&rmsnorm_bwd_kernel<false, false, 0>) kfn;
  if (x_dtype == TCAVT_F32)
    kfn = gf ? (of ? rmsnorm_bwd_kernel<true, true, 0> : rmsnorm_bwd_kernel<true, false, 0>)
             : (of ? rmsnorm_bwd_kernel<false, true, 0> : rmsnorm_bwd_kernel<false, false, 0>);
  else if (x_dtype == TCAVT_F16)
    kfn = gf ? (of ? rmsnorm_bwd_kernel<true, true, 1> : rmsnorm_bwd_kernel<true, false, 1>)
             : (of ? rmsnorm_bwd_kernel<false, true, 1> : rmsnorm_bwd_kernel<false, false, 1>);
  else
    kfn = gf ? (of ? rmsnorm_bwd_kernel<true, true, 2> : rmsnorm_bwd_kernel<true, false, 2>)
             : (of ? rmsnorm_bwd_kernel<false, true, 2> : rmsnorm_bwd_kernel<false, false, 2>);
  hipLaunchKernelGGL(kfn, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, gamma,
                     static_cast<const bf16_t*>(gy_bf16), static_cast<const bf16_t*>(gy2_bf16), eps, gx,
                     static_cast<bf16_t*>(gx_bf16), accumulate, M, H, gy_scale);
  TCAVT_CHECK_LAUNCH("rmsnorm_bwd");
  return TCAVT_OK;
}

extern "C" int tcavt_rope_bwd_pack(const float* g32, void* out_bf16, const float* rope_cos, const float* rope_sin, int64_t M,
                                   int ncols, int rope_cols, int L, int dtype16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(is16(dtype16), "rope_bwd_pack: dtype16 must be TCAVT_BF16 or TCAVT_F16");
  TCAVT_CHECK_ARG(g32 && out_bf16 && rope_cos && rope_sin && M > 0 && L > 0, "rope_bwd_pack: bad args");
  TCAVT_CHECK_ARG(ncols % 64 == 0 && rope_cols % 64 == 0 && rope_cols <= ncols, "rope_bwd_pack: columns come in heads of 64");
  const long n = (long)M * (ncols / 2);
  auto kfn = dtype16 == TCAVT_F16 ? rope_bwd_pack_kernel<true> : rope_bwd_pack_kernel<false>;
  hipLaunchKernelGGL(kfn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     g32, static_cast<bf16_t*>(out_bf16), rope_cos, rope_sin, (long)M, ncols, rope_cols, L);
  TCAVT_CHECK_LAUNCH("rope_bwd_pack");
  return TCAVT_OK;
}

extern "C" int tcavt_attn_causal_gqa_bwd(const void* qkv_bf16, const void* dO_bf16, float* g32, const int32_t* kv_len, int B,
                                         int T, int nq, int nkv, int head_dim, float scale, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(qkv_bf16 && dO_bf16 && g32 && kv_len && B > 0 && T > 0, "attn_causal_gqa_bwd: bad args");
  TCAVT_CHECK_ARG(head_dim == 64 && nkv > 0 && nq % nkv == 0, "attn_causal_gqa_bwd: head_dim 64 and nq %% nkv == 0 required");
  const size_t lds = attn_bwd_lds(T);
  TCAVT_CHECK_ARG(lds <= 160 * 1024, "attn_causal_gqa_bwd: T=%d needs %zu bytes of LDS (limit 160 KB, T <= 280)", T, lds);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_causal_gqa_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      tcavt::set_error("attn_causal_gqa_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return TCAVT_ERR_HIP;
    }
    attr_set = true;
  }
  const int nqb = (T + AB_QB - 1) / AB_QB;
  hipLaunchKernelGGL(attn_causal_gqa_bwd_kernel, dim3((unsigned)(B * nq * nqb)), dim3(256), lds, static_cast<hipStream_t>(stream),
                     static_cast<const bf16_t*>(qkv_bf16), static_cast<const bf16_t*>(dO_bf16), g32, kv_len, T, nq, nkv, scale,
                     nqb);
  TCAVT_CHECK_LAUNCH("attn_causal_gqa_bwd");
  return TCAVT_OK;
}

extern "C" int tcavt_causal_softmax_bwd_rows(const float* S, const float* dP, void* P_bf16, void* dS_bf16,
                                             const int32_t* kv_len, int B, int T, int Tp, int nq, float scale,
                                             tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(S && dP && P_bf16 && dS_bf16 && kv_len && B > 0 && T > 0 && Tp >= T && Tp % 4 == 0 && nq > 0, "causal_softmax_bwd_rows: bad args");
  TCAVT_CHECK_ARG(aligned16(S) && aligned16(dP) && aligned16(P_bf16) && aligned16(dS_bf16), "causal_softmax_bwd_rows: 16-byte alignment required");
  const long rows = (long)B * nq * T;
  hipLaunchKernelGGL(causal_softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), S, dP, static_cast<bf16_t*>(P_bf16), static_cast<bf16_t*>(dS_bf16),
                     kv_len, T, Tp, nq, scale, rows);
  TCAVT_CHECK_LAUNCH("causal_softmax_bwd_rows");
  return TCAVT_OK;
}

extern "C" int tcavt_gqa_rope_bwd_pack(const float* G3, void* out_bf16, const float* rope_cos, const float* rope_sin, int64_t M,
                                       int nq, int nkv, int head_dim, int L, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(G3 && out_bf16 && rope_cos && rope_sin && M > 0 && L > 0, "gqa_rope_bwd_pack: bad args");
  TCAVT_CHECK_ARG(head_dim == 64 && nkv > 0 && nq % nkv == 0, "gqa_rope_bwd_pack: head_dim 64 and nq %% nkv == 0 required");
  const long n = (long)M * (nq + 2 * nkv) * 32;
  hipLaunchKernelGGL(gqa_rope_bwd_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     G3, static_cast<bf16_t*>(out_bf16), rope_cos, rope_sin, (long)M, nq, nkv, L);
  TCAVT_CHECK_LAUNCH("gqa_rope_bwd_pack");
  return TCAVT_OK;
}

extern "C" int tcavt_causal_softmax_bwd_tiles(const float* S, const float* dP, void* dS_bf16, void* PT_bf16, void* dST_bf16,
                                              const int32_t* kv_len, int B, int T, int Tp, int nq, float scale,
                                              tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(S && dP && dS_bf16 && PT_bf16 && dST_bf16 && kv_len && B > 0 && T > 0 && nq > 0, "causal_softmax_bwd_tiles: bad args");
  TCAVT_CHECK_ARG(Tp >= T && Tp - T < 64 && Tp % 64 == 0, "causal_softmax_bwd_tiles: Tp must be T rounded up to a multiple of 64");
  TCAVT_CHECK_ARG(aligned16(S) && aligned16(dP), "causal_softmax_bwd_tiles: 16-byte alignment required");
  hipLaunchKernelGGL(causal_softmax_bwd_tiles_kernel, dim3((unsigned)((long)B * nq * (Tp / 64))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), S, dP, static_cast<bf16_t*>(dS_bf16), static_cast<bf16_t*>(PT_bf16),
                     static_cast<bf16_t*>(dST_bf16), kv_len, T, Tp, nq, scale);
  TCAVT_CHECK_LAUNCH("causal_softmax_bwd_tiles");
  return TCAVT_OK;
}

extern "C" int tcavt_attn_bwd_scores(const void* qkv_bf16, const void* dO_bf16, void* dS_bf16, void* PT_bf16, void* dST_bf16,
                                     float* dQ, int64_t ld_dq, float* stats, const int32_t* kv_len, int B, int T, int Tp,
                                     int nq, int nkv, int head_dim, float scale, int dtype16, const float* lse, const void* att,
                                     tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(qkv_bf16 && dO_bf16 && kv_len && B > 0 && T > 0 && is16(dtype16), "attn_bwd_scores: bad args");
  TCAVT_CHECK_ARG((lse != nullptr) == (att != nullptr) && aligned16(att), "attn_bwd_scores: lse and att come together (att 16-byte aligned)");
  TCAVT_CHECK_ARG((PT_bf16 != nullptr) == (dST_bf16 != nullptr), "attn_bwd_scores: PT and dST come together");
  TCAVT_CHECK_ARG(PT_bf16 || stats, "attn_bwd_scores: give PT/dST (GEMM form of dK, dV) or stats (tcavt_attn_bwd_dkv)");
  TCAVT_CHECK_ARG(aligned16(stats), "attn_bwd_scores: stats must be 16-byte aligned");
  TCAVT_CHECK_ARG(dS_bf16 || dQ, "attn_bwd_scores: give dQ (computed here) or dS (for an external dQ product), or both");
  TCAVT_CHECK_ARG(!dQ || ld_dq >= (int64_t)nq * 64, "attn_bwd_scores: ld_dq must cover nq * 64 columns");
  TCAVT_CHECK_ARG(head_dim == 64 && nkv > 0 && nq % nkv == 0, "attn_bwd_scores: head_dim 64 and nq %% nkv == 0 required");
  TCAVT_CHECK_ARG(Tp >= T && Tp - T < 64 && Tp % 64 == 0, "attn_bwd_scores: Tp must be T rounded up to a multiple of 64");
  TCAVT_CHECK_ARG(aligned16(qkv_bf16) && aligned16(dO_bf16), "attn_bwd_scores: 16-byte alignment required");
  const int group = nq / nkv;
  static const bool no_resident = getenv("TCAVT_ATTN_BWD_NO_RESIDENT") != nullptr;  // (A/B switch)
  if (lse && dQ && stats && !dS_bf16 && !PT_bf16 && Tp <= 256 && ABQ_WAVES % group == 0 && !no_resident) {
    // statistics from the forward, nothing but dQ wanted: K, V, K^T of a key/value head resident in LDS, no key-block loop
    const int lds = (2 * Tp * 72 + 64 * (Tp + 4)) * 2;
    auto kq = dtype16 == TCAVT_F16 ? attn_bwd_dq_kernel<true> : attn_bwd_dq_kernel<false>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[dtype16 == TCAVT_F16]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
      if (e != hipSuccess) {
        tcavt::set_error("attn_bwd_scores: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return TCAVT_ERR_HIP;
      }
      attr_set[dtype16 == TCAVT_F16] = true;
    }
    hipLaunchKernelGGL(kq, dim3((unsigned)(B * nkv)), dim3(ABQ_WAVES * 64), lds, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(qkv_bf16), static_cast<const bf16_t*>(dO_bf16), static_cast<const bf16_t*>(att), lse,
                       dQ, (long)ld_dq, stats, kv_len, T, Tp, nq, nkv, scale, static_cast<bf16_t*>(nullptr),
                       static_cast<const float*>(nullptr), static_cast<const float*>(nullptr));
    TCAVT_CHECK_LAUNCH("attn_bwd_scores(resident)");
    return TCAVT_OK;
  }
  auto kfn = dtype16 == TCAVT_F16 ? attn_bwd_scores_kernel<true> : attn_bwd_scores_kernel<false>;
  hipLaunchKernelGGL(kfn, dim3((unsigned)((long)B * nq * (Tp / 64))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(qkv_bf16), static_cast<const bf16_t*>(dO_bf16),
                     static_cast<bf16_t*>(dS_bf16), static_cast<bf16_t*>(PT_bf16), static_cast<bf16_t*>(dST_bf16), dQ, (long)ld_dq,
                     stats, kv_len, T, Tp, nq, nkv, scale, lse, static_cast<const bf16_t*>(att));
  TCAVT_CHECK_LAUNCH("attn_bwd_scores");
  return TCAVT_OK;
}

extern "C" int tcavt_attn_bwd_dkv(const void* qkv_bf16, const void* dO_bf16, const float* stats, float* g32,
                                  const int32_t* kv_len, int B, int T, int Tp, int nq, int nkv, int head_dim, float scale,
                                  int dtype16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(qkv_bf16 && dO_bf16 && stats && g32 && kv_len && B > 0 && T > 0 && is16(dtype16), "attn_bwd_dkv: bad args");
  TCAVT_CHECK_ARG(head_dim == 64 && nkv > 0 && nq % nkv == 0, "attn_bwd_dkv: head_dim 64 and nq %% nkv == 0 required");
  TCAVT_CHECK_ARG(Tp >= T && Tp - T < 64 && Tp % 64 == 0, "attn_bwd_dkv: Tp must be T rounded up to a multiple of 64");
  TCAVT_CHECK_ARG(aligned16(qkv_bf16) && aligned16(dO_bf16) && aligned16(stats), "attn_bwd_dkv: 16-byte alignment required");
  static const bool no_resident = getenv("TCAVT_ATTN_BWD_NO_RESIDENT") != nullptr;  // (A/B switch)
  if (Tp <= 256 && !no_resident) {  // the query side of a whole head fits in LDS: no query-block loop
    const int lds = (2 * Tp * 72 + 2 * 64 * (Tp + 4)) * 2 + 3 * Tp * 4;
    auto kr = dtype16 == TCAVT_F16 ? attn_bwd_dkv_res_kernel<true> : attn_bwd_dkv_res_kernel<false>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[dtype16 == TCAVT_F16]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      if (e != hipSuccess) {
        tcavt::set_error("attn_bwd_dkv: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return TCAVT_ERR_HIP;
      }
      attr_set[dtype16 == TCAVT_F16] = true;
    }
    hipLaunchKernelGGL(kr, dim3((unsigned)(B * nkv)), dim3(1024), lds, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(qkv_bf16), static_cast<const bf16_t*>(dO_bf16), stats, g32, kv_len, T, Tp, nq, nkv,
                       scale, static_cast<bf16_t*>(nullptr), static_cast<const float*>(nullptr), static_cast<const float*>(nullptr));
    TCAVT_CHECK_LAUNCH("attn_bwd_dkv(resident)");
    return TCAVT_OK;
  }
  auto kfn = dtype16 == TCAVT_F16 ? attn_bwd_dkv_kernel<true> : attn_bwd_dkv_kernel<false>;
  hipLaunchKernelGGL(kfn, dim3((unsigned)((long)B * nkv * (Tp / 64))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(qkv_bf16), static_cast<const bf16_t*>(dO_bf16),
                     stats, g32, kv_len, T, Tp, nq, nkv, scale);
  TCAVT_CHECK_LAUNCH("attn_bwd_dkv");
  return TCAVT_OK;
}

// Whole attention backward of the LoRA-trainable variant in its resident form: two launches, 16-bit output, no fp32 scratch
extern "C" int tcavt_attn_bwd_resident(const void* qkv16, const void* dO16, const void* att16, const float* lse, void* g_qkv16,
                                       float* stats, const float* rope_cos, const float* rope_sin, const int32_t* kv_len, int B,
                                       int T, int nq, int nkv, int head_dim, float scale, int dtype16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(qkv16 && dO16 && att16 && lse && g_qkv16 && stats && rope_cos && rope_sin && kv_len && B > 0 && T > 0 && is16(dtype16),
                  "attn_bwd_resident: bad args");
  TCAVT_CHECK_ARG(head_dim == 64 && nkv > 0 && nq % nkv == 0, "attn_bwd_resident: head_dim 64 and nq %% nkv == 0 required");
  TCAVT_CHECK_ARG(tcavt_attn_bwd_resident_ok(T, nq, nkv), "attn_bwd_resident: T=%d, nq / nkv = %d outside the resident form (T <= 256, 16 %% (nq / nkv) == 0)", T, nq / nkv);
  TCAVT_CHECK_ARG(aligned16(qkv16) && aligned16(dO16) && aligned16(att16) && aligned16(stats) && aligned16(g_qkv16),
                  "attn_bwd_resident: 16-byte alignment required");
  const int Tp = (T + 63) & ~63;
  const bool f16 = dtype16 == TCAVT_F16;
  auto kq = f16 ? attn_bwd_dq_kernel<true> : attn_bwd_dq_kernel<false>;
  auto kr = f16 ? attn_bwd_dkv_res_kernel<true> : attn_bwd_dkv_res_kernel<false>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[f16]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) {
      tcavt::set_error("attn_bwd_resident: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return TCAVT_ERR_HIP;
    }
    attr_set[f16] = true;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bf16_t* q_ = static_cast<const bf16_t*>(qkv16);
  const bf16_t* g_ = static_cast<const bf16_t*>(dO16);
  hipLaunchKernelGGL(kq, dim3((unsigned)(B * nkv)), dim3(ABQ_WAVES * 64), (2 * Tp * 72 + 64 * (Tp + 4)) * 2, st, q_, g_,
                     static_cast<const bf16_t*>(att16), lse, static_cast<float*>(nullptr), 0L, stats, kv_len, T, Tp, nq, nkv, scale,
                     static_cast<bf16_t*>(g_qkv16), rope_cos, rope_sin);
  TCAVT_CHECK_LAUNCH("attn_bwd_resident(dq)");
  hipLaunchKernelGGL(kr, dim3((unsigned)(B * nkv)), dim3(1024), (2 * Tp * 72 + 2 * 64 * (Tp + 4)) * 2 + 3 * Tp * 4, st, q_, g_,
                     static_cast<const float*>(stats), static_cast<float*>(nullptr), kv_len, T, Tp, nq, nkv, scale,
                     static_cast<bf16_t*>(g_qkv16), rope_cos, rope_sin);
  TCAVT_CHECK_LAUNCH("attn_bwd_resident(dkv)");
  return TCAVT_OK;
}

extern "C" int tcavt_attn_bwd_resident_ok(int T, int nq, int nkv) {
  return T > 0 && T <= 256 && nkv > 0 && nq % nkv == 0 && ABQ_WAVES % (nq / nkv) == 0;
}

#define TCAVT_TRY(call)            \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != TCAVT_OK) return rc_; \
  } while (0)

extern "C" int tcavt_llama_stack_backward(const tcavt_llama_backward_args* a, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(a && a->layers && a->h_last && a->gamma_final && a->g_final_a && a->rope_cos && a->rope_sin && a->kv_len && a->scale &&
                      a->scale_scratch && a->g_h && a->g_hb && a->g_xn && a->g_xl && a->g_att && a->g_qkv0 && a->g_qkv1 && a->g_t0 &&
                      a->g_t1 && a->dA && a->dB && a->stats,
                  "llama_stack_backward: null pointer");
  const int B = a->B, L = a->L, H = a->H, I = a->I, nq = a->nq, nkv = a->nkv, dt = a->dtype16, r = a->lora_rank;
  TCAVT_CHECK_ARG(a->n_layers > 0 && B > 0 && L > 0 && is16(dt) && nkv > 0 && nq % nkv == 0 && r > 0 && r <= 16 && a->npart > 0,
                  "llama_stack_backward: bad shape");
  // fp16 tapes only: the 16-bit residual-stream tapes this walk reads exist for fp16 storage alone (bf16 storage keeps fp32
  // streams, whose backward is the per-launch composition of llm_backward.py); head_dim 64, adapters in 16-column groups
  TCAVT_CHECK_ARG(dt == TCAVT_F16, "llama_stack_backward: dtype16 must be TCAVT_F16 (16-bit stream tapes exist for fp16 storage only)");
  const int M = B * L, nqkv = (nq + 2 * nkv) * 64;
  TCAVT_CHECK_ARG(tcavt_attn_bwd_resident_ok(L, nq, nkv) && M % 256 == 0 && I % 256 == 0 && H % 128 == 0,
                  "llama_stack_backward: outside the fused forms (L <= 256, 16 %% (nq / nkv) == 0, M %% 256 == 0, I %% 256 == 0, H %% 128 == 0)");
  TCAVT_CHECK_ARG(!a->leaf_stream || a->events, "llama_stack_backward: a leaf stream needs the four events");
  // every layer's pointers are checked BEFORE anything is launched: an argument error must not leave half a walk enqueued
  // (adapter gradients half-written, the leaf stream forked and never joined -- under a hipGraph capture an unjoined stream)
  for (int li = 0; li < a->n_layers; ++li) {
    const tcavt_llama_bwd_layer& w = a->layers[li];
    TCAVT_CHECK_ARG(w.w_dT && w.w_guT && w.w_oT && w.w_qkvT && w.b_extT && w.a_qT && w.a_vT && w.g1 && w.g2 && w.h_in && w.h_mid && w.qkv &&
                        w.gu && w.att && w.lse && w.part && w.t && w.g_Aq && w.g_Av && w.g_Bq && w.g_Bv,
                    "llama_stack_backward: layer %d: null pointer", li);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipStream_t lf = a->leaf_stream ? static_cast<hipStream_t>(a->leaf_stream) : st;
  const bool two = lf != st;
  const bool f16 = dt == TCAVT_F16;
  const int gdt = f16 ? TCAVT_BF16 : dt;  // type of the incoming gradients
  if (f16) TCAVT_TRY(tcavt_grad_scale_pick(a->g_final_a, a->g_final_b, (int64_t)M * H, TCAVT_BF16, 256.f, a->scale, a->scale_scratch,
                                           a->scale_backoff, stream));
  const float* inv_s = a->scale + 1;
  // final norm: g_h = d(rmsnorm)(h_last) . (g_final_a + g_final_b) * S,  g_hb = its 16-bit copy
  TCAVT_TRY(tcavt_rmsnorm_bwd(a->h_last, a->gamma_final, a->g_final_a, a->g_final_b, a->rms_eps, a->g_h, a->g_hb, 0, M, H, gdt, dt,
                              f16 ? a->scale : nullptr, dt, stream));
  auto gemm = [&](const void* A, int lda, const void* W, int K, void* C, int N, float acc_scale) -> int {
    tcavt_gemm_args g = {};
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K;
    g.out_dtype = dt; g.in_dtype = dt; g.acc_scale = acc_scale;
    return tcavt_gemm_bf16(&g, stream);
  };
  auto hip_ok = [&](hipError_t rc, const char* what) -> int {
    if (rc == hipSuccess) return TCAVT_OK;
    set_error("llama_stack_backward: %s: %s", what, hipGetErrorString(rc));
    return TCAVT_ERR_HIP;
  };
  bool leaf_used[2] = {false, false};
  bool scratch_zeroed = false;
  void* const g_qkv2[2] = {a->g_qkv0, a->g_qkv1};
  void* const g_t2[2] = {a->g_t0, a->g_t1};
  // (the walk is a lambda so that a failure in the middle of it -- a launch error; the arguments were checked above -- still
  //  reaches the join below: the caller's stream then waits for whatever the leaf stream was given)
  auto walk = [&]() -> int {
  for (int li = a->n_layers - 1; li >= 0; --li) {
    const tcavt_llama_bwd_layer& w = a->layers[li];
    // ---- MLP half
    {
      tcavt_gemm_args g = {};
      g.A = a->g_hb; g.lda = H; g.W = w.w_dT; g.ldw = H; g.C = w.gu; g.ldc = 2 * I; g.M = M; g.N = I; g.K = H;
      g.out_dtype = dt; g.in_dtype = dt; g.silu_preact = w.gu; g.ld_preact = 2 * I; g.epilogue = TCAVT_EPI_SILU_BWD;
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
    }
    TCAVT_TRY(gemm(w.gu, 2 * I, w.w_guT, 2 * I, a->g_xn, H, 0.f));
    TCAVT_TRY(tcavt_rmsnorm_bwd(w.h_mid, w.g2, a->g_xn, nullptr, a->rms_eps, a->g_h, a->g_hb, 1, M, H, dt, dt, nullptr, dt, stream));
    // ---- attention half
    TCAVT_TRY(gemm(a->g_hb, H, w.w_oT, H, a->g_att, nq * 64, 0.f));
    const int par = li & 1;
    if (two && leaf_used[par])  // the leaf of layer li + 2 has read g_qkv / g_t of this parity
      TCAVT_TRY(hip_ok(hipStreamWaitEvent(st, static_cast<hipEvent_t>(a->events[2 + par]), 0), "wait(done)"));
    TCAVT_TRY(tcavt_attn_bwd_resident(w.qkv, a->g_att, w.att, w.lse, g_qkv2[par], a->stats, a->rope_cos, a->rope_sin, a->kv_len, B, L,
                                      nq, nkv, 64, 0.125f, dt, stream));
    TCAVT_TRY(gemm(g_qkv2[par], nqkv, w.b_extT, nqkv, g_t2[par], 64, a->lora_scale));
    // ---- leaf: the adapters' weight gradients (nothing downstream reads them)
    if (two) {
      TCAVT_TRY(hip_ok(hipEventRecord(static_cast<hipEvent_t>(a->events[par]), st), "record(ready)"));
      TCAVT_TRY(hip_ok(hipStreamWaitEvent(lf, static_cast<hipEvent_t>(a->events[par]), 0), "wait(ready)"));
    }
    {
      const uint32_t site = a->lora_first_site + 2u * (uint32_t)li;
      if (!scratch_zeroed) {  // (once per walk: adapter_grads_out_kernel leaves the scratch zero for the next layer)
        TCAVT_TRY(hip_ok(hipMemsetAsync(a->dA, 0, (size_t)64 * H * sizeof(float), lf), "memset(dA)"));
        TCAVT_TRY(hip_ok(hipMemsetAsync(a->dB, 0, (size_t)nqkv * 64 * sizeof(float), lf), "memset(dB)"));
        scratch_zeroed = true;
      }
      TCAVT_TRY(tcavt_lora_wgrad_a(w.h_in, w.part, a->npart, a->rms_eps, w.g1, g_t2[par], a->dA, H, M, H, a->lora_dropout_p,
                                   a->dropout_seed, site, site + 1, dt, lf));
      TCAVT_TRY(tcavt_wgrad_tn(w.t, 64, 0, 32, g_qkv2[par], nqkv, dt, a->dB, 64, M, nqkv, 1, dt, w.part, a->npart, H, a->rms_eps, lf));
      const long n_out = (long)64 * H + (long)nqkv * 64;
      hipLaunchKernelGGL(adapter_grads_out_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, lf, a->dA, a->dB, w.g_Aq, w.g_Av,
                         w.g_Bq, w.g_Bv, H, nqkv, nq * 64, (nq + nkv) * 64, r, f16 ? inv_s : nullptr);
      TCAVT_CHECK_LAUNCH("llama_stack_backward(adapter gradients)");
    }
    if (two) {
      TCAVT_TRY(hip_ok(hipEventRecord(static_cast<hipEvent_t>(a->events[2 + par]), lf), "record(done)"));
      leaf_used[par] = true;
    }
    if (li == 0 && !a->input_grad) break;
    {
      const uint32_t site = a->lora_first_site + 2u * (uint32_t)li;
      TCAVT_TRY(tcavt_lora_dgrad(g_t2[par], w.a_qT, w.a_vT, a->g_xl, M, H, a->lora_dropout_p, a->dropout_seed, site, site + 1, dt, stream));
    }
    TCAVT_TRY(gemm(g_qkv2[par], nqkv, w.w_qkvT, nqkv, a->g_xn, H, 0.f));
    TCAVT_TRY(tcavt_rmsnorm_bwd(w.h_in, w.g1, a->g_xn, a->g_xl, a->rms_eps, a->g_h, a->g_hb, 1, M, H, dt, dt, nullptr, dt, stream));
  }
  return TCAVT_OK;
  };
  const int rc_walk = walk();
  int rc_join = TCAVT_OK;
  if (two) {
    if (rc_walk != TCAVT_OK) {
      // the failing layer may have forked the leaf stream without recording its `done` event: one more record covers
      // everything that stream was given, and the join below waits for it
      if (hipEventRecord(static_cast<hipEvent_t>(a->events[2]), lf) == hipSuccess) leaf_used[0] = true;
    }
    for (int par = 0; par < 2; ++par)
      if (leaf_used[par] && hipStreamWaitEvent(st, static_cast<hipEvent_t>(a->events[2 + par]), 0) != hipSuccess && rc_walk == TCAVT_OK) {
        set_error("llama_stack_backward: join(leaf) failed");
        rc_join = TCAVT_ERR_HIP;
      }
  }
  return rc_walk != TCAVT_OK ? rc_walk : rc_join;
}

extern "C" int tcavt_clip_grad_norm(float* g, int64_t n, float max_norm, float grad_scale, float* scratch,
                                    tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(g && scratch && n > 0 && max_norm > 0.f && grad_scale > 0.f, "clip_grad_norm: bad args");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(CLIP_BLOCKS), dim3(256), 0, st, g, (long)n, scratch);
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(64), 0, st, scratch, max_norm, grad_scale);
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_by_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, (long)n, scratch + CLIP_BLOCKS);
  TCAVT_CHECK_LAUNCH("clip_grad_norm");
  return TCAVT_OK;
}

extern "C" int tcavt_wgrad_tn(const void* G, int64_t ldg, int g_col0, int n, const void* X, int64_t ldx, int x_dtype, float* C,
                              int64_t ldc, int M, int H, int trans_out, int g_dtype, const float* rs_part, int rs_npart, int rs_h,
                              float rs_eps, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(!rs_part || (rs_npart > 0 && rs_npart % 4 == 0 && rs_h > 0 && aligned16(rs_part)),
                  "wgrad_tn: rs_part needs rs_npart %% 4 == 0, rs_h > 0 and 16-byte alignment");
  const float rs_inv_h = rs_part ? 1.f / (float)rs_h : 0.f;
  TCAVT_CHECK_ARG(G && X && C && M > 0 && H > 0 && n > 0 && n <= 64 && n % 16 == 0 && g_col0 >= 0 && g_col0 % 8 == 0 &&
                      ldg % 8 == 0 && ldx % 8 == 0 && H % 8 == 0 && is16(x_dtype),
                  "wgrad_tn: n in {16, 32, 48, 64}, g_col0 / ldg / ldx / H multiples of 8");
  TCAVT_CHECK_ARG(g_dtype == 0 || g_dtype == TCAVT_BF16 || (g_dtype == TCAVT_F16 && x_dtype == TCAVT_F16),
                  "wgrad_tn: G is bf16 (g_dtype 0 / TCAVT_BF16), or fp16 together with an fp16 X");
  TCAVT_CHECK_ARG(aligned16(G) && aligned16(X), "wgrad_tn: 16-byte alignment required");
  const int gx = (H + 255) / 256;
  int split = 256 / gx;  // ~one workgroup per CU
  if (split < 1) split = 1;
  int m_per_wg = ((M + split - 1) / split + 31) / 32 * 32;
  split = (M + m_per_wg - 1) / m_per_wg;
  const dim3 grid(gx, split), block(256);
  if (g_dtype == TCAVT_F16)
    hipLaunchKernelGGL((wgrad_tn_kernel<true, true>), grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(G),
                       (long)ldg, g_col0, n / 16, static_cast<const bf16_t*>(X), (long)ldx, C, (long)ldc, M, H, m_per_wg, trans_out,
                       rs_part, rs_npart, rs_inv_h, rs_eps);
  else if (x_dtype == TCAVT_F16)
    hipLaunchKernelGGL(wgrad_tn_kernel<true>, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(G),
                       (long)ldg, g_col0, n / 16, static_cast<const bf16_t*>(X), (long)ldx, C, (long)ldc, M, H, m_per_wg, trans_out,
                       rs_part, rs_npart, rs_inv_h, rs_eps);
  else
    hipLaunchKernelGGL(wgrad_tn_kernel<false>, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(G),
                       (long)ldg, g_col0, n / 16, static_cast<const bf16_t*>(X), (long)ldx, C, (long)ldc, M, H, m_per_wg, trans_out,
                       rs_part, rs_npart, rs_inv_h, rs_eps);
  TCAVT_CHECK_LAUNCH("wgrad_tn");
  return TCAVT_OK;
}

extern "C" int tcavt_lora_wgrad_a(const void* x16, const float* part, int npart, float eps, const float* gamma, const void* g_t,
                                  float* dA, int64_t ldc, int M, int H, float dropout_p, uint64_t dropout_seed, uint32_t site_q,
                                  uint32_t site_v, int dtype16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(x16 && part && gamma && g_t && dA && M > 0 && H > 0 && H % 16 == 0 && npart > 0 && npart % 4 == 0 && ldc >= H &&
                      is16(dtype16) && dropout_p >= 0.f && dropout_p < 1.f,
                  "lora_wgrad_a: bad args (H %% 16 == 0, npart %% 4 == 0, ldc >= H, 0 <= dropout_p < 1)");
  TCAVT_CHECK_ARG(aligned16(x16) && aligned16(part) && aligned16(g_t), "lora_wgrad_a: 16-byte alignment required");
  const DropoutP dq = make_dropout(dropout_p, dropout_seed, site_q), dv = make_dropout(dropout_p, dropout_seed, site_v);
  const int gx = (H + 255) / 256;
  int split = 256 / gx;  // ~one workgroup per CU
  if (split < 1) split = 1;
  int m_per_wg = ((M + split - 1) / split + 31) / 32 * 32;
  split = (M + m_per_wg - 1) / m_per_wg;
  auto kfn = dtype16 == TCAVT_F16 ? lora_wgrad_a_kernel<true> : lora_wgrad_a_kernel<false>;
  hipLaunchKernelGGL(kfn, dim3(gx, split), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(x16), part, npart,
                     1.f / (float)H, eps, gamma, static_cast<const bf16_t*>(g_t), dA, (long)ldc, M, H, m_per_wg, dq, dv);
  TCAVT_CHECK_LAUNCH("lora_wgrad_a");
  return TCAVT_OK;
}

extern "C" int tcavt_grad_scale_pick(const void* g_a, const void* g_b, int64_t n, int dtype16, float target, float* scale,
                                     uint32_t* scratch, const int32_t* backoff, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(g_a && scale && scratch && n > 0 && n % 8 == 0 && is16(dtype16) && target > 0.f && aligned16(g_a) && aligned16(g_b),
                  "grad_scale_pick: bad args (n %% 8 == 0, 16-byte alignment, scratch = one zero-initialised uint32)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  auto kfn = dtype16 == TCAVT_F16 ? grad_amax_kernel<true> : grad_amax_kernel<false>;
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const bf16_t*>(g_a), static_cast<const bf16_t*>(g_b),
                     (long)(n / 8), scratch);
  hipLaunchKernelGGL(grad_scale_set_kernel, dim3(1), dim3(1), 0, st, scratch, target, scale, backoff);
  TCAVT_CHECK_LAUNCH("grad_scale_pick");
  return TCAVT_OK;
}
