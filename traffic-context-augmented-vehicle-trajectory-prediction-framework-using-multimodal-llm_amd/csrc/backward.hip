// Backward kernels for the part of the model that scripts/train.py actually trains
// (lane-polygon encoder + TransformerLTSF incl. its cross-attention; the MLLM is frozen:
// train.py:1140-1145), plus the fused AdamW update (train.py:1145: lr 5e-4, wd 1e-4).
//
// The heavy contractions of the backward (cross-attention projections, 2*8192*2048^2
// each for dW_k, dW_v) reuse the MFMA GEMM of gemm_bf16.hip on physically transposed
// operands (transpose16_kernel below); everything here is the small fp32 glue:
// reductions over the batch, LayerNorm / softmax / small-attention backward, the
// per-channel N-Linear blocks, the loss gradient.
#include "common.hpp"
#include "philox.hpp"

namespace tcavt {

// ---------------------------------------------------------------------------
// out[c][r] = in[r][c] for 16-bit elements, batched, zero-filling r in [rows, rows_pad).
// 64x64 tiles through LDS (+1 pad), 256 threads.
// ---------------------------------------------------------------------------
// F2B: the source is fp16 (a forward activation), the destination bf16 (operand of a gradient-side contraction, whose
// other operand is a bf16 gradient): converted on the way (round-to-nearest-even of the fp16 value).
template <bool F2B>
__global__ __launch_bounds__(256) void transpose16_kernel(const bf16_t* __restrict__ in, long ld_in,
                                                          bf16_t* __restrict__ out, long ld_out, int rows,
                                                          int cols, int rows_pad, long s_in, long s_out) {
  __shared__ bf16_t tile[64][66];
  in += blockIdx.z * s_in;
  out += blockIdx.z * s_out;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows_pad) out[(long)c * ld_out + r] = F2B ? f32_to_bf16(f16_to_f32(tile[tx][i])) : tile[tx][i];
  }
}

// fp32 [rows][cols] -> bf16 transposed [cols][rows_pad] (zero pad)
__global__ __launch_bounds__(256) void transpose_f32_bf16_kernel(const float* __restrict__ in, long ld_in,
                                                                 bf16_t* __restrict__ out, long ld_out, int rows,
                                                                 int cols, int rows_pad) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows_pad) out[(long)c * ld_out + r] = f32_to_bf16(tile[tx][i]);
  }
}

// ---------------------------------------------------------------------------
// out[n] (+)= sum_m g[m][n]   (bias gradients).  One block per 64 columns.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float ldv(const T* p);
template <>
__device__ __forceinline__ float ldv<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldv<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, long ld, float* __restrict__ out,
                                                     int M, int N, int rows_per_block) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int m_lo = blockIdx.y * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
  float s = 0.f;
  if (n < N)
    for (int m = m_lo + ty; m < m_hi; m += 4) s += ldv<T>(g + (long)m * ld + n);
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) atomicAdd(&out[n], part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]);
}

// g[i] = y[i] > 0 ? g[i] : 0      (ReLU backward against the saved post-activation)
template <typename T>
__global__ void relu_bwd_kernel(float* __restrict__ g, const T* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !(ldv<T>(y + i) > 0.f)) g[i] = 0.f;
}

// a[i] += b[i]
__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}

// ---------------------------------------------------------------------------
// LayerNorm backward (one wave per row, D <= 1024, D % 4 == 0):
//   xhat = (x - mean) * rstd;  gx = rstd * (gy*gamma - mean(gy*gamma) - xhat * mean(gy*gamma*xhat))
//   ggamma += sum_rows gy * xhat;  gbeta += sum_rows gy      (block partials -> atomics)
// x is the LayerNorm INPUT (residual already added).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ gy, float eps,
                                                            float* __restrict__ gx, float* __restrict__ ggamma,
                                                            float* __restrict__ gbeta, int M, int D) {
  extern __shared__ float sh[];  // [2][D] partial sums of this block
  float* sg = sh;
  float* sb = sh + D;
  for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) sh[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < M) {
    const float* xr = x + (long)row * D;
    const float* gr = gy + (long)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)D;
    float sq = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = xr[c] - mean; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float gg = gr[c] * gamma[c];
      a += gg;
      b += gg * (xr[c] - mean) * rstd;
    }
    a = wave_sum(a) / (float)D;
    b = wave_sum(b) / (float)D;
    for (int c = lane; c < D; c += 64) {
      const float xh = (xr[c] - mean) * rstd;
      gx[(long)row * D + c] = rstd * (gr[c] * gamma[c] - a - xh * b);
      atomicAdd(&sg[c], gr[c] * xh);
      atomicAdd(&sb[c], gr[c]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    atomicAdd(&ggamma[c], sg[c]);
    atomicAdd(&gbeta[c], sb[c]);
  }
}

// ---------------------------------------------------------------------------
// Small multi-head attention backward (fp32).  One block per (batch, head); P is recomputed
// into LDS; dP and dS reuse a second LDS array.  Lq, Lk <= 64.. (Lq*Lk*8 bytes <= 64 KiB).
//   dP = dO V^T; dS = P o (dP - rowsum(P o dP)); dQ = scale dS K; dK = scale dS^T Q; dV = P^T dO
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mha_small_bwd_kernel(const float* __restrict__ q, long ldq,
                                                            const float* __restrict__ k, long ldk,
                                                            const float* __restrict__ v, long ldv_,
                                                            const float* __restrict__ go, long ldo,
                                                            float* __restrict__ gq, float* __restrict__ gk,
                                                            float* __restrict__ gv, long ldg,
                                                            const int* __restrict__ key_len, int Lq, int Lk, int nh,
                                                            int dh, float scale, DropoutP drop) {
  // With attention-weight dropout (train mode): the forward used P' = P o m / (1 - p) in P' V, m drawn from
  // (seed, site, ((b nh + h) Lq + i) Lk + j) exactly as tcavt_mha does.  Then dV = P'^T dO, dP = (dO V^T) o m / (1 - p),
  // dS = P o (dP - sum_j P_j dP_j): the softmax backward needs the UN-dropped P, which is recomputed here anyway.
  extern __shared__ float sm[];
  float* P = sm;             // [Lq][Lk]
  float* dS = sm + Lq * Lk;  // [Lq][Lk]
  const int b = blockIdx.x / nh, h = blockIdx.x % nh;
  const int klen = key_len ? min(key_len[b], Lk) : Lk;
  const float* qb = q + (long)b * Lq * ldq + h * dh;
  const float* kb = k + (long)b * Lk * ldk + h * dh;
  const float* vb = v + (long)b * Lk * ldv_ + h * dh;
  const float* gob = go + (long)b * Lq * ldo + h * dh;
  float* gqb = gq + (long)b * Lq * ldg + h * dh;
  float* gkb = gk + (long)b * Lk * ldg + h * dh;
  float* gvb = gv + (long)b * Lk * ldg + h * dh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // scores and dP
  for (int ij = tid; ij < Lq * Lk; ij += 256) {
    const int i = ij / Lk, j = ij - i * Lk;
    float s = 0.f, d = 0.f;
    if (j < klen) {
      for (int e = 0; e < dh; ++e) {
        s = fmaf(qb[(long)i * ldq + e], kb[(long)j * ldk + e], s);
        d = fmaf(gob[(long)i * ldo + e], vb[(long)j * ldv_ + e], d);
      }
    }
    P[ij] = j < klen ? s * scale : -1e30f;
    dS[ij] = d;
  }
  __syncthreads();
  for (int i = wave; i < Lq; i += 4) {
    float* row = P + i * Lk;
    float* drow = dS + i * Lk;
    float m = -1e30f;
    for (int j = lane; j < Lk; j += 64) m = fmaxf(m, row[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < Lk; j += 64) {
      const float e = row[j] > -1e29f ? __expf(row[j] - m) : 0.f;
      row[j] = e;
      s += e;
    }
    s = wave_sum(s);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    float dot = 0.f;
    const unsigned long long base = ((unsigned long long)blockIdx.x * Lq + i) * (unsigned long long)Lk;
    for (int j = lane; j < Lk; j += 64) {
      row[j] *= inv;
      if (drop.p > 0.f) drow[j] *= dropout_one(drop, base + j);
      dot += row[j] * drow[j];
    }
    dot = wave_sum(dot);
    for (int j = lane; j < Lk; j += 64) {
      drow[j] = row[j] * (drow[j] - dot) * scale;
      if (drop.p > 0.f) row[j] *= dropout_one(drop, base + j);  // P' for dV below
    }
  }
  __syncthreads();
  for (int id = tid; id < Lq * dh; id += 256) {  // dQ
    const int i = id / dh, e = id - i * dh;
    float a = 0.f;
    for (int j = 0; j < klen; ++j) a = fmaf(dS[i * Lk + j], kb[(long)j * ldk + e], a);
    gqb[(long)i * ldg + e] = a;
  }
  for (int id = tid; id < Lk * dh; id += 256) {  // dK, dV
    const int j = id / dh, e = id - j * dh;
    float a = 0.f, c = 0.f;
    if (j < klen) {
      for (int i = 0; i < Lq; ++i) {
        a = fmaf(dS[i * Lk + j], qb[(long)i * ldq + e], a);
        c = fmaf(P[i * Lk + j], gob[(long)i * ldo + e], c);
      }
    }
    gkb[(long)j * ldg + e] = a;
    gvb[(long)j * ldg + e] = c;
  }
}

// Same computation with q, k, v, dO of the (sample, head) staged in LDS (rows padded to dh + 1 floats) and the inner
// products unrolled so that their LDS loads are issued in batches: one workgroup of 4 waves runs alone on its CU, and
// with one load in flight at a time the global-memory version above spent ~75 us of pure latency on a 64 x 64 x 16
// lane-polygon head.
__global__ __launch_bounds__(256) void mha_small_bwd_lds_kernel(const float* __restrict__ q, long ldq,
                                                                const float* __restrict__ k, long ldk,
                                                                const float* __restrict__ v, long ldv_,
                                                                const float* __restrict__ go, long ldo,
                                                                float* __restrict__ gq, float* __restrict__ gk,
                                                                float* __restrict__ gv, long ldg,
                                                                const int* __restrict__ key_len, int Lq, int Lk, int nh,
                                                                int dh, float scale, DropoutP drop) {
  extern __shared__ float sm[];
  const int ldp = dh + 1;
  float* P = sm;                  // [Lq][Lk]
  float* dS = P + Lq * Lk;        // [Lq][Lk]
  float* qs = dS + Lq * Lk;       // [Lq][ldp]
  float* ks = qs + Lq * ldp;      // [Lk][ldp]
  float* vs = ks + Lk * ldp;      // [Lk][ldp]
  float* gs = vs + Lk * ldp;      // [Lq][ldp]
  const int b = blockIdx.x / nh, h = blockIdx.x % nh;
  const int klen = key_len ? min(key_len[b], Lk) : Lk;
  const float* qb = q + (long)b * Lq * ldq + h * dh;
  const float* kb = k + (long)b * Lk * ldk + h * dh;
  const float* vb = v + (long)b * Lk * ldv_ + h * dh;
  const float* gob = go + (long)b * Lq * ldo + h * dh;
  float* gqb = gq + (long)b * Lq * ldg + h * dh;
  float* gkb = gk + (long)b * Lk * ldg + h * dh;
  float* gvb = gv + (long)b * Lk * ldg + h * dh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int id = tid; id < Lq * dh; id += 256) {
    const int i = id / dh, e = id - i * dh;
    qs[i * ldp + e] = qb[(long)i * ldq + e];
    gs[i * ldp + e] = gob[(long)i * ldo + e];
  }
  for (int id = tid; id < Lk * dh; id += 256) {
    const int j = id / dh, e = id - j * dh;
    ks[j * ldp + e] = kb[(long)j * ldk + e];
    vs[j * ldp + e] = vb[(long)j * ldv_ + e];
  }
  __syncthreads();
  for (int ij = tid; ij < Lq * Lk; ij += 256) {  // scores and dP
    const int i = ij / Lk, j = ij - i * Lk;
    float s = 0.f, d = 0.f;
    if (j < klen) {
      const float* qr = qs + i * ldp;
      const float* kr = ks + j * ldp;
      const float* gr = gs + i * ldp;
      const float* vr = vs + j * ldp;
#pragma unroll 8
      for (int e = 0; e < dh; ++e) {
        s = fmaf(qr[e], kr[e], s);
        d = fmaf(gr[e], vr[e], d);
      }
    }
    P[ij] = j < klen ? s * scale : -1e30f;
    dS[ij] = d;
  }
  __syncthreads();
  for (int i = wave; i < Lq; i += 4) {
    float* row = P + i * Lk;
    float* drow = dS + i * Lk;
    float m = -1e30f;
    for (int j = lane; j < Lk; j += 64) m = fmaxf(m, row[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < Lk; j += 64) {
      const float e = row[j] > -1e29f ? __expf(row[j] - m) : 0.f;
      row[j] = e;
      s += e;
    }
    s = wave_sum(s);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    float dot = 0.f;
    const unsigned long long base = ((unsigned long long)blockIdx.x * Lq + i) * (unsigned long long)Lk;
    for (int j = lane; j < Lk; j += 64) {
      row[j] *= inv;
      if (drop.p > 0.f) drow[j] *= dropout_one(drop, base + j);
      dot += row[j] * drow[j];
    }
    dot = wave_sum(dot);
    for (int j = lane; j < Lk; j += 64) {
      drow[j] = row[j] * (drow[j] - dot) * scale;
      if (drop.p > 0.f) row[j] *= dropout_one(drop, base + j);  // P' for dV below
    }
  }
  __syncthreads();
  for (int id = tid; id < Lq * dh; id += 256) {  // dQ
    const int i = id / dh, e = id - i * dh;
    float a = 0.f;
    const float* dr = dS + i * Lk;
#pragma unroll 8
    for (int j = 0; j < klen; ++j) a = fmaf(dr[j], ks[j * ldp + e], a);
    gqb[(long)i * ldg + e] = a;
  }
  for (int id = tid; id < Lk * dh; id += 256) {  // dK, dV
    const int j = id / dh, e = id - j * dh;
    float a = 0.f, c = 0.f;
    if (j < klen) {
#pragma unroll 8
      for (int i = 0; i < Lq; ++i) {
        a = fmaf(dS[i * Lk + j], qs[i * ldp + e], a);
        c = fmaf(P[i * Lk + j], gs[i * ldp + e], c);
      }
    }
    gkb[(long)j * ldg + e] = a;
    gvb[(long)j * ldg + e] = c;
  }
}

// ---------------------------------------------------------------------------
// Row softmax backward for the batched cross-attention:
//   dS[r][c] = scale * P[r][c] * (dP[r][c] - sum_c' P[r][c'] dP[r][c'])   c < n_valid;  0 up to n_out
// P fp16 (forward's probabilities), dP fp32, dS bf16.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const unsigned short* __restrict__ P, long ldp,
                                                               const float* __restrict__ dP, long ldd,
                                                               bf16_t* __restrict__ dS, long lds, float scale,
                                                               int rows, int n_valid, int n_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const unsigned short* p = P + (long)row * ldp;
  const float* d = dP + (long)row * ldd;
  float dot = 0.f;
  for (int c = lane; c < n_valid; c += 64)
    dot += static_cast<float>(__builtin_bit_cast(_Float16, p[c])) * d[c];
  dot = wave_sum(dot);
  bf16_t* o = dS + (long)row * lds;
  for (int c = lane; c < n_out; c += 64) {
    float v = 0.f;
    if (c < n_valid) v = scale * static_cast<float>(__builtin_bit_cast(_Float16, p[c])) * (d[c] - dot);
    o[c] = f32_to_bf16(v);
  }
}

// ---------------------------------------------------------------------------
// Loss gradient (train.py:945-961): loss = mean_{b,t}(dx^2) + mean_{b,t}(dy^2) in pixels,
// dx = (pred_x - gt_x) * rx  =>  dL/dpred_x = 2 * dx * rx / (B*To)
// ---------------------------------------------------------------------------
__global__ void mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                const float* __restrict__ ns, float* __restrict__ g, int B, int To) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * 2 * To) return;
  const int b = idx / (2 * To), f = (idx / To) & 1;
  const float mn = ns[b * 4 + 2 * f], mx = ns[b * 4 + 2 * f + 1];
  const float r = mx - mn;
  const float d = (pred[idx] * r + mn) - (gt[idx] * r + mn);
  g[idx] = 2.f * d * r / (float)(B * To);
}

// out_head backward: out[b][f][s] = w[f].fused[b][s] + bias[f] (+x_last)
//   gf[b][s][c] = sum_f g[b][f][s] w[f][c];  gw[f][c] = sum_{b,s} g[b][f][s] fused[b][s][c];  gb[f] = sum g
__global__ __launch_bounds__(256) void out_head_bwd_kernel(const float* __restrict__ g, const float* __restrict__ fused,
                                                           const float* __restrict__ w, float* __restrict__ gf, int B,
                                                           int To, int C, int F) {
  const int n = B * To * C;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int c = idx % C, s = (idx / C) % To, b = idx / (C * To);
    float a = 0.f;
    for (int f = 0; f < F; ++f) a = fmaf(g[((long)b * F + f) * To + s], w[f * C + c], a);
    gf[idx] = a;
  }
}

// one wave per (f, c) for gw, one per f for gb
__global__ __launch_bounds__(64) void out_head_bwd_w_kernel(const float* __restrict__ g, const float* __restrict__ fused,
                                                            float* __restrict__ gw, float* __restrict__ gb, int B,
                                                            int To, int C, int F) {
  const int id = blockIdx.x, lane = threadIdx.x;
  float a = 0.f;
  if (id < F * C) {
    const int f = id / C, c = id % C;
    for (int i = lane; i < B * To; i += 64) {
      const int b = i / To, s = i % To;
      a = fmaf(g[((long)b * F + f) * To + s], fused[(long)i * C + c], a);
    }
    a = wave_sum(a);
    if (lane == 0) gw[id] = a;
  } else {
    const int f = id - F * C;
    for (int i = lane; i < B * To; i += 64) a += g[((long)(i / To) * F + f) * To + (i % To)];
    a = wave_sum(a);
    if (lane == 0) gb[f] = a;
  }
}

// ---------------------------------------------------------------------------
// Per-channel N-Linear blocks, backward.  out[b][c][s] = sum_t W[c][s][t] u[b][c][t] + bias[c][s] + last[b][c] (+...)
// with u[b][c][t] = in[b][c][t] - in[b][c][T-1].  One block per channel c.
//   gW[c][s][t] = sum_b g[b][c][s] u[b][c][t];  gbias[c][s] = sum_b g[b][c][s]
//   gin[b][c][t] = sum_s g W[c][s][t]  (t < T-1);  gin[b][c][T-1] = sum_s g - sum_{t<T-1} gin[b][c][t]
// Layouts: `in` and `gin` are token-major [B][T][C]; g is given with strides (gb, gc, gs).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nlinear_bwd_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                          const float* __restrict__ g, long g_sb, long g_sc,
                                                          long g_ss, float* __restrict__ gW, float* __restrict__ gbias,
                                                          float* __restrict__ gin, int B, int C, int T, int S) {
  extern __shared__ float sh[];
  float* u = sh;           // [B][T]
  float* gg = sh + B * T;  // [B][S]
  const int c = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < B * T; i += 256) {
    const int b = i / T, t = i % T;
    u[i] = in[((long)b * T + t) * C + c] - in[((long)b * T + T - 1) * C + c];
  }
  for (int i = tid; i < B * S; i += 256) {
    const int b = i / S, s = i % S;
    gg[i] = g[b * g_sb + c * g_sc + s * g_ss];
  }
  __syncthreads();
  for (int st = tid; st < S * T; st += 256) {
    const int s = st / T, t = st % T;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a = fmaf(gg[b * S + s], u[b * T + t], a);
    gW[((long)c * S + s) * T + t] = a;
  }
  for (int s = tid; s < S; s += 256) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += gg[b * S + s];
    gbias[c * S + s] = a;
  }
  if (gin) {
    // gin[b][t] for t < T-1: one (b, t) pair per thread over the channel's weights in LDS (the first version gave a sample to
    // one thread: B of 256 threads busy, S*(T-1) dependent global loads each); then the last column from the row sums
    float* Ws = gg + B * S;  // [S][T]
    for (int i = tid; i < S * T; i += 256) Ws[i] = W[(long)c * S * T + i];
    __syncthreads();
    for (int bt = tid; bt < B * (T - 1); bt += 256) {
      const int b = bt / (T - 1), t = bt % (T - 1);
      float a = 0.f;
      for (int s = 0; s < S; ++s) a = fmaf(gg[b * S + s], Ws[s * T + t], a);
      gin[((long)b * T + t) * C + c] = a;
      u[b * T + t] = a;  // (u is dead by now: reuse it for the row sums below)
    }
    __syncthreads();
    for (int b = tid; b < B; b += 256) {
      float tot = 0.f, acc_last = 0.f;
      for (int s = 0; s < S; ++s) tot += gg[b * S + s];
      for (int t = 0; t < T - 1; ++t) acc_last += u[b * T + t];
      gin[((long)b * T + T - 1) * C + c] = tot - acc_last;
    }
  }
}

// token_proj (Conv1d k=1) backward from g_xp token-major [B][T][C]:
//   gw[c][f] = sum_{b,t} g[b][t][c] x[b][f][t];  gb[c] = sum g
__global__ __launch_bounds__(64) void conv1x1_bwd_kernel(const float* __restrict__ gxp, const float* __restrict__ x,
                                                         float* __restrict__ gw, float* __restrict__ gb, int B,
                                                         int C, int T, int F) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float sb = 0.f;
  for (int i = lane; i < B * T; i += 64) {
    const int b = i / T, t = i % T;
    const float g = gxp[((long)b * T + t) * C + c];
    sb += g;
    for (int f = 0; f < F && f < 4; ++f) acc[f] = fmaf(g, x[((long)b * F + f) * T + t], acc[f]);
  }
  sb = wave_sum(sb);
  for (int f = 0; f < F && f < 4; ++f) {
    const float a = wave_sum(acc[f]);
    if (lane == 0) gw[c * F + f] = a;
  }
  if (lane == 0) gb[c] = sb;
}

// poly_embed backward: x[b][p][d] = w[d][0]*px + w[d][1]*py + bias[d] + pos[p][d]
__global__ __launch_bounds__(256) void poly_embed_bwd_kernel(const float* __restrict__ g, const float* __restrict__ poly,
                                                             float* __restrict__ gw, float* __restrict__ gb,
                                                             float* __restrict__ gpos, int B, int P, int D) {
  // block d: reductions over (b, p) for gw/gb; gpos[p][d] = sum_b g[b][p][d]
  __shared__ float red[3][256];
  const int d = blockIdx.x, tid = threadIdx.x;
  float a0 = 0.f, a1 = 0.f, ab = 0.f;
  for (int i = tid; i < B * P; i += 256) {
    const float gg = g[(long)i * D + d];
    a0 = fmaf(gg, poly[(long)i * 2], a0);
    a1 = fmaf(gg, poly[(long)i * 2 + 1], a1);
    ab += gg;
  }
  red[0][tid] = a0; red[1][tid] = a1; red[2][tid] = ab;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; red[2][tid] += red[2][tid + o]; }
    __syncthreads();
  }
  if (tid == 0) { gw[d * 2] = red[0][0]; gw[d * 2 + 1] = red[1][0]; gb[d] = red[2][0]; }
  for (int p = tid; p < P; p += 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += g[((long)b * P + p) * D + d];
    gpos[p * D + d] = s;
  }
}

// masked mean backward: genc[b][p][d] = p < len[b] ? gemb[b][d] / len[b] : 0
__global__ void masked_mean_bwd_kernel(const float* __restrict__ gemb, const int* __restrict__ len,
                                       float* __restrict__ genc, int B, int P, int D) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * P * D) return;
  const int d = (int)(idx % D);
  const long bp = idx / D;
  const int p = (int)(bp % P), b = (int)(bp / P);
  const int n = min(len[b], P);
  genc[idx] = (p < n) ? gemb[b * D + d] / (float)n : 0.f;
}

// ---------------------------------------------------------------------------
// Fused AdamW over a flat fp32 parameter vector (torch.optim.AdamW semantics:
// p *= 1 - lr*wd; m,v EMA; p -= lr * mhat / (sqrt(vhat) + eps)); grad_scale folds the
// data-parallel mean (1/world) into the update.
// ---------------------------------------------------------------------------
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2, float grad_scale) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * grad_scale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float mh = mi / bc1, vh = vi / bc2;
    pi -= lr * mh / (sqrtf(vh) + eps);
    p[i] = pi;
  }
}

// Gated form (modify_scripts/modify_train.py:1190-1196: clip + step only `if torch.isfinite(loss)`, otherwise the update
// is skipped) without a host round trip: one thread decides from the device-resident loss (and, optionally, the
// exchanged gradient's norm) and keeps the optimizer's step count on the device, so that the bias corrections of a
// later step are those of torch.optim.AdamW after the same number of APPLIED updates.
//   ctl[0] = applied steps, ctl[1] = skipped steps, ctl[2] = 1 if this call is skipped, ctl[4], ctl[5] = bc1, bc2 (float bits)
//   ctl[6] = scale back-off of the fp16 backward (tcavt_grad_scale_pick reads it), ctl[7] = applied updates since it last changed:
//   dynamic loss scaling as mixed-precision training does it, decided on the device -- a skipped update (the 16-bit gradient chain
//   left the half range somewhere, the norm came out non-finite) buys the NEXT steps four more binary orders of headroom; 256
//   applied updates in a row give one back.  Without it the same overflow recurred step after step until the loss had moved
//   (full-size whole-set variant: updates 2 and 3 of a run skipped), and nothing would ever recover a run that stays there.
__global__ void adamw_gate_kernel(const float* __restrict__ loss, const float* __restrict__ norm, int* __restrict__ ctl,
                                  float b1, float b2) {
  const bool ok = isfinite(*loss) && (norm == nullptr || isfinite(*norm));
  if (ok) {
    const int step = ++ctl[0];
    ctl[2] = 0;
    ctl[4] = __float_as_int(1.f - powf(b1, (float)step));
    ctl[5] = __float_as_int(1.f - powf(b2, (float)step));
    if (ctl[6] > 0 && ++ctl[7] >= 256) {
      --ctl[6];
      ctl[7] = 0;
    }
  } else {
    ++ctl[1];
    ctl[2] = 1;
    ctl[6] = min(ctl[6] + 4, 24);
    ctl[7] = 0;
  }
}
__global__ void adamw_gated_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                                   const int* __restrict__ ctl, float grad_scale) {
  if (ctl[2]) return;  // (uniform) non-finite loss: parameters and moments stay as they are
  const float bc1 = __int_as_float(ctl[4]), bc2 = __int_as_float(ctl[5]);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * grad_scale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float mh = mi / bc1, vh = vi / bc2;
    pi -= lr * mh / (sqrtf(vh) + eps);
    p[i] = pi;
  }
}

}  // namespace tcavt

using namespace tcavt;
#define S_(x) static_cast<hipStream_t>(x)

extern "C" int tcavt_transpose16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols,
                                 int rows_pad, int batch, int64_t s_in, int64_t s_out, int f16_to_bf16,
                                 tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(in && out && rows > 0 && cols > 0 && rows_pad >= rows && batch >= 1 && ld_in >= cols &&
                      ld_out >= rows_pad && batch <= 65535,
                  "transpose16: bad args");
  dim3 grid((cols + 63) / 64, (rows_pad + 63) / 64, batch);
  if (f16_to_bf16)
    hipLaunchKernelGGL(transpose16_kernel<true>, grid, dim3(256), 0, S_(stream), static_cast<const bf16_t*>(in), (long)ld_in,
                       static_cast<bf16_t*>(out), (long)ld_out, rows, cols, rows_pad, (long)s_in, (long)s_out);
  else
    hipLaunchKernelGGL(transpose16_kernel<false>, grid, dim3(256), 0, S_(stream), static_cast<const bf16_t*>(in), (long)ld_in,
                       static_cast<bf16_t*>(out), (long)ld_out, rows, cols, rows_pad, (long)s_in, (long)s_out);
  TCAVT_CHECK_LAUNCH("transpose16");
  return TCAVT_OK;
}

extern "C" int tcavt_transpose_f32_bf16(const float* in, int64_t ld_in, void* out, int64_t ld_out, int rows,
                                        int cols, int rows_pad, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(in && out && rows > 0 && cols > 0 && rows_pad >= rows && ld_in >= cols && ld_out >= rows_pad,
                  "transpose_f32_bf16: bad args");
  dim3 grid((cols + 63) / 64, (rows_pad + 63) / 64);
  hipLaunchKernelGGL(transpose_f32_bf16_kernel, grid, dim3(256), 0, S_(stream), in, (long)ld_in,
                     static_cast<bf16_t*>(out), (long)ld_out, rows, cols, rows_pad);
  TCAVT_CHECK_LAUNCH("transpose_f32_bf16");
  return TCAVT_OK;
}

extern "C" int tcavt_colsum(const void* g, int64_t ld, int dtype, float* out, int M, int N, int accumulate,
                            tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(g && out && M > 0 && N > 0 && ld >= N, "colsum: bad args");
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)N * sizeof(float), S_(stream));
    if (e != hipSuccess) {
      set_error("colsum: hipMemsetAsync failed: %s", hipGetErrorString(e));
      return TCAVT_ERR_HIP;
    }
  }
  const int rpb = 128;  // rows per block: partial sums meet in `out` through float atomics
  dim3 grid((N + 63) / 64, (M + rpb - 1) / rpb);
  if (dtype == TCAVT_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, S_(stream), static_cast<const float*>(g), (long)ld, out,
                       M, N, rpb);
  else if (dtype == TCAVT_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, S_(stream), static_cast<const bf16_t*>(g), (long)ld,
                       out, M, N, rpb);
  else {
    set_error("colsum: dtype must be f32 or bf16");
    return TCAVT_ERR_ARG;
  }
  TCAVT_CHECK_LAUNCH("colsum");
  return TCAVT_OK;
}

extern "C" int tcavt_relu_bwd(float* g, const void* y, int y_dtype, int64_t n, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(g && y && n > 0, "relu_bwd: bad args");
  dim3 grid((unsigned)((n + 255) / 256));
  if (y_dtype == TCAVT_F32)
    hipLaunchKernelGGL(relu_bwd_kernel<float>, grid, dim3(256), 0, S_(stream), g, static_cast<const float*>(y), (long)n);
  else
    hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, grid, dim3(256), 0, S_(stream), g, static_cast<const bf16_t*>(y), (long)n);
  TCAVT_CHECK_LAUNCH("relu_bwd");
  return TCAVT_OK;
}

extern "C" int tcavt_add_inplace(float* a, const float* b, int64_t n, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(a && b && n > 0, "add_inplace: bad args");
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), a, b, (long)n);
  TCAVT_CHECK_LAUNCH("add_inplace");
  return TCAVT_OK;
}

extern "C" int tcavt_layernorm_bwd(const float* x, const float* gamma, const float* gy, float eps, float* gx,
                                   float* ggamma, float* gbeta, int M, int D, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(x && gamma && gy && gx && ggamma && gbeta && M > 0 && D > 0 && D <= 4096, "layernorm_bwd: bad args");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((M + 3) / 4), dim3(256), 2 * D * sizeof(float), S_(stream), x, gamma,
                     gy, eps, gx, ggamma, gbeta, M, D);
  TCAVT_CHECK_LAUNCH("layernorm_bwd");
  return TCAVT_OK;
}

extern "C" int tcavt_mha_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                             const float* go, int64_t ldo, float* gq, float* gk, float* gv, int64_t ldg,
                             const int32_t* key_len, int B, int Lq, int Lk, int nh, int dh, float scale,
                             float dropout_p, uint64_t dropout_seed, uint32_t dropout_site, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(q && k && v && go && gq && gk && gv && B > 0 && Lq > 0 && Lk > 0 && nh > 0 && dh > 0, "mha_bwd: bad args");
  TCAVT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "mha_bwd: dropout_p must be in [0, 1)");
  long lds = 2L * Lq * Lk * 4;
  TCAVT_CHECK_ARG(lds <= 64 * 1024, "mha_bwd: 2*Lq*Lk*4 = %ld bytes exceeds 64 KiB", lds);
  const long stage_bytes = 2L * (Lq + Lk) * (dh + 1) * 4;
  const int staged = lds + stage_bytes <= 64 * 1024;
  if (staged) lds += stage_bytes;
  if (staged)
    hipLaunchKernelGGL(mha_small_bwd_lds_kernel, dim3(B * nh), dim3(256), lds, S_(stream), q, (long)ldq, k, (long)ldk, v,
                       (long)ldv, go, (long)ldo, gq, gk, gv, (long)ldg, key_len, Lq, Lk, nh, dh, scale,
                       make_dropout(dropout_p, dropout_seed, dropout_site));
  else
    hipLaunchKernelGGL(mha_small_bwd_kernel, dim3(B * nh), dim3(256), lds, S_(stream), q, (long)ldq, k, (long)ldk, v,
                       (long)ldv, go, (long)ldo, gq, gk, gv, (long)ldg, key_len, Lq, Lk, nh, dh, scale,
                       make_dropout(dropout_p, dropout_seed, dropout_site));
  TCAVT_CHECK_LAUNCH("mha_bwd");
  return TCAVT_OK;
}

extern "C" int tcavt_softmax_bwd_rows(const void* P_f16, int64_t ldp, const float* dP, int64_t ldd, void* dS_bf16,
                                      int64_t lds, float scale, int rows, int n_valid, int n_out,
                                      tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(P_f16 && dP && dS_bf16 && rows > 0 && n_valid > 0 && n_out >= n_valid, "softmax_bwd_rows: bad args");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, S_(stream),
                     static_cast<const unsigned short*>(P_f16), (long)ldp, dP, (long)ldd, static_cast<bf16_t*>(dS_bf16),
                     (long)lds, scale, rows, n_valid, n_out);
  TCAVT_CHECK_LAUNCH("softmax_bwd_rows");
  return TCAVT_OK;
}

extern "C" int tcavt_mse_grad(const float* pred, const float* gt, const float* norm_stat, float* g, int B, int To,
                              tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(pred && gt && norm_stat && g && B > 0 && To > 0, "mse_grad: bad args");
  const int n = B * 2 * To;
  hipLaunchKernelGGL(mse_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, S_(stream), pred, gt, norm_stat, g, B, To);
  TCAVT_CHECK_LAUNCH("mse_grad");
  return TCAVT_OK;
}

extern "C" int tcavt_out_head_bwd(const float* g, const float* fused, const float* w, float* gf, float* gw, float* gb,
                                  int B, int To, int C, int F, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(g && fused && w && gf && gw && gb && B > 0 && To > 0 && C > 0 && F > 0, "out_head_bwd: bad args");
  const int n = B * To * C;
  hipLaunchKernelGGL(out_head_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, S_(stream), g, fused, w, gf, B, To, C, F);
  hipLaunchKernelGGL(out_head_bwd_w_kernel, dim3(F * C + F), dim3(64), 0, S_(stream), g, fused, gw, gb, B, To, C, F);
  TCAVT_CHECK_LAUNCH("out_head_bwd");
  return TCAVT_OK;
}

extern "C" int tcavt_nlinear_bwd(const float* in_tok, const float* W, const float* g, int64_t g_sb, int64_t g_sc,
                                 int64_t g_ss, float* gW, float* gbias, float* gin_tok, int B, int C, int T, int S,
                                 tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(in_tok && W && g && gW && gbias && B > 0 && C > 0 && T > 0 && S > 0, "nlinear_bwd: bad args");
  const long lds = ((long)B * T + (long)B * S + (long)S * T) * 4;
  TCAVT_CHECK_ARG(lds <= 64 * 1024, "nlinear_bwd: (B*(T+S) + S*T)*4 = %ld bytes exceeds 64 KiB", lds);
  hipLaunchKernelGGL(nlinear_bwd_kernel, dim3(C), dim3(256), lds, S_(stream), in_tok, W, g, (long)g_sb, (long)g_sc,
                     (long)g_ss, gW, gbias, gin_tok, B, C, T, S);
  TCAVT_CHECK_LAUNCH("nlinear_bwd");
  return TCAVT_OK;
}

extern "C" int tcavt_conv1x1_bwd(const float* gxp_tok, const float* x, float* gw, float* gb, int B, int C, int T,
                                 int F, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(gxp_tok && x && gw && gb && B > 0 && C > 0 && T > 0 && F > 0 && F <= 4, "conv1x1_bwd: bad args (F <= 4)");
  hipLaunchKernelGGL(conv1x1_bwd_kernel, dim3(C), dim3(64), 0, S_(stream), gxp_tok, x, gw, gb, B, C, T, F);
  TCAVT_CHECK_LAUNCH("conv1x1_bwd");
  return TCAVT_OK;
}

extern "C" int tcavt_poly_embed_bwd(const float* g, const float* polygon, float* gw, float* gb, float* gpos, int B,
                                    int P, int D, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(g && polygon && gw && gb && gpos && B > 0 && P > 0 && D > 0, "poly_embed_bwd: bad args");
  hipLaunchKernelGGL(poly_embed_bwd_kernel, dim3(D), dim3(256), 0, S_(stream), g, polygon, gw, gb, gpos, B, P, D);
  TCAVT_CHECK_LAUNCH("poly_embed_bwd");
  return TCAVT_OK;
}

extern "C" int tcavt_masked_mean_bwd(const float* gemb, const int32_t* len, float* genc, int B, int P, int D,
                                     tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(gemb && len && genc && B > 0 && P > 0 && D > 0, "masked_mean_bwd: bad args");
  const long n = (long)B * P * D;
  hipLaunchKernelGGL(masked_mean_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_(stream), gemb, len,
                     genc, B, P, D);
  TCAVT_CHECK_LAUNCH("masked_mean_bwd");
  return TCAVT_OK;
}

extern "C" int tcavt_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int step, float grad_scale, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adamw: bad args");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, S_(stream), p, g, m, v, (long)n, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2, grad_scale);
  TCAVT_CHECK_LAUNCH("adamw");
  return TCAVT_OK;
}

extern "C" int tcavt_adamw_gated(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, float grad_scale, const float* loss,
                                 const float* grad_norm, int32_t* ctl, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(p && g && m && v && n > 0 && loss && ctl, "adamw_gated: bad args");
  hipLaunchKernelGGL(adamw_gate_kernel, dim3(1), dim3(1), 0, S_(stream), loss, grad_norm, ctl, beta1, beta2);
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adamw_gated_kernel, dim3((unsigned)blocks), dim3(256), 0, S_(stream), p, g, m, v, (long)n, lr,
                     beta1, beta2, eps, weight_decay, ctl, grad_scale);
  TCAVT_CHECK_LAUNCH("adamw_gated");
  return TCAVT_OK;
}
