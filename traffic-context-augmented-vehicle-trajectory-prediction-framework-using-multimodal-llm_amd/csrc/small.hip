// fp32 kernels for the parts of the path that must not drop to bf16: the
// lane-polygon encoder sees raw pixel coordinates up to 3839 (scripts/train.py:364,
// scripts/graph.py:7-216), and the LTSF head / metrics work in pixel space
// (scripts/train.py:945-962).  All of them are tiny next to the decoder stack, so
// they are written for clarity and coalesced access, not for MFMA.
#include "common.hpp"
#include "philox.hpp"

namespace tcavt {

// ---------------------------------------------------------------------------
// C[M,N] = A[M,K] . W[N,K]^T (+bias)(+relu)(+residual) in exact fp32 on the matrix cores:
// v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate; bit-for-bit an fmaf chain per output).
// General element strides: A[m][k] = A[m*rsA + k*csA], W[n][k] = W[n*rsW + k*csW], so the same
// kernel serves y = x W^T (forward), gx = gy W and gW = gy^T x (backward) without transposes.
// 64x64 tile, 4 waves (2x2) of 32x32, K-step 16 staged through LDS ([row][k], 17-float rows);
// global->register prefetch of step s+1 overlaps the MFMAs of step s.  (Measured and rejected in round 2: K-step 64 with 32
// unconditional loads in flight per thread -- 10-40 % slower on the head's shapes, whose cost is the K-serial chain of one or
// a few workgroups, not the loads.)  The global read pattern
// follows the operand's contiguous direction (lanes walk k when csX == 1, rows otherwise).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gemm_f32_fetch(const float* __restrict__ X, long rs, long cs, int r0, int R, int k0,
                                               int K, int tid, bool kfast, float (&v)[4], int (&row)[4], int (&kk)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = e * 256 + tid;  // 0..1023 over the 64x16 tile
    const int r = kfast ? (idx >> 4) : (idx & 63);
    const int k = kfast ? (idx & 15) : (idx >> 6);
    row[e] = r;
    kk[e] = k;
    const int gr = r0 + r, gk = k0 + k;
    v[e] = (gr < R && gk < K) ? X[(long)gr * rs + (long)gk * cs] : 0.f;
  }
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long rsA, long csA,
                                                       const float* __restrict__ W, long rsW, long csW,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ res, long ldr,
                                                       float* __restrict__ C, long ldc, int M, int N,
                                                       int K, int flags, int k_chunk, DropoutP drop) {
  // split-K: blockIdx.z owns k in [kb, ke); partial sums meet in C (zeroed by the launcher) through float
  // atomics, split 0 adds bias / residual.  Used when the output has too few tiles to fill the chip.
  const int kb = blockIdx.z * k_chunk, ke = min(K, kb + k_chunk);
  __shared__ float As[64][17];
  __shared__ float Ws[64][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const bool akf = csA <= rsA, wkf = csW <= rsW;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float va[4], vw[4];
  int ra[4], ka[4], rw[4], kw[4];
  gemm_f32_fetch(A, rsA, csA, m0, M, kb, ke, tid, akf, va, ra, ka);
  gemm_f32_fetch(W, rsW, csW, n0, N, kb, ke, tid, wkf, vw, rw, kw);
  const int lr = lane & 15, lk = lane >> 4;
  for (int k0 = kb; k0 < ke; k0 += 16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      As[ra[e]][ka[e]] = va[e];
      Ws[rw[e]][kw[e]] = vw[e];
    }
    __syncthreads();
    if (k0 + 16 < ke) {
      gemm_f32_fetch(A, rsA, csA, m0, M, k0 + 16, ke, tid, akf, va, ra, ka);
      gemm_f32_fetch(W, rsW, csW, n0, N, k0 + 16, ke, tid, wkf, vw, rw, kw);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = As[wm * 32 + i * 16 + lr][ks * 4 + lk];
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = Ws[wn * 32 + j * 16 + lr][ks * 4 + lk];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], wf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D layout: column (n) = lane & 15, row (m) = 4 * (lane >> 4) + reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + j * 16 + lr;
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + 4 * lk + r;
        if (m >= M) continue;
        float v = acc[i][j][r];
        if (gridDim.z > 1) {
          if (blockIdx.z == 0) {
            if (flags & TCAVT_EPI_BIAS) v += bias[n];
            if (flags & TCAVT_EPI_RESIDUAL) v += res[(long)m * ldr + n];
          }
          atomicAdd(&C[(long)m * ldc + n], v);
          continue;
        }
        if (flags & TCAVT_EPI_ACCUM) v += C[(long)m * ldc + n];
        if (flags & TCAVT_EPI_BIAS) v += bias[n];
        if (flags & TCAVT_EPI_RELU) v = relu_nan(v);
        if (drop.p > 0.f) v *= dropout_one(drop, (unsigned long long)m * (unsigned long long)N + (unsigned long long)n);
        if (flags & TCAVT_EPI_RESIDUAL) v += res[(long)m * ldr + n];
        C[(long)m * ldc + n] = v;
      }
    }
}

// x[b][p][:] = W_in[:, 0]*px + W_in[:, 1]*py + b_in + pos[p]   (train.py:364-365)
__global__ void poly_embed_kernel(const float* __restrict__ poly, const float* __restrict__ w,
                                  const float* __restrict__ bi, const float* __restrict__ pos,
                                  float* __restrict__ x, int B, int P, int D) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * P * D) return;
  const int d = (int)(idx % D);
  const long bp = idx / D;
  const int p = (int)(bp % P);
  const float px = poly[bp * 2], py = poly[bp * 2 + 1];
  x[idx] = fmaf(w[d * 2 + 1], py, fmaf(w[d * 2], px, 0.f)) + bi[d] + pos[p * D + d];
}

// emb[b][d] = mean_{p<len[b]} enc[b][p][d], zeros when len == 0   (train.py:372-382)
__global__ void masked_mean_kernel(const float* __restrict__ enc, const int* __restrict__ len,
                                   float* __restrict__ emb, int B, int P, int D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * D) return;
  const int b = idx / D, d = idx % D;
  const int n = min(len[b], P);
  float s = 0.f;
  for (int p = 0; p < n; ++p) s += enc[((long)b * P + p) * D + d];
  emb[idx] = n > 0 ? s / (float)n : 0.f;
}

// TransformerLTSF front: Conv1d(k=1) token projection + per-channel N-Linear
// encoder + positional term (train.py:837-839, 701-716).  One block per sample,
// one thread per (c, s).  Output token-major tok[b][s][c].
__global__ void ltsf_front_kernel(const float* __restrict__ x, const float* __restrict__ cw,
                                  const float* __restrict__ cb, const float* __restrict__ ew,
                                  const float* __restrict__ eb, const float* __restrict__ pos,
                                  float* __restrict__ tok, float* __restrict__ xp_tok, int B, int C, int T) {
  extern __shared__ float xs[];  // [2][T]
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) xs[i] = x[(long)b * 2 * T + i];
  __syncthreads();
  for (int cs = threadIdx.x; cs < C * T; cs += blockDim.x) {
    const int c = cs / T, s = cs % T;
    const float w0 = cw[c * 2], w1 = cw[c * 2 + 1], bb = cb[c];
    const float last = fmaf(w1, xs[T + T - 1], fmaf(w0, xs[T - 1], 0.f)) + bb;
    float acc = 0.f;
    const float* wr = ew + ((long)c * T + s) * T;
    for (int t = 0; t < T; ++t) {
      const float xp = fmaf(w1, xs[T + t], fmaf(w0, xs[t], 0.f)) + bb;
      acc = fmaf(wr[t], xp - last, acc);
    }
    tok[((long)b * T + s) * C + c] = acc + eb[c * T + s] + last + pos[c * T + s];
    // training: also keep the projected input xp[b][s][c] (token-major) for the backward
    if (xp_tok) xp_tok[((long)b * T + s) * C + c] = fmaf(w1, xs[T + s], fmaf(w0, xs[s], 0.f)) + bb;
  }
}

// LTSF_NLinearDecoder front (train.py:768-785): per-channel Linear(T -> To) on
// (e - e_last), + e_last + lane_adj.  e given token-major [B][T][C].
__global__ void ltsf_decode_kernel(const float* __restrict__ e, const float* __restrict__ dw,
                                   const float* __restrict__ db, const float* __restrict__ lane,
                                   float* __restrict__ dec, int B, int C, int T, int To) {
  // grid (chunks of 256 outputs, B): one output per thread, its T weights loaded in one burst (the first version gave a
  // sample to one workgroup, 8 outputs x T dependent weight loads per thread: 95 us on the critical path of the head)
  extern __shared__ float es[];  // [T][C]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < T * C; i += blockDim.x) es[i] = e[(long)b * T * C + i];
  __syncthreads();
  const int cs = blockIdx.x * blockDim.x + threadIdx.x;
  if (cs >= C * To) return;
  const int c = cs / To, s = cs % To;
  const float last = es[(T - 1) * C + c];
  const float* wr = dw + ((long)c * To + s) * T;
  float acc = 0.f;
  int t = 0;
  for (; t + 6 <= T; t += 6) {
    float w[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) w[u] = wr[t + u];
#pragma unroll
    for (int u = 0; u < 6; ++u) acc = fmaf(w[u], es[(t + u) * C + c] - last, acc);
  }
  for (; t < T; ++t) acc = fmaf(wr[t], es[t * C + c] - last, acc);
  dec[(long)b * C * To + cs] = acc + db[c * To + s] + last + lane[(long)b * C * To + cs];
}

__global__ void transpose_ct_kernel(const float* __restrict__ in, float* __restrict__ of,
                                    bf16_t* __restrict__ ob, int B, int C, int To, int f16) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over [B][To][C]
  if (idx >= (long)B * C * To) return;
  const int c = (int)(idx % C);
  const long bs = idx / C;
  const int s = (int)(bs % To);
  const long b = bs / To;
  const float v = in[(b * C + c) * To + s];
  if (of) of[idx] = v;
  if (ob) ob[idx] = f16 ? f32_to_f16(v) : f32_to_bf16(v);
}

// out[b][f][s] = w[f] . fused[b][s] + bias[f] + x[b][f][T-1]   (train.py:804-805,941-943)
__global__ void out_head_kernel(const float* __restrict__ fused, const float* __restrict__ w,
                                const float* __restrict__ bias, const float* __restrict__ x,
                                float* __restrict__ out, int B, int To, int C, int F, int T, int add_last) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * F * To) return;
  const int s = idx % To, f = (idx / To) % F, b = idx / (To * F);
  const float* fr = fused + ((long)b * To + s) * C;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) acc = fmaf(w[f * C + c], fr[c], acc);
  out[idx] = acc + bias[f] + (add_last ? x[((long)b * F + f) * T + T - 1] : 0.f);
}

// ---------------------------------------------------------------------------
// De-normalise + squared error + ADE/FDE/RMSE with min over K candidates.
// ONE workgroup of 16 waves: wave w takes samples w, w + 16, ... (lanes over time steps) and keeps its five partial
// sums in registers; the waves' partials meet in LDS and are added in wave order, then once to sums[] -- a fixed
// summation order, so the loss and the metrics are bit-reproducible (a block per sample with float atomics was not).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void traj_metrics_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ gt,
                                                          const float* __restrict__ ns,
                                                          float* __restrict__ sums,
                                                          int* __restrict__ argmin,
                                                          float* __restrict__ per_sample, int B, int K,
                                                          int To) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  __shared__ float part[16][5];
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, acc4 = 0.f;
  for (int b = wave; b < B; b += nwave) {
  const float minx = ns[b * 4], maxx = ns[b * 4 + 1], miny = ns[b * 4 + 2], maxy = ns[b * 4 + 3];
  const float rx = maxx - minx, ry = maxy - miny;
  float best_ade = 0.f, best_fde = 0.f, best_rmse = 0.f;
  int ia = 0, ifd = 0, ir = 0;
  float sx_tot = 0.f, sy_tot = 0.f;
  for (int k = 0; k < K; ++k) {
    const float* px = pred + (((long)b * K + k) * 2) * To;
    const float* py = px + To;
    const float* gx = gt + (long)b * 2 * To;
    const float* gy = gx + To;
    float sx = 0.f, sy = 0.f, se = 0.f, last = 0.f;
    for (int t = lane; t < To; t += 64) {
      const float dx = (px[t] * rx + minx) - (gx[t] * rx + minx);
      const float dy = (py[t] * ry + miny) - (gy[t] * ry + miny);
      sx += dx * dx;
      sy += dy * dy;
      const float err = sqrtf(dx * dx + dy * dy);
      se += err;
      if (t == To - 1) last = err;
    }
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    se = wave_sum(se);
    last = wave_sum(last);
    const float ade = se / (float)To;
    const float rmse = sqrtf((sx + sy) / (float)(2 * To));
    if (k == 0 || ade < best_ade) { best_ade = ade; ia = k; }
    if (k == 0 || last < best_fde) { best_fde = last; ifd = k; }
    if (k == 0 || rmse < best_rmse) { best_rmse = rmse; ir = k; }
    if (k == 0) { sx_tot = sx; sy_tot = sy; }
  }
  acc0 += sx_tot; acc1 += sy_tot; acc2 += best_ade; acc3 += best_fde; acc4 += best_rmse;
  if (lane == 0) {
    if (argmin) { argmin[b * 3] = ia; argmin[b * 3 + 1] = ifd; argmin[b * 3 + 2] = ir; }
    if (per_sample) { per_sample[b * 3] = best_ade; per_sample[b * 3 + 1] = best_fde; per_sample[b * 3 + 2] = best_rmse; }
  }
  }  // samples of this wave
  if (lane == 0) { part[wave][0] = acc0; part[wave][1] = acc1; part[wave][2] = acc2; part[wave][3] = acc3; part[wave][4] = acc4; }
  __syncthreads();
  if (threadIdx.x < 5) {
    float t = 0.f;
    for (int w = 0; w < nwave; ++w) t += part[w][threadIdx.x];
    sums[threadIdx.x] += t;  // sums[] accumulates over calls (one call per batch), single writer
  }
}

}  // namespace tcavt

using namespace tcavt;

// Launch with split-K when the 64x64 output tiles cannot fill the chip and K is long (lane-polygon FFN:
// K = 2048 or a contraction over 2048 tokens with 32 output tiles).  Not with ReLU (needs the full sum),
// not when C aliases the residual (split 0 would read what other splits already added to).
// max_splits = 2 keeps the result run-to-run reproducible (0 + a + b == 0 + b + a in IEEE arithmetic; three or
// more addends are order dependent): the forward entry point uses that, because a 1-ulp difference in the
// lane-polygon embedding is amplified to ~1e-4 by the bf16 decoder downstream; the backward entry point
// (gradients, tolerance-checked) uses up to 8.
static int launch_gemm_f32(int max_splits, const float* A, long rsA, long csA, const float* W, long rsW, long csW, const float* bias,
                           const float* residual, long ldr, float* C, long ldc, int M, int N, int K, int flags,
                           DropoutP drop, hipStream_t stream, const char* what) {
  const int tiles = ((N + 63) / 64) * ((M + 63) / 64);
  int splits = 1;
  if (tiles < 128 && K >= 512 && !(flags & TCAVT_EPI_RELU) && residual != C && ldc == N && drop.p == 0.f) {
    splits = K / 256;
    if (splits > max_splits) splits = max_splits;
    while (splits > 1 && tiles * splits > 512) splits >>= 1;
  }
  int k_chunk = K;
  if (splits > 1) {
    k_chunk = (((K + splits - 1) / splits) + 15) / 16 * 16;
    splits = (K + k_chunk - 1) / k_chunk;
    hipError_t e = (flags & TCAVT_EPI_ACCUM) ? hipSuccess : hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), stream);
    if (e != hipSuccess) {
      set_error("%s: hipMemsetAsync failed: %s", what, hipGetErrorString(e));
      return TCAVT_ERR_HIP;
    }
  }
  dim3 grid((N + 63) / 64, (M + 63) / 64, splits), block(256);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, block, 0, stream, A, rsA, csA, W, rsW, csW, bias, residual, ldr, C, ldc, M,
                     N, K, flags, k_chunk, drop);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return TCAVT_ERR_HIP;
  }
  return TCAVT_OK;
}

extern "C" int tcavt_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw,
                              const float* bias, const float* residual, int64_t ldr, float* C,
                              int64_t ldc, int M, int N, int K, int flags, float dropout_p,
                              uint64_t dropout_seed, uint32_t dropout_site, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0, "gemm_f32: null pointer or bad shape");
  TCAVT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "gemm_f32: dropout_p must be in [0, 1)");
  TCAVT_CHECK_ARG(!(flags & TCAVT_EPI_BIAS) || bias, "gemm_f32: BIAS without bias");
  TCAVT_CHECK_ARG(!(flags & TCAVT_EPI_RESIDUAL) || residual, "gemm_f32: RESIDUAL without residual");
  TCAVT_CHECK_ARG(!(flags & ~(TCAVT_EPI_BIAS | TCAVT_EPI_RELU | TCAVT_EPI_RESIDUAL)), "gemm_f32: unsupported flag");
  return launch_gemm_f32(2, A, (long)lda, 1L, W, (long)ldw, 1L, bias, residual, (long)ldr, C, (long)ldc, M, N, K, flags,
                         make_dropout(dropout_p, dropout_seed, dropout_site), static_cast<hipStream_t>(stream), "gemm_f32");
}

extern "C" int tcavt_gemm_f32_strided(const float* A, int64_t rsA, int64_t csA, const float* W, int64_t rsW,
                                      int64_t csW, const float* bias, const float* residual, int64_t ldr, float* C,
                                      int64_t ldc, int M, int N, int K, int flags, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0, "gemm_f32_strided: null pointer or bad shape");
  TCAVT_CHECK_ARG(!(flags & TCAVT_EPI_BIAS) || bias, "gemm_f32_strided: BIAS without bias");
  TCAVT_CHECK_ARG(!(flags & TCAVT_EPI_RESIDUAL) || residual, "gemm_f32_strided: RESIDUAL without residual");
  TCAVT_CHECK_ARG(!(flags & ~(TCAVT_EPI_BIAS | TCAVT_EPI_RELU | TCAVT_EPI_RESIDUAL | TCAVT_EPI_ACCUM)), "gemm_f32_strided: unsupported flag");
  TCAVT_CHECK_ARG(!((flags & TCAVT_EPI_ACCUM) && (flags & TCAVT_EPI_RELU)), "gemm_f32_strided: ACCUM excludes RELU");
  return launch_gemm_f32(8, A, (long)rsA, (long)csA, W, (long)rsW, (long)csW, bias, residual, (long)ldr, C, (long)ldc, M,
                         N, K, flags, make_dropout(0.f, 0, 0), static_cast<hipStream_t>(stream), "gemm_f32_strided");
}

extern "C" int tcavt_poly_embed(const float* polygon, const float* w_in, const float* b_in,
                                const float* pos, float* x, int B, int P, int D, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(polygon && w_in && b_in && pos && x && B > 0 && P > 0 && D > 0, "poly_embed: bad args");
  const long n = (long)B * P * D;
  hipLaunchKernelGGL(poly_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), polygon, w_in, b_in, pos, x, B, P, D);
  TCAVT_CHECK_LAUNCH("poly_embed");
  return TCAVT_OK;
}

extern "C" int tcavt_masked_mean(const float* enc, const int32_t* len, float* emb, int B, int P, int D,
                                 tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(enc && len && emb && B > 0 && P > 0 && D > 0, "masked_mean: bad args");
  hipLaunchKernelGGL(masked_mean_kernel, dim3((B * D + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), enc, len, emb, B, P, D);
  TCAVT_CHECK_LAUNCH("masked_mean");
  return TCAVT_OK;
}

extern "C" int tcavt_ltsf_front(const float* x, const float* conv_w, const float* conv_b,
                                const float* enc_w, const float* enc_b, const float* pos,
                                float* enc_tok, float* xp_tok, int B, int C, int T, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(x && conv_w && conv_b && enc_w && enc_b && pos && enc_tok && B > 0 && C > 0 && T > 0,
                  "ltsf_front: bad args");
  hipLaunchKernelGGL(ltsf_front_kernel, dim3(B), dim3(256), 2 * T * sizeof(float),
                     static_cast<hipStream_t>(stream), x, conv_w, conv_b, enc_w, enc_b, pos, enc_tok, xp_tok, B, C, T);
  TCAVT_CHECK_LAUNCH("ltsf_front");
  return TCAVT_OK;
}

extern "C" int tcavt_ltsf_decode(const float* e_tok, const float* dec_w, const float* dec_b,
                                 const float* lane_adj, float* dec, int B, int C, int T, int To,
                                 tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(e_tok && dec_w && dec_b && lane_adj && dec && B > 0 && C > 0 && T > 0 && To > 0,
                  "ltsf_decode: bad args");
  TCAVT_CHECK_ARG((long)T * C * 4 <= 48 * 1024, "ltsf_decode: T*C too large for LDS");
  hipLaunchKernelGGL(ltsf_decode_kernel, dim3((C * To + 255) / 256, B), dim3(256), T * C * sizeof(float),
                     static_cast<hipStream_t>(stream), e_tok, dec_w, dec_b, lane_adj, dec, B, C, T, To);
  TCAVT_CHECK_LAUNCH("ltsf_decode");
  return TCAVT_OK;
}

extern "C" int tcavt_transpose_ct(const float* in, float* out_f32, void* out_bf16, int B, int C, int To,
                                  int dtype16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(in && (out_f32 || out_bf16) && B > 0 && C > 0 && To > 0, "transpose_ct: bad args");
  TCAVT_CHECK_ARG(is16(dtype16), "transpose_ct: dtype16 must be TCAVT_BF16 or TCAVT_F16");
  const long n = (long)B * C * To;
  hipLaunchKernelGGL(transpose_ct_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out_f32, static_cast<bf16_t*>(out_bf16), B, C, To,
                     dtype16 == TCAVT_F16 ? 1 : 0);
  TCAVT_CHECK_LAUNCH("transpose_ct");
  return TCAVT_OK;
}

extern "C" int tcavt_out_head(const float* fused, const float* w, const float* bias, const float* x,
                              float* out, int B, int To, int C, int F, int T, int add_last,
                              tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(fused && w && bias && x && out && B > 0 && To > 0 && C > 0 && F > 0 && T > 0,
                  "out_head: bad args");
  const int n = B * F * To;
  hipLaunchKernelGGL(out_head_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     fused, w, bias, x, out, B, To, C, F, T, add_last);
  TCAVT_CHECK_LAUNCH("out_head");
  return TCAVT_OK;
}

extern "C" int tcavt_traj_metrics(const float* pred, const float* gt, const float* norm_stat,
                                  float* sums, int32_t* argmin, float* per_sample, int B, int K, int To,
                                  tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(pred && gt && norm_stat && sums && B > 0 && K > 0 && To > 0, "traj_metrics: bad args");
  hipLaunchKernelGGL(traj_metrics_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), pred, gt,
                     norm_stat, sums, argmin, per_sample, B, K, To);
  TCAVT_CHECK_LAUNCH("traj_metrics");
  return TCAVT_OK;
}
