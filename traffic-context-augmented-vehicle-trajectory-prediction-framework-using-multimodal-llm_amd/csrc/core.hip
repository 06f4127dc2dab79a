// Library plumbing: error channel, device check, and the HBM-bound row kernels
// (RMSNorm, LayerNorm(+residual), fp32->bf16 cast, fused embedding build).
// These are bandwidth-bound: every access is 16 bytes per lane, one row per
// wave (H = 2048 fp32 = 8 KiB = 8 float4 per lane) so the only cross-lane
// traffic is one wave reduction.
#include "common.hpp"
#include "philox.hpp"
#include <string.h>

namespace tcavt {

static thread_local char g_err[512] = "";

static const unsigned long long* g_dropout_epoch = nullptr;
const unsigned long long* dropout_epoch_ptr() { return g_dropout_epoch; }

__global__ void epoch_advance_kernel(unsigned long long* e) { *e += 1ull; }

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------
// RMSNorm (HF modeling_llama.py:62-67): var = mean(x^2) in fp32,
// y = x * rsqrt(var + eps) * gamma.  One wave per row, NV float4 per lane.
// ---------------------------------------------------------------------------
// NV > 0: H == NV * 256 exactly, the row lives in NV float4 per lane (no bounds logic, one HBM pass).
// NV == 0: any H % 4 == 0; two passes over the row (the second one hits L2).
template <int NV, bool F16>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ gamma, float eps,
                                                      bf16_t* __restrict__ out_bf16,
                                                      float* __restrict__ out_f32, int M, int H,
                                                      bf16_t* __restrict__ out_drop, DropoutP drop) {
  // out_drop (optional): dropout(out_bf16) with the mask of (seed, site, row * H + column) -- bit-identical to running
  // tcavt_dropout on out_bf16 afterwards (the LoRA branch input, train.py:433-440), without re-reading it
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  auto store_drop = [&](const f32x4& y, int idx) {
    float sc[4];
    dropout_quad(drop, ((unsigned long long)row * (unsigned long long)H + 4ull * idx) >> 2, sc);
    float z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = from16<F16>(to16<F16>(y[e])) * sc[e];
    u32x2 o = {pack16x2<F16>(z[0], z[1]), pack16x2<F16>(z[2], z[3])};
    *reinterpret_cast<u32x2*>(out_drop + (long)row * H + idx * 4) = o;
  };
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + (long)row * H);
  const f32x4* gr = reinterpret_cast<const f32x4*>(gamma);
  if constexpr (NV > 0) {
    f32x4 v[NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = xr[i * 64 + lane];
      ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = i * 64 + lane;
      const f32x4 g = gr[idx];
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = v[i][e] * rs * g[e];
      if (out_bf16) {
        u32x2 o = {pack16x2<F16>(y[0], y[1]), pack16x2<F16>(y[2], y[3])};
        *reinterpret_cast<u32x2*>(out_bf16 + (long)row * H + idx * 4) = o;
      }
      if (out_f32) *reinterpret_cast<f32x4*>(out_f32 + (long)row * H + idx * 4) = y;
      if (out_drop) store_drop(y, idx);
    }
  } else {
    const int nvec = H >> 2;
    float ss = 0.f;
    for (int idx = lane; idx < nvec; idx += 64) {
      const f32x4 t = xr[idx];
      ss += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)H + eps);
    for (int idx = lane; idx < nvec; idx += 64) {
      const f32x4 t = xr[idx];
      const f32x4 g = gr[idx];
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = t[e] * rs * g[e];
      if (out_bf16) {
        u32x2 o = {pack16x2<F16>(y[0], y[1]), pack16x2<F16>(y[2], y[3])};
        *reinterpret_cast<u32x2*>(out_bf16 + (long)row * H + idx * 4) = o;
      }
      if (out_f32) *reinterpret_cast<f32x4*>(out_f32 + (long)row * H + idx * 4) = y;
      if (out_drop) store_drop(y, idx);
    }
  }
}

// ---------------------------------------------------------------------------
// LayerNorm with optional residual add (torch.nn.LayerNorm semantics: biased
// variance, eps inside the sqrt).  One wave per row.
// ---------------------------------------------------------------------------
template <int NV, bool F16>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ res,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ out_f32,
                                                        bf16_t* __restrict__ out_bf16, int M, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + (long)row * D);
  const f32x4* rr = res ? reinterpret_cast<const f32x4*>(res + (long)row * D) : nullptr;
  const int nvec = D >> 2;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = i * 64 + lane;
    if (idx < nvec) {
      v[i] = xr[idx];
      if (rr) v[i] += rr[idx];
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = i * 64 + lane;
    if (idx < nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rs = rsqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = i * 64 + lane;
    if (idx < nvec) {
      const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[idx];
      const f32x4 b = reinterpret_cast<const f32x4*>(beta)[idx];
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rs * g[e] + b[e];
      if (out_f32) *reinterpret_cast<f32x4*>(out_f32 + (long)row * D + idx * 4) = y;
      if (out_bf16) {
        u32x2 o = {pack16x2<F16>(y[0], y[1]), pack16x2<F16>(y[2], y[3])};
        *reinterpret_cast<u32x2*>(out_bf16 + (long)row * D + idx * 4) = o;
      }
    }
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x,
                                                   bf16_t* __restrict__ out, long n) {
  const long nvec = n >> 3;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i];
    const f32x4 b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
    u32x4 o = {pack16x2<F16>(a[0], a[1]), pack16x2<F16>(a[2], a[3]), pack16x2<F16>(b[0], b[1]),
               pack16x2<F16>(b[2], b[3])};
    reinterpret_cast<u32x4*>(out)[i] = o;
  }
  // tail (n % 8 elements)
  const long tail0 = nvec << 3;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < n - tail0) out[tail0 + gid] = to16<F16>(x[tail0 + gid]);
}

// ---------------------------------------------------------------------------
// Fused embedding build (scripts/train.py:521-528): one wave per output row,
// 8 elements per lane per step (16-byte bf16 table reads, 2 x float4 writes).
// ---------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void embed_fuse_kernel(const bf16_t* __restrict__ table,
                                                         const int64_t* __restrict__ ids,
                                                         const float* __restrict__ img,
                                                         const float* __restrict__ vis_mod,
                                                         const float* __restrict__ txt_mod,
                                                         float* __restrict__ h, int B, int Nq, int Lt,
                                                         int H, int V, int* bad_flag, bf16_t* __restrict__ h16,
                                                         float* __restrict__ part, int npart, float sscale, int frag16) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int L = Nq + Lt;
  if (row >= (long)B * L) return;
  const int b = (int)(row / L), i = (int)(row % L);
  float* out = h ? h + row * H : nullptr;  // (null: the residual stream is the 16-bit h16 itself)
  bf16_t* out16 = h16 ? h16 + row * H : nullptr;
  const bool frag = frag16 && h16;  // (decode step: the <= 32 rows in the skinny GEMM's operand order, common.hpp frag16_off)
  float ss = 0.f;  // sum of squares of the row (first decoder layer's fused RMSNorm)
  auto emit = [&](int c, f32x4 o) {
    if (out) *reinterpret_cast<f32x4*>(out + c) = o;
    o *= sscale;  // (scaled 16-bit image of the stream: tcavt_llama_stack_args.stream_scale; 1 by default, exact for powers of two)
    if (out16) {
      const u32x2 w = u32x2{pack16x2<F16>(o[0], o[1]), pack16x2<F16>(o[2], o[3])};
      *reinterpret_cast<u32x2*>(frag ? h16 + frag_off((int)row, c, H, frag16) : out16 + c) = w;
      if (!out) o = f32x4{from16_lo<F16>(w[0]), from16_hi<F16>(w[0]), from16_lo<F16>(w[1]), from16_hi<F16>(w[1])};  // sums of what is stored
    }
    ss += o[0] * o[0];
    ss += o[1] * o[1];
    ss += o[2] * o[2];
    ss += o[3] * o[3];
  };
  if (i < Nq) {
    const float* src = img + ((long)b * Nq + i) * H;
    for (int c = lane * 4; c < H; c += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + c);
      const f32x4 m = *reinterpret_cast<const f32x4*>(vis_mod + c);
      emit(c, a + m);
    }
  } else {
    int64_t id = ids[(long)b * Lt + (i - Nq)];
    if (id < 0 || id >= V) {
      if (lane == 0) *bad_flag = 1;
      id = 0;
    }
    const bf16_t* src = table + id * H;
    for (int c = lane * 8; c < H; c += 512) {
      const u32x4 t = *reinterpret_cast<const u32x4*>(src + c);
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(txt_mod + c);
      const f32x4 m1 = *reinterpret_cast<const f32x4*>(txt_mod + c + 4);
      f32x4 o0, o1;
      o0[0] = from16_lo<F16>(t[0]) + m0[0];
      o0[1] = from16_hi<F16>(t[0]) + m0[1];
      o0[2] = from16_lo<F16>(t[1]) + m0[2];
      o0[3] = from16_hi<F16>(t[1]) + m0[3];
      o1[0] = from16_lo<F16>(t[2]) + m1[0];
      o1[1] = from16_hi<F16>(t[2]) + m1[1];
      o1[2] = from16_lo<F16>(t[3]) + m1[2];
      o1[3] = from16_hi<F16>(t[3]) + m1[3];
      emit(c, o0);
      emit(c + 4, o1);
    }
  }
  if (part) {
    ss = wave_sum(ss);
    for (int k = lane; k < npart; k += 64) part[row * npart + k] = k == 0 ? ss : 0.f;
  }
}

// 16-bit copy + partial sums of squares of arbitrary fp32 rows: the inputs of a fused RMSNorm (TCAVT_EPI_ROWSCALE) when
// the rows do not come from tcavt_embed_fuse or a TCAVT_EPI_NORM_OUT epilogue (inputs_embeds given by the caller).
template <bool F16>
__global__ __launch_bounds__(256) void rownorm_prep_kernel(const float* __restrict__ x, bf16_t* __restrict__ x16,
                                                           float* __restrict__ part, long M, int H, int npart, int rounded,
                                                           float sscale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float ss = 0.f;
  for (int c = lane * 4; c < H; c += 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + row * H + c) * sscale;
    const u32x2 w = u32x2{pack16x2<F16>(v[0], v[1]), pack16x2<F16>(v[2], v[3])};
    *reinterpret_cast<u32x2*>(x16 + row * H + c) = w;
    if (rounded) v = f32x4{from16_lo<F16>(w[0]), from16_hi<F16>(w[0]), from16_lo<F16>(w[1]), from16_hi<F16>(w[1])};
    ss += v[0] * v[0];
    ss += v[1] * v[1];
    ss += v[2] * v[2];
    ss += v[3] * v[3];
  }
  ss = wave_sum(ss);
  for (int k = lane; k < npart; k += 64) part[row * npart + k] = k == 0 ? ss : 0.f;
}

// Row softmax (one wave per row): P = softmax(S[:, :n_valid]), zero-padded to n_out columns.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, long lds,
                                                           bf16_t* __restrict__ P, long ldp, int f16, int rows,
                                                           int n_valid, int n_out, DropoutP drop) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = S + (long)row * lds;
  float m = -1e30f;
  for (int c = lane; c < n_valid; c += 64) m = fmaxf(m, s[c]);
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < n_valid; c += 64) sum += __expf(s[c] - m);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  bf16_t* o = P + (long)row * ldp;
  for (int c = lane; c < n_out; c += 64) {
    float v = c < n_valid ? __expf(s[c] - m) * inv : 0.f;
    if (drop.p > 0.f && c < n_valid) v *= dropout_one(drop, (unsigned long long)row * n_out + c);
    o[c] = f16 ? __builtin_bit_cast(unsigned short, static_cast<_Float16>(v)) : f32_to_bf16(v);
  }
}

// Elementwise dropout, one Philox call per octet of elements (16-byte loads / stores of 16-bit tensors when aligned).
template <typename T, bool F16 = false>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ out, long n, DropoutP drop,
                                                      const T* __restrict__ add, int vec_ok) {
  const long no = (n + 7) >> 3;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < no; o += stride) {
    float sc[8];
    dropout_oct(drop, (unsigned long long)o, sc);
    const long i0 = 8 * o;
    if constexpr (sizeof(T) == 2) {
      if (vec_ok && i0 + 8 <= n) {
        const u32x4 xv = *reinterpret_cast<const u32x4*>(x + i0);
        u32x4 av = {0u, 0u, 0u, 0u};
        if (add) av = *reinterpret_cast<const u32x4*>(add + i0);
        u32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          ov[e] = pack16x2<F16>(from16_lo<F16>(xv[e]) * sc[2 * e] + (add ? from16_lo<F16>(av[e]) : 0.f),
                                from16_hi<F16>(xv[e]) * sc[2 * e + 1] + (add ? from16_hi<F16>(av[e]) : 0.f));
        *reinterpret_cast<u32x4*>(out + i0) = ov;
        continue;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const long i = i0 + e;
      if (i < n) {
        if constexpr (sizeof(T) == 2) out[i] = to16<F16>(from16<F16>(x[i]) * sc[e] + (add ? from16<F16>(add[i]) : 0.f));
        else out[i] = x[i] * sc[e] + (add ? add[i] : 0.f);
      }
    }
  }
}

// kv_len[b] = Nq + popcount(mask[b]); flags masks that are not a prefix of ones.
__global__ __launch_bounds__(64) void mask_to_kvlen_kernel(const int64_t* __restrict__ mask, int Lt, int Nq,
                                                           int* __restrict__ kv_len, int* __restrict__ flag) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float cnt = 0.f, bad = 0.f;
  for (int j = lane; j < Lt; j += 64) {
    const bool on = mask[(long)b * Lt + j] != 0;
    cnt += on ? 1.f : 0.f;
    if (on && j > 0 && mask[(long)b * Lt + j - 1] == 0) bad += 1.f;
  }
  cnt = wave_sum(cnt);
  bad = wave_sum(bad);
  if (lane == 0) {
    kv_len[b] = Nq + (int)cnt;
    if (bad > 0.f) *flag = 1;
  }
}

}  // namespace tcavt

using namespace tcavt;

extern "C" int tcavt_softmax_rows(const float* S, int64_t lds, void* P, int64_t ldp, int out_dtype, int rows,
                                  int n_valid, int n_out, float dropout_p, uint64_t dropout_seed,
                                  uint32_t dropout_site, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "softmax_rows: dropout_p must be in [0, 1)");
  TCAVT_CHECK_ARG(S && P && rows > 0 && n_valid > 0 && n_out >= n_valid && lds >= n_valid && ldp >= n_out,
                  "softmax_rows: bad args");
  TCAVT_CHECK_ARG(out_dtype == TCAVT_BF16 || out_dtype == TCAVT_F16, "softmax_rows: out_dtype must be bf16 or fp16");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), S,
                     (long)lds, static_cast<bf16_t*>(P), (long)ldp, out_dtype == TCAVT_F16 ? 1 : 0, rows, n_valid,
                     n_out, make_dropout(dropout_p, dropout_seed, dropout_site));
  TCAVT_CHECK_LAUNCH("softmax_rows");
  return TCAVT_OK;
}

extern "C" int tcavt_dropout(const void* x, void* out, int64_t n, int dtype, float p, uint64_t seed, uint32_t site,
                             const void* add, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(x && out && n > 0 && p >= 0.f && p < 1.f, "dropout: bad args");
  TCAVT_CHECK_ARG(dtype == TCAVT_F32 || is16(dtype), "dropout: dtype must be f32, bf16 or fp16");
  long blocks = ((n + 7) / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const DropoutP d = make_dropout(p, seed, site);
  const int vec_ok = aligned16(x) && aligned16(out) && (!add || aligned16(add));
  if (dtype == TCAVT_F32)
    hipLaunchKernelGGL(dropout_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(x), static_cast<float*>(out), (long)n, d, static_cast<const float*>(add), vec_ok);
  else if (dtype == TCAVT_F16)
    hipLaunchKernelGGL((dropout_kernel<bf16_t, true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(x), static_cast<bf16_t*>(out), (long)n, d, static_cast<const bf16_t*>(add), vec_ok);
  else
    hipLaunchKernelGGL((dropout_kernel<bf16_t, false>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(x), static_cast<bf16_t*>(out), (long)n, d, static_cast<const bf16_t*>(add), vec_ok);
  TCAVT_CHECK_LAUNCH("dropout");
  return TCAVT_OK;
}

extern "C" int tcavt_mask_to_kvlen(const int64_t* mask, int B, int Lt, int Nq, int32_t* kv_len,
                                   int* not_prefix_flag, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(mask && kv_len && not_prefix_flag && B > 0 && Lt > 0 && Nq >= 0, "mask_to_kvlen: bad args");
  hipLaunchKernelGGL(mask_to_kvlen_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), mask, Lt, Nq,
                     kv_len, not_prefix_flag);
  TCAVT_CHECK_LAUNCH("mask_to_kvlen");
  return TCAVT_OK;
}

extern "C" int tcavt_set_dropout_epoch(const uint64_t* epoch_dev) {
  g_dropout_epoch = reinterpret_cast<const unsigned long long*>(epoch_dev);
  return TCAVT_OK;
}

extern "C" int tcavt_dropout_epoch_advance(uint64_t* epoch_dev, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(epoch_dev, "dropout_epoch_advance: null pointer");
  hipLaunchKernelGGL(epoch_advance_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<unsigned long long*>(epoch_dev));
  TCAVT_CHECK_LAUNCH("dropout_epoch_advance");
  return TCAVT_OK;
}

extern "C" int tcavt_abi_version(void) { return TCAVT_ABI_VERSION; }

extern "C" const char* tcavt_last_error(void) { return g_err; }

extern "C" int tcavt_init(int device, int* num_cus) {
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) {
    set_error("tcavt_init: hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
    return TCAVT_ERR_HIP;
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    set_error("tcavt_init: hipGetDeviceProperties failed: %s", hipGetErrorString(e));
    return TCAVT_ERR_HIP;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("tcavt_init: device %d is %s; this library is built for gfx950 only", device,
              prop.gcnArchName);
    return TCAVT_ERR_ARCH;
  }
  if (num_cus) *num_cus = prop.multiProcessorCount;
  return TCAVT_OK;
}

extern "C" int tcavt_rmsnorm(const float* x, const float* gamma, float eps, void* out_bf16,
                             float* out_f32, int M, int H, void* out_drop_bf16, float dropout_p,
                             uint64_t dropout_seed, uint32_t dropout_site, int dtype16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(x && gamma && (out_bf16 || out_f32), "rmsnorm: null pointer");
  TCAVT_CHECK_ARG(is16(dtype16), "rmsnorm: dtype16 must be TCAVT_BF16 or TCAVT_F16");
  TCAVT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f && (!out_drop_bf16 || dropout_p > 0.f),
                  "rmsnorm: out_drop needs 0 < dropout_p < 1");
  bf16_t* od = static_cast<bf16_t*>(out_drop_bf16);
  const DropoutP drop = make_dropout(dropout_p, dropout_seed, dropout_site);
  TCAVT_CHECK_ARG(M > 0 && H > 0 && H % 8 == 0, "rmsnorm: H=%d must be a multiple of 8", H);
  TCAVT_CHECK_ARG(aligned16(x) && aligned16(gamma), "rmsnorm: unaligned input");
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((M + 3) / 4), block(256);
  bf16_t* ob = static_cast<bf16_t*>(out_bf16);
#define TCAVT_RMS(NV)                                                                                                  \
  do {                                                                                                                 \
    if (dtype16 == TCAVT_F16) hipLaunchKernelGGL((rmsnorm_kernel<NV, true>), grid, block, 0, s, x, gamma, eps, ob, out_f32, M, H, od, drop); \
    else hipLaunchKernelGGL((rmsnorm_kernel<NV, false>), grid, block, 0, s, x, gamma, eps, ob, out_f32, M, H, od, drop); \
  } while (0)
  switch (H % 256 == 0 ? H / 256 : 0) {
    case 1: TCAVT_RMS(1); break;
    case 2: TCAVT_RMS(2); break;
    case 3: TCAVT_RMS(3); break;
    case 4: TCAVT_RMS(4); break;
    case 8: TCAVT_RMS(8); break;
    case 16: TCAVT_RMS(16); break;
    default: TCAVT_RMS(0); break;
  }
#undef TCAVT_RMS
  TCAVT_CHECK_LAUNCH("rmsnorm");
  return TCAVT_OK;
}

extern "C" int tcavt_layernorm(const float* x, const float* residual, const float* gamma,
                               const float* beta, float eps, float* out_f32, void* out_bf16, int M,
                               int D, int dtype16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(x && gamma && beta && (out_f32 || out_bf16), "layernorm: null pointer");
  TCAVT_CHECK_ARG(is16(dtype16), "layernorm: dtype16 must be TCAVT_BF16 or TCAVT_F16");
  TCAVT_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 4096, "layernorm: D=%d must be a multiple of 4 and <= 4096", D);
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((M + 3) / 4), block(256);
  bf16_t* ob = static_cast<bf16_t*>(out_bf16);
  const int nvec = D / 4;
#define TCAVT_LN(NV)                                                                                                    \
  do {                                                                                                                  \
    if (dtype16 == TCAVT_F16) hipLaunchKernelGGL((layernorm_kernel<NV, true>), grid, block, 0, s, x, residual, gamma, beta, eps, out_f32, ob, M, D); \
    else hipLaunchKernelGGL((layernorm_kernel<NV, false>), grid, block, 0, s, x, residual, gamma, beta, eps, out_f32, ob, M, D); \
  } while (0)
  if (nvec <= 64) TCAVT_LN(1);
  else if (nvec <= 256) TCAVT_LN(4);
  else TCAVT_LN(16);
#undef TCAVT_LN
  TCAVT_CHECK_LAUNCH("layernorm");
  return TCAVT_OK;
}

extern "C" int tcavt_cast_f32_16(const float* x, void* out_bf16, int64_t n, int dtype16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(x && out_bf16 && n > 0 && is16(dtype16), "cast: null pointer, n <= 0 or bad dtype16");
  TCAVT_CHECK_ARG(aligned16(x) && aligned16(out_bf16), "cast: unaligned pointer");
  long blocks = (n / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  if (dtype16 == TCAVT_F16)
    hipLaunchKernelGGL(cast_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       static_cast<bf16_t*>(out_bf16), (long)n);
  else
    hipLaunchKernelGGL(cast_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       static_cast<bf16_t*>(out_bf16), (long)n);
  TCAVT_CHECK_LAUNCH("cast_f32_16");
  return TCAVT_OK;
}

// ---------------------------------------------------------------------------
// Batched host -> device copy AS A KERNEL: up to 16 (dst, src, bytes) items in one launch, src = pinned (device-mapped) host
// memory read over the host link by the copying lanes.  The training loop uploads a ~1.4 MB batch of nine small tensors per step
// (scripts/train.py:1153-1166): one launch in the stream's queue, ordered like any other kernel, instead of nine copy-engine
// transfers with their engine hand-offs.  At this size the link's latency, not its bandwidth, is what the copy costs (tens of
// microseconds, on a stream that is idle then).
// ---------------------------------------------------------------------------
struct CopyBatch {
  void* dst[16];
  const void* src[16];
  long bytes[16];
};

__global__ __launch_bounds__(256) void copy_batch_kernel(CopyBatch b) {
  const int it = blockIdx.y;
  const long n = b.bytes[it];
  char* d = static_cast<char*>(b.dst[it]);
  const char* s_ = static_cast<const char*>(b.src[it]);
  const long n16 = n >> 4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
    reinterpret_cast<u32x4*>(d)[i] = reinterpret_cast<const u32x4*>(s_)[i];
  const long tail0 = n16 << 4;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < n - tail0) d[tail0 + gid] = s_[tail0 + gid];
}

extern "C" int tcavt_copy_batch(void* const* dst, const void* const* src, const int64_t* bytes, int n, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(dst && src && bytes && n > 0 && n <= 16, "copy_batch: 1 .. 16 items");
  CopyBatch b;
  long most = 0;
  for (int i = 0; i < 16; ++i) {
    const bool on = i < n;
    b.dst[i] = on ? dst[i] : nullptr;
    b.src[i] = on ? src[i] : nullptr;
    b.bytes[i] = on ? (long)bytes[i] : 0;
    if (on) {
      TCAVT_CHECK_ARG(dst[i] && src[i] && bytes[i] >= 0 && aligned16(dst[i]) && aligned16(src[i]), "copy_batch: item %d: null / unaligned / negative", i);
      most = bytes[i] > most ? bytes[i] : most;
    }
  }
  long blocks = (most / 16 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  hipLaunchKernelGGL(copy_batch_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream), b);
  TCAVT_CHECK_LAUNCH("copy_batch");
  return TCAVT_OK;
}

extern "C" int tcavt_embed_fuse(const void* table_bf16, const int64_t* ids, const float* img,
                                const float* vis_mod, const float* txt_mod, float* h, int B, int Nq,
                                int Lt, int H, int V, int* bad_id_flag, int table_dtype, void* h16, float* part,
                                int npart, float stream_scale, tcavt_stream_t stream) {
  return tcavt::embed_fuse_impl(table_bf16, ids, img, vis_mod, txt_mod, h, B, Nq, Lt, H, V, bad_id_flag, table_dtype, h16, part, npart,
                                stream_scale, 0, stream);
}

int tcavt::embed_fuse_impl(const void* table_bf16, const int64_t* ids, const float* img, const float* vis_mod, const float* txt_mod,
                           float* h, int B, int Nq, int Lt, int H, int V, int32_t* bad_id_flag, int table_dtype, void* h16, float* part,
                           int npart, float stream_scale, int frag16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(!frag16 || ((frag16 == 1 || frag16 == 2) && (long)B * (Nq + Lt) <= (frag16 == 2 ? 8 : 32) && H % 32 == 0),
                  "embed_fuse: fragment-major rows: mode 1 (at most 32 rows) or 2 (at most 8), H %% 32 == 0");
  TCAVT_CHECK_ARG(table_bf16 && ids && img && vis_mod && txt_mod && (h || h16) && bad_id_flag, "embed_fuse: null pointer");
  TCAVT_CHECK_ARG(stream_scale >= 0.f && stream_scale <= 1.f, "embed_fuse: stream_scale must be in (0, 1] (0 means 1)");
  const float sscale = stream_scale == 0.f ? 1.f : stream_scale;
  TCAVT_CHECK_ARG(is16(table_dtype), "embed_fuse: table_dtype must be TCAVT_BF16 or TCAVT_F16");
  TCAVT_CHECK_ARG((h16 == nullptr) == (part == nullptr) && (!part || npart > 0), "embed_fuse: h16 and part go together (npart > 0)");
  TCAVT_CHECK_ARG(B > 0 && Nq >= 0 && Lt >= 0 && Nq + Lt > 0 && H % 8 == 0 && V > 0, "embed_fuse: bad shape");
  const long rows = (long)B * (Nq + Lt);
  if (table_dtype == TCAVT_F16)
    hipLaunchKernelGGL(embed_fuse_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(table_bf16), ids, img,
                       vis_mod, txt_mod, h, B, Nq, Lt, H, V, bad_id_flag, static_cast<bf16_t*>(h16), part, npart, sscale, frag16);
  else
    hipLaunchKernelGGL(embed_fuse_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(table_bf16), ids, img,
                       vis_mod, txt_mod, h, B, Nq, Lt, H, V, bad_id_flag, static_cast<bf16_t*>(h16), part, npart, sscale, frag16);
  TCAVT_CHECK_LAUNCH("embed_fuse");
  return TCAVT_OK;
}

extern "C" int tcavt_rownorm_prep(const float* x, void* x16, float* part, int64_t M, int H, int npart, int dtype16,
                                  int rounded_sums, float stream_scale, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(x && x16 && part && M > 0 && H > 0 && H % 4 == 0 && npart > 0 && is16(dtype16), "rownorm_prep: bad args");
  TCAVT_CHECK_ARG(stream_scale >= 0.f && stream_scale <= 1.f, "rownorm_prep: stream_scale must be in (0, 1] (0 means 1)");
  const float sscale = stream_scale == 0.f ? 1.f : stream_scale;
  TCAVT_CHECK_ARG(aligned16(x) && aligned16(x16), "rownorm_prep: unaligned pointer");
  const dim3 grid((unsigned)((M + 3) / 4)), block(256);
  if (dtype16 == TCAVT_F16)
    hipLaunchKernelGGL(rownorm_prep_kernel<true>, grid, block, 0, static_cast<hipStream_t>(stream), x,
                       static_cast<bf16_t*>(x16), part, (long)M, H, npart, rounded_sums, sscale);
  else
    hipLaunchKernelGGL(rownorm_prep_kernel<false>, grid, block, 0, static_cast<hipStream_t>(stream), x,
                       static_cast<bf16_t*>(x16), part, (long)M, H, npart, rounded_sums, sscale);
  TCAVT_CHECK_LAUNCH("rownorm_prep");
  return TCAVT_OK;
}
