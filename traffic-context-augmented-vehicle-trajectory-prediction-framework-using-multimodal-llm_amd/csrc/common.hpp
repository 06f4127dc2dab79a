// Shared host/device helpers for the gfx950 kernels behind include/tcavt.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/tcavt.h"

namespace tcavt {

typedef unsigned short bf16_t;  // raw storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- error plumbing -------------------------------------------------------
void set_error(const char* fmt, ...);
#define TCAVT_CHECK_ARG(cond, ...)                 \
  do {                                             \
    if (!(cond)) {                                 \
      tcavt::set_error(__VA_ARGS__);               \
      return TCAVT_ERR_ARG;                        \
    }                                              \
  } while (0)
#define TCAVT_CHECK_LAUNCH(name)                                            \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      tcavt::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return TCAVT_ERR_HIP;                                                 \
    }                                                                       \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- bf16 <-> f32 ---------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(static_cast<unsigned int>(v) << 16);
}
// round-to-nearest-even; plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = static_cast<__bf16>(f);
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  return static_cast<unsigned int>(f32_to_bf16(lo)) |
         (static_cast<unsigned int>(f32_to_bf16(hi)) << 16);
}

// ---- fp16 storage (the default 16-bit storage type of the forward path; bf16 stays for gradient-side tensors) ----------
// IEEE half, round-to-nearest-even; a value beyond +-65504 becomes inf and surfaces as a non-finite loss (it is outside
// the storage contract, DESIGN.md "Precision contract") rather than being clamped silently.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ bf16_t f32_to_f16(float f) {
  return __builtin_bit_cast(unsigned short, static_cast<_Float16>(f));
}
__device__ __forceinline__ float f16_to_f32(bf16_t v) {
  return static_cast<float>(__builtin_bit_cast(_Float16, v));
}
__device__ __forceinline__ unsigned int pack_f16x2(float lo, float hi) {
  // one v_cvt_pk_f16_f32 (round-to-nearest-even, overflow -> inf): two scalar casts + shift/or were 3 instructions per pair
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(f2_t{lo, hi}, h2_t));
}
// 16-bit storage type chosen at compile time: F16 = IEEE half, otherwise bf16 (raw storage is `bf16_t` = uint16 either way)
template <bool F16>
__device__ __forceinline__ unsigned int pack16x2(float lo, float hi) {
  if constexpr (F16) return pack_f16x2(lo, hi);
  else return pack_bf16x2(lo, hi);
}
template <bool F16>
__device__ __forceinline__ bf16_t to16(float f) {
  if constexpr (F16) return f32_to_f16(f);
  else return f32_to_bf16(f);
}
template <bool F16>
__device__ __forceinline__ float from16(bf16_t v) {
  if constexpr (F16) return f16_to_f32(v);
  else return bf16_to_f32(v);
}
// low / high half of a packed pair
template <bool F16>
__device__ __forceinline__ float from16_lo(unsigned int w) { return from16<F16>(static_cast<bf16_t>(w & 0xffffu)); }
template <bool F16>
__device__ __forceinline__ float from16_hi(unsigned int w) { return from16<F16>(static_cast<bf16_t>(w >> 16)); }

// nonzero when either half of a packed fp16 pair is +-inf or NaN (exponent field all ones)
__device__ __forceinline__ unsigned int half_is_inf2(unsigned int w) {
  return (unsigned)((w & 0x7C00u) == 0x7C00u) | (unsigned)((w & 0x7C000000u) == 0x7C000000u);
}

// ReLU as torch computes it: NaN stays NaN (fmaxf(NaN, 0) = 0 would turn an upstream overflow into finite garbage -- the
// LTSF head's fusion layer is LN -> Linear -> ReLU -> Linear, so a NaN from the decoder would come out as a finite trajectory)
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }

static inline bool is16(int dtype) { return dtype == TCAVT_BF16 || dtype == TCAVT_F16; }

// tcavt_gemm_bf16 runs its skinny form (csrc/gemm_bf16.hip) for these shapes when no tile is forced ...
static inline bool skinny_shape(int M, int K) { return M <= 32 && K % 256 == 0; }
// ... and TCAVT_EPI_NORM_OUT then writes one partial sum of squares per 16 output columns instead of one per 64:
// the number of partials per row a consumer (TCAVT_EPI_ROWSCALE: rowscale_npart) has to add up
// (32 columns per workgroup for M > 16, to halve the activation re-reads, made the B = 32 decode step slower: 2.05 vs 1.91 ms --
// half as many workgroups streaming weights costs more than the activation bytes save)
// (8 columns per workgroup for the N = 2048 projections, so that all 256 CUs get one: also slower, 1.37 vs 1.27 ms at B = 8 --
// every workgroup pays the same load instructions, LDS reduction and barrier for half the columns, and the consumers add up
// twice the partial sums)
static inline int norm_out_npart(int M, int N, int K) { return skinny_shape(M, K) ? N / 16 : N / 64; }

// Fragment-major activations of the decode step (tcavt.h: TCAVT_ACT_*_FRAG16): element (m, f) of a [<= 32][K] 16-bit operand.
// Tokens in blocks of 16, columns in steps of 32: block (m >> 4) at 16 K elements, step (f >> 5) a 1 KiB chunk, inside it lane
// 16 q + r (r = m & 15, q = (f >> 3) & 3) holds columns 8 q .. 8 q + 7 of the step -- the B-operand layout of v_mfma_f32_16x16x32.
__host__ __device__ __forceinline__ long frag16_off(int m, int f, int K) {
  return (long)(m >> 4) * 16 * K + (long)(f >> 5) * 512 + ((f >> 3) & 3) * 128 + (m & 15) * 8 + (f & 7);
}
// ... and for at most 8 tokens (TCAVT_ACT_BLOCK8): one block of 8, a k-step is 512 bytes -- lanes r and r + 8 of the consuming wave
// read the same 16 bytes (the empty token slots repeat the real ones), so an instruction touches 512 consecutive bytes.
// mode: 1 = blocks of 16 tokens, 2 = one block of 8.
__host__ __device__ __forceinline__ long frag_off(int m, int f, int K, int mode) {
  return mode == 2 ? (long)(f >> 5) * 256 + ((f >> 3) & 3) * 64 + (m & 7) * 8 + (f & 7) : frag16_off(m, f, K);
}

// ---- wave / block reductions (wave = 64 lanes) ----------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// csrc-internal forms of tcavt_embed_fuse / tcavt_rmsnorm16 with a layout switch for their 16-bit rows (frag16 != 0: the rows
// are written / read through frag16_off; B * (Nq + Lt) resp. M <= 32).  The exported entry points call them with 0.
int embed_fuse_impl(const void* table16, const int64_t* ids, const float* img, const float* vis_mod, const float* txt_mod, float* h,
                    int B, int Nq, int Lt, int H, int V, int32_t* bad_id_flag, int table_dtype, void* h16, float* part, int npart,
                    float stream_scale, int frag16, tcavt_stream_t stream);
int rmsnorm16_impl(const void* x16, const float* gamma, float eps, void* out16, float* out_f32, int M, int H, int dtype16, int frag16,
                   tcavt_stream_t stream);

}  // namespace tcavt
