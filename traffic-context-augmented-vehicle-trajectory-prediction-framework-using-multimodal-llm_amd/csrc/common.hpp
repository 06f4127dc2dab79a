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

}  // namespace tcavt
