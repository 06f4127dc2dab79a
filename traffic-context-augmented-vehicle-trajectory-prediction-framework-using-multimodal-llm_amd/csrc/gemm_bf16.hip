// bf16 x bf16 -> fp32-accumulate contraction for gfx950 (MI355X), hand-tiled for
// 64-wide waves and v_mfma_f32_16x16x32_bf16.
//
//   C[M,N] = A[M,K] . W[N,K]^T (+ A2[M,K2] . W2[N,K2]^T)   + fused epilogue
//
// Both operands are K-contiguous (activations [tokens][features], nn.Linear
// weights [out][in]), so they are staged with the same code: 64-deep K-tiles go
// global -> LDS by 16-byte LDS-DMA (global_load_lds_dwordx4, no VGPR round trip),
// double-buffered.  An LDS row is one tile row's 128 bytes; the eight 16-byte
// chunks of a row are XOR-swizzled with (row>>1)&7 so that every ds_read_b128
// fragment read is bank-conflict free.  Because LDS-DMA writes lane-linear, the
// swizzle is applied to the per-lane SOURCE address and to the fragment READ
// address (same involution), never to the destination.
//
// The MFMA is issued "swapped": the weight rows are the A operand and the token
// rows the B operand, so each lane ends with 4 consecutive output FEATURES of
// one token in a register quad -> 8/16-byte row-contiguous epilogue accesses,
// and RoPE / SiLU*up partners are lane-local.
//
// Replaces the nn.Linear contractions listed in include/tcavt.h (reference:
// scripts/train.py:401-406,446-452,493,754-757 and HF modeling_llama.py
// :174-176,254-256,279-280).
#include "common.hpp"
#include "philox.hpp"
#include <stdlib.h>
#include <type_traits>

namespace tcavt {

struct GemmP {
  const bf16_t* A;
  const bf16_t* W;
  const bf16_t* A2;
  const bf16_t* W2;
  void* C;
  const float* bias;
  const float* residual;
  const float* cosT;
  const float* sinT;
  long lda, ldw, lda2, ldw2, ldc, ldr;
  int M, N, K, K2;
  int out_kind, flags, rope_L, rope_cols;  // out_kind: TCAVT_F32 / TCAVT_BF16 / TCAVT_F16
  int tiles_m, tiles_n;
  float acc_scale;
  int batch_inner;
  bf16_t* aux;  // EPI_SILU_SAVE: gate|up pre-activations [M, N] bf16
  long ldaux;
  int w_group;  // >= 1: the inner batch index is divided by this for W (grouped-query heads share one W)
  long sAo, sAi, sWo, sWi, sCo, sCi;
  int xcd_gx;  // XCD partition of the tile grid (block_to_tile)
  DropoutP drop;  // epilogue dropout (generic epilogue only)
  int pers_tiles;  // > 0: persistent launch of the 4-wave kernel, workgroup w runs tiles w, w + gridDim.x, ... < pers_tiles
  int prio;  // wave-priority experiment: 0 none, 1 static s_setprio(1) for the upper half of the waves, 2 around MFMA clusters
  // ---- RMSNorm fused into the GEMMs around it (TCAVT_EPI_NORM_OUT / TCAVT_EPI_ROWSCALE, include/tcavt.h)
  bf16_t* norm_h16;       // NORM_OUT: 16-bit copy of the fp32 output rows (leading dimension ldc)
  float* norm_part;       // NORM_OUT: [M][N / 64] sums of squares of the fp32 output, one per 64-column group
  const float* rs_part;   // ROWSCALE: [M][rs_npart] sums of squares of the row the A operand was rounded from
  int rs_npart;
  float rs_eps, rs_inv_h;
  const int* rope_pos;    // ROPE: position of row m (decode step: one row per sample); NULL: m % rope_L
  const bf16_t* res16;    // NORM16: where the 16-bit residual is read from (norm_h16 itself unless the caller keeps every layer's stream)
  // skinny form, split K across workgroups (decode step): S = sk_split workgroups share one block of output columns, each
  // over K / S; partial sums meet in sk_slab, the last arriver (ticket in sk_cnt) adds them in slice order and finishes
  // skinny form (decode step): the NEXT layer's LoRA down-projection folded into this layer's residual GEMM and q|k|v GEMM.
  // Producer (NORM_OUT forms, lp_a != nullptr): every workgroup also writes t_part[blk][m][16] = its 16 output columns of the
  // rounded stream times the 16 adapter rows (A_q rows 0..7, A_v rows LORA_V..LORA_V+7 of a_cat).  Consumer (RoPE form,
  // lp_np > 0): t = round16(lp_scale * sum over the lp_np partials, in index order) replaces the A2 operand (K2 = 32).
  float* lp_part;
  const bf16_t* lp_a;
  long lp_lda;
  int lp_np;
  float lp_scale;
  // skinny form, M > 16: the two 16-token blocks of a column block go to TWO workgroups (sk_msplit = 2) instead of one that
  // loads both blocks' activation rows for every weight fragment (twice the weight bytes through the CU's load path)
  int sk_msplit;
  int a_frag, o_frag;  // skinny form: A / the 16-bit result in fragment-major order (tcavt_gemm_args.act_layout): 0, 1 = blocks of 16
                       // tokens, 2 = one block of 8 (common.hpp frag_off)
  int w_frag;  // skinny form: W is the fragment-major copy of tcavt_pack_weight16 (tcavt_gemm_args.w_layout)
  int sk_split;
  float* sk_slab;
  int* sk_cnt;
  long sk_slab_bytes;
  int sk_cnt_n;
  int* nf_flag;           // NORM_OUT: receives nf_tag (CAS from 0) when a partial sum / rounded element is not finite
  int nf_tag;
  float norm_scale;       // NORM_OUT: the 16-bit image of the stream (and its partial sums) holds norm_scale * x (tcavt_gemm_args.norm_scale)
};

// a * s + b, one rounding (s = 1: exactly a + b, so the default scale leaves every result bit for bit as it was)
__device__ __forceinline__ f32x4 fma4(const f32x4& a, float s, const f32x4& b) {
  return __builtin_elementwise_fma(a, f32x4{s, s, s, s}, b);
}

// a non-finite partial sum of squares (inf: a rounded element overflowed; NaN: inf / NaN came in from upstream)
__device__ __forceinline__ void flag_nonfinite(const GemmP& p, float ss) {
  if (p.nf_flag && !(ss <= 3.0e38f)) atomicCAS(p.nf_flag, 0, p.nf_tag);
}

// 1 / rms of row m from its partial sums of squares, added in index order (bit-reproducible; rs_npart % 4 == 0)
__device__ __forceinline__ float row_rscale(const GemmP& p, long m) {
  const f32x4* q = reinterpret_cast<const f32x4*>(p.rs_part + m * p.rs_npart);
  float ss = 0.f;
  const int nq = p.rs_npart >> 2;
  // eight quads (H = 2048: all of them) in flight together -- one load per step, each waited for, was eight dependent
  // L2 round trips per output tile of the persistent kernel
  for (int i0 = 0; i0 < nq; i0 += 8) {
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = q[min(i0 + i, nq - 1)];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i0 + i < nq) {
        ss += v[i][0];
        ss += v[i][1];
        ss += v[i][2];
        ss += v[i][3];
      }
    }
  }
  return rsqrtf(ss * p.rs_inv_h + p.rs_eps);
}

// EPI_DROP = EPI_GENERIC + Philox dropout.  A separate instantiation: with the mask code inside the generic
// epilogue the 256x256 kernel spilled its accumulators (528 B/lane of scratch, 3x slower).
// EPI_SILU_SAVE = EPI_SILU + a bf16 copy of the gate|up pre-activations (LoRA-trainable variant: the backward of
// silu(gate)*up needs them); its own instantiation so that the production SiLU kernel keeps its register allocation.
// EPI_NORM = TCAVT_EPI_NORM_OUT (fp32 residual output + 16-bit copy + per-row partial sums of squares): its own
// instantiation as well -- inside EPI_GENERIC it pushed the 4-wave kernel's generic form into 460 bytes of scratch.
// EPI_NORM16 = the same with C == NULL (16-bit residual stream, updated in place): again its own instantiation (both bodies in
// one kernel spilled 150-500 bytes per lane in the 4-wave kernel).
// EPI_SILUBWD = TCAVT_EPI_SILU_BWD (4-wave kernel only).
enum { EPI_GENERIC = 0, EPI_SILU = 1, EPI_ROPE = 2, EPI_DROP = 3, EPI_SILU_SAVE = 4, EPI_NORM = 5, EPI_NORM16 = 6, EPI_SILUBWD = 7 };

__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)src,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void store_quad(const GemmP& p, int m, int n, f32x4 v) {
  if (p.out_kind == TCAVT_BF16) {
    u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n) = o;
  } else if (p.out_kind == TCAVT_F16) {
    u32x2 o = {pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n) = o;
  } else {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = v;
  }
}

// ---------------------------------------------------------------------------
// Epilogue shared by both main-loop variants.  acc[i][j] holds, for n-tile i and m-tile j of this
// wave, features n..n+3 (n = n_base + 16 i + 4 (lane >> 4)) of token m = m_base + 16 j + (lane & 15).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float silu_mul(float g, float u) {
  // g * sigmoid(g) * u with v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE division here cost ~10 VALU
  // instructions per output, 128 outputs per lane, with nothing to overlap them (one tile per CU at a time)
  return g * __builtin_amdgcn_rcpf(1.f + __expf(-g)) * u;
}

typedef __attribute__((ext_vector_type(2))) float f32x2;

// silu(g) * u for the lane's four consecutive features, written on natural register pairs so that the products are
// v_pk_mul_f32 (two outputs per issue slot; left to itself the vectoriser paired (0,2),(1,3) and paid for it in v_mov and
// re-interleaving instructions: ~50 instructions per quad, this is ~24).  Same arithmetic, same order as silu_mul.
template <bool F16>
__device__ __forceinline__ u32x2 silu_mul_quad(const f32x4& g, const f32x4& u) {
  const f32x2 g0 = {g[0], g[1]}, g1 = {g[2], g[3]}, u0 = {u[0], u[1]}, u1 = {u[2], u[3]};
  const f32x2 t0 = g0 * -1.44269504088896340736f, t1 = g1 * -1.44269504088896340736f;  // __expf(-g) = exp2(-g log2 e)
  const f32x2 d0 = f32x2{__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])} + 1.f;
  const f32x2 d1 = f32x2{__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])} + 1.f;
  const f32x2 r0 = {__builtin_amdgcn_rcpf(d0[0]), __builtin_amdgcn_rcpf(d0[1])};
  const f32x2 r1 = {__builtin_amdgcn_rcpf(d1[0]), __builtin_amdgcn_rcpf(d1[1])};
  const f32x2 o0 = g0 * r0 * u0, o1 = g1 * r1 * u1;
  return u32x2{pack16x2<F16>(o0[0], o0[1]), pack16x2<F16>(o1[0], o1[1])};
}

// ---- 16-byte epilogue accesses -------------------------------------------------------------------------------------------
// A lane holds four consecutive features (8 bytes as 16-bit values) of one token per 16x16 MFMA tile; the lane 16 further on
// holds the next four.  v_permlane16_swap (odd 16-lane rows of the first operand <-> even rows of the second) applied to the
// packed quads of two column-adjacent tiles a, b leaves EIGHT consecutive features in every lane -- even rows: tile a,
// features 4q .. 4q+7; odd rows: tile b, features 4(q-1) .. 4(q-1)+7 -- i.e. one global_store_dwordx4 instead of two
// dwordx2 (the store tail of these epilogues is issue-bound: half the instructions, same bytes, same addresses).  The swap is
// an involution, so a 16-byte LOAD from the same address followed by the same swap returns the two quads.
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
// element offset of this lane's 16 bytes relative to column 0 of tile a (tile b follows at column 16)
__device__ __forceinline__ int pair16_off(int lane) {
  const int q = lane >> 4;
  return (q & 1) ? 16 + 4 * (q - 1) : 4 * q;
}
__device__ __forceinline__ void store_pair16(bf16_t* row_pair, int off, const u32x2& a, const u32x2& b) {
  unsigned a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
  swap16(a0, b0);
  swap16(a1, b1);
  *reinterpret_cast<u32x4*>(row_pair + off) = u32x4{a0, a1, b0, b1};
}
__device__ __forceinline__ void unswap_pair16(const u32x4& v, u32x2& a, u32x2& b) {
  unsigned a0 = v[0], a1 = v[1], b0 = v[2], b1 = v[3];
  swap16(a0, b0);
  swap16(a1, b1);
  a = u32x2{a0, a1};
  b = u32x2{b0, b1};
}

// WHOLE_ONLY: the caller guarantees whole tiles (the 4-wave kernel); the bounds-checked paths are compiled out where
// a fast path covers the form.
// F16: the operands' 16-bit type; the fast paths below write 16-bit outputs of that same type (OUT16).
// rs_lds (ROWSCALE, optional): the row scales of this wave's rows already in LDS (rs_lds[16 j + (lane & 15)] for m-tile j;
// the 4-wave kernel computes them once per output tile while the first operands are in flight); otherwise they are
// summed here from the partials, all TM rows' loads in flight together.
// LEGACY (experiments build only, A/B): 1 = the 8-byte-access epilogues of rounds 1-2
// (the accumulator argument is the kernel's own f32x4 [TN][TM] array, or -- experiments build -- a view that hands out quads of a
//  differently shaped accumulator file; pin_acc re-pins the registers behind element (i, j) in front of a row's arithmetic)
template <int TN, int TM>
__device__ __forceinline__ void pin_acc(f32x4 (&acc)[TN][TM], int i, int j) {
  asm volatile("" : "+a"(acc[i][j]));
}
#ifdef TCAVT_EXPERIMENTS
struct Acc32View {  // TIMING ONLY: quad ((i & 1) * 2 + (j & 1)) of the 32x32 accumulator [i / 2][j / 2] (not where that MFMA leaves (i, j))
  f32x16 (&a)[4][4];
  struct Row {
    f32x16 (&a)[4][4];
    int i;
    __device__ __forceinline__ f32x4 operator[](int j) const {
      const f32x16& t = a[i >> 1][j >> 1];
      const int q = (i & 1) * 2 + (j & 1);
      return f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
    }
  };
  __device__ __forceinline__ Row operator[](int i) const { return Row{a, i}; }
};
__device__ __forceinline__ void pin_acc(Acc32View& v, int i, int j) {
  if ((i & 1) == 0 && (j & 1) == 0) asm volatile("" : "+a"(v.a[i >> 1][j >> 1]));
}
#endif
template <int TM, int TN, int EPI, bool WHOLE_ONLY = false, bool F16 = false, int LEGACY = 0, class ACC>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, ACC& acc, int m_base, int n_base, int lane,
                                              const float* rs_lds = nullptr) {
  constexpr int OUT16 = F16 ? TCAVT_F16 : TCAVT_BF16;
  float rsv[TM];
  if constexpr (EPI == EPI_SILU || EPI == EPI_SILU_SAVE || EPI == EPI_ROPE) {
    if (rs_lds) {
#pragma unroll
      for (int j = 0; j < TM; ++j) rsv[j] = rs_lds[j * 16 + (lane & 15)];
    } else if (p.rs_part) {
      // same summation order as row_rscale (four partials per step, in index order): bit-identical to the LDS path
      float ss[TM];
      const f32x4* q[TM];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        ss[j] = 0.f;
        q[j] = reinterpret_cast<const f32x4*>(p.rs_part + (long)min(m_base + j * 16 + (lane & 15), p.M - 1) * p.rs_npart);
      }
      for (int i = 0; i < (p.rs_npart >> 2); ++i) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const f32x4 v = q[j][i];
          ss[j] += v[0];
          ss[j] += v[1];
          ss[j] += v[2];
          ss[j] += v[3];
        }
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) rsv[j] = rsqrtf(ss[j] * p.rs_inv_h + p.rs_eps);
    } else {
#pragma unroll
      for (int j = 0; j < TM; ++j) rsv[j] = 1.f;
    }
  }
  // ---- epilogue: lane holds features n..n+3 of token m in acc[i][j]
  const int nq = 4 * (lane >> 4);
  const int ml = lane & 15;
  // Fast paths for the forms the decoder launches (whole wave tile inside the matrix): row pointers hoisted,
  // no per-quad flag tests -- the general path below costs ~55 instructions per quad, these ~8.
  const bool whole = WHOLE_ONLY || (m_base + TM * 16 <= p.M && n_base + TN * 16 <= p.N);  // wave-uniform
  if constexpr ((EPI == EPI_NORM || EPI == EPI_NORM16) && TN % 4 == 0) {
    // o_proj / down_proj of the decoder with the NEXT RMSNorm's input side fused in: besides the fp32 residual stream
    // the epilogue leaves its 16-bit copy (the next projection's A operand; gamma is folded into that projection's
    // weights) and, per 64-column group, the row's partial sum of squares -- the consumer adds the N / 64 partials in
    // index order and applies rsqrt(mean + eps) as a row scale (TCAVT_EPI_ROWSCALE).  No float atomics anywhere.
    // The residual loads and the stores go to the same buffer (in place), so the compiler keeps them in program order:
    // written load-add-store per group, every group cost one full memory latency (16 groups per wave, ~25 us per tile
    // with every CU in its epilogue at once).  Hence the loads are issued up front, D rows (m-tiles) ahead of their use:
    // all of them in the 4-wave kernel, whose accumulators sit in AGPRs and whose operand registers are dead by now.
    const bool res = p.flags & TCAVT_EPI_RESIDUAL;
    const int npart = p.N >> 6;
    if constexpr (EPI == EPI_NORM16) {
      // 16-bit residual stream (eval / frozen-decoder passes): norm_h16 IS the stream -- read, added to and rewritten in
      // place by the lane that owns the element; the partial sums are of the rounded values, i.e. of what the consumer
      // multiplies.  4 bytes per element instead of 10.
      if constexpr (WHOLE_ONLY && LEGACY == 0) {
        {  // (the 4-wave kernel is dispatched for ldc % 8 == 0 only: launch_w4)
          // 16-byte form (pair16 helpers above): the residual pieces are requested DW rows ahead of their use (TN / 2 loads
          // of 16 bytes per row: half the instructions of the 8-byte form for the same lines), the stores are 16 bytes as well
          constexpr int DW = TM > 5 ? 5 : TM;
          const int off16 = pair16_off(lane);
          u32x4 oldw[TM][TN / 2];
          auto fetchw = [&](int j) {
            const bf16_t* hrow = p.res16 + (long)(m_base + j * 16 + ml) * p.ldc + n_base + off16;
#pragma unroll
            for (int k = 0; k < TN / 2; ++k)
              oldw[j][k] = res ? *reinterpret_cast<const u32x4*>(hrow + k * 32) : u32x4{0u, 0u, 0u, 0u};
          };
#pragma unroll
          for (int j = 0; j < DW; ++j) fetchw(j);
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            if (j + DW < TM) fetchw(j + DW < TM ? j + DW : 0);
            const long m = m_base + j * 16 + ml;
            bf16_t* hrow = p.norm_h16 + m * p.ldc + n_base;
#pragma unroll
            for (int i = 0; i < TN; ++i) pin_acc(acc, i, j);  // (see the SiLU epilogue: no hoisted accumulator reads)
#pragma unroll
            for (int g = 0; g < TN / 4; ++g) {
              float ss = 0.f;
#pragma unroll
              for (int k = g * 2; k < g * 2 + 2; ++k) {
                u32x2 o[2], w[2];
                unswap_pair16(oldw[j][k], o[0], o[1]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const f32x4 v = fma4(acc[2 * k + h][j], p.norm_scale,
                                       f32x4{from16_lo<F16>(o[h][0]), from16_hi<F16>(o[h][0]), from16_lo<F16>(o[h][1]), from16_hi<F16>(o[h][1])});
                  w[h] = u32x2{pack16x2<F16>(v[0], v[1]), pack16x2<F16>(v[2], v[3])};
                  const float r0 = from16_lo<F16>(w[h][0]), r1 = from16_hi<F16>(w[h][0]), r2 = from16_lo<F16>(w[h][1]),
                              r3 = from16_hi<F16>(w[h][1]);
                  ss += r0 * r0;
                  ss += r1 * r1;
                  ss += r2 * r2;
                  ss += r3 * r3;
                }
                store_pair16(hrow + k * 32, off16, w[0], w[1]);
              }
              ss += __shfl_xor(ss, 16, 64);
              ss += __shfl_xor(ss, 32, 64);
              if (lane < 16) {
                p.norm_part[m * npart + ((n_base >> 6) + g)] = ss;
                flag_nonfinite(p, ss);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          return;
        }
      }
      constexpr int D = WHOLE_ONLY ? (TM > 4 ? 4 : TM) : 1;
      u32x2 old[TM][TN];
      auto fetch = [&](int j) {
        const long m = m_base + j * 16 + ml;
        const long mm = (WHOLE_ONLY || m < p.M) ? m : 0;
        const bf16_t* hrow = p.res16 + mm * p.ldc + n_base + nq;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const bool colok = WHOLE_ONLY || n_base + (i >> 2) * 64 < p.N;
          old[j][i] = (res && colok) ? *reinterpret_cast<const u32x2*>(hrow + i * 16) : u32x2{0u, 0u};
        }
      };
#pragma unroll
      for (int j = 0; j < D; ++j) fetch(j);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        if (j + D < TM) fetch(j + D < TM ? j + D : 0);
        const long m = m_base + j * 16 + ml;
        const bool rowok = WHOLE_ONLY || m < p.M;
        const long mm = rowok ? m : 0;  // (rows beyond M read row 0 and store nothing: the shuffles below need every lane)
        bf16_t* hrow = p.norm_h16 + mm * p.ldc + n_base + nq;
#pragma unroll
        for (int g = 0; g < TN / 4; ++g) {
          const bool colok = WHOLE_ONLY || n_base + g * 64 < p.N;  // (N % 64 == 0: a group is inside or outside)
          float ss = 0.f;
#pragma unroll
          for (int i = g * 4; i < g * 4 + 4; ++i) {
            const u32x2 o = old[j][i];
            const f32x4 v = fma4(acc[i][j], p.norm_scale,
                                 f32x4{from16_lo<F16>(o[0]), from16_hi<F16>(o[0]), from16_lo<F16>(o[1]), from16_hi<F16>(o[1])});
            const u32x2 w = u32x2{pack16x2<F16>(v[0], v[1]), pack16x2<F16>(v[2], v[3])};
            if (rowok && colok) *reinterpret_cast<u32x2*>(hrow + i * 16) = w;
            const float r0 = from16_lo<F16>(w[0]), r1 = from16_hi<F16>(w[0]), r2 = from16_lo<F16>(w[1]), r3 = from16_hi<F16>(w[1]);
            ss += r0 * r0;
            ss += r1 * r1;
            ss += r2 * r2;
            ss += r3 * r3;
          }
          ss += __shfl_xor(ss, 16, 64);
          ss += __shfl_xor(ss, 32, 64);
          if (lane < 16 && rowok && colok) {
            p.norm_part[m * npart + ((n_base >> 6) + g)] = ss;
            flag_nonfinite(p, ss);
          }
        }
      }
      return;
    } else {
      constexpr int D = WHOLE_ONLY ? 2 : 1;
      f32x4 rv[TM][TN];
      auto fetch = [&](int j) {
        const long m = m_base + j * 16 + ml;
        const long mm = (WHOLE_ONLY || m < p.M) ? m : 0;
        const float* rrow = p.residual + mm * p.ldr + n_base + nq;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const bool colok = WHOLE_ONLY || n_base + (i >> 2) * 64 < p.N;
          rv[j][i] = (res && colok) ? *reinterpret_cast<const f32x4*>(rrow + i * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      };
#pragma unroll
      for (int j = 0; j < D && j < TM; ++j) fetch(j);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        if (j + D < TM) fetch(j + D < TM ? j + D : 0);
        const long m = m_base + j * 16 + ml;
        const bool rowok = WHOLE_ONLY || m < p.M;
        const long mm = rowok ? m : 0;  // (rows beyond M read row 0 and store nothing: the shuffles below need every lane)
        float* crow = reinterpret_cast<float*>(p.C) + mm * p.ldc + n_base + nq;
        bf16_t* hrow = p.norm_h16 + mm * p.ldc + n_base + nq;
        unsigned ovf = 0u;
#pragma unroll
        for (int g = 0; g < TN / 4; ++g) {
          const bool colok = WHOLE_ONLY || n_base + g * 64 < p.N;  // (N % 64 == 0: a group is inside or outside)
          float ss = 0.f;
#pragma unroll
          for (int i = g * 4; i < g * 4 + 4; ++i) {
            const f32x4 vt = acc[i][j] + rv[j][i];
            const f32x4 v = vt * p.norm_scale;  // (the 16-bit copy and its sums of squares are kept at norm_scale; 1: unchanged)
            const u32x2 w = u32x2{pack16x2<F16>(v[0], v[1]), pack16x2<F16>(v[2], v[3])};
            if (rowok && colok) {
              *reinterpret_cast<f32x4*>(crow + i * 16) = vt;
              *reinterpret_cast<u32x2*>(hrow + i * 16) = w;
            }
            if constexpr (F16) ovf |= half_is_inf2(w[0]) | half_is_inf2(w[1]);  // the fp32 value may be fine, its fp16 copy not
            ss += v[0] * v[0];
            ss += v[1] * v[1];
            ss += v[2] * v[2];
            ss += v[3] * v[3];
          }
          ss += __shfl_xor(ss, 16, 64);
          ss += __shfl_xor(ss, 32, 64);
          if (lane < 16 && rowok && colok) {
            p.norm_part[m * npart + ((n_base >> 6) + g)] = ss;
            flag_nonfinite(p, ss);
          }
        }
        if (F16 && ovf && rowok && p.nf_flag) atomicCAS(p.nf_flag, 0, p.nf_tag);
      }
      return;
    }
  }
  if constexpr (EPI == EPI_GENERIC) {
    if (whole && p.acc_scale == 1.f && p.out_kind == TCAVT_F32 && p.flags == TCAVT_EPI_RESIDUAL) {
      // (C and residual may be one buffer: load-add-store per quad would serialise on the memory latency -- a row's
      // residual quads are loaded together, one row ahead of their use)
      f32x4 rv[TM][TN];
      auto fetch = [&](int j) {
        const float* rrow = p.residual + (long)(m_base + j * 16 + ml) * p.ldr + n_base + nq;
#pragma unroll
        for (int i = 0; i < TN; ++i) rv[j][i] = *reinterpret_cast<const f32x4*>(rrow + i * 16);
      };
      fetch(0);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        if (j + 1 < TM) fetch(j + 1 < TM ? j + 1 : 0);
        const long m = m_base + j * 16 + ml;
        float* crow = reinterpret_cast<float*>(p.C) + m * p.ldc + n_base + nq;
#pragma unroll
        for (int i = 0; i < TN; ++i) *reinterpret_cast<f32x4*>(crow + i * 16) = acc[i][j] + rv[j][i];
      }
      return;
    }
    if (whole && p.acc_scale == 1.f && p.out_kind == OUT16 && p.flags == 0) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const long m = m_base + j * 16 + ml;
        bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n_base + nq;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const f32x4 v = acc[i][j];
          *reinterpret_cast<u32x2*>(crow + i * 16) = u32x2{pack16x2<F16>(v[0], v[1]), pack16x2<F16>(v[2], v[3])};
        }
      }
      return;
    }
  }
  if constexpr (EPI == EPI_SILUBWD) {
    // d(silu(gate) * up) straight from the accumulator of down_proj's dgrad GEMM (include/tcavt.h: TCAVT_EPI_SILU_BWD): for
    // the lane's four features of a 16-column tile, gate and up sit in two column-adjacent 16-column blocks of the
    // interleaved pre-activation row, and so do dgate and dup in the output row -- one 16-byte load and one 16-byte store
    // per tile (pair16 helpers above).  Same arithmetic as silu_mul_bwd_kernel, on the un-rounded d.
    static_assert(WHOLE_ONLY, "the SiLU-backward epilogue exists in the 4-wave kernel only");
    constexpr int DW = 2;  // rows of pre-activations requested ahead (TN x 16 bytes per lane and row)
    const int off16 = pair16_off(lane);
    u32x4 pre[TM][TN];
    auto fetchp = [&](int j) {
      const bf16_t* arow = p.aux + (long)(m_base + j * 16 + ml) * p.ldaux + 2 * n_base + off16;
#pragma unroll
      for (int i = 0; i < TN; ++i) pre[j][i] = *reinterpret_cast<const u32x4*>(arow + i * 32);
    };
#pragma unroll
    for (int j = 0; j < DW && j < TM; ++j) fetchp(j);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      if (j + DW < TM) fetchp(j + DW < TM ? j + DW : 0);
      bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + (long)(m_base + j * 16 + ml) * p.ldc + 2 * n_base;
#pragma unroll
      for (int i = 0; i < TN; ++i) pin_acc(acc, i, j);  // (no hoisted accumulator reads: see the SiLU epilogue)
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        u32x2 gq, uq;
        unswap_pair16(pre[j][i], gq, uq);
        const f32x4 d = acc[i][j];
        float dg[4], du[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = (e & 1) ? from16_hi<F16>(gq[e >> 1]) : from16_lo<F16>(gq[e >> 1]);
          const float u = (e & 1) ? from16_hi<F16>(uq[e >> 1]) : from16_lo<F16>(uq[e >> 1]);
          const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-g));
          dg[e] = d[e] * u * sg * (1.f + g * (1.f - sg));
          du[e] = d[e] * g * sg;
        }
        store_pair16(crow + i * 32, off16, u32x2{pack16x2<F16>(dg[0], dg[1]), pack16x2<F16>(dg[2], dg[3])},
                     u32x2{pack16x2<F16>(du[0], du[1]), pack16x2<F16>(du[2], du[3])});
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  if constexpr ((EPI == EPI_SILU || EPI == EPI_SILU_SAVE) && TN % 4 == 0) {
    // (the 4-wave kernel is dispatched for this form only -- silu16_ok() on the host -- so that its general path, and
    // the registers it costs around the persistent loop, compile away)
    if ((WHOLE_ONLY && LEGACY == 0) || (whole && p.out_kind == OUT16 && (p.ldc & 7) == 0 && LEGACY == 0)) {
      // two gate|up tile pairs -> two adjacent 16-column output tiles -> one 16-byte store per lane
      const int off16 = pair16_off(lane);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const long m = m_base + j * 16 + ml;
        bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + (n_base >> 1);
        const float rs = rsv[j];  // fused RMSNorm: 1 / rms of the row (gamma is in W)
        if constexpr (WHOLE_ONLY) {
          // (4-wave kernel: the accumulators live in AGPRs; re-pinning this row's here keeps their v_accvgpr_reads from
          // being hoisted over the rows before it -- 150 hoisted reads cost spills that were reloaded behind the stores)
#pragma unroll
          for (int i = 0; i < TN; ++i) pin_acc(acc, i, j);
        }
#pragma unroll
        for (int i = 0; i < TN; i += 4) {
          u32x2 o[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 g = acc[i + 2 * h][j] * rs, u = acc[i + 2 * h + 1][j] * rs;
            if constexpr (EPI == EPI_SILU_SAVE) {
              bf16_t* arow = p.aux + m * p.ldaux + n_base + (i + 2 * h) * 16 + nq;
              *reinterpret_cast<u32x2*>(arow) = u32x2{pack16x2<F16>(g[0], g[1]), pack16x2<F16>(g[2], g[3])};
              *reinterpret_cast<u32x2*>(arow + 16) = u32x2{pack16x2<F16>(u[0], u[1]), pack16x2<F16>(u[2], u[3])};
            }
            o[h] = silu_mul_quad<F16>(g, u);
          }
          store_pair16(crow + (i >> 1) * 16, off16, o[0], o[1]);
        }
        // (one row of MFMA tiles at a time: left free, the scheduler hoists the accumulator reads of later rows over this
        // one's arithmetic, runs out of registers and spills -- and a scratch reload waits for every store issued so far)
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
  }
  if constexpr (EPI == EPI_SILU || EPI == EPI_SILU_SAVE) {
    if (whole && p.out_kind == OUT16) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const long m = m_base + j * 16 + ml;
        bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + (n_base >> 1) + nq;
        const float rs = rsv[j];  // fused RMSNorm: 1 / rms of the row (gamma is in W)
#pragma unroll
        for (int i = 0; i < TN; i += 2) {
          const f32x4 g = acc[i][j] * rs, u = acc[i + 1][j] * rs;
          if constexpr (EPI == EPI_SILU_SAVE) {
            bf16_t* arow = p.aux + m * p.ldaux + n_base + i * 16 + nq;
            *reinterpret_cast<u32x2*>(arow) = u32x2{pack16x2<F16>(g[0], g[1]), pack16x2<F16>(g[2], g[3])};
            *reinterpret_cast<u32x2*>(arow + 16) = u32x2{pack16x2<F16>(u[0], u[1]), pack16x2<F16>(u[2], u[3])};
          }
          *reinterpret_cast<u32x2*>(crow + (i >> 1) * 16) =
              u32x2{pack16x2<F16>(silu_mul(g[0], u[0]), silu_mul(g[1], u[1])),
                    pack16x2<F16>(silu_mul(g[2], u[2]), silu_mul(g[3], u[3]))};
        }
      }
      return;
    }
  }
  if constexpr (EPI == EPI_ROPE && WHOLE_ONLY) {
    if constexpr (LEGACY == 0) {
      // 4-wave kernel (dispatched for 16-bit outputs with ldc % 8 == 0): 16-byte stores (pair16 helpers above); the cos / sin
      // rows of one row of MFMA tiles are loaded one row ahead
      const int off16 = pair16_off(lane);
      f32x4 cs[2][2], sn[2][2];
      auto fetch = [&](int j, int slot) {
        const int m = m_base + j * 16 + ml;
        const int pos = p.rope_pos ? p.rope_pos[m] : m % p.rope_L;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          cs[slot][i] = *reinterpret_cast<const f32x4*>(p.cosT + pos * 32 + nq + i * 16);
          sn[slot][i] = *reinterpret_cast<const f32x4*>(p.sinT + pos * 32 + nq + i * 16);
        }
      };
      fetch(0, 0);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        if (j + 1 < TM) fetch(j + 1 < TM ? j + 1 : 0, (j + 1) & 1);
        const int m = m_base + j * 16 + ml;
        bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n_base;
        const float rs = rsv[j];  // fused RMSNorm: 1 / rms of the row (gamma is in W)
#pragma unroll
        for (int i = 0; i < TN; ++i) pin_acc(acc, i, j);  // (see the SiLU epilogue: no hoisted accumulator reads)
#pragma unroll
        for (int hh = 0; hh < TN / 4; ++hh) {
          const bool rot = n_base + hh * 64 < p.rope_cols;  // uniform: q and k heads rotate, v heads do not
          u32x2 o[4];
          if (rot) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const f32x4 lo = acc[hh * 4 + i][j] * rs, hi = acc[hh * 4 + i + 2][j] * rs;
              const f32x4 c = cs[j & 1][i];
              const f32x4 s = sn[j & 1][i];
              const f32x4 l2 = lo * c - hi * s;
              const f32x4 h2 = hi * c + lo * s;
              o[i] = u32x2{pack16x2<F16>(l2[0], l2[1]), pack16x2<F16>(l2[2], l2[3])};
              o[i + 2] = u32x2{pack16x2<F16>(h2[0], h2[1]), pack16x2<F16>(h2[2], h2[3])};
            }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const f32x4 v = acc[hh * 4 + i][j] * rs;
              o[i] = u32x2{pack16x2<F16>(v[0], v[1]), pack16x2<F16>(v[2], v[3])};
            }
          }
          store_pair16(crow + hh * 64, off16, o[0], o[1]);
          store_pair16(crow + hh * 64 + 32, off16, o[2], o[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
  }
  if constexpr (EPI == EPI_ROPE) {
    if (whole && p.out_kind == OUT16) {
      // cos / sin rows are loaded D rows ahead of their use (all of them in the 4-wave kernel): issued between the stores
      // of the output, which the compiler must assume they alias, every load cost a full L2 latency (32 of them per tile)
      constexpr int D = WHOLE_ONLY ? TM : 1;
      f32x4 cs[TM][2], sn[TM][2];
      int pos[TM];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int m = m_base + j * 16 + ml;
        pos[j] = p.rope_pos ? p.rope_pos[m] : m % p.rope_L;
      }
      auto fetch = [&](int j) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          cs[j][i] = *reinterpret_cast<const f32x4*>(p.cosT + pos[j] * 32 + nq + i * 16);
          sn[j][i] = *reinterpret_cast<const f32x4*>(p.sinT + pos[j] * 32 + nq + i * 16);
        }
      };
#pragma unroll
      for (int j = 0; j < D; ++j) fetch(j);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        if (j + D < TM) fetch(j + D < TM ? j + D : 0);
        const int m = m_base + j * 16 + ml;
        bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n_base + nq;
        const float rs = rsv[j];  // fused RMSNorm: 1 / rms of the row (gamma is in W)
#pragma unroll
        for (int hh = 0; hh < TN / 4; ++hh) {
          const bool rot = n_base + hh * 64 < p.rope_cols;  // uniform: q and k heads rotate, v heads do not
          // (two separate bodies: merging rotated temporaries with the un-rotated accumulators in one variable made the
          // compiler shuttle ~1000 values through v_accvgpr_write / _mov per tile)
          if (rot) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const f32x4 lo = acc[hh * 4 + i][j] * rs, hi = acc[hh * 4 + i + 2][j] * rs;
              const f32x4 c = cs[j][i];
              const f32x4 s = sn[j][i];
              const f32x4 l2 = lo * c - hi * s;
              const f32x4 h2 = hi * c + lo * s;
              *reinterpret_cast<u32x2*>(crow + hh * 64 + i * 16) = u32x2{pack16x2<F16>(l2[0], l2[1]), pack16x2<F16>(l2[2], l2[3])};
              *reinterpret_cast<u32x2*>(crow + hh * 64 + 32 + i * 16) = u32x2{pack16x2<F16>(h2[0], h2[1]), pack16x2<F16>(h2[2], h2[3])};
            }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const f32x4 v = acc[hh * 4 + i][j] * rs;
              *reinterpret_cast<u32x2*>(crow + hh * 64 + i * 16) = u32x2{pack16x2<F16>(v[0], v[1]), pack16x2<F16>(v[2], v[3])};
            }
          }
        }
      }
      return;
    }
  }
  if constexpr (WHOLE_ONLY && EPI == EPI_ROPE) {
    return;  // the 4-wave kernel is dispatched for bf16 output only on this epilogue (launch_w4 checks)
  }
  if constexpr (EPI == EPI_NORM || EPI == EPI_NORM16) {
    return;  // (TN % 4 != 0: the 64x64 form, never dispatched for this epilogue)
  }
  if constexpr (EPI == EPI_GENERIC || EPI == EPI_DROP) {
    // Loads first, stores after: bias / residual loads written between the stores of C (which they may alias as far as
    // the compiler knows) each waited for a full memory latency, TM x TN times per wave.  The column biases are loaded
    // once, the row biases for all rows, the residual quads one row ahead of their use.
    const bool has_bias = p.flags & TCAVT_EPI_BIAS, has_brow = p.flags & TCAVT_EPI_BIAS_ROW, has_res = p.flags & TCAVT_EPI_RESIDUAL;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bq[TN];
    float bm[TM];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n_base + i * 16 + nq;
      bq[i] = (has_bias && n < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + n) : zero4;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m_base + j * 16 + ml;
      bm[j] = (has_brow && m < p.M) ? p.bias[m] : 0.f;
    }
    f32x4 rv[TM][TN];
    auto fetch = [&](int j) {
      const int m = m_base + j * 16 + ml;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int n = n_base + i * 16 + nq;
        rv[j][i] = (has_res && m < p.M && n < p.N) ? *reinterpret_cast<const f32x4*>(p.residual + (long)m * p.ldr + n) : zero4;
      }
    };
    fetch(0);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      if (j + 1 < TM) fetch(j + 1 < TM ? j + 1 : 0);
      const int m = m_base + j * 16 + ml;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int n = n_base + i * 16 + nq;
        if (n >= p.N) continue;
        f32x4 v = acc[i][j] * p.acc_scale;
        if (has_bias) v += bq[i];
        if (has_brow) {
          v[0] += bm[j]; v[1] += bm[j]; v[2] += bm[j]; v[3] += bm[j];
        }
        if (p.flags & TCAVT_EPI_RELU) {
          v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]);
          v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]);
        }
        if constexpr (EPI == EPI_DROP) {
          float sc[4];
          dropout_quad(p.drop, ((unsigned long long)m * (unsigned long long)p.N + (unsigned long long)n) >> 2, sc);
          v[0] *= sc[0]; v[1] *= sc[1]; v[2] *= sc[2]; v[3] *= sc[3];
        }
        if (has_res) v += rv[j][i];
        store_quad(p, m, n, v);
      }
    }
  } else if constexpr (EPI == EPI_SILU || EPI == EPI_SILU_SAVE) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m_base + j * 16 + ml;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < TN; i += 2) {
        if (n_base + i * 16 >= p.N) continue;  // partial last tile column
        const int n = ((n_base) >> 1) + (i >> 1) * 16 + nq;
        const float rs = rsv[j];
        const f32x4 g = acc[i][j] * rs, u = acc[i + 1][j] * rs;
        if constexpr (EPI == EPI_SILU_SAVE) {
          bf16_t* arow = p.aux + (long)m * p.ldaux + n_base + i * 16 + nq;
          *reinterpret_cast<u32x2*>(arow) = u32x2{pack16x2<F16>(g[0], g[1]), pack16x2<F16>(g[2], g[3])};
          *reinterpret_cast<u32x2*>(arow + 16) = u32x2{pack16x2<F16>(u[0], u[1]), pack16x2<F16>(u[2], u[3])};
        }
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_mul(g[e], u[e]);
        store_quad(p, m, n, v);
      }
    }
  } else {  // EPI_ROPE
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m_base + j * 16 + ml;
      if (m >= p.M) continue;
      const int pos = p.rope_pos ? p.rope_pos[m] : m % p.rope_L;
#pragma unroll
      for (int hh = 0; hh < TN / 4; ++hh) {
        const int nb = n_base + hh * 64;
        if (nb >= p.N) continue;  // partial last tile column (N % BN != 0)
        const bool rot = nb < p.rope_cols;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int d = i * 16 + nq;
          const float rs = rsv[j];
          f32x4 lo = acc[hh * 4 + i][j] * rs, hi = acc[hh * 4 + i + 2][j] * rs;
          if (rot) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(p.cosT + pos * 32 + d);
            const f32x4 s = *reinterpret_cast<const f32x4*>(p.sinT + pos * 32 + d);
            const f32x4 l2 = lo * c - hi * s;
            const f32x4 h2 = hi * c + lo * s;
            lo = l2; hi = h2;
          }
          store_quad(p, m, nb + d, lo);
          store_quad(p, m, nb + 32 + d, hi);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Workgroup -> output tile.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the
// XCD, each with a private 4 MiB L2), so the grid is cut into gx x gy rectangles of tiles, one per XCD
// (gx * gy = 8): an XCD then streams 1/gx of the activation rows and 1/gy of the weight rows, and the
// fabric / Infinity-Cache traffic of the launch is  gy * |A| + gx * |W|.  The host picks (gx, gy) that
// minimises it (p.xcd_gx; 8 = row bands, the right choice whenever |A| >= |W|).  Inside its rectangle an
// XCD walks 4-tile-tall super rows so that its 32 CUs work on a 4 x 8 patch at any time.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void block_to_tile(const GemmP& p, int& tile_m, int& tile_n, int bid, int nwg) {
  constexpr int GM = 4;
  const int gx = p.xcd_gx;
  if (gx != 8) {  // 2-D partition; the host guarantees tiles_m % gx == 0 and tiles_n % (8 / gx) == 0
    const int gy = 8 / gx;
    const int xcd = bid & 7, local = bid >> 3;
    const int xi = xcd / gy, xj = xcd - xi * gy;
    const int sm = p.tiles_m / gx, sn = p.tiles_n / gy;
    const int per_group = GM * sn;
    const int g = local / per_group, in_g = local - g * per_group;
    const int gsz = min(GM, sm - g * GM);
    tile_m = xi * sm + g * GM + in_g % gsz;
    tile_n = xj * sn + in_g / gsz;
    return;
  }
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int per_group = GM * p.tiles_n;
  const int g = wgid / per_group, in_g = wgid - g * per_group;
  const int gsz = min(GM, p.tiles_m - g * GM);
  tile_m = g * GM + in_g % gsz;
  tile_n = in_g / gsz;
}

__device__ __forceinline__ void block_to_tile(const GemmP& p, int& tile_m, int& tile_n) {
  block_to_tile(p, tile_m, tile_n, blockIdx.x, gridDim.x);
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int EPI, bool F16, int PIPE>
__global__ __launch_bounds__(WARPS_M* WARPS_N * 64) void gemm_bf16_kernel(GemmP p) {
  constexpr int NW = WARPS_M * WARPS_N;
  constexpr int ROWS = BM + BN;
  constexpr int TILE_BYTES = ROWS * 128;
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int ROUNDS = ROWS / (8 * NW);
  static_assert(ROWS % (8 * NW) == 0, "staging rounds must be whole");
  static_assert(BM % 16 == 0 && BN % 16 == 0, "tile rows");
  static_assert(EPI != EPI_ROPE || WTN % 64 == 0, "RoPE needs whole heads per wave");
  static_assert((EPI != EPI_SILU && EPI != EPI_SILU_SAVE) || WTN % 32 == 0, "SiLU needs gate/up pairs per wave");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;

  // ---- block -> tile (XCD-aware; any bijection is correct, this one is for L2 / fabric traffic)
  int tile_m, tile_n;
  block_to_tile(p, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (gridDim.y > 1) {  // batched form: product blockIdx.y
    const int bo = blockIdx.y / p.batch_inner, bi = blockIdx.y - bo * p.batch_inner;
    p.A += bo * p.sAo + bi * p.sAi;
    p.W += bo * p.sWo + (bi / p.w_group) * p.sWi;
    const long co = bo * p.sCo + bi * p.sCi;
    p.C = p.out_kind == TCAVT_F32 ? static_cast<void*>(reinterpret_cast<float*>(p.C) + co)
                                  : static_cast<void*>(reinterpret_cast<bf16_t*>(p.C) + co);
  }

  // ---- per-lane staging sources (main K source), one per round
  const bf16_t* src[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int g = r * NW + wave;
    const int row = g * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    if (g * 8 < BM) {
      const int m = min(m0 + row, p.M - 1);
      src[r] = p.A + (long)m * p.lda + c * 8;
    } else {
      const int n = min(n0 + row - BM, p.N - 1);
      src[r] = p.W + (long)n * p.ldw + c * 8;
    }
  }
  const int nt1 = p.K >> 6, nt = nt1 + (p.K2 >> 6);

  auto stage = [&](int buf, int t) {
    char* base = smem + buf * TILE_BYTES;
    if (t < nt1) {
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) glds16(src[r] + t * 64, base + (r * NW + wave) * 1024);
    } else {
      const int k0 = (t - nt1) * 64;
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int g = r * NW + wave;
        const int row = g * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const bf16_t* s;
        if (g * 8 < BM) {
          const int m = min(m0 + row, p.M - 1);
          s = p.A2 + (long)m * p.lda2 + k0 + c * 8;
        } else {
          const int n = min(n0 + row - BM, p.N - 1);
          s = p.W2 + (long)n * p.ldw2 + k0 + c * 8;
        }
        glds16(s, base + g * 1024);
      }
    }
  };

  // ---- fragment read addressing
  const int fsw = (lane >> 1) & 7;  // == (row>>1)&7 for row = 16*j + (lane&15)
  const int off0 = (((lane >> 4)) ^ fsw) * 16;
  const int off1 = ((4 + (lane >> 4)) ^ fsw) * 16;
  const int xrow = (wm * WTM + (lane & 15)) * 128;
  const int wrow = (BM + wn * WTN + (lane & 15)) * 128;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char* base = smem + buf * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = ks ? off1 : off0;
      bf16x8 wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i)
        wf[i] = *reinterpret_cast<const bf16x8*>(base + wrow + i * 2048 + off);
#pragma unroll
      for (int j = 0; j < TM; ++j)
        xf[j] = *reinterpret_cast<const bf16x8*>(base + xrow + j * 2048 + off);
      if (p.prio == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          if constexpr (F16)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[i]),
                                                               __builtin_bit_cast(f16x8, xf[j]), acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      if (p.prio == 2) __builtin_amdgcn_s_setprio(0);
    }
  };

  // Variant (prio == 3): the DMA pieces of tile t+1 are issued one at a time BETWEEN the MFMAs of the
  // first k-step of tile t instead of in one burst ahead of them (each piece costs the issuing wave
  // ~60-180 cycles of issue time; spread out, the other wave of the SIMD keeps the matrix pipe busy).
  auto compute_interleaved = [&](int buf, int nbuf, int tn) {
    const char* base = smem + buf * TILE_BYTES;
    char* nbase = smem + nbuf * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = ks ? off1 : off0;
      bf16x8 wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(base + wrow + i * 2048 + off);
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(base + xrow + j * 2048 + off);
      if (p.prio == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          if constexpr (F16)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[i]),
                                                               __builtin_bit_cast(f16x8, xf[j]), acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
          constexpr int PER = (TN * TM) / ROUNDS;  // MFMAs per DMA piece
          const int idx = i * TM + j;
          if (ks == 0 && (idx % PER) == PER - 1) {
            const int r = idx / PER;
            glds16(src[r] + tn * 64, nbase + (r * NW + wave) * 1024);
          }
        }
      if (p.prio == 2) __builtin_amdgcn_s_setprio(0);
    }
  };

  if constexpr (PIPE == 2) {
    // ---- deep main loop for launches that cannot fill the chip (a few dozen workgroups, each walking its K
    // range alone): four LDS stages, up to three K-tiles of DMA in flight, so that a K-tile costs its issue
    // time instead of a full HBM / L2 round trip (the weights of these layers are HBM-cold inside the model).
    // Counted vmcnt (the wave's own pieces of the NEWER tiles stay in flight) and a raw barrier per K-tile:
    // the barrier publishes tile t and proves everyone is done with tile t-1, whose stage the next DMA reuses.
    constexpr int NS = 4;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < nt) stage(s, s);
    for (int t = 0; t < nt; ++t) {
      const int rem = nt - 1 - t;
      if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ROUNDS) : "memory");
      else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ROUNDS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      if (t + NS - 1 < nt) stage((t + NS - 1) % NS, t + NS - 1);
      compute(t % NS);
    }
    gemm_epilogue<TM, TN, EPI, false, F16>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lane);
    return;
  }
  if (p.prio == 1 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  // ---- main loop: stage t+1 while computing t; one drain+barrier per K-tile
  stage(0, 0);
  __syncthreads();
  int cur = 0;
  if constexpr (PIPE == 1) {
    int t = 0;
    for (; t + 1 < nt1; ++t) {
      compute_interleaved(cur, cur ^ 1, t + 1);
      __syncthreads();
      cur ^= 1;
    }
    for (; t < nt - 1; ++t) {  // second K-source (LoRA): burst staging
      stage(cur ^ 1, t + 1);
      compute(cur);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    for (int t = 0; t < nt - 1; ++t) {
      stage(cur ^ 1, t + 1);
      compute(cur);
      __syncthreads();
      cur ^= 1;
    }
  }
  compute(cur);

  gemm_epilogue<TM, TN, EPI, false, F16>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lane);
}

// (gx, gy) minimising gy * |A| + gx * |W| -- the bytes the eight XCDs pull over the fabric -- among the partitions the
// tile grid divides evenly into whole super rows; ties go to the larger gx (row bands).  Round 1 weighted W three times
// (HBM-cold weights vs activations served from the Infinity Cache); re-measured in the model in round 2 with the fused-norm
// epilogues (forward pass, one box, us per launch for gx = 1 / 2 / 4 / 8):
//     q|k|v 116.2 / 113.4 / 112.5 / 118.1    o 104.4 / 97.2 / 96.1 / 89.5    gate|up 448.7 / 441.0 / 439.9 / 440.0
//     down 249.4 / 245.9 / 245.6 / 243.8
// which the unweighted byte count reproduces (q|k|v -> 4, o -> 8, gate|up -> 2, down -> 8).
// TCAVT_GEMM_XCD_GX=<1|2|4|8> forces a partition (A/B runs).
static int choose_xcd_partition(const GemmP& p) {
  static const int forced = [] {
    const char* e = getenv("TCAVT_GEMM_XCD_GX");
    return e ? atoi(e) : 0;
  }();
  const double a_bytes = (double)p.M * p.K, w_bytes = (double)p.N * p.K;
  int best = 8;
  double best_cost = 1.0 * a_bytes + 8.0 * w_bytes;
  for (int gx = 4; gx >= 1; gx /= 2) {
    const int gy = 8 / gx;
    if (p.tiles_m % gx || p.tiles_n % gy || (p.tiles_m / gx) % 4 || (long)p.tiles_m * p.tiles_n % 8) continue;
    if (forced == gx) return gx;
    const double cost = gy * a_bytes + gx * w_bytes;
    if (cost < best_cost) {
      best_cost = cost;
      best = gx;
    }
  }
  return forced == 8 ? 8 : best;
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int EPI, bool F16, int PIPE = 0>
static int launch(const GemmP& p0, int batch, hipStream_t stream) {
  GemmP p = p0;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.xcd_gx = choose_xcd_partition(p);
  constexpr int lds = (PIPE == 2 ? 4 : 2) * (BM + BN) * 128;
  auto kfn = gemm_bf16_kernel<BM, BN, WARPS_M, WARPS_N, EPI, F16, PIPE>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      set_error("gemm_bf16: hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e));
      return TCAVT_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, batch), block(WARPS_M * WARPS_N * 64);
  hipLaunchKernelGGL(kfn, grid, block, lds, stream, p);
  TCAVT_CHECK_LAUNCH("gemm_bf16");
  return TCAVT_OK;
}

#ifdef TCAVT_EXPERIMENTS  // measured-and-rejected form: only in the experiments build (tools/), never in the product library
// ===========================================================================
// Main-loop variant 2 ("ring", 256x256 tile only): the K dimension is cut into 32-deep slabs that
// live in a 4-slot LDS ring (4 x 32 KiB).  Up to three slabs of LDS-DMA stay in flight ACROSS the
// workgroup barriers: the loop never drains vmcnt to 0 in steady state (counted s_waitcnt vmcnt(8)),
// uses raw s_barrier (a __syncthreads() would drain the DMA queue), and has one barrier per slab:
//
//     wait(slab s landed for my own DMA) ; barrier ; issue DMA for slab s+3 ; ds_read + 32 MFMA on slab s
//
// The barrier both publishes slab s (every wave waited for its own pieces before arriving) and
// proves every wave has finished reading slab s-1, whose slot the new DMA overwrites.
// An LDS row is a tile row's 64 bytes of the slab; the four 16-byte chunks are XOR-swizzled with
// F[(row>>2)&3], F = {0,2,3,1}, which makes every ds_read_b128 lane group hit 16 distinct slots of
// the 256-byte bank row (swizzle on the DMA source address and on the fragment read, never on the
// DMA destination, which is lane-linear).
// ===========================================================================
template <int EPI, bool F16>
__global__ __launch_bounds__(512) void gemm_bf16_ring_kernel(GemmP p) {
  constexpr int BM = 256, BN = 256, WARPS_N = 4, NW = 8;
  constexpr int WTM = 128, WTN = 64, TM = 8, TN = 4;
  constexpr int SLAB = (BM + BN) * 64;  // bytes per ring slot
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  int tile_m, tile_n;
  block_to_tile(p, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (gridDim.y > 1) {
    const int bo = blockIdx.y / p.batch_inner, bi = blockIdx.y - bo * p.batch_inner;
    p.A += bo * p.sAo + bi * p.sAi;
    p.W += bo * p.sWo + (bi / p.w_group) * p.sWi;
    const long co = bo * p.sCo + bi * p.sCi;
    p.C = p.out_kind == TCAVT_F32 ? static_cast<void*>(reinterpret_cast<float*>(p.C) + co)
                                  : static_cast<void*>(reinterpret_cast<bf16_t*>(p.C) + co);
  }

  // ---- staging: 4 DMA pieces per thread per slab; piece r of wave w covers rows 16*(8r+w) .. +15
  const bf16_t* src[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int g = r * NW + wave;
    const int row = g * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
    if (g * 16 < BM) {
      const int m = min(m0 + row, p.M - 1);
      src[r] = p.A + (long)m * p.lda + c * 8;
    } else {
      const int n = min(n0 + row - BM, p.N - 1);
      src[r] = p.W + (long)n * p.ldw + c * 8;
    }
  }
  const int ns1 = p.K >> 5, ns = ns1 + (p.K2 >> 5);

  auto stage = [&](int slot, int s) {
    char* base = smem + slot * SLAB;
    if (s < ns1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) glds16(src[r] + s * 32, base + (r * NW + wave) * 1024);
    } else {
      const int k0 = (s - ns1) * 32;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int g = r * NW + wave;
        const int row = g * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
        const bf16_t* sp;
        if (g * 16 < BM) {
          const int m = min(m0 + row, p.M - 1);
          sp = p.A2 + (long)m * p.lda2 + k0 + c * 8;
        } else {
          const int n = min(n0 + row - BM, p.N - 1);
          sp = p.W2 + (long)n * p.ldw2 + k0 + c * 8;
        }
        glds16(sp, base + g * 1024);
      }
    }
  };

  // ---- fragment read addressing inside a slab
  const int r16 = lane & 15;
  const int foff = r16 * 64 + ((((lane >> 4)) ^ ((0x78 >> (((r16 >> 2) & 3) * 2)) & 3)) << 4);
  const int xoff = wm * WTM * 64 + foff;
  const int woff = (BM + wn * WTN) * 64 + foff;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: three slabs in flight
#pragma unroll
  for (int s = 0; s < 3; ++s)
    if (s < ns) stage(s, s);

  for (int s = 0; s < ns; ++s) {
    const int rem = ns - 1 - s;  // slabs issued after slab s (capped at 2)
    if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 3 < ns) stage((s + 3) & 3, s + 3);
    const char* base = smem + (s & 3) * SLAB;
    bf16x8 wf[TN], xf[TM];
#pragma unroll
    for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(base + woff + i * 1024);
#pragma unroll
    for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(base + xoff + j * 1024);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
        if constexpr (F16)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[i]),
                                                             __builtin_bit_cast(f16x8, xf[j]), acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
  }
  gemm_epilogue<TM, TN, EPI, false, F16>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lane);
}

#endif  // TCAVT_EXPERIMENTS

// ===========================================================================
// Main-loop variant 3 ("w4", 256x256 tile, whole tiles only): FOUR waves, one per SIMD, each owning a
// 128x128 quadrant (8x8 MFMA tiles = 256 accumulator registers, which the unified 512-entry file holds as
// AGPRs when a SIMD runs a single wave).  Compared with the 2x4-wave kernel above this reads a third less
// LDS per MFMA (16 fragment loads per 64 MFMAs instead of 12 per 32) and software-pipelines the fragment
// loads inside the wave: the fragments of the next 32-deep K-step are loaded into a second register set
// while the 64 MFMAs of the current one run, across the tile barrier as well -- the barrier sits 16 MFMAs
// before the end of a K-tile, and those 16 cover the first fragment loads of the next tile:
//
//   phase A : 64 MFMA on F0(t)  | ds_read F1(t)   | DMA pieces 4..15 of tile t+1 (one per 5 MFMAs)
//   phase B1: 48 MFMA on F1(t)
//   vmcnt(0) + barrier           (tile t+1 landed for everyone; everyone is done reading tile t)
//   phase B2: 16 MFMA on F1(t)  | ds_read F0(t+1) | DMA pieces 0..3 of tile t+2
//
// Same LDS image as the kernel above (128-byte rows, XOR-swizzled 16-byte chunks, two 64 KiB buffers).
// ===========================================================================
// MFMA with the accumulator pinned to AGPRs and tied in place.  Written as inline asm because the compiler's
// VGPR/AGPR rewriting turned the 256-register accumulator of the 4-wave kernel into ~350 v_accvgpr copies per K-tile.
template <bool F16>
__device__ __forceinline__ void mfma_agpr(f32x4& c, const bf16x8& a, const bf16x8& b) {
  if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// DBG (timing experiments only, results are wrong): 1 = no DMA in the loop, 2 = no barrier, 4 = no fragment loads
// BUF: stage with buffer_load ... lds (SGPR resource + one constant per-lane VGPR offset + scalar per-piece offset)
// instead of global_load ... lds (64-bit per-lane address, two VALU ops per piece).
// BN = 256: 2 x 2 waves of 128 x 128.  BN = 192: 4 x 1 waves of 64 x 192 (4 x 12 MFMA tiles, 192 accumulator
// registers) for N = 3072 (fused q|k|v): 512 tiles = two full waves of 256 CUs instead of 384 = one and a half.
// Instantiations that may run persistently (one workgroup per CU walking its tiles as one K-tile stream): not the buffer-load
// experiment; not RoPE (falls apart into 1.7 KB of scratch with it); not the generic epilogue in the two-barrier DEEP form
// (DBG & 64: 924 bytes of scratch with it -- the backward's long-K dgrad GEMMs are one tile per CU anyway).
constexpr bool w4_pers_ok(int epi, int dbg, bool buf) {
  return !buf && epi != EPI_ROPE && !((dbg & 64) && epi == EPI_GENERIC);
}

template <int EPI, int B2R, int DBG = 0, bool BUF = false, int BN = 256, bool F16 = false>
__global__ __launch_bounds__(256) void gemm_bf16_w4_kernel(GemmP p) {
  constexpr int BM = 256, NW = 4;
  constexpr int WN_ = BN == 256 ? 2 : 1, WM_ = NW / WN_;
  constexpr int TM = BM / WM_ / 16, TN = BN / WN_ / 16;
  static_assert(TM + TN == 16, "the fragment pipeline assumes 16 fragment loads per 32-deep K-step");
  constexpr int TILE_BYTES = (BM + BN) * 128;
  constexpr int NP = (BM + BN) / 32;  // DMA pieces (8 rows x 128 B per wave-instruction) per thread and K-tile
  constexpr int NB2 = B2R * TM;     // MFMAs after the barrier (phase B2)
  constexpr bool S1 = DBG & 16, S2 = DBG & 32;  // schedule variants (valid results)
  // DEEP (DBG & 64, valid results): a second barrier in the middle of phase A, where every wave holds all
  // fragments of tile t in registers, frees tile t's LDS buffer a whole K-tile earlier; the 16 pieces of tile t+2 are
  // issued behind it (one per 4 MFMAs over the rest of phase A and phase B1) and stay in flight ACROSS the end-of-B1
  // barrier, which waits with a counted vmcnt(NP) for the older tile t+1 only.  Every piece gets >= one full K-tile
  // (2048 MFMA cycles) to land instead of 0.4-1.2.
  constexpr bool DEEP = DBG & 64;
  static_assert(!DEEP || (TM == 8 && TN == 8), "DEEP is laid out for the 2x2-wave form");
  constexpr int EARLY = S2 ? 0 : NB2 / 4;    // pieces of tile t+2 issued in phase B2 of tile t
  constexpr int SPREAD = (TM * TN) / (NP - EARLY);  // phase A: one DMA piece per SPREAD MFMAs
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN_, wn = wave % WN_;
  // fused RMSNorm (TCAVT_EPI_ROWSCALE): the 256 row scales of the output tile, behind the two tile buffers
  constexpr bool RS = EPI == EPI_SILU || EPI == EPI_SILU_SAVE || EPI == EPI_ROPE;
  // (two sets, used alternately by consecutive output tiles of the persistent form: the next tile's scales are written
  // BEFORE this tile's epilogue, so that nothing has to be loaded, waited for or published behind the epilogue's stores)
  float* const rs_base = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);
  int rs_sel = 0;  // (uniform) the set the current output tile reads
  // Persistent form (p.pers_tiles > 0): this workgroup walks tiles vb = blockIdx.x, + gridDim.x, ... as ONE stream of
  // K-tiles -- the look-ahead of the pipeline (fragments of the next K-tile, DMA of the next two) simply continues into
  // the next output tile, so its first operands arrive while this tile's epilogue runs (no per-tile prologue).
  // (round 4: the two-barrier DEEP form walks its tiles as one K-tile stream as well -- it used to start every output tile with a
  //  burst prologue, which is what it lost to the one-barrier form at K = 2048, 8 tiles per CU)
  constexpr bool PERS_OK = w4_pers_ok(EPI, DBG, BUF);
  const bool pers = PERS_OK && p.pers_tiles > 0;
  const int total_tiles = pers ? p.pers_tiles : (int)gridDim.x;
  int vb = blockIdx.x;
  int tile_m, tile_n;
  block_to_tile(p, tile_m, tile_n, vb, total_tiles);
  int m0 = tile_m * BM, n0 = tile_n * BN;

  // DMA sources: piece r of a K-tile is the 8-row group g = 4 r + wave (r < 8: activation rows, r >= 8:
  // weight rows).  The swizzle term of a lane does not depend on r (32-row steps), so one base per operand.
  const int rl = lane >> 3;
  const int csw = (lane & 7) ^ (((wave & 1) * 4 + (rl >> 1)) & 7);
  // WT (DBG & 128, experiments build, TIMING ONLY): W addressed as if stored tile-major -- every 256-row x 64-column K-tile of a
  // column tile a contiguous 32 KiB, a DMA piece 1 KiB of consecutive bytes instead of 8 rows x 128 bytes ldw apart
  constexpr bool WT = (DBG & 128) != 0;
  constexpr int WKT = WT ? BN * 64 : 64;  // elements from one K-tile of W to the next
  const bf16_t* srcA = p.A + (long)(m0 + wave * 8 + rl) * p.lda + csw * 8;
  const bf16_t* srcW = WT ? p.W + (long)n0 * p.K + (wave * 8 + rl) * 64 + csw * 8 : p.W + (long)(n0 + wave * 8 + rl) * p.ldw + csw * 8;
  // next output tile of this workgroup (persistent form): where the look-ahead continues
  bool has_next = pers && vb + (int)gridDim.x < total_tiles;
  int nm0 = m0, nn0 = n0;
  const bf16_t* nxtA = srcA;
  const bf16_t* nxtW = srcW;
  auto locate_next = [&]() {
    if (has_next) {
      int tm, tn;
      block_to_tile(p, tm, tn, vb + (int)gridDim.x, total_tiles);
      nm0 = tm * BM;
      nn0 = tn * BN;
      nxtA = p.A + (long)(nm0 + wave * 8 + rl) * p.lda + csw * 8;
      nxtW = WT ? p.W + (long)nn0 * p.K + (wave * 8 + rl) * 64 + csw * 8 : p.W + (long)(nn0 + wave * 8 + rl) * p.ldw + csw * 8;
    }
  };
  locate_next();
  const long stepA = 32 * p.lda, stepW = WT ? 32 * 64 : 32 * p.ldw;
  // second K source (LoRA: A2 = x.A_cat^T, W2 = B_ext): its 64-deep tiles follow the main ones
  constexpr bool HASK2 = EPI == EPI_ROPE;  // only the fused q|k|v projection uses it
  const bf16_t* srcA2 = nullptr;
  const bf16_t* srcW2 = nullptr;
  long stepA2 = 0, stepW2 = 0;
  const int nt1 = p.K >> 6;
  int nt = nt1;
  auto locate_k2 = [&]() {
    if constexpr (HASK2) {
      if (p.K2 > 0) {
        srcA2 = p.A2 + (long)(m0 + wave * 8 + rl) * p.lda2 + csw * 8;
        srcW2 = p.W2 + (long)(n0 + wave * 8 + rl) * p.ldw2 + csw * 8;
      }
    }
  };
  if constexpr (HASK2) {
    if (p.K2 > 0) {
      stepA2 = 32 * p.lda2;
      stepW2 = 32 * p.ldw2;
      nt += p.K2 >> 6;
    }
  }
  locate_k2();

  struct Src {
    const bf16_t* a;
    const bf16_t* w;
    long sa, sw;
  };
  auto tsrc = [&](int t) -> Src {
    if constexpr (PERS_OK) {
      if (t >= nt && has_next) return Src{nxtA + (t - nt) * 64, nxtW + (t - nt) * WKT, stepA, stepW};  // next tile's first K-tiles
    }
    t = min(t, nt - 1);  // the last two K-tiles re-fetch the last tile into a free buffer (keeps the loop body uniform)
    if constexpr (BUF) return Src{nullptr, nullptr, (long)t * 128, 0};  // only the tile's byte offset along K
    if constexpr (HASK2) {
      if (t >= nt1) return Src{srcA2 + (t - nt1) * 64, srcW2 + (t - nt1) * 64, stepA2, stepW2};
    }
    return Src{srcA + t * 64, srcW + t * WKT, stepA, stepW};
  };
  // buffer form: resources based at the tile's first row, byte offsets in 32 bits (launch_w4 checks the range)
  __amdgpu_buffer_rsrc_t rsA, rsW;
  int voffA = 0, voffW = 0, stepAb = 0, stepWb = 0;
  if constexpr (BUF) {
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A + (long)m0 * p.lda), 0, 0x7ffffffe, 0x00020000);
    rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W + (long)n0 * p.ldw), 0, 0x7ffffffe, 0x00020000);
    voffA = ((wave * 8 + rl) * (int)p.lda + csw * 8) * 2;
    voffW = ((wave * 8 + rl) * (int)p.ldw + csw * 8) * 2;
    stepAb = 64 * (int)p.lda;  // 32 rows, bytes
    stepWb = 64 * (int)p.ldw;
  }
  auto piece = [&](int buf, const Src& s, int r) {
    char* dst = smem + buf * TILE_BYTES + (r * NW + wave) * 1024;
    if constexpr (BUF) {
      auto l = (__attribute__((address_space(3))) void*)dst;
      if (r < 8)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, l, 16, voffA, (int)s.sa + r * stepAb, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, l, 16, voffW, (int)s.sa + (r - 8) * stepWb, 0, 0);
    } else {
      if (r < 8)
        glds16(s.a + r * s.sa, dst);
      else
        glds16(s.w + (r - 8) * s.sw, dst);
    }
  };

  const int fsw = (lane >> 1) & 7;
  const int off0 = ((lane >> 4) ^ fsw) * 16;
  const int off1 = ((4 + (lane >> 4)) ^ fsw) * 16;
  const int xrow = (wm * TM * 16 + (lane & 15)) * 128;
  const int wrow = (BM + wn * TN * 16 + (lane & 15)) * 128;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 w0[TN], x0[TM], w1[TN], x1[TM];
  auto ldx = [&](const char* base, int off, int j) { return *reinterpret_cast<const bf16x8*>(base + xrow + j * 2048 + off); };
  auto ldw = [&](const char* base, int off, int i) { return *reinterpret_cast<const bf16x8*>(base + wrow + i * 2048 + off); };

  // ---- prologue: tile 0 (burst), publish, first pieces of tile 1, fragments F0(0)
  if constexpr (RS) {  // (the partial-sum loads are in flight together with tile 0's DMA; written before the barrier below)
    if (p.rs_part) rs_base[threadIdx.x] = row_rscale(p, m0 + threadIdx.x);
  }
  {
    const Src s0 = tsrc(0);
#pragma unroll
    for (int r = 0; r < NP; ++r) piece(0, s0, r);
  }
  __syncthreads();
  Src sn1 = tsrc(1);  // source of tile t+1, carried from iteration to iteration (sn1(t+1) = sn2(t))
  if constexpr (DEEP) {
#pragma unroll
    for (int r = 0; r < NP; ++r) piece(1, sn1, r);  // all of tile 1 (tsrc clamps when there is none: harmless re-fetch)
  } else if (nt > 1) {
#pragma unroll
    for (int r = 0; r < EARLY; ++r) piece(1, sn1, r);
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) x0[j] = ldx(smem, off0, j);
#pragma unroll
  for (int i = 0; i < TN; ++i) w0[i] = ldw(smem, off0, i);

  // One K-tile.  MORE / MORE2 (tile t+1 / t+2 exist) are compile-time so that the steady-state loop body is one
  // branch-free scheduling region; the last two tiles run peeled copies.
  int cur = 0;
  auto ktile = [&](auto more_c, auto more2_c, int t) {
    constexpr bool more = decltype(more_c)::value, more2 = decltype(more2_c)::value;
    constexpr bool dma = !(DBG & 1), bar = !(DBG & 2), frd = !(DBG & 4);
    const char* base = smem + cur * TILE_BYTES;
    const char* nbase = smem + (cur ^ 1) * TILE_BYTES;
    Src sn2;  // source of tile t+2, put together step by step in the shadow of the bare MFMAs of phase B1
    int koff2 = 0;
    bool second2 = false, into_next2 = false;
    // ---- phase A: MFMAs on F0 | load F1 (second 32-deep half of tile t) | rest of the DMA for tile t+1
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        mfma_agpr<F16>(acc[i][j], w0[i], x0[j]);
        const int idx = i * TM + j;
        if (frd && !S1 && idx < 32 && (idx & 1) == 0) {  // 16 fragment loads, one per 2 MFMAs
          const int f = idx >> 1;
          if (f < TM) x1[f] = ldx(base, off1, f);
          else w1[f - TM] = ldw(base, off1, f - TM);
        }
        if (frd && S1 && idx < 16) {  // S1: one per MFMA, the DMA pieces only after them
          if (idx < TM) x1[idx] = ldx(base, off1, idx);
          else w1[idx - TM] = ldw(base, off1, idx - TM);
        }
        if constexpr (DEEP) {
          if (idx == 0) {
            into_next2 = PERS_OK && has_next && t + 2 >= nt;   // the look-ahead crosses into the next output tile
            const int tt = into_next2 ? t + 2 - nt : min(t + 2, nt - 1);  // clamp: see tsrc
            second2 = HASK2 && !into_next2 && tt >= nt1;
            koff2 = (second2 ? tt - nt1 : tt) * 64;
          }
          if (idx == 3) sn2.a = (into_next2 ? nxtA : second2 ? srcA2 : srcA) + koff2;
          if (idx == 6) sn2.w = (into_next2 ? nxtW : second2 ? srcW2 : srcW) + (WT ? koff2 * (WKT / 64) : koff2);
          if (idx == 9) {
            sn2.sa = second2 ? stepA2 : stepA;
            sn2.sw = second2 ? stepW2 : stepW;
          }
          if (idx == 39 && bar) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // tile t is in registers everywhere
          if (dma && idx >= 43 && (idx & 3) == 3) piece(cur, sn2, (idx - 43) / 4);  // pieces 0..5 of tile t+2
        } else if (!S1) {
          if (dma && more && idx % SPREAD == SPREAD - 1 && EARLY + idx / SPREAD < NP) piece(cur ^ 1, sn1, EARLY + idx / SPREAD);
        } else {
          if (dma && more && idx >= 16 && (idx & 3) == 3 && EARLY + (idx - 16) / 4 < NP) piece(cur ^ 1, sn1, EARLY + (idx - 16) / 4);
        }
      }
    }
    // ---- phase B1: first 48 MFMAs on F1
#pragma unroll
    for (int i = 0; i < TN - B2R; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        mfma_agpr<F16>(acc[i][j], w1[i], x1[j]);
        const int idx = i * TM + j;
        if constexpr (DEEP) {
          if (dma && (idx & 3) == 3 && 6 + idx / 4 < NP) piece(cur, sn2, 6 + idx / 4);  // pieces 6..15 of tile t+2
          continue;
        }
        if (idx == 0) {
          into_next2 = PERS_OK && has_next && t + 2 >= nt;   // the look-ahead crosses into the next output tile
          const int tt = into_next2 ? t + 2 - nt : min(t + 2, nt - 1);  // clamp: see tsrc
          second2 = HASK2 && !into_next2 && tt >= nt1;
          koff2 = (second2 ? tt - nt1 : tt) * 64;
        }
        if constexpr (BUF) {
          if (idx == 3) sn2 = Src{nullptr, nullptr, (long)koff2 * 2, 0};
        } else {
          if (idx == 3) sn2.a = (into_next2 ? nxtA : second2 ? srcA2 : srcA) + koff2;
          if (idx == 6) sn2.w = (into_next2 ? nxtW : second2 ? srcW2 : srcW) + (WT ? koff2 * (WKT / 64) : koff2);
          if (idx == 9) {
            sn2.sa = second2 ? stepA2 : stepA;
            sn2.sw = second2 ? stepW2 : stepW;
          }
        }
      }
    if (DEEP && bar) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NP) : "memory");  // tile t+1 landed; t+2 in flight
    else if (more && bar && (DBG & 8)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // no DMA wait (timing only)
    else if (more && bar) __syncthreads();  // tile t+1 has landed for everyone; nobody reads tile t any more
    // ---- phase B2: last 16 MFMAs on F1 | load F0 of tile t+1 | first DMA pieces of tile t+2
#pragma unroll
    for (int i = TN - B2R; i < TN; ++i) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        mfma_agpr<F16>(acc[i][j], w1[i], x1[j]);
        const int idx = (i - (TN - B2R)) * TM + j;  // 0..NB2-1
        if (frd && more && !S2 && TM == TN && idx < TM) {  // the 16 fragment loads go first, two per MFMA
          x0[idx] = ldx(nbase, off0, idx);
          w0[idx] = ldw(nbase, off0, idx);
        }
        if (frd && more && (S2 || TM != TN) && idx < 16) {  // S2 / unequal fragment counts: one per MFMA
          if (idx < TM) x0[idx] = ldx(nbase, off0, idx);
          else w0[idx - TM] = ldw(nbase, off0, idx - TM);
        }
        if (!DEEP && dma && more2 && (idx & 3) == 3 && (idx >> 2) < EARLY) piece(cur, sn2, idx >> 2);
      }
    }
    // Last K-tile of an output tile: the accumulators are read next (the epilogue's v_accvgpr_read, but also AGPR-to-AGPR
    // copies the register allocator may place on the loop-exit edge, BEFORE any statement that follows the loop).  The
    // compiler cannot see the MFMA write latency behind the inline asm, so the wait states sit here, inside the loop
    // body, where nothing can be scheduled between them and the last MFMA (one scalar compare + branch per K-tile).
    if (t == nt - 1) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    cur ^= 1;
    sn1 = sn2;
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
#if TCAVT_W4_PEEL
  int t = 0;
  for (; t + 2 < nt; ++t) ktile(T_{}, T_{}, t);
  if (t + 1 < nt) { ktile(T_{}, F_{}, t); ++t; }
  ktile(F_{}, F_{}, t);
#else
  for (;;) {
    for (int t = 0; t < nt; ++t) ktile(T_{}, T_{}, t);
    // the accumulators are read by VALU next: cover the MFMA write latency the compiler cannot see behind the asm
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // ... and pin every accumulator read behind those nops: an empty volatile asm that redefines the register is
    // ordered after the s_nop asm, and the epilogue's v_accvgpr_read depends on it (without this the scheduler is
    // free to hoist the reads above the nops -- one instantiation did, and read stale values)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) asm volatile("" : "+a"(acc[i][j]));
    // ---- everything of the NEXT output tile that reads memory comes BEFORE this tile's stores: vmcnt counts loads and
    // stores in one queue, so a load waited for after the epilogue (the next tile's row scales; any register the allocator
    // chose to spill around the K loop) would first wait for the whole store tail to drain -- per output tile.
    const int em0 = m0, en0 = n0;
    const bool cont = PERS_OK && has_next;  // (uniform) non-persistent launches leave after the epilogue
    if (cont) {  // F0 already holds the next tile's first fragments, its second K-tile is in flight
      vb += gridDim.x;
      if constexpr (RS) {
        // the next tile's row scales, into the set this tile does not read (last read in the previous tile's epilogue:
        // K-tile barriers have passed since; published by the next tile's K-tile barriers)
        if (p.rs_part) rs_base[(rs_sel ^ 1) * 256 + threadIdx.x] = row_rscale(p, nm0 + threadIdx.x);
      }
      m0 = nm0;
      n0 = nn0;
      srcA = nxtA;
      srcW = nxtW;
      locate_k2();
      has_next = vb + (int)gridDim.x < total_tiles;
      locate_next();
    }
    __builtin_amdgcn_sched_barrier(0);
    gemm_epilogue<TM, TN, EPI, true, F16>(p, acc, em0 + wm * TM * 16, en0 + wn * TN * 16, lane,
                                          (RS && p.rs_part) ? rs_base + rs_sel * 256 + wm * TM * 16 : nullptr);
    if (!cont) {
      if constexpr (DEEP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA (the clamped re-fetches) may outlive the workgroup
      break;
    }
    rs_sel ^= 1;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+a"(acc[i][j]));  // the zeroing stays in front of ...
      }
    asm volatile("s_nop 7" ::: "memory");    // ... the wait states a VALU write needs before an MFMA reads it as SrcC
  }
#endif
}

#ifdef TCAVT_EXPERIMENTS
// ===========================================================================
// Experiment (round 4, tile code 273): the 4-wave kernel's main loop on v_mfma_f32_32x32x16 instead of 16x16x32.
// Why: the loop above is ISSUE-bound, not matrix-pipe-bound (round 1's elimination runs: no DMA -58 us, no fragment loads -38 us of
// 428; matrix pipe busy 0.68) -- one wave per SIMD must issue 128 MFMAs + 32 ds_read_b128 + 16 LDS-DMA pieces + address arithmetic
// per K-tile, and a 16x16x32 MFMA holds the issue port for 8 of its 16 cycles.  A 32x32x16 MFMA does the same FLOPs per cycle
// and holds the port for 8 of 32: 64 MFMAs per K-tile leave 3x the issue slack for the same loads.  Against it: the guide's
// measurement that the chip sustains a ~13 % lower clock on the 32x32 shape in a bare loop.  Same LDS image, same DMA, same
// fragment count (32 per K-tile); per wave a 128x128 quadrant = 4x4 tiles of 32x32 (16 accumulators of 16 registers).
// TIMING FIRST: the epilogue below is fed the accumulators in the 16x16 kernel's quad order, which is NOT where this MFMA leaves
// them (results are wrong); a real epilogue mapping is written only if the loop is faster.
// ===========================================================================
template <bool F16>
__device__ __forceinline__ void mfma32_agpr(f32x16& c, const bf16x8& a, const bf16x8& b) {
  if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int EPI, bool F16>
__global__ __launch_bounds__(256) void gemm_w4m32_kernel(GemmP p) {
  constexpr int BM = 256, BN = 256, NW = 4;
  constexpr int TILE_BYTES = (BM + BN) * 128;
  constexpr int NP = 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr bool RS = EPI == EPI_SILU;
  float* const rs_base = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);
  int rs_sel = 0;
  const bool pers = p.pers_tiles > 0;
  const int total_tiles = pers ? p.pers_tiles : (int)gridDim.x;
  int vb = blockIdx.x;
  int tile_m, tile_n;
  block_to_tile(p, tile_m, tile_n, vb, total_tiles);
  int m0 = tile_m * BM, n0 = tile_n * BN;
  const int rl = lane >> 3;
  const int csw = (lane & 7) ^ (((wave & 1) * 4 + (rl >> 1)) & 7);
  const bf16_t* srcA = p.A + (long)(m0 + wave * 8 + rl) * p.lda + csw * 8;
  const bf16_t* srcW = p.W + (long)(n0 + wave * 8 + rl) * p.ldw + csw * 8;
  bool has_next = pers && vb + (int)gridDim.x < total_tiles;
  int nm0 = m0, nn0 = n0;
  const bf16_t* nxtA = srcA;
  const bf16_t* nxtW = srcW;
  auto locate_next = [&]() {
    if (has_next) {
      int tm, tn;
      block_to_tile(p, tm, tn, vb + (int)gridDim.x, total_tiles);
      nm0 = tm * BM;
      nn0 = tn * BN;
      nxtA = p.A + (long)(nm0 + wave * 8 + rl) * p.lda + csw * 8;
      nxtW = p.W + (long)(nn0 + wave * 8 + rl) * p.ldw + csw * 8;
    }
  };
  locate_next();
  const long stepA = 32 * p.lda, stepW = 32 * p.ldw;
  const int nt = p.K >> 6;
  struct Src { const bf16_t* a; const bf16_t* w; };
  auto tsrc = [&](int t) -> Src {
    if (t >= nt && has_next) return Src{nxtA + (t - nt) * 64, nxtW + (t - nt) * 64};
    t = min(t, nt - 1);
    return Src{srcA + t * 64, srcW + t * 64};
  };
  auto piece = [&](int buf, const Src& s_, int r) {
    char* dst = smem + buf * TILE_BYTES + (r * NW + wave) * 1024;
    if (r < 8) glds16(s_.a + r * stepA, dst);
    else glds16(s_.w + (r - 8) * stepW, dst);
  };
  // fragment read: 32 rows per MFMA tile (row = lane & 31), 8 k-values per lane: 16-byte chunk (2 ks + (lane >> 5)) of the row
  const int fsw = ((lane & 31) >> 1) & 7;
  int off[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) off[ks] = ((2 * ks + (lane >> 5)) ^ fsw) * 16;
  const int xrow = (wm * 128 + (lane & 31)) * 128;
  const int wrow = (BM + wn * 128 + (lane & 31)) * 128;
  auto ldx = [&](const char* base, int ks, int j) { return *reinterpret_cast<const bf16x8*>(base + xrow + j * 4096 + off[ks]); };
  auto ldw = [&](const char* base, int ks, int i) { return *reinterpret_cast<const bf16x8*>(base + wrow + i * 4096 + off[ks]); };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bf16x8 w0[2][4], x0[2][4], w1[2][4], x1[2][4];  // F0 = k-steps 0, 1; F1 = k-steps 2, 3 of a K-tile

  if constexpr (RS) {
    if (p.rs_part) rs_base[threadIdx.x] = row_rscale(p, m0 + threadIdx.x);
  }
  {
    const Src s0 = tsrc(0);
#pragma unroll
    for (int r = 0; r < NP; ++r) piece(0, s0, r);
  }
  __syncthreads();
  Src sn1 = tsrc(1);
  if (nt > 1 || has_next) {
#pragma unroll
    for (int r = 0; r < 4; ++r) piece(1, sn1, r);
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      x0[ks][q] = ldx(smem, ks, q);
      w0[ks][q] = ldw(smem, ks, q);
    }
  int cur = 0;
  auto ktile = [&](int t) {
    const char* base = smem + cur * TILE_BYTES;
    const char* nbase = smem + (cur ^ 1) * TILE_BYTES;
    // ---- phase A: 32 MFMAs on F0 | 16 fragment loads of F1 (one per 2 MFMAs) | DMA pieces 4..15 of tile t+1 (three per 8 MFMAs)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mfma32_agpr<F16>(acc[i][j], w0[ks][i], x0[ks][j]);
          const int idx = ks * 16 + i * 4 + j;
          if ((idx & 1) == 0) {
            const int f = idx >> 1;  // 0..15
            const int fk = f >> 3, fq = f & 3;
            if ((f & 4) == 0) x1[fk][fq] = ldx(base, 2 + fk, fq);
            else w1[fk][fq] = ldw(base, 2 + fk, fq);
          }
          const int o = idx & 7;
          if (o == 2 || o == 5 || o == 7) {
            const int pc = 4 + (idx >> 3) * 3 + (o == 2 ? 0 : o == 5 ? 1 : 2);
            piece(cur ^ 1, sn1, pc);
          }
        }
    // ---- phase B1: 24 MFMAs on F1; the source of tile t+2 in their shadow
    Src sn2;
#pragma unroll
    for (int idx = 0; idx < 24; ++idx) {
      const int ks = idx >> 4, i = (idx >> 2) & 3, j = idx & 3;
      mfma32_agpr<F16>(acc[i][j], w1[ks][i], x1[ks][j]);
      if (idx == 1) sn2 = tsrc(t + 2);
    }
    __syncthreads();  // tile t+1 has landed for everyone; nobody reads tile t any more
    // ---- phase B2: last 8 MFMAs on F1 | 16 fragment loads of F0(t+1), two per MFMA | DMA pieces 0..3 of tile t+2
#pragma unroll
    for (int idx = 24; idx < 32; ++idx) {
      const int ks = 1, i = (idx >> 2) & 3, j = idx & 3;
      mfma32_agpr<F16>(acc[i][j], w1[ks][i], x1[ks][j]);
      const int f = idx - 24;  // 0..7
      x0[f >> 2][f & 3] = ldx(nbase, f >> 2, f & 3);
      w0[f >> 2][f & 3] = ldw(nbase, f >> 2, f & 3);
      if (f & 1) piece(cur, sn2, f >> 1);
    }
    if (t == nt - 1) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    cur ^= 1;
    sn1 = sn2;
  };
  for (;;) {
    for (int t = 0; t < nt; ++t) ktile(t);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
    const int em0 = m0, en0 = n0;
    const bool cont = has_next;
    if (cont) {
      vb += gridDim.x;
      if constexpr (RS) {
        if (p.rs_part) rs_base[(rs_sel ^ 1) * 256 + threadIdx.x] = row_rscale(p, nm0 + threadIdx.x);
      }
      m0 = nm0;
      n0 = nn0;
      srcA = nxtA;
      srcW = nxtW;
      has_next = vb + (int)gridDim.x < total_tiles;
      locate_next();
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // TIMING-ONLY epilogue: the 16x16 kernel's quad order over this kernel's accumulator registers (wrong element mapping)
      Acc32View a4{acc};
      gemm_epilogue<8, 8, EPI, true, F16>(p, a4, em0 + wm * 128, en0 + wn * 128, lane,
                                          (RS && p.rs_part) ? rs_base + rs_sel * 256 + wm * 128 : nullptr);
    }
    if (!cont) break;
    rs_sel ^= 1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        asm volatile("" : "+a"(acc[i][j]));
      }
    asm volatile("s_nop 7" ::: "memory");
  }
}

template <int EPI, bool F16>
static int launch_w4m32(const GemmP& p0, hipStream_t stream) {
  GemmP p = p0;
  p.tiles_m = p.M / 256;
  p.tiles_n = p.N / 256;
  p.xcd_gx = choose_xcd_partition(p);
  constexpr int lds = 2 * 512 * 128 + 2048;
  auto kfn = gemm_w4m32_kernel<EPI, F16>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      set_error("gemm_bf16(w4m32): hipFuncSetAttribute failed");
      return TCAVT_ERR_HIP;
    }
    attr_set = true;
  }
  const int tiles = p.tiles_m * p.tiles_n;
  int wgs = tiles;
  p.pers_tiles = 0;
  if (tiles > 256 && p.K >= 128) {
    p.pers_tiles = tiles;
    wgs = 256;
  }
  hipLaunchKernelGGL(kfn, dim3(wgs), dim3(256), lds, stream, p);
  TCAVT_CHECK_LAUNCH("gemm_bf16(w4m32)");
  return TCAVT_OK;
}
#endif  // TCAVT_EXPERIMENTS

// the forms of the SiLU*up epilogue the 4-wave kernel is built for (16-byte stores of the 16-bit operand type)
template <bool F16>
static bool silu16_ok(const GemmP& p) {
  return p.out_kind == (F16 ? TCAVT_F16 : TCAVT_BF16) && (p.ldc & 7) == 0;
}

template <int EPI, int B2R, int DBG = 0, bool BUF = false, int BN = 256, bool F16 = false>
static int launch_w4(const GemmP& p0, hipStream_t stream) {
  if constexpr (EPI == EPI_SILU || EPI == EPI_SILU_SAVE) {
    if (!silu16_ok<F16>(p0)) {
      set_error("gemm_bf16(w4): the SiLU epilogue of the 4-wave kernel writes the 16-bit operand type with ldc %% 8 == 0");
      return TCAVT_ERR_ARG;
    }
  }
  if constexpr (EPI == EPI_SILUBWD) {
    if (p0.out_kind != (F16 ? TCAVT_F16 : TCAVT_BF16) || (p0.ldc & 7) || (p0.ldaux & 7) || p0.aux == nullptr || p0.ldc < 2 * p0.N ||
        p0.ldaux < 2 * p0.N) {
      set_error("gemm_bf16(w4): SILU_BWD needs silu_preact, 16-bit output of the operand type, ldc / ld_preact %% 8 == 0 and >= 2 N");
      return TCAVT_ERR_ARG;
    }
  }
  if constexpr (EPI == EPI_ROPE) {
    if (p0.ldc & 7) {
      set_error("gemm_bf16(w4): the RoPE epilogue of the 4-wave kernel needs ldc %% 8 == 0");
      return TCAVT_ERR_ARG;
    }
  }
  if constexpr (EPI == EPI_NORM16) {
    if (p0.ldc & 7) {
      set_error("gemm_bf16(w4): the in-place 16-bit residual epilogue of the 4-wave kernel needs ldc %% 8 == 0");
      return TCAVT_ERR_ARG;
    }
  }
  if (BUF && ((long)256 * p0.lda * 2 + (long)p0.K * 2 >= (1L << 31) || (long)256 * p0.ldw * 2 + (long)p0.K * 2 >= (1L << 31))) {
    set_error("gemm_bf16(w4, buffer loads): a 256-row operand panel must span < 2 GiB");
    return TCAVT_ERR_ARG;
  }
  GemmP p = p0;
  p.tiles_m = p.M / 256;
  p.tiles_n = p.N / BN;
  p.xcd_gx = choose_xcd_partition(p);
  constexpr int lds = 2 * (256 + BN) * 128 + 2048;  // two tile buffers + two sets of 256 row scales (TCAVT_EPI_ROWSCALE)
  auto kfn = gemm_bf16_w4_kernel<EPI, B2R, DBG, BUF, BN, F16>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      set_error("gemm_bf16(w4): hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e));
      return TCAVT_ERR_HIP;
    }
    attr_set = true;
  }
  // more tiles than CUs: one persistent workgroup per CU walking its tiles as one K-tile stream (multiple of 8 so that
  // tile ids keep their XCD); TCAVT_GEMM_NO_PERSIST=1 launches one workgroup per tile (A/B)
  const int tiles = p.tiles_m * p.tiles_n;
  static const bool no_pers = getenv("TCAVT_GEMM_NO_PERSIST") != nullptr;
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n / 8 * 8;
  }();
  int wgs = tiles;
  p.pers_tiles = 0;
  // (the two-K-tile look-ahead may reach into the NEXT output tile only: at least two K-tiles per tile)
  if (w4_pers_ok(EPI, DBG, BUF) && !no_pers && n_cu >= 8 && tiles > n_cu && p.K + p.K2 >= 128) {
    p.pers_tiles = tiles;
    wgs = n_cu;
  }
  dim3 grid(wgs), block(256);
  hipLaunchKernelGGL(kfn, grid, block, lds, stream, p);
  TCAVT_CHECK_LAUNCH("gemm_bf16(w4)");
  return TCAVT_OK;
}

#ifdef TCAVT_EXPERIMENTS
template <int EPI, bool F16>
static int launch_ring(const GemmP& p0, int batch, hipStream_t stream) {
  GemmP p = p0;
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  p.xcd_gx = 8;
  constexpr int lds = 4 * 512 * 64;
  auto kfn = gemm_bf16_ring_kernel<EPI, F16>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      set_error("gemm_bf16(ring): hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e));
      return TCAVT_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, batch), block(512);
  hipLaunchKernelGGL(kfn, grid, block, lds, stream, p);
  TCAVT_CHECK_LAUNCH("gemm_bf16(ring)");
  return TCAVT_OK;
}
#endif  // TCAVT_EXPERIMENTS

// Launches below one wave of 256x256 tiles (tile = 128, 64 or 0 = pick).
template <int EPI, bool F16>
static int launch_small(const GemmP& p, int tile, int batch, hipStream_t stream) {
  GemmP q = p;
  q.prio = 0;
  // grids that leave CUs idle: the 4-stage loop (latency bound, one workgroup per CU is no loss);
  // fuller grids: interleaved DMA issue, 64 KiB of LDS so that two workgroups share a CU
  const long wgs = (long)((q.M + 127) / 128) * ((q.N + 127) / 128) * batch;
  static const bool no_deep = getenv("TCAVT_GEMM_NO_DEEP") != nullptr;  // A/B switches
  static const bool no_64 = getenv("TCAVT_GEMM_NO_64") != nullptr;
  if constexpr (EPI != EPI_ROPE && EPI != EPI_NORM && EPI != EPI_NORM16) {
    // very small grids (Q-Former projections, LoRA down-projection): 64x64 tiles, four times the workgroups,
    // each K-tile costing a quarter of the DMA issue and MFMA time
    // (not for EPI_NORM: the 64x64 form's waves cover 32 columns, no whole 64-column group)
    if ((tile == 64 || (tile == 0 && wgs <= 128 && !no_64)) && !no_deep) return launch<64, 64, 2, 2, EPI, F16, 2>(q, batch, stream);
  }
  if (wgs <= 256 && !no_deep) return launch<128, 128, 2, 2, EPI, F16, 2>(q, batch, stream);
  return launch<128, 128, 2, 2, EPI, F16, 1>(q, batch, stream);
}

// tile codes (tcavt_gemm_args.tile): 0 auto | 64, 128, 256 (8-wave), 257 (4-wave 256x256), 271 (4-wave 256x192),
// 272 (4-wave two-barrier form): the production variants, all bit-identical in results.
// Only with -DTCAVT_EXPERIMENTS (libtcavt_hip_exp.so, built and used by tools/ alone): the measured-and-rejected A/B
// variants 255 / 253 / 252 (burst DMA issue, static / no wave priority), 250 (32-deep ring), 258 / 259 / 268 / 269 / 270,
// 124-127, and the timing-only elimination experiments 261-267, which compute WRONG results.
template <int EPI, bool F16>
static int dispatch_tile(const GemmP& p, int tile, int batch, hipStream_t stream) {
  GemmP q = p;
  switch (tile) {
    case 256:  // DMA pieces interleaved with the MFMAs + s_setprio(1) around MFMA clusters (fastest measured)
      q.prio = 2;
      return launch<256, 256, 2, 4, EPI, F16, 1>(q, batch, stream);
#ifdef TCAVT_EXPERIMENTS
    case 255: q.prio = 2; return launch<256, 256, 2, 4, EPI, F16, 0>(q, batch, stream);
    case 253: q.prio = 1; return launch<256, 256, 2, 4, EPI, F16, 0>(q, batch, stream);
    case 252: q.prio = 0; return launch<256, 256, 2, 4, EPI, F16, 0>(q, batch, stream);
    case 250: return launch_ring<EPI, F16>(q, batch, stream);
#endif
    case 271:  // 256 x 192 tiles (N % 192 == 0)
      if (batch == 1 && (q.K2 == 0 || EPI == EPI_ROPE) && q.M % 256 == 0 && q.N % 192 == 0 &&
          (EPI != EPI_ROPE || q.out_kind == (F16 ? TCAVT_F16 : TCAVT_BF16)))
        return launch_w4<EPI, 4, 0, false, 192, F16>(q, stream);
      set_error("gemm_bf16: tile 271 (4-wave kernel, 256x192) needs M %% 256 == 0, N %% 192 == 0, no batch");
      return TCAVT_ERR_ARG;
#ifdef TCAVT_EXPERIMENTS
    case 273:  // 32x32x16 main loop, TIMING ONLY (wrong results): SiLU / in-place residual / plain 16-bit forms
      if (batch == 1 && q.K2 == 0 && q.M % 256 == 0 && q.N % 256 == 0) {
        if constexpr (EPI == EPI_SILU || EPI == EPI_NORM16 || EPI == EPI_GENERIC) return launch_w4m32<EPI, F16>(q, stream);
      }
      set_error("gemm_bf16: tile 273 (32x32 MFMA experiment) needs whole 256x256 tiles and the SiLU / NORM16 / generic epilogue");
      return TCAVT_ERR_ARG;
#endif
    case 257: case 272:
#ifdef TCAVT_EXPERIMENTS
    case 258: case 259: case 268: case 269: case 270: case 266: case 274:
#endif
      if (batch == 1 && (q.K2 == 0 || EPI == EPI_ROPE) && q.M % 256 == 0 && q.N % 256 == 0 &&
          (EPI != EPI_ROPE || q.out_kind == (F16 ? TCAVT_F16 : TCAVT_BF16))) {
        if (tile == 272) {
          if constexpr (EPI != EPI_ROPE) return launch_w4<EPI, 2, 64, false, 256, F16>(q, stream);
        }
#ifdef TCAVT_EXPERIMENTS
        if constexpr (!F16) {
          if (tile == 270) {
            if constexpr (EPI != EPI_ROPE) return launch_w4<EPI, 2, 0, true>(q, stream);
          }
          if (tile == 268) return launch_w4<EPI, 2, 16>(q, stream);
          if (tile == 269) return launch_w4<EPI, 2, 32>(q, stream);
          if (tile == 258) return launch_w4<EPI, 3>(q, stream);
          if (tile == 259) return launch_w4<EPI, 4>(q, stream);
        }
        if constexpr (EPI != EPI_ROPE) {  // tile-major W addressing, TIMING ONLY (wrong results): one-barrier / deep form
          if (tile == 266) return launch_w4<EPI, 2, 128, false, 256, F16>(q, stream);
          if (tile == 274) return launch_w4<EPI, 2, 128 + 64, false, 256, F16>(q, stream);
        }
#endif
        return launch_w4<EPI, 2, 0, false, 256, F16>(q, stream);
      }
      set_error("gemm_bf16: tile %d (4-wave kernel) needs whole 256x256 tiles, one K source, no batch", tile);
      return TCAVT_ERR_ARG;
#ifdef TCAVT_EXPERIMENTS
    case 261: case 262: case 263: case 264: case 265: case 267: {  // timing experiments (wrong results)
      static const bool allow = getenv("TCAVT_GEMM_TIMING_EXPERIMENTS") != nullptr;
      if (!allow) {
        set_error("gemm_bf16: tile codes 261-267 are timing-only elimination experiments that compute WRONG results; "
                  "set TCAVT_GEMM_TIMING_EXPERIMENTS=1 to run them (tools/ab_w4_dbg.py)");
        return TCAVT_ERR_ARG;
      }
      if (!(q.M % 256 == 0 && q.N % 256 == 0 && batch == 1 && q.K2 == 0)) {
        set_error("gemm_bf16: timing experiments need whole 256x256 tiles");
        return TCAVT_ERR_ARG;
      }
      if constexpr (!F16 && EPI == EPI_SILU) {
        if (tile == 261) return launch_w4<EPI, 2, 1>(q, stream);
        if (tile == 262) return launch_w4<EPI, 2, 2>(q, stream);
        if (tile == 263) return launch_w4<EPI, 2, 3>(q, stream);
        if (tile == 264) return launch_w4<EPI, 2, 4>(q, stream);
        if (tile == 265) return launch_w4<EPI, 2, 8>(q, stream);
        return launch_w4<EPI, 2, 7>(q, stream);
      }
      set_error("gemm_bf16: the timing experiments exist for the SiLU epilogue only");
      return TCAVT_ERR_ARG;
    }
    case 127: q.prio = 2; return launch<128, 128, 2, 2, EPI, F16, 0>(q, batch, stream);
    case 126: q.prio = 0; return launch<128, 128, 2, 2, EPI, F16, 0>(q, batch, stream);
    case 125: q.prio = 0; return launch<128, 128, 2, 2, EPI, F16, 2>(q, batch, stream);
    case 124: q.prio = 0; return launch<128, 128, 2, 2, EPI, F16, 1>(q, batch, stream);
#endif  // TCAVT_EXPERIMENTS
    default: return launch_small<EPI, F16>(q, tile, batch, stream);  // 128 / 64 / 0 (auto)
  }
}

// ===========================================================================
// Skinny form (M <= 32 rows: the decode step of text generation, one row per sample).  The contraction is a stream of
// the weight matrix through the chip, HBM-bound; the 256-row tiles above would leave all but a handful of CUs idle
// (N / 128 workgroups) and spend 8x the MFMA work on padding rows.  Here a workgroup owns NCB blocks of 16 output
// columns and ALL rows; its eight waves split K, each streaming its slice of the 16 x K weight panel straight from
// global memory into MFMA A fragments (16 bytes per lane, 64 contiguous bytes per weight row and instruction), with the
// <= 32 activation rows (L2-resident) as B fragments; the eight partial accumulators meet in LDS and are added in wave
// order (bit-reproducible).  Epilogues as above; TCAVT_EPI_NORM_OUT writes one partial sum of squares per workgroup
// (16 columns): norm_out_npart() in common.hpp tells producers and consumers the count.
// ===========================================================================
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int SK_WAVES = 8;

// agent-scope store / load of four floats (relaxed atomics: global_store / global_load ... sc1, coherent across the XCDs)
__device__ __forceinline__ void sk_store(float* ptr, f32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) __hip_atomic_store(ptr + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 sk_load(const float* ptr) {
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = __hip_atomic_load(ptr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}

template <int EPI, int NCB, bool F16, bool NT>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_kernel(GemmP p) {
  __shared__ f32x4 red[SK_WAVES][NCB * 2][64];
  // fused LoRA down-projection, consumer side (RoPE form): group sums of the partials, then t as 16-bit rows [32][32]
  __shared__ float lp_sum[EPI == EPI_ROPE ? 512 : 1];
  __shared__ __attribute__((aligned(16))) bf16_t lp_t[EPI == EPI_ROPE ? 32 * 32 : 8];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool lp_in = EPI == EPI_ROPE && p.lp_np > 0;
  if constexpr (EPI == EPI_ROPE) {
    if (lp_in) reinterpret_cast<unsigned int*>(lp_t)[threadIdx.x] = 0u;  // 512 threads x 4 B = the whole tile
  }
  // Split K (p.sk_split = S > 1): the S slices of one column block get ids that are congruent mod 8 -- workgroups are dealt
  // round-robin to the 8 XCDs, so the slabs the last arriver reads were written through its own XCD's L2 (a speed choice
  // only: the hand-off below is correct for any placement).  Host: (number of column blocks) % 8 == 0 when S > 1.
  const int S = p.sk_split;
  int blk = blockIdx.x, ks = 0, mrow0 = 0;
  if (S > 1) {
    const int q = blockIdx.x >> 3;
    ks = q % S;
    blk = (q / S) * 8 + (blockIdx.x & 7);
  } else if (p.sk_msplit > 1) {  // (the two token blocks of a column block: ids congruent mod 8 -> one XCD, the weights' second read is an L2 hit)
    const int q = blockIdx.x >> 3;
    mrow0 = 16 * (q & 1);
    blk = (q >> 1) * 8 + (blockIdx.x & 7);
  }
  const int n0 = blk * (16 * NCB);
  const int r16 = lane & 15, kq = lane >> 4;
  // first output column of column block c.  RoPE: a workgroup owns the two 16-column blocks of one head that rotate
  // together (dimensions d and d + 32), so that two workgroups share a head (96 workgroups for the fused q|k|v instead of 48)
  int ncol[NCB];
#pragma unroll
  for (int c = 0; c < NCB; ++c)
    ncol[c] = EPI == EPI_ROPE ? (blk >> 1) * 64 + (blk & 1) * 16 + c * 32 : n0 + c * 16;
  f32x4 acc[NCB][2];
#pragma unroll
  for (int c = 0; c < NCB; ++c) acc[c][0] = acc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this wave's K slice: K / (8 S) (a multiple of 32), 32 per MFMA step
  const int kper = p.K / (SK_WAVES * S);
  const int kbeg = (ks * SK_WAVES + wave) * kper;
  // Weight fragments: row-major W -> 16 rows x 64 bytes per instruction, K * 2 bytes apart; fragment-major copy (w_frag,
  // tcavt_pack_weight16) -> the same 1 KiB as consecutive bytes, the wave's K slice one contiguous run (k advances 16 x as fast)
  const bf16_t* wp[NCB];
  const int wstep = p.w_frag ? 16 : 1;
#pragma unroll
  for (int c = 0; c < NCB; ++c)
    wp[c] = p.w_frag ? p.W + (long)ncol[c] * p.K + (long)kbeg * 16 + lane * 8 : p.W + (long)(ncol[c] + r16) * p.ldw + kbeg + kq * 8;
  const bf16_t* xp0 = p.A + (long)min(mrow0 + r16, p.M - 1) * p.lda + kbeg + kq * 8;
  const bf16_t* xp1 = p.A + (long)min(16 + r16, p.M - 1) * p.lda + kbeg + kq * 8;
  // fragment-major activations: the 16 tokens' fragments of a k-step are one 1 KiB run as well (rows >= M of a block hold
  // whatever the producer left: they only reach output columns >= M, which nobody stores)
  // (one block of 8 tokens, M <= 8: a k-step is 512 bytes, lanes r and r + 8 read the same 16)
  const int xstep = p.a_frag == 2 ? 8 : p.a_frag ? 16 : 1;
  if (p.a_frag == 2) {
    xp0 = p.A + (long)kbeg * 8 + kq * 64 + (r16 & 7) * 8;
  } else if (p.a_frag) {
    xp0 = p.A + (long)mrow0 * p.K + (long)kbeg * 16 + lane * 8;
    xp1 = p.A + (long)16 * p.K + (long)kbeg * 16 + lane * 8;
  }
  const bool two = p.M > 16 && p.sk_msplit <= 1;
  constexpr int U = 4;  // k-steps in flight (8 made the decode step slower: 3.08 vs 2.60 ms in round 2, and again in round 3 for the residual forms alone: 1.155 vs 1.138; so did 16 waves with K / 16 slices each: 1.55 vs 1.34 ms)
  // Epilogue operands of the two finishing waves (wave mb completes token block mb), fetched while the first batch of weight
  // loads is in flight instead of after the K loop: the row scale's partial sums, the RoPE position -> cos / sin rows, the
  // 16-bit residual, and (wave 0) the LoRA second source.  Each of these was one more dependent global-memory round trip
  // at the tail of a kernel that is a few microseconds long (decode step).
  const int pm = mrow0 + wave * 16 + r16;  // (meaningful for wave < 2)
  const long pmm = pm < p.M ? pm : 0;
  float rs = 1.f;
  f32x4 rope_c = {1.f, 1.f, 1.f, 1.f}, rope_s = {0.f, 0.f, 0.f, 0.f};
  u32x2 old16[NCB];
  u32x4 l_a0[2], l_a1[2], l_w[2][NCB];
  // Row scale of the fused RMSNorm (finishing waves): the H / 16 partial sums of a token are split over the four lanes that
  // share it (kq), eight quads each and ALL requested before the first weight batch -- row_rscale's index-order loop was four
  // dependent round trips at the head of a launch that lasts ten microseconds.  (Sum order: per lane in index order, then
  // the four lanes; the tiled kernels add in index order throughout -- same value up to fp32 summation order.)
  constexpr bool RSK = EPI == EPI_SILU || EPI == EPI_ROPE;
  f32x4 rsq[RSK ? 8 : 1];
  int rope_pos_v = 0;
  if constexpr (RSK) {
    if (wave < 2 && p.rs_part) {
      const f32x4* q = reinterpret_cast<const f32x4*>(p.rs_part + pmm * p.rs_npart);
      const int nq4 = p.rs_npart >> 2, per = (nq4 + 3) >> 2;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = kq * per + i;
        rsq[i] = q[min(idx, nq4 - 1)];
        if (i >= per || idx >= nq4) rsq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if constexpr (EPI == EPI_ROPE) {
      if (wave < 2) rope_pos_v = p.rope_pos ? p.rope_pos[pmm] : (int)(pmm % p.rope_L);
    }
  }
  constexpr bool NORMF = EPI == EPI_NORM || EPI == EPI_NORM16;
  u32x2 lp_af[NORMF ? NCB : 1][NORMF ? 16 : 1];  // producer: this lane's 4 columns of the 16 adapter rows
  // fused LoRA down-projection, consumer side: t = lp_scale * sum of the lp_np partials the previous residual GEMM left --
  // value (m, j) by thread m * 16 + j of group g, groups of lp_np / G consecutive partials (G = a power of two that divides
  // lp_np), combined in group order (this workgroup's tokens: all M, or the 16-token block it owns when the token blocks
  // are split, sk_msplit).  The first 32 partials of a thread are requested HERE, before the first weight batch (they are
  // back before it; added up under it), the rest (M > 8) in the same place as before
  float lp_tv[EPI == EPI_ROPE ? 32 : 1];
  const float* lp_src = nullptr;
  int lp_per = 0, lp_nall = 0;
  if constexpr (EPI == EPI_ROPE) {
    if (lp_in) {
      const int mtok = p.sk_msplit > 1 ? min(16, p.M - mrow0) : p.M;
      const int nv = mtok * 16;
      lp_nall = p.M * 16;
      int G = 1;
      while (2 * G * nv <= SK_WAVES * 64 && p.lp_np % (2 * G) == 0) G *= 2;
      const int tid = threadIdx.x;
      if (tid < G * nv) {
        const int g = tid / nv, v = tid - g * nv;
        lp_per = p.lp_np / G;
        lp_src = p.lp_part + (long)g * lp_per * lp_nall + mrow0 * 16 + v;
#pragma unroll
        for (int u = 0; u < 32; ++u) lp_tv[u] = lp_src[(long)min(u, lp_per - 1) * lp_nall];
      }
    }
  }
  for (int k = 0; k < kper; k += 32 * U) {
    u32x4 wf[U][NCB], x0[U], x1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k + 32 * u < kper) {
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          const u32x4* wsrc = reinterpret_cast<const u32x4*>(wp[c] + (k + 32 * u) * wstep);
          wf[u][c] = NT ? __builtin_nontemporal_load(wsrc) : *wsrc;
        }
        x0[u] = *reinterpret_cast<const u32x4*>(xp0 + (k + 32 * u) * xstep);
        if (two) x1[u] = *reinterpret_cast<const u32x4*>(xp1 + (k + 32 * u) * xstep);
      }
    }
    if constexpr (EPI == EPI_ROPE) {
      if (k == 0 && lp_src) {
        float acc_t = 0.f;
#pragma unroll
        for (int u = 0; u < 32; ++u) acc_t += u < lp_per ? lp_tv[u] : 0.f;
        for (int i = 32; i < lp_per; i += 32) {
          float tv[32];
#pragma unroll
          for (int u = 0; u < 32; ++u) tv[u] = lp_src[(long)min(i + u, lp_per - 1) * lp_nall];
#pragma unroll
          for (int u = 0; u < 32; ++u) acc_t += i + u < lp_per ? tv[u] : 0.f;
        }
        lp_sum[threadIdx.x] = acc_t;
      }
    }
    if (k == 0 && wave < 2) {
      if constexpr (EPI == EPI_SILU || EPI == EPI_ROPE) {
        if (p.rs_part) {
          const int nq4 = p.rs_npart >> 2, per = (nq4 + 3) >> 2;
          float ss = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            ss += rsq[i][0];
            ss += rsq[i][1];
            ss += rsq[i][2];
            ss += rsq[i][3];
          }
          if (per > 8) {  // (more than 128 partials per row: the rest in further batches)
            const f32x4* q = reinterpret_cast<const f32x4*>(p.rs_part + pmm * p.rs_npart);
            for (int i = 8; i < per; ++i) {
              const int idx = kq * per + i;
              if (idx < nq4) {
                const f32x4 v4 = q[idx];
                ss += v4[0];
                ss += v4[1];
                ss += v4[2];
                ss += v4[3];
              }
            }
          }
          ss += __shfl_xor(ss, 16, 64);
          ss += __shfl_xor(ss, 32, 64);
          rs = rsqrtf(ss * p.rs_inv_h + p.rs_eps);
        }
      }
      if constexpr (EPI == EPI_ROPE) {
        if (ncol[0] < p.rope_cols) {
          const int pos = rope_pos_v;
          const int d = (blk & 1) * 16 + 4 * kq;
          rope_c = *reinterpret_cast<const f32x4*>(p.cosT + pos * 32 + d);
          rope_s = *reinterpret_cast<const f32x4*>(p.sinT + pos * 32 + d);
        }
        if (lp_in) {
          // (the partial sums: below, by all eight waves)
          if (wave == 0) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) l_w[0][c] = *reinterpret_cast<const u32x4*>(p.W2 + (long)(ncol[c] + r16) * p.ldw2 + kq * 8);
          }
        } else if (wave == 0 && p.K2 > 0 && ks == 0) {  // (the second K source is added once: by slice 0)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (32 * u < p.K2) {
              l_a0[u] = *reinterpret_cast<const u32x4*>(p.A2 + (long)min(mrow0 + r16, p.M - 1) * p.lda2 + 32 * u + kq * 8);
              l_a1[u] = *reinterpret_cast<const u32x4*>(p.A2 + (long)min(16 + r16, p.M - 1) * p.lda2 + 32 * u + kq * 8);
#pragma unroll
              for (int c = 0; c < NCB; ++c)
                l_w[u][c] = *reinterpret_cast<const u32x4*>(p.W2 + (long)(ncol[c] + r16) * p.ldw2 + 32 * u + kq * 8);
            }
          }
        }
      }
      if constexpr (EPI == EPI_NORM16) {
        if (p.flags & TCAVT_EPI_RESIDUAL) {
#pragma unroll
          for (int c = 0; c < NCB; ++c)
            old16[c] = *reinterpret_cast<const u32x2*>(p.res16 + (p.o_frag ? frag_off((int)pmm, n0 + c * 16 + 4 * kq, p.N, p.o_frag)
                                                                            : pmm * p.ldc + n0 + c * 16 + 4 * kq));
        }
      }
      if constexpr (NORMF) {
        if (p.lp_a) {
#pragma unroll
          for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int j = 0; j < 16; ++j)
              lp_af[c][j] = *reinterpret_cast<const u32x2*>(p.lp_a + (long)(j < 8 ? j : 8 + j) * p.lp_lda + n0 + c * 16 + 4 * kq);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k + 32 * u < kper) {
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          acc[c][0] = mfma16<F16>(wf[u][c], x0[u], acc[c][0]);
          if (two) acc[c][1] = mfma16<F16>(wf[u][c], x1[u], acc[c][1]);
        }
      }
    }
  }
  if constexpr (EPI == EPI_ROPE) {
    if (lp_in) {  // (uniform) fused LoRA down-projection: group sums -> t rows in LDS -> wave 0's B fragments, one 32-deep step
      __syncthreads();
      const int mtok = p.sk_msplit > 1 ? min(16, p.M - mrow0) : p.M;  // (t rows in LDS: local token index)
      const int nv = mtok * 16;
      if ((int)threadIdx.x < nv) {
        int G = 1;
        while (2 * G * nv <= SK_WAVES * 64 && p.lp_np % (2 * G) == 0) G *= 2;
        float tot = lp_sum[threadIdx.x];
        for (int g = 1; g < G; ++g) tot += lp_sum[g * nv + threadIdx.x];
        const int m_ = threadIdx.x >> 4, j = threadIdx.x & 15;
        lp_t[m_ * 32 + (j < 8 ? j : 8 + j)] = to16<F16>(tot * p.lp_scale);
      }
      __syncthreads();
      if (wave == 0 && ks == 0) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(lp_t + min(r16, mtok - 1) * 32 + kq * 8);
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(lp_t + min(16 + r16, mtok - 1) * 32 + kq * 8);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          acc[c][0] = mfma16<F16>(l_w[0][c], a0, acc[c][0]);
          if (two) acc[c][1] = mfma16<F16>(l_w[0][c], a1, acc[c][1]);
        }
      }
    }
  }
  if constexpr (EPI == EPI_ROPE) {  // LoRA second K source (K2 = 64: two steps), done by wave 0 (of slice 0)
    if (!lp_in && p.K2 > 0 && wave == 0 && ks == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {  // the first two steps come from the registers filled under the first weight batch
        if (32 * u < p.K2) {
#pragma unroll
          for (int c = 0; c < NCB; ++c) {
            acc[c][0] = mfma16<F16>(l_w[u][c], l_a0[u], acc[c][0]);
            if (two) acc[c][1] = mfma16<F16>(l_w[u][c], l_a1[u], acc[c][1]);
          }
        }
      }
      for (int k = 64; k < p.K2; k += 32) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(p.A2 + (long)min(mrow0 + r16, p.M - 1) * p.lda2 + k + kq * 8);
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(p.A2 + (long)min(16 + r16, p.M - 1) * p.lda2 + k + kq * 8);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          const u32x4 w2 = *reinterpret_cast<const u32x4*>(p.W2 + (long)(ncol[c] + r16) * p.ldw2 + k + kq * 8);
          acc[c][0] = mfma16<F16>(w2, a0, acc[c][0]);
          if (two) acc[c][1] = mfma16<F16>(w2, a1, acc[c][1]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NCB; ++c) {
    red[wave][c * 2][lane] = acc[c][0];
    red[wave][c * 2 + 1][lane] = acc[c][1];
  }
  __syncthreads();
  const bool finisher = wave == 0 || (wave == 1 && two);
  if (S <= 1 && !finisher) return;
  // ---- wave mb (0 / 1) finishes token block mb: lane holds features 4 (lane >> 4) .. + 3 of every column block for
  // token m = 16 mb + (lane & 15); the partials are added in wave order
  const int mb = wave & 1;
  const int m = mrow0 + mb * 16 + r16, nq = 4 * kq;
  const bool rowok = m < p.M;
  const long mm = rowok ? m : 0;
  f32x4 v[NCB];
  if (finisher) {
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      f32x4 t = red[0][c * 2 + mb][lane];
#pragma unroll
      for (int w = 1; w < SK_WAVES; ++w) t += red[w][c * 2 + mb][lane];
      v[c] = t;
    }
  }
  if (S > 1) {
    // ---- cross-workgroup combine: every slice writes its partial sums to its slab, every storing wave drains its stores,
    // barrier, ONE agent-scope ticket; the workgroup that draws S - 1 adds the S slabs in slice order (bit-reproducible
    // whatever the arrival order) and runs the epilogue.  The counter is re-armed by the last arriver (zeroed once by the
    // caller before first use).  The slabs move with agent-scope (sc1) stores and loads -- write-through to / read from the
    // point where the 8 XCDs' L2s agree -- and the order "slab stores complete -> ticket" is the s_waitcnt + barrier: the
    // release / acquire FENCES that plain stores would need write back and invalidate a whole L2 per workgroup
    // (buffer_wbl2 / buffer_inv: measured + 10 us per launch, twice what the split gains).
    float* slab = p.sk_slab + ((long)(blk * S + ks) * 2 * NCB) * 256;  // [mb][c][64 lanes][4]
    if (finisher) {
#pragma unroll
      for (int c = 0; c < NCB; ++c) sk_store(slab + ((mb * NCB + c) * 64 + lane) * 4, v[c]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* last_flag = reinterpret_cast<int*>(&red[0][0][0]);  // (the one LDS array: all waves are past their reads of it)
    if (threadIdx.x == 0) {
      const int ticket = __hip_atomic_fetch_add(p.sk_cnt + blk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == S - 1;
      if (last) __hip_atomic_store(p.sk_cnt + blk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
      *last_flag = last;
    }
    __syncthreads();
    if (*last_flag == 0 || !finisher) return;
    const float* base = p.sk_slab + ((long)(blk * S) * 2 * NCB) * 256;
    f32x4 part[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) v[c] = sk_load(base + ((mb * NCB + c) * 64 + lane) * 4);
    for (int s2 = 1; s2 < S; ++s2) {
#pragma unroll
      for (int c = 0; c < NCB; ++c) part[c] = sk_load(base + (long)s2 * 2 * NCB * 256 + ((mb * NCB + c) * 64 + lane) * 4);
#pragma unroll
      for (int c = 0; c < NCB; ++c) v[c] += part[c];
    }
  }
  constexpr int OUT16 = F16 ? TCAVT_F16 : TCAVT_BF16;
  if constexpr (EPI == EPI_GENERIC) {
    if (!rowok) return;
#pragma unroll
    for (int c = 0; c < NCB; ++c) store_quad(p, m, n0 + c * 16 + nq, v[c] * p.acc_scale);
  } else if constexpr (EPI == EPI_NORM || EPI == EPI_NORM16) {
    const bool res = p.flags & TCAVT_EPI_RESIDUAL;
    float ss = 0.f;
    f32x4 hq[NCB];  // the 16-bit stream's values (what the next layer's projections read)
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      f32x4 o = v[c];
      const long off = mm * p.ldc + n0 + c * 16 + nq;
      if constexpr (EPI == EPI_NORM16) {  // 16-bit residual stream: in place, sums of the rounded values (see gemm_epilogue)
        f32x4 oldv = {0.f, 0.f, 0.f, 0.f};
        if (res) {
          const u32x2 old = old16[c];
          oldv = f32x4{from16_lo<F16>(old[0]), from16_hi<F16>(old[0]), from16_lo<F16>(old[1]), from16_hi<F16>(old[1])};
        }
        o = fma4(o, p.norm_scale, oldv);
        const u32x2 w = u32x2{pack16x2<F16>(o[0], o[1]), pack16x2<F16>(o[2], o[3])};
        if (rowok) *reinterpret_cast<u32x2*>(p.norm_h16 + (p.o_frag ? frag_off(m, n0 + c * 16 + nq, p.N, p.o_frag) : off)) = w;
        o = f32x4{from16_lo<F16>(w[0]), from16_hi<F16>(w[0]), from16_lo<F16>(w[1]), from16_hi<F16>(w[1])};
        hq[c] = o;
      } else {
        if (res) o += *reinterpret_cast<const f32x4*>(p.residual + mm * p.ldr + n0 + c * 16 + nq);
        if (rowok) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + off) = o;
        o *= p.norm_scale;  // (the 16-bit copy and the sums below are kept at norm_scale)
        const u32x2 w = u32x2{pack16x2<F16>(o[0], o[1]), pack16x2<F16>(o[2], o[3])};
        if (rowok) *reinterpret_cast<u32x2*>(p.norm_h16 + off) = w;
        hq[c] = f32x4{from16_lo<F16>(w[0]), from16_hi<F16>(w[0]), from16_lo<F16>(w[1]), from16_hi<F16>(w[1])};
      }
      ss += o[0] * o[0];
      ss += o[1] * o[1];
      ss += o[2] * o[2];
      ss += o[3] * o[3];
    }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lane < 16 && rowok) {
      p.norm_part[(long)m * (p.N / (16 * NCB)) + blk] = ss;
      flag_nonfinite(p, ss);
    }
    if (p.lp_a) {  // (uniform) this workgroup's share of the next layer's LoRA down-projection
      float pj[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float a_ = 0.f;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          const u32x2 w = lp_af[c][j];
          a_ += hq[c][0] * from16_lo<F16>(w[0]);
          a_ += hq[c][1] * from16_hi<F16>(w[0]);
          a_ += hq[c][2] * from16_lo<F16>(w[1]);
          a_ += hq[c][3] * from16_hi<F16>(w[1]);
        }
        a_ += __shfl_xor(a_, 16, 64);
        a_ += __shfl_xor(a_, 32, 64);
        pj[j] = a_;
      }
      if (lane < 16 && rowok) {
        float* dst = p.lp_part + ((long)blk * p.M + m) * 16;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          *reinterpret_cast<f32x4*>(dst + 4 * q4) = f32x4{pj[4 * q4], pj[4 * q4 + 1], pj[4 * q4 + 2], pj[4 * q4 + 3]};
      }
    }
  } else if constexpr (EPI == EPI_SILU) {
    static_assert(EPI != EPI_SILU || NCB == 2, "gate block + up block");
    if (!rowok) return;
    const f32x4 g = v[0] * rs, u = v[NCB - 1] * rs;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = silu_mul(g[e], u[e]);
    if (p.o_frag)  // (16-bit output of the operand type: checked by the host)
      *reinterpret_cast<u32x2*>(static_cast<bf16_t*>(p.C) + frag_off(m, (n0 >> 1) + nq, p.N >> 1, p.o_frag)) =
          u32x2{pack16x2<F16>(o[0], o[1]), pack16x2<F16>(o[2], o[3])};
    else store_quad(p, m, (n0 >> 1) + nq, o);
  } else {  // EPI_ROPE: dimensions d = 16 half + nq .. + 3 and d + 32 of one head
    static_assert(EPI != EPI_ROPE || NCB == 2, "two partner blocks per workgroup");
    if (!rowok) return;
    const bool rot = ncol[0] < p.rope_cols;
    f32x4 lo = v[0] * rs, hi = v[NCB - 1] * rs;
    if (rot) {
      const f32x4 c = rope_c, sn = rope_s;
      const f32x4 l2 = lo * c - hi * sn, h2 = hi * c + lo * sn;
      lo = l2;
      hi = h2;
    }
    store_quad(p, m, ncol[0] + nq, lo);
    store_quad(p, m, ncol[NCB - 1] + nq, hi);
  }
  (void)OUT16;
}

template <int EPI, int NCB, bool F16>
static int launch_skinny(const GemmP& p, hipStream_t stream) {
  GemmP q = p;
  const int nblk = p.N / (16 * NCB);  // (RoPE: N / 64 heads x 2 halves = N / 32)
  // split K over S workgroups per column block when the caller lent a workspace: the decode step's projections are streams of
  // their weights, and N / 16 workgroups of 8 waves (128 for N = 2048: half the CUs, 4 KB per wave in flight) cannot keep the
  // memory system busy -- S is chosen so that ~two workgroups per CU stream, each wave's K slice staying a multiple of 32
  int S = 1;
  if (p.sk_slab && p.sk_cnt && nblk % 8 == 0 && !p.lp_a && p.lp_np == 0) {
    static const int max_wg = [] { const char* e = getenv("TCAVT_SK_MAXWG"); return e ? atoi(e) : 640; }();
    while (S < 8 && nblk * S * 2 <= max_wg && p.K % (SK_WAVES * S * 2 * 32) == 0) S *= 2;
    if ((long)nblk * S * 2 * NCB * 256 * 4 > p.sk_slab_bytes || nblk > p.sk_cnt_n) S = 1;
  }
  q.sk_split = S;
  static const bool no_msplit = getenv("TCAVT_SK_NO_MSPLIT") != nullptr;  // (A/B switch)
  // (only where the column blocks alone leave CUs idle -- o, down, q|k|v: 96-128 of 256; with more workgroups than CUs the
  //  second read of every weight row costs more than the activation rows it saves: gate|up 25.8 -> 33.2 us, lm_head likewise)
  const int msplit = (S == 1 && p.M > 16 && nblk % 8 == 0 && nblk <= 256 && !no_msplit) ? 2 : 1;
  q.sk_msplit = msplit;
  // Non-temporal weight loads where every weight byte is read ONCE per launch (one workgroup per column block) from the
  // fragment-major copy: 0.925 -> 0.885 ms per decode step at B = 8.  (On row-major weights nt was slower, 1.15 vs 1.09 ms --
  // the two 64-byte halves of a 128-byte line are fetched by different instructions there; with the token blocks on two
  // workgroups the second reader wants the L2 copy.)
  static const bool no_nt = getenv("TCAVT_SK_NO_NT") != nullptr;  // (A/B switch)
  const dim3 grid(nblk * S * msplit), block(SK_WAVES * 64);
  if (q.w_frag && msplit == 1 && !no_nt) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, NCB, F16, true>), grid, block, 0, stream, q);
  else hipLaunchKernelGGL((gemm_skinny_kernel<EPI, NCB, F16, false>), grid, block, 0, stream, q);
  TCAVT_CHECK_LAUNCH("gemm_bf16(skinny)");
  return TCAVT_OK;
}

// ---------------------------------------------------------------------------
// Fragment-major copy of a weight matrix for the skinny form (tcavt.h: tcavt_pack_weight16).  One thread per 16-byte piece:
// piece (b, j, l) <- W[16 b + (l & 15)][32 j + 8 (l >> 4) .. + 7]; the writes are consecutive, the reads 16 rows x 64 bytes.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight16_kernel(const bf16_t* __restrict__ W, long ldw, bf16_t* __restrict__ out, int N, int K) {
  const long piece = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)N * K / 8;
  if (piece >= total) return;
  const int l = (int)(piece & 63);
  const long chunk = piece >> 6;
  const int ksteps = K >> 5;
  const long b = chunk / ksteps;
  const int j = (int)(chunk - b * ksteps);
  const u32x4 v = *reinterpret_cast<const u32x4*>(W + (16 * b + (l & 15)) * ldw + 32 * j + 8 * (l >> 4));
  *reinterpret_cast<u32x4*>(out + piece * 8) = v;
}

// ---------------------------------------------------------------------------
// Split K for residual GEMMs that cannot fill the chip (round 4).  At M = 1024 (BASELINE config 4's low end: B = 8, L = 128) the
// o / down projections are 128 tiles of 128 x 128 -- half the CUs, one 4-wave workgroup each walking 32 / 128 K-tiles alone at
// ~1 us per K-tile (its waves issue DMA, fragment loads and MFMAs one after the other; nothing else is resident to overlap
// them): 30 / 90 us where the arithmetic is worth 7 / 29.  With a workspace the launch becomes TWO: (1) the S partial products
// over K / S as a batched launch of the generic fp32 form into S slabs [M][N] -- 4 x the workgroups, two per CU on the 64 KiB
// form, each a quarter of the chain -- and (2) this kernel: slabs added in slice order (bit-reproducible), then exactly the
// in-place 16-bit residual epilogue of TCAVT_EPI_NORM_OUT (round(norm_scale * acc + h16), partial sums of squares of the rounded
// values per 64 columns, range flag).  No cross-workgroup hand-off inside a kernel: the launch boundary is the reduction's barrier.
// ---------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void splitk_norm16_kernel(const float* __restrict__ slabs, int S, long slab_stride,
                                                            const bf16_t* __restrict__ res16, bf16_t* __restrict__ out16,
                                                            float* __restrict__ part, int M, int N, long ldc, int npart, float nscale,
                                                            int has_res, int* __restrict__ nf_flag, int nf_tag) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one thread: 8 consecutive columns of one row
  const int per_row = N >> 3;
  const long m = idx / per_row;
  const int c0 = (int)(idx - m * per_row) * 8;
  const bool on = m < M;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  u32x4 old = {0u, 0u, 0u, 0u};
  if (on) {
    const float* sp = slabs + m * N + c0;
    if (has_res) old = *reinterpret_cast<const u32x4*>(res16 + m * ldc + c0);
    for (int s_ = 0; s_ < S; ++s_) {  // (slice order: the sum does not depend on which workgroup finished first)
      a0 += *reinterpret_cast<const f32x4*>(sp + s_ * slab_stride);
      a1 += *reinterpret_cast<const f32x4*>(sp + s_ * slab_stride + 4);
    }
  }
  const f32x4 v0 = fma4(a0, nscale, f32x4{from16_lo<F16>(old[0]), from16_hi<F16>(old[0]), from16_lo<F16>(old[1]), from16_hi<F16>(old[1])});
  const f32x4 v1 = fma4(a1, nscale, f32x4{from16_lo<F16>(old[2]), from16_hi<F16>(old[2]), from16_lo<F16>(old[3]), from16_hi<F16>(old[3])});
  const u32x4 w = {pack16x2<F16>(v0[0], v0[1]), pack16x2<F16>(v0[2], v0[3]), pack16x2<F16>(v1[0], v1[1]), pack16x2<F16>(v1[2], v1[3])};
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float lo = from16_lo<F16>(w[e]), hi = from16_hi<F16>(w[e]);
    ss += lo * lo;
    ss += hi * hi;
  }
  if (on) *reinterpret_cast<u32x4*>(out16 + m * ldc + c0) = w;
  if (!on) ss = 0.f;
  ss += __shfl_xor(ss, 1, 64);  // the 8 lanes of a 64-column group are consecutive (N % 64 == 0)
  ss += __shfl_xor(ss, 2, 64);
  ss += __shfl_xor(ss, 4, 64);
  if (on && (threadIdx.x & 7) == 0) {
    part[m * npart + (c0 >> 6)] = ss;
    if (nf_flag && !(ss <= 3.0e38f)) atomicCAS(nf_flag, 0, nf_tag);
  }
}

// slices for the two-launch split: enough 128 x 128 tiles for ~two workgroups per CU, slices of whole K-tiles and >= 2048 deep --
// measured at M = 1024 (one box, in the model): down (K = 8192) 87.6 -> 53.7 us with S = 4; o (K = 2048) 29.8 -> 29.7 with S = 4
// and 31.4 -> 40.0 at M = 2048 with S = 2: a 512-deep slice is all pipeline fill, and the reduce kernel costs what the split saves
static int splitk_slices(int M, int N, int K) {
  const long wg = (long)((M + 127) / 128) * (N / 128);
  int S = 1;
  while (S < 8 && wg * S * 2 <= 512 && K % (S * 2 * 64) == 0 && K / (S * 2) >= 2048) S *= 2;
  return S;
}

}  // namespace tcavt

using namespace tcavt;

extern "C" int tcavt_pack_weight16(const void* W, int64_t ldw, void* out, int N, int K, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(W && out && W != out && N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0 && ldw >= K && ldw % 8 == 0,
                  "pack_weight16: N %% 16 == 0, K %% 32 == 0, ldw >= K, ldw %% 8 == 0, out != W");
  TCAVT_CHECK_ARG(aligned16(W) && aligned16(out), "pack_weight16: 16-byte alignment required");
  const long pieces = (long)N * K / 8;
  hipLaunchKernelGGL(pack_weight16_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const bf16_t*>(W), (long)ldw, static_cast<bf16_t*>(out), N, K);
  TCAVT_CHECK_LAUNCH("pack_weight16");
  return TCAVT_OK;
}

extern "C" int tcavt_gemm_bf16(const tcavt_gemm_args* a, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(a != nullptr, "gemm_bf16: null args");
  const bool stream16 = (a->epilogue & TCAVT_EPI_NORM_OUT) && a->C == nullptr;  // 16-bit residual stream in norm_h16
  TCAVT_CHECK_ARG(a->A && a->W && (a->C || stream16), "gemm_bf16: null A/W/C");
  TCAVT_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "gemm_bf16: bad M/N/K %d/%d/%d", a->M, a->N, a->K);
  TCAVT_CHECK_ARG(a->K % 64 == 0, "gemm_bf16: K=%d must be a multiple of 64", a->K);
  TCAVT_CHECK_ARG(a->K < (1 << 26), "gemm_bf16: K too large");
  TCAVT_CHECK_ARG(a->N % 16 == 0, "gemm_bf16: N=%d must be a multiple of 16", a->N);
  TCAVT_CHECK_ARG(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->lda >= a->K && a->ldw >= a->K,
                  "gemm_bf16: lda/ldw must be >= K and multiples of 8");
  TCAVT_CHECK_ARG(aligned16(a->A) && aligned16(a->W) && aligned16(a->C), "gemm_bf16: A/W/C must be 16-byte aligned");
  TCAVT_CHECK_ARG(a->out_dtype == TCAVT_F32 || a->out_dtype == TCAVT_BF16 || a->out_dtype == TCAVT_F16,
                  "gemm_bf16: bad out_dtype");
  TCAVT_CHECK_ARG(a->in_dtype == 0 || a->in_dtype == TCAVT_BF16 || a->in_dtype == TCAVT_F16, "gemm_bf16: bad in_dtype");
  const bool f16 = a->in_dtype == TCAVT_F16;
  const int batch = a->batch > 1 ? a->batch : 1;
  if (batch > 1)
    TCAVT_CHECK_ARG(!(a->epilogue & (TCAVT_EPI_SILU_MUL | TCAVT_EPI_ROPE)),
                    "gemm_bf16: batching is supported by the generic epilogue only");
  if (batch > 1) {
    TCAVT_CHECK_ARG(a->batch_inner >= 1 && batch % a->batch_inner == 0, "gemm_bf16: batch must be a multiple of batch_inner");
    TCAVT_CHECK_ARG(a->sAo % 8 == 0 && a->sAi % 8 == 0 && a->sWo % 8 == 0 && a->sWi % 8 == 0 && a->sCo % 4 == 0 &&
                        a->sCi % 4 == 0 && batch <= 65535,
                    "gemm_bf16: batch strides must keep 16-byte (A/W) and 4-element (C) alignment; batch <= 65535");
  }
  const int n_out = (a->epilogue & TCAVT_EPI_SILU_MUL) ? a->N / 2 : a->N;
  TCAVT_CHECK_ARG(a->ldc >= n_out && a->ldc % 4 == 0, "gemm_bf16: ldc=%ld too small or not a multiple of 4", (long)a->ldc);
  int K2 = 0;
  const bool lp_consumer = a->lora_part && (a->epilogue & TCAVT_EPI_ROPE);  // (W2 without A2: checked with lora_part below)
  if ((a->A2 || a->W2 || a->K2) && !lp_consumer) {
    TCAVT_CHECK_ARG(a->A2 && a->W2 && a->K2 > 0 && a->K2 % 64 == 0,
                    "gemm_bf16: second K-source needs A2, W2 and K2 %% 64 == 0");
    TCAVT_CHECK_ARG(a->lda2 % 8 == 0 && a->ldw2 % 8 == 0 && a->lda2 >= a->K2 && a->ldw2 >= a->K2 &&
                        aligned16(a->A2) && aligned16(a->W2),
                    "gemm_bf16: bad lda2/ldw2/alignment");
    K2 = a->K2;
  }
  int epi = a->epilogue;
  if (epi & TCAVT_EPI_BIAS) TCAVT_CHECK_ARG(a->bias && aligned16(a->bias), "gemm_bf16: BIAS needs an aligned bias pointer");
  if (epi & TCAVT_EPI_BIAS_ROW) TCAVT_CHECK_ARG(a->bias && !(epi & TCAVT_EPI_BIAS), "gemm_bf16: BIAS_ROW needs bias and excludes BIAS");
  if ((epi & TCAVT_EPI_RESIDUAL) && !stream16)  // (stream16: the residual is norm_h16 itself)
    TCAVT_CHECK_ARG(a->residual && aligned16(a->residual) && a->ldr >= a->N && a->ldr % 4 == 0,
                    "gemm_bf16: RESIDUAL needs residual pointer and ldr >= N");
  if (epi & TCAVT_EPI_SILU_MUL)
    TCAVT_CHECK_ARG(a->N % 128 == 0 && !(epi & ~(TCAVT_EPI_SILU_MUL | TCAVT_EPI_ROWSCALE)),
                    "gemm_bf16: SILU_MUL needs N %% 128 == 0 and no other flag but ROWSCALE");
  if (epi & TCAVT_EPI_ROPE) {
    TCAVT_CHECK_ARG(a->N % 128 == 0 && !(epi & ~(TCAVT_EPI_ROPE | TCAVT_EPI_ROWSCALE)),
                    "gemm_bf16: ROPE needs N %% 128 == 0 and no other flag but ROWSCALE");
    TCAVT_CHECK_ARG(a->rope_cos && a->rope_sin && a->rope_L > 0 && a->rope_cols % 64 == 0 &&
                        aligned16(a->rope_cos) && aligned16(a->rope_sin),
                    "gemm_bf16: ROPE needs cos/sin tables, rope_L > 0, rope_cols %% 64 == 0");
  }
#ifdef TCAVT_EXPERIMENTS
  TCAVT_CHECK_ARG(a->tile == 0 || a->tile == 64 || a->tile == 128 || a->tile == 256 || (a->tile >= 250 && a->tile <= 274 && a->tile != 251 && a->tile != 254) || (a->tile >= 124 && a->tile <= 127),
                  "gemm_bf16: tile must be 0 (auto), 128 or 256 (or an A/B code: 250, 252, 253, 255, 126, 127)");
#else
  TCAVT_CHECK_ARG(a->tile == 0 || a->tile == 64 || a->tile == 128 || a->tile == 256 || a->tile == 257 || a->tile == 271 || a->tile == 272,
                  "gemm_bf16: tile must be 0 (auto), 64, 128, 256, 257, 271 or 272 (A/B and timing-experiment codes exist in the "
                  "experiments build only: python -m tcavt_amd.build --experiments)");
#endif

  GemmP p;
  p.A = static_cast<const bf16_t*>(a->A);
  p.W = static_cast<const bf16_t*>(a->W);
  p.A2 = static_cast<const bf16_t*>(a->A2);
  p.W2 = static_cast<const bf16_t*>(a->W2);
  p.C = a->C;
  p.bias = a->bias;
  p.residual = a->residual;
  p.cosT = a->rope_cos;
  p.sinT = a->rope_sin;
  p.lda = a->lda; p.ldw = a->ldw; p.lda2 = a->lda2; p.ldw2 = a->ldw2; p.ldc = a->ldc; p.ldr = a->ldr;
  p.M = a->M; p.N = a->N; p.K = a->K; p.K2 = K2;
  p.out_kind = a->out_dtype;
  p.batch_inner = batch > 1 ? a->batch_inner : 1;
  p.w_group = a->batch_w_group > 1 ? a->batch_w_group : 1;
  p.sAo = a->sAo; p.sAi = a->sAi; p.sWo = a->sWo; p.sWi = a->sWi; p.sCo = a->sCo; p.sCi = a->sCi;
  p.flags = epi;
  p.rope_L = a->rope_L; p.rope_cols = a->rope_cols;
  p.tiles_m = p.tiles_n = 0;
  p.prio = 0;
  p.pers_tiles = 0;
  p.xcd_gx = 8;
  p.norm_h16 = nullptr;
  p.norm_part = nullptr;
  p.res16 = nullptr;
  p.lp_part = nullptr;
  p.lp_a = nullptr;
  p.lp_lda = 0;
  p.lp_np = 0;
  p.lp_scale = 1.f;
  p.w_frag = p.a_frag = p.o_frag = 0;
  if (a->act_layout != 0) {
    TCAVT_CHECK_ARG((a->act_layout & ~(TCAVT_ACT_A_FRAG16 | TCAVT_ACT_OUT_FRAG16 | TCAVT_ACT_BLOCK8)) == 0 && a->tile == 0 &&
                        skinny_shape(a->M, a->K) && batch == 1 && a->dropout_p == 0.f,
                    "gemm_bf16: act_layout (fragment-major activations) goes with the skinny form only (M <= 32, K %% 256 == 0, tile 0)");
    const int fmode = (a->act_layout & TCAVT_ACT_BLOCK8) ? 2 : 1;
    TCAVT_CHECK_ARG(fmode == 1 || a->M <= 8, "gemm_bf16: TCAVT_ACT_BLOCK8 holds at most 8 rows");
    p.a_frag = (a->act_layout & TCAVT_ACT_A_FRAG16) ? fmode : 0;
    if (a->act_layout & TCAVT_ACT_OUT_FRAG16) {
      const bool silu = (a->epilogue & ~TCAVT_EPI_ROWSCALE) == TCAVT_EPI_SILU_MUL && a->out_dtype == (f16 ? TCAVT_F16 : TCAVT_BF16) &&
                        a->N % 64 == 0;
      const bool stream = (a->epilogue & TCAVT_EPI_NORM_OUT) && stream16 && a->N % 32 == 0;
      TCAVT_CHECK_ARG(silu || stream, "gemm_bf16: TCAVT_ACT_OUT_FRAG16 needs SILU_MUL with a 16-bit output of the operand type, or "
                                      "NORM_OUT with C == NULL (the in-place 16-bit stream)");
      p.o_frag = fmode;
    }
  }
  if (a->w_layout != 0) {
    TCAVT_CHECK_ARG(a->w_layout == TCAVT_W_FRAG16 && a->tile == 0 && skinny_shape(a->M, a->K) && batch == 1 && a->dropout_p == 0.f &&
                        a->lda >= a->K && a->ldw == a->K && a->N % 16 == 0,
                    "gemm_bf16: w_layout = TCAVT_W_FRAG16 (tcavt_pack_weight16 copy) goes with the skinny form only (M <= 32, "
                    "K %% 256 == 0, N %% 16 == 0, tile 0, ldw == K)");
    p.w_frag = 1;
  }
  if (a->lora_part) {
    // (decode step only: the skinny form; anything else is a caller error rather than a silent no-op)
    TCAVT_CHECK_ARG(a->tile == 0 && skinny_shape(a->M, a->K) && batch == 1 && a->dropout_p == 0.f && aligned16(a->lora_part),
                    "gemm_bf16: lora_part goes with the skinny form only (M <= 32, tile 0)");
    p.lp_part = static_cast<float*>(a->lora_part);
    if (epi & TCAVT_EPI_NORM_OUT) {
      TCAVT_CHECK_ARG(a->lora_part_a && a->lora_part_lda >= a->N && a->lora_part_lda % 4 == 0 && ((uintptr_t)a->lora_part_a & 7) == 0,
                      "gemm_bf16: lora_part with NORM_OUT needs lora_part_a [32, lda >= N]");
      p.lp_a = static_cast<const bf16_t*>(a->lora_part_a);
      p.lp_lda = a->lora_part_lda;
    } else {
      TCAVT_CHECK_ARG((epi & TCAVT_EPI_ROPE) && a->lora_part_np > 0 && a->W2 && !a->A2 && a->K2 == 0 && a->ldw2 >= 32 && a->ldw2 % 8 == 0 &&
                          aligned16(a->W2) && a->M * 16 <= 512,
                      "gemm_bf16: lora_part with ROPE needs lora_part_np > 0, W2 (ldw2 >= 32) and no A2");
      p.lp_np = a->lora_part_np;
      p.lp_scale = a->lora_part_scale;
      p.W2 = static_cast<const bf16_t*>(a->W2);
      p.ldw2 = a->ldw2;
    }
  }
  p.sk_msplit = 1;
  p.sk_split = 1;
  p.sk_slab = nullptr;
  p.sk_cnt = nullptr;
  p.sk_slab_bytes = 0;
  p.sk_cnt_n = 0;
  if (a->splitk_ws && a->splitk_ws_bytes >= (64 << 10)) {  // [0, 16 KiB): 4096 tickets; the rest: slabs
    TCAVT_CHECK_ARG(aligned16(a->splitk_ws), "gemm_bf16: splitk_ws must be 16-byte aligned");
    p.sk_cnt = static_cast<int*>(a->splitk_ws);
    p.sk_cnt_n = 4096;
    p.sk_slab = reinterpret_cast<float*>(static_cast<char*>(a->splitk_ws) + (16 << 10));
    p.sk_slab_bytes = a->splitk_ws_bytes - (16 << 10);
  }
  p.nf_flag = (epi & TCAVT_EPI_NORM_OUT) ? a->nonfinite_flag : nullptr;
  p.nf_tag = a->nonfinite_tag;
  p.norm_scale = 1.f;
  if (epi & TCAVT_EPI_NORM_OUT) {
    TCAVT_CHECK_ARG(a->norm_scale >= 0.f && a->norm_scale <= 1.f, "gemm_bf16: norm_scale must be in (0, 1] (0 means 1)");
    if (a->norm_scale != 0.f) p.norm_scale = a->norm_scale;
  }
  p.rs_part = nullptr;
  p.rope_pos = (epi & TCAVT_EPI_ROPE) ? a->rope_pos : nullptr;
  p.rs_npart = 0;
  p.rs_eps = p.rs_inv_h = 0.f;
  if (epi & TCAVT_EPI_NORM_OUT) {
    TCAVT_CHECK_ARG(!(epi & ~(TCAVT_EPI_NORM_OUT | TCAVT_EPI_RESIDUAL)) && a->out_dtype == TCAVT_F32 && batch == 1 && K2 == 0 &&
                        a->N % 64 == 0 && a->norm_h16 && a->norm_part && aligned16(a->norm_h16) && a->dropout_p == 0.f &&
                        (a->acc_scale == 0.f || a->acc_scale == 1.f) && a->tile != 64,
                    "gemm_bf16: NORM_OUT goes with an fp32 output (+ RESIDUAL) only, N %% 64 == 0, and needs norm_h16 / norm_part");
    TCAVT_CHECK_ARG(!stream16 || a->residual == nullptr,
                    "gemm_bf16: NORM_OUT with C == NULL keeps the residual stream in norm_h16 (updated in place): residual must be NULL");
    p.norm_h16 = static_cast<bf16_t*>(a->norm_h16);
    p.norm_part = a->norm_part;
    TCAVT_CHECK_ARG(a->norm_res16 == nullptr || (stream16 && aligned16(a->norm_res16)),
                    "gemm_bf16: norm_res16 goes with the 16-bit residual stream (NORM_OUT, C == NULL) and needs 16-byte alignment");
    p.res16 = a->norm_res16 ? static_cast<const bf16_t*>(a->norm_res16) : p.norm_h16;
  }
  if (epi & TCAVT_EPI_ROWSCALE) {
    TCAVT_CHECK_ARG((epi & (TCAVT_EPI_SILU_MUL | TCAVT_EPI_ROPE)) && a->rowscale_part && aligned16(a->rowscale_part) &&
                        a->rowscale_npart > 0 && a->rowscale_npart % 4 == 0 && a->rowscale_h > 0,
                    "gemm_bf16: ROWSCALE goes with the ROPE / SILU_MUL epilogues and needs rowscale_part, npart %% 4 == 0, h > 0");
    p.rs_part = a->rowscale_part;
    p.rs_npart = a->rowscale_npart;
    p.rs_eps = a->rowscale_eps;
    p.rs_inv_h = 1.f / (float)a->rowscale_h;
  }
  epi &= ~TCAVT_EPI_ROWSCALE;  // (carried by p.rs_part from here on)
  TCAVT_CHECK_ARG(a->dropout_p >= 0.f && a->dropout_p < 1.f, "gemm_bf16: dropout_p must be in [0, 1)");
  if (a->dropout_p > 0.f)
    TCAVT_CHECK_ARG(!(epi & (TCAVT_EPI_SILU_MUL | TCAVT_EPI_ROPE)) && batch == 1 && a->N % 4 == 0,
                    "gemm_bf16: dropout is supported by the un-batched generic epilogue only");
  p.drop = make_dropout(a->dropout_p, a->dropout_seed, a->dropout_site);
  p.acc_scale = a->acc_scale == 0.f ? 1.f : a->acc_scale;
  if (epi & (TCAVT_EPI_SILU_MUL | TCAVT_EPI_ROPE))
    TCAVT_CHECK_ARG(p.acc_scale == 1.f, "gemm_bf16: acc_scale is only supported by the generic epilogue");

  hipStream_t s = static_cast<hipStream_t>(stream);
  // ---- two-launch split K (splitk_norm16_kernel above): the in-place 16-bit residual form on a grid that leaves CUs idle
  static const bool no_splitk2 = getenv("TCAVT_GEMM_NO_SPLITK2") != nullptr;  // (A/B switch)
  if (a->tile == 0 && stream16 && !no_splitk2 && p.sk_slab && a->M > 32 && !skinny_shape(a->M, a->K) && batch == 1 && K2 == 0 &&
      a->N % 128 == 0 && a->ldc % 8 == 0 && !a->lora_part && a->dropout_p == 0.f) {
    const int S = splitk_slices(a->M, a->N, a->K);
    if (S > 1 && (long)S * a->M * a->N * 4 <= p.sk_slab_bytes) {
      GemmP q = p;
      q.C = p.sk_slab; q.ldc = a->N; q.out_kind = TCAVT_F32; q.flags = 0; q.K = a->K / S;
      q.norm_h16 = nullptr; q.norm_part = nullptr; q.res16 = nullptr; q.nf_flag = nullptr; q.residual = nullptr;
      q.batch_inner = S; q.w_group = 1;
      q.sAo = 0; q.sAi = a->K / S; q.sWo = 0; q.sWi = a->K / S; q.sCo = 0; q.sCi = (long)a->M * a->N;
      const int rc = f16 ? dispatch_tile<EPI_GENERIC, true>(q, 0, S, s) : dispatch_tile<EPI_GENERIC, false>(q, 0, S, s);
      if (rc != TCAVT_OK) return rc;
      const long threads = (long)a->M * (a->N / 8);
      auto kfn = f16 ? splitk_norm16_kernel<true> : splitk_norm16_kernel<false>;
      hipLaunchKernelGGL(kfn, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, p.sk_slab, S, (long)a->M * a->N, p.res16, p.norm_h16,
                         p.norm_part, a->M, a->N, (long)a->ldc, a->N >> 6, p.norm_scale, (epi & TCAVT_EPI_RESIDUAL) ? 1 : 0, p.nf_flag, p.nf_tag);
      TCAVT_CHECK_LAUNCH("gemm_bf16(split-K reduce)");
      return TCAVT_OK;
    }
  }
  // ---- skinny form: M <= 32 rows (decode step), auto-selected only (tile == 0); norm_out_npart() mirrors this rule
  if (a->tile == 0 && skinny_shape(a->M, a->K) && batch == 1 && a->dropout_p == 0.f && a->lda >= a->K) {
    const int e0 = a->epilogue & ~TCAVT_EPI_ROWSCALE;
    if (e0 == TCAVT_EPI_ROPE && a->N % 64 == 0 && (K2 == 0 || K2 % 32 == 0))
      return f16 ? launch_skinny<EPI_ROPE, 2, true>(p, s) : launch_skinny<EPI_ROPE, 2, false>(p, s);
    if (e0 == TCAVT_EPI_SILU_MUL && !a->silu_preact && a->N % 32 == 0)
      return f16 ? launch_skinny<EPI_SILU, 2, true>(p, s) : launch_skinny<EPI_SILU, 2, false>(p, s);
    if ((e0 & TCAVT_EPI_NORM_OUT) && a->N % 16 == 0) {
      if (stream16) return f16 ? launch_skinny<EPI_NORM16, 1, true>(p, s) : launch_skinny<EPI_NORM16, 1, false>(p, s);
      return f16 ? launch_skinny<EPI_NORM, 1, true>(p, s) : launch_skinny<EPI_NORM, 1, false>(p, s);
    }
    if (e0 == 0 && K2 == 0 && a->N % 16 == 0) {
      // (lm_head with more than 16 rows: two column blocks per workgroup share the activation fragments -- with 16 columns
      //  the 32 activation rows are twice the weight bytes of every step, and the launch is bound by the CUs' load paths)
      if (a->M > 16 && a->N % 256 == 0 && a->N >= 8192)
        return f16 ? launch_skinny<EPI_GENERIC, 2, true>(p, s) : launch_skinny<EPI_GENERIC, 2, false>(p, s);
      return f16 ? launch_skinny<EPI_GENERIC, 1, true>(p, s) : launch_skinny<EPI_GENERIC, 1, false>(p, s);
    }
  }
  TCAVT_CHECK_ARG(!p.w_frag && !p.a_frag && !p.o_frag, "gemm_bf16: w_layout / act_layout: this epilogue / shape has no skinny form");
  if (epi & TCAVT_EPI_SILU_BWD) {  // dgrad of down_proj with d(silu(gate) * up) in the epilogue: the 4-wave kernel only
    TCAVT_CHECK_ARG(epi == TCAVT_EPI_SILU_BWD && batch == 1 && K2 == 0 && a->M % 256 == 0 && a->N % 256 == 0 && a->K >= 128 &&
                        a->silu_preact && aligned16(a->silu_preact) && a->dropout_p == 0.f && p.acc_scale == 1.f &&
                        a->out_dtype == (f16 ? TCAVT_F16 : TCAVT_BF16) && (a->tile == 0 || a->tile == 257),
                    "gemm_bf16: SILU_BWD runs on whole 256x256 tiles with silu_preact, a 16-bit output of the operand type and no other flag");
    p.aux = static_cast<bf16_t*>(a->silu_preact);
    p.ldaux = a->ld_preact;
    return f16 ? launch_w4<EPI_SILUBWD, 2, 0, false, 256, true>(p, s) : launch_w4<EPI_SILUBWD, 2, 0, false, 256, false>(p, s);
  }
  int tile = a->tile;
  if (tile == 0) {
    // 256x256 tiles when whole waves of 256 workgroups stay >= 75 % full (the fused q|k|v projection,
    // 32 x 12 = 384 tiles = 1.5 waves, is the boundary case: 111 us on 256x256 vs 114 us on 128x128).
    const long t256 = (long)((a->M + 255) / 256) * ((a->N + 255) / 256) * batch;
    const long waves = (t256 + 255) / 256;
    tile = (t256 >= 256 && (double)t256 / (double)(waves * 256) >= 0.75) ? 256 : 0;  // 0: dispatch_tile picks 128 / 64
    // whole 256x256 tiles, one K source, bf16, no RoPE: the 4-wave kernel (gate|up 406 vs 434 us, down 204 vs 218,
    // o 57.5 vs 60 on the 8-wave kernel)
    const bool silu_ok = (!(epi & TCAVT_EPI_SILU_MUL) || (a->out_dtype == (f16 ? TCAVT_F16 : TCAVT_BF16) && a->ldc % 8 == 0)) &&
                         (!(stream16 || (epi & TCAVT_EPI_ROPE)) || a->ldc % 8 == 0);  // (16-byte epilogue accesses of the 4-wave kernel)
    if (tile == 256 && batch == 1 && a->M % 256 == 0 && a->N % 256 == 0 && silu_ok &&
        ((epi & TCAVT_EPI_ROPE) ? a->out_dtype == (f16 ? TCAVT_F16 : TCAVT_BF16) : K2 == 0)) {
      tile = 257;
      // 256 x 192 tiles where they fill whole waves of 256 CUs and 256 x 256 tiles do not (q|k|v: N = 3072)
      if (a->N % 192 == 0) {
        const long t192 = (long)(a->M / 256) * (a->N / 192);
        const double f256 = (double)t256 / (double)(waves * 256), f192 = (double)t192 / (double)((t192 + 255) / 256 * 256);
        if (f192 > f256 + 0.1) tile = 271;
      }
      // long K (down projection, K = 8192): the two-barrier form with a whole K-tile of DMA in flight (190 vs 200 us);
      // neutral to slightly worse at K = 2048
      // (TCAVT_GEMM_DEEP_MINK=<K>: A/B switch for the threshold)
      static const int deep_mink = [] { const char* e = getenv("TCAVT_GEMM_DEEP_MINK"); return e ? atoi(e) : 4096; }();
      if (tile == 257 && !(epi & TCAVT_EPI_ROPE) && a->K >= deep_mink) tile = 272;
    }
  }
  if (epi & TCAVT_EPI_SILU_MUL) {
    if (a->silu_preact) {
      TCAVT_CHECK_ARG(aligned16(a->silu_preact) && a->ld_preact >= a->N && a->ld_preact % 4 == 0,
                      "gemm_bf16: silu_preact needs 16-byte alignment and ld_preact >= N, %% 4 == 0");
      p.aux = static_cast<bf16_t*>(a->silu_preact);
      p.ldaux = a->ld_preact;
      return f16 ? dispatch_tile<EPI_SILU_SAVE, true>(p, tile, 1, s) : dispatch_tile<EPI_SILU_SAVE, false>(p, tile, 1, s);
    }
    return f16 ? dispatch_tile<EPI_SILU, true>(p, tile, 1, s) : dispatch_tile<EPI_SILU, false>(p, tile, 1, s);
  }
  if (epi & TCAVT_EPI_ROPE) return f16 ? dispatch_tile<EPI_ROPE, true>(p, tile, 1, s) : dispatch_tile<EPI_ROPE, false>(p, tile, 1, s);
  if ((epi & TCAVT_EPI_NORM_OUT) && stream16)
    return f16 ? dispatch_tile<EPI_NORM16, true>(p, tile, 1, s) : dispatch_tile<EPI_NORM16, false>(p, tile, 1, s);
  if (epi & TCAVT_EPI_NORM_OUT) return f16 ? dispatch_tile<EPI_NORM, true>(p, tile, 1, s) : dispatch_tile<EPI_NORM, false>(p, tile, 1, s);
  if (a->dropout_p > 0.f) {  // small layers only (Q-Former, polygon encoder, LTSF): one 128x128 variant
    const int t = a->tile == 64 || a->tile == 128 ? a->tile : 0;
    return f16 ? launch_small<EPI_DROP, true>(p, t, 1, s) : launch_small<EPI_DROP, false>(p, t, 1, s);
  }
  if (f16) return dispatch_tile<EPI_GENERIC, true>(p, tile, batch, s);
  return dispatch_tile<EPI_GENERIC, false>(p, tile, batch, s);
}
