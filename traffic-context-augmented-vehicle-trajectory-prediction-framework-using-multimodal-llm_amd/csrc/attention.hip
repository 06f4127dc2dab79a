// Attention kernels for gfx950.
//
// (1) attn_causal_gqa_kernel -- the Llama decoder attention (SURVEY.md row F4;
//     HF modeling_llama.py:191-213 with the sdpa causal AND key-valid mask).
//     One workgroup per (sample, kv head); its 2 * `group` waves cover the query heads
//     that share that kv head (two waves per head, alternating 32-query blocks), so K and V
//     are fetched from HBM exactly once per (sample, kv head) and live in LDS for the
//     whole workgroup:
//       K   [Lp][64]  bf16, 128-byte rows, 16-byte chunks XOR-swizzled by (row>>1)&7
//       V^T [64][Lp+4] bf16 (transposed while staging so the PV product reads
//                           k-contiguous 8-byte pieces; +4 pad => conflict-free)
//     Per 32-query block a wave walks the 32-key tiles up to the diagonal:
//       S^T = K . Q^T        v_mfma_f32_32x32x16_bf16, key on the accumulator
//                            row, query on the lane -> row softmax is lane-local
//       O^T += V^T . P^T     v_mfma_f32_32x32x16_f16: the S^T accumulator, converted
//                            in place, IS the B operand (no LDS round trip for P).
//                            P lives in [0,1], so it is carried as fp16 (11-bit
//                            significand) rather than bf16 (8-bit): bf16 P alone costs
//                            1.3e-3 relative on a decoder layer's output.  V (bf16 from
//                            the QKV epilogue) converts to fp16 exactly (clamped to the
//                            fp16 range) while it is transposed into LDS.
//     Online softmax in fp32 (exp2 with log2e folded into the scale).
//
// (2) mha_small_kernel -- generic fp32-softmax multi-head attention for the short
//     sequences of the Q-Former, the lane-polygon encoder, the LTSF block and the
//     head_dim-1024 cross-attention (scripts/train.py:359,403,406,663,754).
#include "common.hpp"
#include "philox.hpp"

namespace tcavt {

__device__ __forceinline__ f16x8 cvt8_f16(const float* v) {
  f16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = static_cast<_Float16>(v[i]);  // RNE
  return o;
}

// the two half-waves' copies of x (lanes l and l ^ 32) in every lane, through v_permlane32_swap (gfx950): a register
// permute, where __shfl_xor(x, 32) is a ds_bpermute round trip through the LDS crossbar with an lgkmcnt wait
__device__ __forceinline__ void halves(float x, float& lo, float& hi) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  lo = __uint_as_float(r[0]);
  hi = __uint_as_float(r[1]);
}

// two bf16 (same feature, keys r0 and r1) -> packed fp16 pair, clamped to the fp16 range
__device__ __forceinline__ unsigned int bf16pair_to_f16pair(unsigned int k0_bits, unsigned int k1_bits) {
  const float a = fminf(fmaxf(__uint_as_float(k0_bits << 16), -65504.f), 65504.f);
  const float b = fminf(fmaxf(__uint_as_float(k1_bits << 16), -65504.f), 65504.f);
  const unsigned short ha = __builtin_bit_cast(unsigned short, static_cast<_Float16>(a));
  const unsigned short hb = __builtin_bit_cast(unsigned short, static_cast<_Float16>(b));
  return static_cast<unsigned int>(ha) | (static_cast<unsigned int>(hb) << 16);
}

// MAXT = launch bound: 512 threads (GQA groups up to 4, the Llama-3.2-1B case) leaves the compiler 256 VGPRs per
// lane -- with the 1024-thread bound (groups up to 8) the kernel is held to 128 and spills.
// F16: q|k|v and the output are fp16 (the forward path's default storage): S^T = K . Q^T runs on the f16 MFMA and V needs
// no conversion while it is transposed into LDS.
// STAMP (experiments build, tools/attn_stamps.py): s_memtime at the phase boundaries of every wave, written to `stamps`
// ([workgroup][wave][16] uint64) -- a buffer nothing else reads; the product instantiation has no such code.
template <int MAXT, bool F16, bool STAMP = false>
__global__ __launch_bounds__(MAXT) void attn_causal_gqa_kernel(const bf16_t* __restrict__ qkv,
                                                              bf16_t* __restrict__ out,
                                                              const int* __restrict__ kv_len_p,
                                                              int L, int Lp, int nq, int nkv,
                                                              float scale_log2e, int ot_bytes,
                                                              unsigned long long* __restrict__ stamps = nullptr,
                                                              float* __restrict__ lse = nullptr) {
  int n_stamp = 0;
  auto stamp = [&]() {
    if constexpr (STAMP) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if ((threadIdx.x & 63) == 0 && n_stamp < 16) stamps[((long)blockIdx.x * (MAXT / 64) + (threadIdx.x >> 6)) * 16 + n_stamp] = t;
      ++n_stamp;
    }
  };
  stamp();  // 0: entry
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int group = nq / nkv;
  const int b = blockIdx.x / nkv, kvh = blockIdx.x % nkv;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ld = (nq + 2 * nkv) * 64;
  const int koff = nq * 64 + kvh * 64, voff = (nq + nkv) * 64 + kvh * 64;
  const int vstride = Lp + 4;  // elements
  char* Ks = smem;
  bf16_t* Vt = reinterpret_cast<bf16_t*>(smem + Lp * 128);
  const int ot_off = (Lp * 128 + 64 * (Lp + 4) * 2 + 15) & ~15;  // output tiles of the waves (ot_bytes > 0), behind K and V^T
  const bf16_t* base = qkv + (long)b * L * ld;

  // ---- this wave's query blocks.  2 * group waves: wave w serves query head w % group and the 32-query blocks of parity
  // w / group, so every SIMD holds two waves whose MFMA / softmax / LDS phases interleave (one wave per SIMD left the matrix
  // pipe idle during every softmax).  Query blocks are dealt to the two waves of a head in PAIRS (k, nqb-1-k): a short causal
  // block with a long one, every pair costs nqb + 1 key tiles, wave parity p takes pairs p, p + 2, ...  (8 blocks: {0,7,2,5}
  // and {1,6,3,4}, 18 key tiles each, instead of 16 / 20 with alternating blocks).
  const int head = kvh * group + (wave % group);
  const int qb0 = wave / group;
  const int r = lane & 31, hh = lane >> 5;
  const int nqb = (L + 31) >> 5;
  const int npair = (nqb + 1) >> 1;
  // block number bi of this wave -> query block (or -1: none): bi = 2 * (pair index) + member
  auto block_of = [&](int bi) {
    const int pr = qb0 + 2 * (bi >> 1);
    if (pr >= npair) return -1;
    const int qb = (bi & 1) ? nqb - 1 - pr : pr;
    return ((bi & 1) && qb == pr) ? -1 : qb;  // odd block count: the middle block is its own pair
  };
  auto load_q = [&](int qb, bf16x8 (&qf)[4]) {
    const int qrow = min(qb * 32 + r, L - 1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      qf[s] = *reinterpret_cast<const bf16x8*>(base + (long)qrow * ld + head * 64 + s * 16 + hh * 8);
  };
  // The whole problem of a workgroup is 192 KB in, 128 KB out and ~7 us of arithmetic on a CU that has nothing else to
  // do: it is a streaming problem.  Everything a wave will read is requested up front -- the query rows of its first NPRE
  // blocks (16 VGPRs each) BEFORE the K / V staging loads -- so the memory latency is paid once, not once per block (loading
  // each block's queries at its start left the wave waiting ~2 us per block; L <= 256 has at most 4 blocks per wave).
  constexpr int NPRE = 4;
  bf16x8 qpre[NPRE][4];
#pragma unroll
  for (int bi = 0; bi < NPRE; ++bi) {
    const int qb = block_of(bi);
    load_q(qb < 0 ? 0 : qb, qpre[bi]);  // (uniform; a block that does not exist loads block 0's rows and is never used)
  }

  // ---- stage K (swizzled rows) and V^T.  The loads of the first KIT / VIT passes over the rows (all of them for
  // L <= 256 with 512 threads) are issued together and written to LDS afterwards: written load-then-store per pass, every
  // pass waited for its own memory round trip -- six in a row, a third of the kernel's time.
  const int nthreads = blockDim.x;
  constexpr int KIT = 4, VIT = 2;
  auto stage_k = [&](int idx, const u32x4& v) {
    const int row = idx >> 3, c = idx & 7;
    *reinterpret_cast<u32x4*>(Ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = v;
  };
  auto stage_v = [&](int idx, const u32x4& a, const u32x4& bb) {
    const int kp = idx >> 3, c = idx & 7;
    const int r0 = 2 * kp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned int lo, hi;
      if constexpr (F16) {
        lo = (a[e] & 0xffffu) | (bb[e] << 16);
        hi = (a[e] >> 16) | (bb[e] & 0xffff0000u);
      } else {
        lo = bf16pair_to_f16pair(a[e] & 0xffffu, bb[e] & 0xffffu);
        hi = bf16pair_to_f16pair(a[e] >> 16, bb[e] >> 16);
      }
      *reinterpret_cast<unsigned int*>(Vt + (c * 8 + 2 * e) * vstride + r0) = lo;
      *reinterpret_cast<unsigned int*>(Vt + (c * 8 + 2 * e + 1) * vstride + r0) = hi;
    }
  };
  // (unconditional loads from a clamped row: a "load or zero" select on a runtime condition makes the compiler branch
  // around every load and wait for it at the join; rows >= L are zeroed when they are staged)
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_k = [&](int idx) {
    const int row = min(idx >> 3, L - 1), c = idx & 7;
    return *reinterpret_cast<const u32x4*>(base + (long)row * ld + koff + c * 8);
  };
  auto load_v = [&](int idx, int which) {
    const int rr = min(2 * (idx >> 3) + which, L - 1), c = idx & 7;
    return *reinterpret_cast<const u32x4*>(base + (long)rr * ld + voff + c * 8);
  };
  auto krow_ok = [&](int idx) { return (idx >> 3) < L; };
  auto vrow_ok = [&](int idx, int which) { return 2 * (idx >> 3) + which < L; };
  {
    u32x4 kreg[KIT], va[VIT], vb[VIT];
#pragma unroll
    for (int it = 0; it < KIT; ++it) kreg[it] = load_k(threadIdx.x + it * nthreads);
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      va[it] = load_v(threadIdx.x + it * nthreads, 0);
      vb[it] = load_v(threadIdx.x + it * nthreads, 1);
    }
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int idx = threadIdx.x + it * nthreads;
      if (idx < Lp * 8) stage_k(idx, krow_ok(idx) ? kreg[it] : zero4);
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int idx = threadIdx.x + it * nthreads;
      if (idx < (Lp >> 1) * 8) stage_v(idx, vrow_ok(idx, 0) ? va[it] : zero4, vrow_ok(idx, 1) ? vb[it] : zero4);
    }
  }
  for (int idx = threadIdx.x + KIT * nthreads; idx < Lp * 8; idx += nthreads)  // (L > 256)
    stage_k(idx, krow_ok(idx) ? load_k(idx) : zero4);
  for (int idx = threadIdx.x + VIT * nthreads; idx < (Lp >> 1) * 8; idx += nthreads)
    stage_v(idx, vrow_ok(idx, 0) ? load_v(idx, 0) : zero4, vrow_ok(idx, 1) ? load_v(idx, 1) : zero4);
  stamp();  // 1: staging stores issued
  __syncthreads();
  stamp();  // 2: K / V visible

  const int kv_len = min(kv_len_p[b], L);
  const int kv_tiles = (kv_len + 31) >> 5;
  const int kswz = (r >> 1) & 7;

  auto do_block = [&](int qb, const bf16x8 (&qf)[4]) {
    const int qi = qb * 32 + r;  // this lane's query row
    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float mrun = -1e30f, lrun = 0.f;

    const int ntile = min(qb + 1, kv_tiles);
    for (int kt = 0; kt < ntile; ++kt) {
      f32x16 sacc;
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
      const char* krow = Ks + (kt * 32 + r) * 128;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + (((2 * s + hh) ^ kswz) << 4));
        if constexpr (F16)
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf), __builtin_bit_cast(f16x8, qf[s]), sacc, 0, 0, 0);
        else
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc, 0, 0, 0);
      }
      // scale + mask + tile max.  Only the diagonal tile and a tile that straddles kv_len need per-element masks
      // (wave-uniform test); every other tile is all-valid for all 32 queries of the block.
      float p[16];
      float tmax = -1e30f;
      const bool full = (kt < qb) && (kt * 32 + 32 <= kv_len);
      if (full) {
        // all 32 x 32 scores valid: running maximum over the RAW scores (the scale is positive), scale folded into the
        // exponent's fma -> max3 / fma / exp2 / add per score instead of mul / max / sub / exp2 / add
        float rmax = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
#pragma unroll
        for (int i = 3; i + 1 < 16; i += 2) rmax = fmaxf(fmaxf(rmax, sacc[i]), sacc[i + 1]);
        rmax = fmaxf(rmax, sacc[15]);
        { float a, b; halves(rmax, a, b); rmax = fmaxf(a, b); }
        const float mnew = fmaxf(mrun, rmax * scale_log2e);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          p[i] = __builtin_amdgcn_exp2f(fmaf(sacc[i], scale_log2e, -mnew));
          psum += p[i];
        }
        { float a, b; halves(psum, a, b); psum = a + b; }
        lrun = lrun * alpha + psum;
        mrun = mnew;
        // (unconditional: skipping the multiplication when no maximum moved -- a wave-uniform branch -- made the running
        // output a phi of two register sets, and the compiler paid for it with 32 v_mov_b64 per key tile, half of them
        // BETWEEN the dependent P.V MFMAs, whose latency they exposed)
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int kk = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          const bool ok = (kk <= qi) && (kk < kv_len);
          p[i] = ok ? sacc[i] * scale_log2e : -1e30f;
          tmax = fmaxf(tmax, p[i]);
        }
        { float a, b; halves(tmax, a, b); tmax = fmaxf(a, b); }
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          p[i] = (p[i] > -1e29f) ? __builtin_amdgcn_exp2f(p[i] - mnew) : 0.f;
          psum += p[i];
        }
        { float a, b; halves(psum, a, b); psum = a + b; }
        lrun = lrun * alpha + psum;
        mrun = mnew;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
      }
      // O^T += V^T . P^T  (two 16-key k-steps; P registers 8*s2.. are the B operand)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const f16x8 pf = cvt8_f16(&p[8 * s2]);
        const int kbase = kt * 32 + 16 * s2 + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16_t* vrow = Vt + (dt * 32 + r) * vstride + kbase;
          const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow);
          const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + 8);
          const u32x4 vv = {lo[0], lo[1], hi[0], hi[1]};
          const f16x8 vf = __builtin_bit_cast(f16x8, vv);
          if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o0, 0, 0, 0);
          else o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o1, 0, 0, 0);
        }
      }
    }
    // ---- normalise + store: lane holds d = dt*32 + (i&3) + 8*(i>>2) + 4*hh of query qi, i.e. per 8-dim group g the lower
    // half-wave has dims 8g .. 8g+3 and the upper 8g+4 .. 8g+7 of the same row.  v_permlane32_swap on the packed words of
    // two neighbouring groups leaves 16 consecutive bytes in every lane (lower half: group g whole, upper half: group g + 1
    // whole) -> four 16-byte stores per row instead of eight 8-byte ones (the store tail is issue-bound)
    const float inv = lrun > 0.f ? 1.f / lrun : 0.f;
    // (LoRA-trainable variant) log-sum-exp of the row's scaled scores, natural log: the backward rebuilds P = exp(s - lse)
    // from it instead of sweeping the keys once more for the row maximum and sum
    if (lse && hh == 0 && qi < L) lse[((long)b * nq + head) * L + qi] = lrun > 0.f ? (mrun + __log2f(lrun)) * 0.6931471805599453f : 0.f;
    if (ot_bytes) {
      // Through a wave-private LDS tile [32 queries][64 dims] (rows padded to 144 bytes) so that every store instruction
      // writes 8 whole 128-byte rows instead of 32-byte pieces of 32 rows: the row-per-lane form cost ~2.3k cycles per block
      // (a quarter of the compute phase) in the store path, whose price goes by row segments, not bytes.
      char* ot = smem + ot_off + wave * (32 * 144);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f32x16& o = half ? o1 : o0;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<u32x2*>(ot + r * 144 + (half * 32 + 8 * g + 4 * hh) * 2) =
              u32x2{pack16x2<F16>(o[4 * g] * inv, o[4 * g + 1] * inv), pack16x2<F16>(o[4 * g + 2] * inv, o[4 * g + 3] * inv)};
      }
      // (same wave, in-order LDS queue: the reads below see the writes above)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = (lane >> 3) + 8 * k, q = qb * 32 + row;
        const u32x4 v = *reinterpret_cast<const u32x4*>(ot + row * 144 + (lane & 7) * 16);
        if (q < L) *reinterpret_cast<u32x4*>(out + ((long)b * L + q) * (nq * 64) + head * 64 + (lane & 7) * 8) = v;
      }
      stamp();
      return;
    }
    bf16_t* orow = out + ((long)b * L + min(qi, L - 1)) * (nq * 64) + head * 64;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const f32x16& o = half ? o1 : o0;
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        unsigned int a0 = pack16x2<F16>(o[4 * g] * inv, o[4 * g + 1] * inv), a1 = pack16x2<F16>(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
        unsigned int b0 = pack16x2<F16>(o[4 * g + 4] * inv, o[4 * g + 5] * inv), b1 = pack16x2<F16>(o[4 * g + 6] * inv, o[4 * g + 7] * inv);
        auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        // lower half-wave: [own g | upper's g] = dims 8g .. 8g+7; upper: [lower's g+1 | own g+1] = dims 8(g+1) .. 8(g+1)+7
        if (qi < L) *reinterpret_cast<u32x4*>(orow + half * 32 + 8 * g + 8 * hh) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
    }
    stamp();  // 3 + block: block done (stores issued)
  };

#pragma unroll
  for (int bi = 0; bi < NPRE; ++bi) {
    const int qb = block_of(bi);
    if (qb >= 0) do_block(qb, qpre[bi]);  // (uniform)
  }
  for (int bi = NPRE;; ++bi) {  // L > 256: further blocks load their queries when they start
    if (qb0 + 2 * (bi >> 1) >= npair) break;
    const int qb = block_of(bi);
    if (qb < 0) continue;
    bf16x8 qf[4];
    load_q(qb, qf);
    do_block(qb, qf);
  }
}

// ---------------------------------------------------------------------------
// Generic small attention.  One workgroup (256 threads) per (batch, head).
// scores [Lq][Lk] fp32 live in LDS.  Phase 1: thread t owns score (i, j) pairs
// strided over Lq*Lk and walks the head dim; phase 2: one wave per query row does
// the softmax; phase 3: thread t owns (i, d) outputs strided over Lq*dh.
// K and V rows are read from global/L2 (sequences here are <= 544 keys, the
// per-(b,h) K/V footprint is <= 2 MB and is re-read from L2).
// ---------------------------------------------------------------------------
// element types of the small attention kernels: float, or 16-bit storage tagged with its format
struct b16 { bf16_t v; };  // bf16
struct h16 { bf16_t v; };  // fp16
template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <>
__device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldf<b16>(const b16* p) { return bf16_to_f32(p->v); }
template <>
__device__ __forceinline__ float ldf<h16>(const h16* p) { return f16_to_f32(p->v); }
template <typename T>
__device__ __forceinline__ void stf(T* p, float x);
template <>
__device__ __forceinline__ void stf<float>(float* p, float x) { *p = x; }
template <>
__device__ __forceinline__ void stf<b16>(b16* p, float x) { p->v = f32_to_bf16(x); }
template <>
__device__ __forceinline__ void stf<h16>(h16* p, float x) { p->v = f32_to_f16(x); }

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void mha_small_kernel(const TI* __restrict__ q, long ldq,
                                                        const TI* __restrict__ k, long ldk,
                                                        const TI* __restrict__ v, long ldv,
                                                        TO* __restrict__ out, long ldo,
                                                        const int* __restrict__ key_len, int Lq,
                                                        int Lk, int nh, int dh, float scale, DropoutP drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc = reinterpret_cast<float*>(smem);  // [Lq][Lk]
  const int b = blockIdx.x / nh, h = blockIdx.x % nh;
  const int klen = key_len ? min(key_len[b], Lk) : Lk;
  const TI* qb = q + (long)b * Lq * ldq + h * dh;
  const TI* kb = k + (long)b * Lk * ldk + h * dh;
  const TI* vb = v + (long)b * Lk * ldv + h * dh;
  const int tid = threadIdx.x;

  // phase 1: scores.  A wave handles one (i, j) at a time with lanes over d so
  // that K/Q reads are contiguous; reduction by wave_sum.
  const int lane = tid & 63, wave = tid >> 6;
  for (int ij = wave; ij < Lq * Lk; ij += 4) {
    const int i = ij / Lk, j = ij - i * Lk;
    float acc = 0.f;
    if (j < klen) {
      const TI* qr = qb + (long)i * ldq;
      const TI* kr = kb + (long)j * ldk;
      for (int d = lane; d < dh; d += 64) acc += ldf<TI>(qr + d) * ldf<TI>(kr + d);
      acc = wave_sum(acc);
    }
    if (lane == 0) sc[ij] = (j < klen) ? acc * scale : -1e30f;
  }
  __syncthreads();
  // phase 2: softmax per row (one wave per row)
  for (int i = wave; i < Lq; i += 4) {
    float* row = sc + i * Lk;
    float m = -1e30f;
    for (int j = lane; j < Lk; j += 64) m = fmaxf(m, row[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < Lk; j += 64) {
      const float e = (row[j] > -1e29f) ? __expf(row[j] - m) : 0.f;
      row[j] = e;
      s += e;
    }
    s = wave_sum(s);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    for (int j = lane; j < Lk; j += 64) {
      float pj = row[j] * inv;
      if (drop.p > 0.f)
        pj *= dropout_one(drop, ((unsigned long long)blockIdx.x * Lq + i) * (unsigned long long)Lk + j);
      row[j] = pj;
    }
  }
  __syncthreads();
  // phase 3: out[i][d] = sum_j P[i][j] V[j][d]; consecutive threads -> consecutive d
  for (int id = tid; id < Lq * dh; id += 256) {
    const int i = id / dh, d = id - i * dh;
    const float* row = sc + i * Lk;
    float acc = 0.f;
    for (int j = 0; j < klen; ++j) acc += row[j] * ldf<TI>(vb + (long)j * ldv + d);
    TO* o = out + ((long)b * Lq + i) * ldo + h * dh + d;
    stf<TO>(o, acc);
  }
}

// ---------------------------------------------------------------------------
// Same contract as mha_small_kernel for the cases whose whole (batch, head) problem fits in LDS
// (Q, K, V rows padded to dh+1 floats + the score matrix): one global read of q/k/v, everything
// else out of LDS.  Used for the lane-polygon encoder (64x64, dh 16), the Q-Former (<= 18 keys,
// dh 96) and the LTSF block (<= 30 tokens, dh 32).
// ---------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void mha_lds_kernel(const TI* __restrict__ q, long ldq,
                                                      const TI* __restrict__ k, long ldk,
                                                      const TI* __restrict__ v, long ldv,
                                                      TO* __restrict__ out, long ldo,
                                                      const int* __restrict__ key_len, int Lq, int Lk,
                                                      int nh, int dh, float scale, DropoutP drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ds = dh + 1;
  float* Qs = reinterpret_cast<float*>(smem);
  float* Ks = Qs + Lq * ds;
  float* Vs = Ks + Lk * ds;
  float* sc = Vs + Lk * ds;  // [Lq][Lk]
  const int b = blockIdx.x / nh, h = blockIdx.x % nh;
  const int klen = key_len ? min(key_len[b], Lk) : Lk;
  const TI* qb = q + (long)b * Lq * ldq + h * dh;
  const TI* kb = k + (long)b * Lk * ldk + h * dh;
  const TI* vb = v + (long)b * Lk * ldv + h * dh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int id = tid; id < Lq * dh; id += 256) {
    const int i = id / dh, d = id - i * dh;
    Qs[i * ds + d] = ldf<TI>(qb + (long)i * ldq + d);
  }
  for (int id = tid; id < Lk * dh; id += 256) {
    const int j = id / dh, d = id - j * dh;
    Ks[j * ds + d] = ldf<TI>(kb + (long)j * ldk + d);
    Vs[j * ds + d] = ldf<TI>(vb + (long)j * ldv + d);
  }
  __syncthreads();
  for (int ij = tid; ij < Lq * Lk; ij += 256) {
    const int i = ij / Lk, j = ij - i * Lk;
    float acc = 0.f;
    if (j < klen) {
      const float* qr = Qs + i * ds;
      const float* kr = Ks + j * ds;
      for (int d = 0; d < dh; ++d) acc = fmaf(qr[d], kr[d], acc);
    }
    sc[ij] = (j < klen) ? acc * scale : -1e30f;
  }
  __syncthreads();
  for (int i = wave; i < Lq; i += 4) {
    float* row = sc + i * Lk;
    float m = -1e30f;
    for (int j = lane; j < Lk; j += 64) m = fmaxf(m, row[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < Lk; j += 64) {
      const float e = (row[j] > -1e29f) ? __expf(row[j] - m) : 0.f;
      row[j] = e;
      s += e;
    }
    s = wave_sum(s);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    for (int j = lane; j < Lk; j += 64) {
      float pj = row[j] * inv;
      if (drop.p > 0.f)
        pj *= dropout_one(drop, ((unsigned long long)blockIdx.x * Lq + i) * (unsigned long long)Lk + j);
      row[j] = pj;
    }
  }
  __syncthreads();
  for (int id = tid; id < Lq * dh; id += 256) {
    const int i = id / dh, d = id - i * dh;
    const float* row = sc + i * Lk;
    float acc = 0.f;
    for (int j = 0; j < klen; ++j) acc = fmaf(row[j], Vs[j * ds + d], acc);
    TO* o = out + ((long)b * Lq + i) * ldo + h * dh + d;
    stf<TO>(o, acc);
  }
}

}  // namespace tcavt

using namespace tcavt;

extern "C" int tcavt_attn_causal_gqa(const void* qkv, void* out, const int32_t* kv_len, int B, int L,
                                     int nq, int nkv, float scale, int dtype16, tcavt_stream_t stream) {
  return tcavt_attn_causal_gqa_lse(qkv, out, nullptr, kv_len, B, L, nq, nkv, scale, dtype16, stream);
}

extern "C" int tcavt_attn_causal_gqa_lse(const void* qkv, void* out, float* lse, const int32_t* kv_len, int B, int L,
                                         int nq, int nkv, float scale, int dtype16, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(qkv && out && kv_len, "attn_causal_gqa: null pointer");
  TCAVT_CHECK_ARG(is16(dtype16), "attn_causal_gqa: dtype16 must be TCAVT_BF16 or TCAVT_F16");
  TCAVT_CHECK_ARG(B > 0 && L > 0 && L <= 544, "attn_causal_gqa: L=%d must be in [1, 544]", L);
  TCAVT_CHECK_ARG(nkv > 0 && nq % nkv == 0 && nq / nkv <= 8, "attn_causal_gqa: nq/nkv must be an integer <= 8");
  TCAVT_CHECK_ARG(aligned16(qkv) && aligned16(out), "attn_causal_gqa: unaligned pointer");
  const int Lp = (L + 31) & ~31;
  int lds = Lp * 128 + 64 * (Lp + 4) * 2;
  const int group = nq / nkv;
  const bool small = 2 * group * 64 <= 512;
  // per-wave output tiles (32 x 144 bytes) behind K and V^T when the CU's LDS has room (L <= 384 for groups of 4)
  int ot_bytes = 2 * group * 32 * 144;
  if (((lds + 15) & ~15) + ot_bytes <= 160 * 1024 - 256) lds = ((lds + 15) & ~15) + ot_bytes;
  else ot_bytes = 0;
  const bool f16 = dtype16 == TCAVT_F16;
  const void* fns[4] = {reinterpret_cast<const void*>(attn_causal_gqa_kernel<1024, false>),
                        reinterpret_cast<const void*>(attn_causal_gqa_kernel<512, false>),
                        reinterpret_cast<const void*>(attn_causal_gqa_kernel<1024, true>),
                        reinterpret_cast<const void*>(attn_causal_gqa_kernel<512, true>)};
  const int which = (f16 ? 2 : 0) + (small ? 1 : 0);
  static bool attr_set[4] = {false, false, false, false};
  if (!attr_set[which]) {
    hipError_t e = hipFuncSetAttribute(fns[which], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    if (e != hipSuccess) {
      set_error("attn_causal_gqa: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return TCAVT_ERR_HIP;
    }
    attr_set[which] = true;
  }
#define TCAVT_ATTN(MAXT, F)                                                                                       \
  hipLaunchKernelGGL((attn_causal_gqa_kernel<MAXT, F>), dim3(B * nkv), dim3(2 * group * 64), lds,                  \
                     static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(qkv), static_cast<bf16_t*>(out), \
                     kv_len, L, Lp, nq, nkv, scale * 1.4426950408889634f, ot_bytes, nullptr, lse)
  if (which == 0) TCAVT_ATTN(1024, false);
  else if (which == 1) TCAVT_ATTN(512, false);
  else if (which == 2) TCAVT_ATTN(1024, true);
  else TCAVT_ATTN(512, true);
#undef TCAVT_ATTN
  TCAVT_CHECK_LAUNCH("attn_causal_gqa");
  return TCAVT_OK;
}

extern "C" int tcavt_mha(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                         int64_t ldv, void* out, int64_t ldo, const int32_t* key_len, int B, int Lq,
                         int Lk, int nh, int dh, float scale, int in_dtype, int out_dtype, float dropout_p,
                         uint64_t dropout_seed, uint32_t dropout_site, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "mha: dropout_p must be in [0, 1)");
  const DropoutP drop = make_dropout(dropout_p, dropout_seed, dropout_site);
  TCAVT_CHECK_ARG(q && k && v && out, "mha: null pointer");
  TCAVT_CHECK_ARG(B > 0 && Lq > 0 && Lk > 0 && nh > 0 && dh > 0, "mha: bad shape");
  const long lds = (long)Lq * Lk * 4;
  TCAVT_CHECK_ARG(lds <= 64 * 1024, "mha: Lq*Lk*4 = %ld bytes exceeds the 64 KiB score buffer", lds);
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(B * nh), block(256);
  const long lds_all = ((long)(Lq + 2 * Lk) * (dh + 1) + (long)Lq * Lk) * 4;
  const bool whole = lds_all <= 60 * 1024;  // whole problem in LDS
  const long lds_use = whole ? lds_all : lds;
#define TCAVT_MHA(TI, TO)                                                                                          \
  do {                                                                                                             \
    if (whole)                                                                                                     \
      hipLaunchKernelGGL((mha_lds_kernel<TI, TO>), grid, block, lds_use, s, (const TI*)q, ldq, (const TI*)k, ldk, \
                         (const TI*)v, ldv, (TO*)out, ldo, key_len, Lq, Lk, nh, dh, scale, drop);                  \
    else                                                                                                           \
      hipLaunchKernelGGL((mha_small_kernel<TI, TO>), grid, block, lds_use, s, (const TI*)q, ldq, (const TI*)k, ldk, \
                         (const TI*)v, ldv, (TO*)out, ldo, key_len, Lq, Lk, nh, dh, scale, drop);                  \
  } while (0)
#define TCAVT_MHA_IN(TI)                                  \
  do {                                                    \
    if (out_dtype == TCAVT_F32) TCAVT_MHA(TI, float);     \
    else if (out_dtype == TCAVT_BF16) TCAVT_MHA(TI, b16); \
    else TCAVT_MHA(TI, h16);                              \
  } while (0)
  TCAVT_CHECK_ARG((in_dtype == TCAVT_F32 || is16(in_dtype)) && (out_dtype == TCAVT_F32 || is16(out_dtype)),
                  "mha: bad dtype %d/%d", in_dtype, out_dtype);
  if (in_dtype == TCAVT_F32) TCAVT_MHA_IN(float);
  else if (in_dtype == TCAVT_BF16) TCAVT_MHA_IN(b16);
  else TCAVT_MHA_IN(h16);
#undef TCAVT_MHA_IN
#undef TCAVT_MHA
  TCAVT_CHECK_LAUNCH("mha");
  return TCAVT_OK;
}

#ifdef TCAVT_EXPERIMENTS
// tools/attn_stamps.py: the fp16, group-4 instantiation with s_memtime stamps (stamps: [B * nkv][8][16] uint64)
extern "C" int tcavt_attn_causal_gqa_stamped(const void* qkv, void* out, const int32_t* kv_len, int B, int L, int nq, int nkv,
                                             float scale, unsigned long long* stamps, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(qkv && out && kv_len && stamps && nq / nkv == 4 && L <= 544, "attn_causal_gqa_stamped: bad args");
  const int Lp = (L + 31) & ~31;
  int lds = Lp * 128 + 64 * (Lp + 4) * 2;
  int ot_bytes = 8 * 32 * 144;
  if (((lds + 15) & ~15) + ot_bytes <= 160 * 1024 - 256) lds = ((lds + 15) & ~15) + ot_bytes;
  else ot_bytes = 0;
  auto kfn = attn_causal_gqa_kernel<512, true, true>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipLaunchKernelGGL(kfn, dim3(B * nkv), dim3(512), lds, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(qkv),
                     static_cast<bf16_t*>(out), kv_len, L, Lp, nq, nkv, scale * 1.4426950408889634f, ot_bytes, stamps,
                     static_cast<float*>(nullptr));
  TCAVT_CHECK_LAUNCH("attn_causal_gqa_stamped");
  return TCAVT_OK;
}
#endif
