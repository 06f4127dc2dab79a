// Autoregressive text generation on the decoder stack (reference: LlamaMultiModal.generate_batch,
// scripts/train.py:577-654, and scripts/check_generation.py:152-222).
//
// The reference hands HF `generate` a prefix of image tokens + prompt embeddings through a patched embedding layer and
// samples with temperature 0.9, top-k 40, top-p 0.9, repetition penalty 1.2 and no-repeat-3-grams (train.py:628-642).
// Here the prefix is a first-class argument: the PREFILL is the ordinary batched forward (tcavt_llama_stack_forward
// with k_cache / v_cache set), and every following token is ONE call of tcavt_llama_decode_step:
//
//   embed (table[id] + text modality embedding)  ->  16 x [ LoRA down | q|k|v + RoPE at the sample's own position |
//   attention of the one query over the cached keys (which also appends the new k, v) | o_proj | gate|up | down ]  ->
//   final RMSNorm  ->  lm_head (tied embedding table)  ->  logits processors + token selection (tcavt_sample_logits)
//
// All per-step state (positions, current tokens, token history, step counter, finished flags) lives on the device and
// is advanced by the selection kernel, so the launch sequence of a step never changes: it is captured in a hipGraph
// once and replayed per token (BASELINE.json configs[4] "hipGraph-captured decode").
// The projections reuse the MFMA GEMM (M = batch rows: weight-streaming, HBM-bound); RMSNorm is fused exactly as in the
// prefill (csrc/stack.hip).
#include "common.hpp"
#include "philox.hpp"

namespace tcavt {

// One query per (sample, query head) over the cached keys 0 .. pos[b] (the new token's own key included).
// One workgroup per (sample, query head) -- B * nq workgroups, so a batch of 8 already covers the 256 CUs; the heads of a
// GQA group re-read their kv head's rows through L2 -- with KS waves that split the keys in chunks of whole 64-key rounds.
// A round is read COALESCED: lane = (key g = lane / 8 of an 8-key row group, 16-byte column c = lane % 8), so one load
// instruction covers eight whole 128-byte rows and all eight loads of a round are in flight together (the first version gave
// a lane a whole key row -- 64 cache lines per load instruction -- and walked the values one key at a time: 26 us per layer,
// latency bound).  Scores: 8-feature partial dot products, added over the 8 lanes of a key; the KS (max, sum) pairs meet in
// LDS; the probabilities are normalised with the head's global sum and carried in fp16 as in the prefill kernel; output:
// 8 features per lane over the lane's keys, added over the 8 key groups by shuffles and over the KS waves in split order.
// The new token's own key / value (row b of qkv, position pos[b]) are read from qkv and written to the cache by the
// workgroup of the kv head's first query head: no separate append launch, and nothing written here is read back here.
// PFV: the first round of value rows is requested before the softmax statistics meet (its latency passes under the reductions
// and the barrier).  Costs 32 registers, i.e. resident waves: a gain when the grid is one workgroup per CU (B = 8: 1.138 ->
// 1.129 ms per step), a loss when several workgroups per CU hide each other's latencies anyway (B = 32: 1.45 -> 1.56).
template <bool F16, bool PFV>
__global__ __launch_bounds__(1024) void attn_decode_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ kc,
                                                           bf16_t* __restrict__ vc, const int* __restrict__ pos,
                                                           bf16_t* __restrict__ out, int lmax, int nq, int nkv, float scale,
                                                           int KS, int out_frag) {
  extern __shared__ float sc[];  // [lmax] scores | [KS][2] (max, sum) | [KS][64] partial outputs
  const int group = nq / nkv;
  const int b = blockIdx.x / nq, head = blockIdx.x % nq, kvh = head / group;
  const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int g = lane >> 3, c = lane & 7;
  const int n = min(pos[b] + 1, lmax);
  const int ld = (nq + 2 * nkv) * 64, w = nkv * 64;
  const bf16_t* kb = kc + (long)b * lmax * w + kvh * 64;
  const bf16_t* vb = vc + (long)b * lmax * w + kvh * 64;
  const int jn = n - 1;  // the new token's position
  const bf16_t* knew = qkv + (long)b * ld + (nq + kvh) * 64;
  const bf16_t* vnew = qkv + (long)b * ld + (nq + nkv + kvh) * 64;
  if (head % group == 0 && ks == 0 && lane < 16) {  // append: 8 lanes x 16 bytes of k, 8 of v
    const int cc = (lane & 7) * 8;
    if (lane < 8) *reinterpret_cast<u32x4*>(kc + ((long)b * lmax + jn) * w + kvh * 64 + cc) = *reinterpret_cast<const u32x4*>(knew + cc);
    else *reinterpret_cast<u32x4*>(vc + ((long)b * lmax + jn) * w + kvh * 64 + cc) = *reinterpret_cast<const u32x4*>(vnew + cc);
  }
  float* s = sc;
  float* st = sc + lmax;
  float* po = st + KS * 2;
  const int chunk = ((n + KS - 1) / KS + 63) / 64 * 64;  // keys per split, whole 64-key rounds
  const int j0 = ks * chunk, j1 = min(j0 + chunk, n);
  float q[8];
  {
    const u32x4 t = *reinterpret_cast<const u32x4*>(qkv + (long)b * ld + head * 64 + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      q[2 * e] = from16_lo<F16>(t[e]) * scale;
      q[2 * e + 1] = from16_hi<F16>(t[e]) * scale;
    }
  }
  float mx = -1e30f;
  for (int jb = j0; jb < j1; jb += 64) {
    u32x4 kr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = min(jb + i * 8 + g, jn);
      kr[i] = *reinterpret_cast<const u32x4*>((j == jn ? knew : kb + (long)j * w) + c * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d = fmaf(q[2 * e], from16_lo<F16>(kr[i][e]), d);
        d = fmaf(q[2 * e + 1], from16_hi<F16>(kr[i][e]), d);
      }
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      const int j = jb + i * 8 + g;
      if (j < j1) {
        if (c == 0) s[j] = d;
        mx = fmaxf(mx, d);
      }
    }
  }
  u32x4 vr0[PFV ? 8 : 1];
  if constexpr (PFV) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = min(j0 + i * 8 + g, jn);
      vr0[i] = *reinterpret_cast<const u32x4*>((j == jn ? vnew : vb + (long)j * w) + c * 8);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = j0 + lane; j < j1; j += 64) sum += __expf(s[j] - mx);
  sum = wave_sum(sum);
  if (lane == 0) { st[ks * 2] = mx; st[ks * 2 + 1] = sum; }
  __syncthreads();
  float gm = -1e30f, gl = 0.f;
  for (int k = 0; k < KS; ++k) gm = fmaxf(gm, st[k * 2]);
  for (int k = 0; k < KS; ++k) gl += st[k * 2 + 1] * __expf(st[k * 2] - gm);
  const float inv = 1.f / gl;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int jb = j0; jb < j1; jb += 64) {
    u32x4 vr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (PFV && jb == j0) {  // (uniform)
        vr[i] = vr0[PFV ? i : 0];
      } else {
        const int j = min(jb + i * 8 + g, jn);
        vr[i] = *reinterpret_cast<const u32x4*>((j == jn ? vnew : vb + (long)j * w) + c * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = jb + i * 8 + g;
      const float pr = j < j1 ? f16_to_f32(f32_to_f16(__expf(s[j] - gm) * inv)) : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] = fmaf(pr, from16_lo<F16>(vr[i][e]), acc[2 * e]);
        acc[2 * e + 1] = fmaf(pr, from16_hi<F16>(vr[i][e]), acc[2 * e + 1]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    acc[e] += __shfl_xor(acc[e], 8, 64);
    acc[e] += __shfl_xor(acc[e], 16, 64);
    acc[e] += __shfl_xor(acc[e], 32, 64);
  }
  if (g == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) po[ks * 64 + c * 8 + e] = acc[e];
  }
  __syncthreads();
  if (ks == 0) {
    float o = po[lane];
    for (int k = 1; k < KS; ++k) o += po[k * 64 + lane];
    // (out_frag: the o projection's operand order, common.hpp frag16_off)
    out[out_frag ? frag_off(b, head * 64 + lane, nq * 64, out_frag) : (long)b * nq * 64 + head * 64 + lane] = to16<F16>(o);
  }
}

// out[b] = src[b * L + kv_len[b] - 1]  (the last valid position's hidden state of each sample after the prefill)
__global__ __launch_bounds__(256) void gather_last_kernel(const bf16_t* __restrict__ src, const int* __restrict__ kv_len,
                                                          bf16_t* __restrict__ out, int L, int H) {
  const int b = blockIdx.x;
  const int l = min(max(kv_len[b], 1), L) - 1;
  const bf16_t* s = src + ((long)b * L + l) * H;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8)
    *reinterpret_cast<u32x4*>(out + (long)b * H + c) = *reinterpret_cast<const u32x4*>(s + c);
}

// ---------------------------------------------------------------------------
// Logits processors + token selection, one workgroup per sample (HF generation: RepetitionPenaltyLogitsProcessor,
// NoRepeatNGramLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper, in that order; the warpers
// only when sampling).  `logits` is modified in place (penalty / bans).  The candidates that survive top-k are gathered
// in descending (value, then ascending index) order by k rounds of a block-wide arg-max "next after the previous one" --
// exact ties at the k-th value are all kept, as `scores < topk[-1]` keeps them.
// ---------------------------------------------------------------------------
struct SampleP {
  float temperature, top_p, rep_penalty;
  int top_k, no_repeat_ngram, do_sample;
  long eos, pad;
  unsigned int seed_lo, seed_hi;
};

constexpr int SMP_T = 1024;   // threads
constexpr int SMP_CAP = 256;  // candidate list capacity (top_k + ties)
constexpr int SMP_BINS = 2048;  // histogram of (max - score) * 64: covers 32 units below the maximum
constexpr int SMP_LIST = 1024;  // pre-selected scores (everything down to the threshold bin)
static_assert(SMP_LIST <= SMP_T && (SMP_BINS / 64 & (SMP_BINS / 64 - 1)) == 0, "one thread per list entry; bins per lane a power of two");

__device__ __forceinline__ bool after(float v, int i, float pv, int pi) {  // (v, i) comes after (pv, pi) in (value desc, index asc) order
  return v < pv || (v == pv && i > pi);
}
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {  // (v, i) precedes (bv, bi)
  return v > bv || (v == bv && i < bi);
}

// ---------------------------------------------------------------------------
// Two-stage form (round 4; tcavt_sample_logits with a workspace): one workgroup per sample walks 128 256 logits in three
// dependent passes -- 64 us on 8 of 256 CUs at B = 8, latency bound.  Stage 1 cuts every sample's vocabulary into SMP_G
// contiguous slices, one workgroup each (B x 16 workgroups): the slice's processors (a token's penalty / ban is applied by the
// workgroup that owns the token), then the slice's <= 8 scores per thread live in REGISTERS and the three passes (max / min,
// histogram, collection) cost no further memory traffic; out come the slice's candidates -- everything at or above the slice's
// own top-k threshold bin -- and the smallest of them, lb: at least k scores of the slice are >= lb, so the sample's k-th
// largest score is >= lb too.  Stage 2 (sample_kernel with `pre`) keeps the candidates >= max over the slices of lb (a superset
// of the global top-k and of every tie with the k-th value), and ranks / keeps / draws exactly as the one-stage form does:
// the selected token is the same.  A slice whose threshold bin overflows its list (e.g. constant logits) raises ovf[b] and the
// sample falls back to the three passes of the one-stage form (its processors are done).
// ---------------------------------------------------------------------------
// Does token t occur among the first n entries of an LDS-resident history?  Four entries per LDS read and NO early exit: written
// as "scan until found", thread i's loop is a chain of up to i dependent LDS round trips (one ~100-cycle latency per entry:
// 15 us for a 300-token history -- a quarter of the whole selection kernel); without the data-dependent exit the reads pipeline.
__device__ __forceinline__ bool occurs_before(const int* __restrict__ lds_tok, int n, int t) {
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  bool dup = false;
  int j = 0;
#pragma unroll 4
  for (; j + 4 <= n; j += 4) {
    const i32x4 q = *reinterpret_cast<const i32x4*>(lds_tok + j);
    dup |= (q[0] == t) | (q[1] == t) | (q[2] == t) | (q[3] == t);
  }
  for (; j < n; ++j) dup |= lds_tok[j] == t;
  return dup;
}

constexpr int SMP_G = 16;      // slices (stage-1 workgroups) per sample
constexpr int SMP_LCAP = 512;  // candidates per slice
constexpr int SMP_E4 = 4;      // f32x4 registers per thread: slices of up to 4 * 1024 * 4 scores (V <= 262 144)

struct SliceWs {  // views into the caller's workspace (tcavt_sample_logits: layout and size)
  int* ticket;    // [1]   stage 2: workgroups that have read *step (the last one increments it); zero between calls
  int* ovf;       // [B]   a slice of the sample overflowed its list; zero between calls
  int* cand_n;    // [B][G]
  float* lb;      // [B][G]
  float* cand_v;  // [B][G][SMP_LCAP]
  int* cand_i;    // [B][G][SMP_LCAP]
};

__global__ __launch_bounds__(SMP_T) void sample_slice_kernel(float* __restrict__ logits, int V, const long* __restrict__ history,
                                                             int hist_cap, const int* __restrict__ hist_len, SampleP sp, SliceWs ws) {
  __shared__ int hist_bins[SMP_BINS];
  __shared__ float lv[SMP_LCAP];
  __shared__ __attribute__((aligned(16))) int li[SMP_LCAP];
  __shared__ float rv[SMP_T / 64], rm[SMP_T / 64];
  __shared__ int ri[SMP_T / 64];
  __shared__ float s_max, s_min;
  __shared__ int s_bi, list_n, thr_bin;
  const int b = blockIdx.x / SMP_G, g = blockIdx.x % SMP_G, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* x = logits + (long)b * V;
  const long* hist = history + (long)b * hist_cap;
  const int hl = min(hist_len[b], hist_cap);
  const int V4 = V >> 2, chunk4 = (V4 + SMP_G - 1) / SMP_G;
  const int lo4 = g * chunk4, hi4 = min(lo4 + chunk4, V4);
  const long lo = 4L * lo4, hi = 4L * hi4;
  // ---- processors, for the tokens this slice owns (same arithmetic and order as the one-stage form: penalty, then ban)
  // (the history goes to LDS once when it fits, and both processors read it there: every global access the processors make is
  //  one more dependent round trip at the head of a kernel that is a handful of them long)
  const bool in_lds = hl <= SMP_LCAP;
  const int ng = sp.no_repeat_ngram;
  const bool do_rep = sp.rep_penalty != 1.f, do_ban = ng > 0 && hl + 1 >= ng;
  if (in_lds && (do_rep || do_ban)) {
    for (int i = tid; i < hl; i += SMP_T) li[i] = (int)min(max(hist[i], -1L), 0x7fffffffL);  // (ids >= 2^31 - 1 are out of range anyway)
    __syncthreads();
  }
  auto tok_at = [&](int i) -> long { return in_lds ? (long)li[i] : hist[i]; };
  if (do_rep) {
    for (int i = tid; i < hl; i += SMP_T) {
      const long t = tok_at(i);
      bool first = t >= lo && t < hi;
      if (in_lds) {
        first = first && !occurs_before(li, i, (int)t);
      } else {
        for (int j = 0; j < i && first; ++j) first = hist[j] != t;
      }
      if (first) {
        const float s_ = x[t];
        x[t] = s_ < 0.f ? s_ * sp.rep_penalty : s_ / sp.rep_penalty;
      }
    }
    __syncthreads();
  }
  if (do_ban) {
    for (int i = tid; i + ng - 1 < hl; i += SMP_T) {
      bool match = true;
      for (int k = 0; k < ng - 1 && match; ++k) match = tok_at(i + k) == tok_at(hl - (ng - 1) + k);
      const long t = tok_at(i + ng - 1);
      if (match && t >= lo && t < hi) x[t] = -INFINITY;
    }
    __syncthreads();
  }
  // ---- the slice, in registers (scaled by 1 / temperature when sampling: the order of the scores is what counts)
  const float sc = sp.do_sample ? 1.f / sp.temperature : 1.f;
  f32x4 v4[SMP_E4];
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
#pragma unroll
  for (int e = 0; e < SMP_E4; ++e) {
    const int i4 = lo4 + tid + e * SMP_T;
    v4[e] = i4 < hi4 ? x4[i4] * sc : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // (a padding -inf never precedes a real entry: its index is larger)
  }
  // best (value desc, index asc) and smallest finite score of the slice
  float bv = -INFINITY, lmin = INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int e = 0; e < SMP_E4; ++e) {
    const int i4 = lo4 + tid + e * SMP_T;
    if (i4 < hi4) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float v = v4[e][c];
        const int i = 4 * i4 + c;
        if (better(v, i, bv, bi)) { bv = v; bi = i; }
        if (v > -INFINITY) lmin = fminf(lmin, v);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  lmin = -wave_max(-lmin);
  if (lane == 0) { rv[wv] = bv; ri[wv] = bi; rm[wv] = lmin; }
  __syncthreads();
  if (tid == 0) {
    float v = rv[0], m_ = rm[0];
    int i = ri[0];
    for (int k = 1; k < SMP_T / 64; ++k) {
      if (better(rv[k], ri[k], v, i)) { v = rv[k]; i = ri[k]; }
      m_ = fminf(m_, rm[k]);
    }
    s_max = v; s_bi = i; s_min = m_;
    list_n = 0;
  }
  for (int i = tid; i < SMP_BINS; i += SMP_T) hist_bins[i] = 0;
  __syncthreads();
  const long slot0 = ((long)b * SMP_G + g) * SMP_LCAP;
  if (!sp.do_sample) {  // greedy: the slice's first maximum
    if (tid == 0) {
      ws.cand_v[slot0] = s_max;
      ws.cand_i[slot0] = s_bi;
      ws.cand_n[b * SMP_G + g] = 1;
      ws.lb[b * SMP_G + g] = -INFINITY;
    }
    return;
  }
  const float gmax = s_max;
  if (!(gmax > -INFINITY)) {  // (uniform) nothing finite in this slice
    if (tid == 0) { ws.cand_n[b * SMP_G + g] = 0; ws.lb[b * SMP_G + g] = -INFINITY; }
    return;
  }
  const float bscale = (float)(SMP_BINS - 1) / fmaxf(gmax - s_min, 1e-20f);
  const int k = min(max(sp.top_k, 1), SMP_CAP);
#pragma unroll
  for (int e = 0; e < SMP_E4; ++e) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = (gmax - v4[e][c]) * bscale;
      if (d < (float)SMP_BINS) atomicAdd(&hist_bins[(int)d], 1);  // (-inf: d = +inf, skipped)
    }
  }
  __syncthreads();
  if (wv == 0) {  // first bin at which the running count reaches k (as in the one-stage form)
    constexpr int PER = SMP_BINS / 64;
    int loc = 0;
#pragma unroll 8
    for (int t_ = 0; t_ < PER; ++t_) loc += hist_bins[lane * PER + ((t_ + lane) & (PER - 1))];
    int pre = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(pre, o, 64);
      if (lane >= o) pre += up;
    }
    const bool hit = pre >= k && pre - loc < k;
    if (__builtin_amdgcn_ballot_w64(hit) == 0ull) {
      if (lane == 0) thr_bin = -1;  // fewer than k finite scores in the slice: all of them are candidates, no bound
    } else if (hit) {
      int cum = pre - loc, tb = lane * PER;
      for (int t_ = 0; t_ < PER; ++t_) {
        cum += hist_bins[lane * PER + t_];
        if (cum >= k) { tb = lane * PER + t_; break; }
      }
      thr_bin = cum <= SMP_LCAP ? tb : -2;  // -2: the threshold bin overflows the slice's list
    }
  }
  __syncthreads();
  const int tb = thr_bin;
  if (tb == -2) {  // (uniform)
    if (tid == 0) { atomicOr(&ws.ovf[b], 1); ws.cand_n[b * SMP_G + g] = 0; ws.lb[b * SMP_G + g] = -INFINITY; }
    return;
  }
  const float lim = tb < 0 ? (float)SMP_BINS : (float)(tb + 1);
  float mymin = INFINITY;
#pragma unroll
  for (int e = 0; e < SMP_E4; ++e) {
    const int i4 = lo4 + tid + e * SMP_T;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v = v4[e][c];
      if ((gmax - v) * bscale < lim) {
        const int slot = atomicAdd(&list_n, 1);
        if (slot < SMP_LCAP) { lv[slot] = v; li[slot] = 4 * i4 + c; }
        mymin = fminf(mymin, v);
      }
    }
  }
  mymin = -wave_max(-mymin);
  if (lane == 0) rm[wv] = mymin;
  __syncthreads();
  const int n = min(list_n, SMP_LCAP);  // (tb >= 0: list_n = the running count at the threshold bin <= SMP_LCAP; tb < 0: < k <= SMP_CAP)
  for (int t_ = tid; t_ < n; t_ += SMP_T) {
    ws.cand_v[slot0 + t_] = lv[t_];
    ws.cand_i[slot0 + t_] = li[t_];
  }
  if (tid == 0) {
    float m_ = rm[0];
    for (int w_ = 1; w_ < SMP_T / 64; ++w_) m_ = fminf(m_, rm[w_]);
    ws.cand_n[b * SMP_G + g] = n;
    ws.lb[b * SMP_G + g] = tb >= 0 ? m_ : -INFINITY;  // (>= k scores of the slice are >= the smallest collected one)
  }
}

// `pre` (ws.cand_v != nullptr): stage 2 of the two-stage form -- the processors are done and the candidates come from the slices
__global__ __launch_bounds__(SMP_T) void sample_kernel(float* __restrict__ logits, int V, long* __restrict__ history,
                                                       int hist_cap, int* __restrict__ hist_len, SampleP sp,
                                                       int* __restrict__ step_p, long* __restrict__ cur_tok,
                                                       int* __restrict__ pos, int* __restrict__ finished,
                                                       long* __restrict__ out_tokens, int out_cap, int advance_pos, SliceWs ws,
                                                       int n_samples) {
  __shared__ float cv[SMP_CAP];
  __shared__ int ci[SMP_CAP];
  __shared__ float rv[SMP_T / 64];
  __shared__ int ri[SMP_T / 64];
  __shared__ float rm[SMP_T / 64];
  __shared__ float bestv;
  __shared__ int besti;
  __shared__ int hist_bins[SMP_BINS];
  __shared__ float list_v[SMP_LIST];
  __shared__ __attribute__((aligned(16))) int list_i[SMP_LIST];
  __shared__ int list_n, thr_bin, keep_n;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* x = logits + (long)b * V;
  long* hist = history + (long)b * hist_cap;
  const int hl = min(hist_len[b], hist_cap);
  const int step = *step_p;
  const bool pre = ws.cand_v != nullptr;
  __shared__ int s_ovf;
  if (pre) {
    if (tid == 0) s_ovf = ws.ovf[b];
    __syncthreads();
    if (tid == 0 && s_ovf) ws.ovf[b] = 0;  // (re-armed for the next call)
  }
  const bool pre_ok = pre && s_ovf == 0;  // (uniform) the slices' candidates are complete
  // ---- repetition penalty: once per distinct token of the history (scatter semantics of the reference processor)
  if (!pre && sp.rep_penalty != 1.f) {
    // (the history is staged in LDS when it fits: thread i compares its token with all earlier ones)
    const bool in_lds = hl <= SMP_LIST;
    if (in_lds) {
      for (int i = tid; i < hl; i += SMP_T) list_i[i] = (int)min(max(hist[i], -1L), 0x7fffffffL);  // (ids >= 2^31 - 1 are out of range anyway)
      __syncthreads();
    }
    for (int i = tid; i < hl; i += SMP_T) {
      const long t = hist[i];
      bool first = t >= 0 && t < V;
      if (in_lds) {
        first = first && !occurs_before(list_i, i, (int)t);
      } else {
        for (int j = 0; j < i && first; ++j) first = hist[j] != t;
      }
      if (first) {
        const float s = x[t];
        x[t] = s < 0.f ? s * sp.rep_penalty : s / sp.rep_penalty;
      }
    }
    __syncthreads();
  }
  // ---- no-repeat n-gram: ban every token that would complete an n-gram already in the history
  const int ng = sp.no_repeat_ngram;
  if (!pre && ng > 0 && hl + 1 >= ng) {
    for (int i = tid; i + ng - 1 < hl; i += SMP_T) {
      bool match = true;
      for (int k = 0; k < ng - 1 && match; ++k) match = hist[i + k] == hist[hl - (ng - 1) + k];
      const long t = hist[i + ng - 1];
      if (match && t >= 0 && t < V) x[t] = -INFINITY;
    }
    __syncthreads();
  }
  // every logit of the sample, 16 bytes per lane and load when the row allows (V % 4 == 0, aligned) and several loads in flight:
  // the passes below are latency bound (one workgroup per sample), so fewer and wider loads are what shortens them
  auto walk = [&](auto&& f) {
    if ((V & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
      const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
      const int V4 = V >> 2;
#pragma unroll 4
      for (int i4 = tid; i4 < V4; i4 += SMP_T) {
        const f32x4 t = x4[i4];
        f(t[0], 4 * i4);
        f(t[1], 4 * i4 + 1);
        f(t[2], 4 * i4 + 2);
        f(t[3], 4 * i4 + 3);
      }
    } else {
#pragma unroll 4
      for (int i = tid; i < V; i += SMP_T) f(x[i], i);
    }
  };
  // block-wide arg-max of the elements that come after (pv, pi) in (value desc, index asc) order; with_min: also the
  // smallest finite value (left in rv[0] region -> minv) in the same pass
  __shared__ float minv;
  auto next_best = [&](float pv, int pi, float scale, bool with_min) {
    float bv = -INFINITY, lmin = INFINITY;
    int bi = 0x7fffffff;
    walk([&](float raw, int i) {
      const float v = raw * scale;
      if (after(v, i, pv, pi) && better(v, i, bv, bi)) { bv = v; bi = i; }
      if (with_min && v > -INFINITY) lmin = fminf(lmin, v);
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (with_min) lmin = -wave_max(-lmin);
    if (lane == 0) { rv[wv] = bv; ri[wv] = bi; rm[wv] = lmin; }
    __syncthreads();
    if (tid == 0) {
      float v = rv[0], m_ = rm[0];
      int i = ri[0];
      for (int k = 1; k < SMP_T / 64; ++k) {
        if (better(rv[k], ri[k], v, i)) { v = rv[k]; i = ri[k]; }
        m_ = fminf(m_, rm[k]);
      }
      bestv = v;
      besti = i;
      minv = m_;
    }
    __syncthreads();
  };
  long tok;
  if (!sp.do_sample && pre_ok) {  // greedy, two-stage form: the best of the slices' first maxima
    if (tid < 64) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      if (tid < SMP_G && ws.cand_n[b * SMP_G + tid] > 0) {
        bv = ws.cand_v[((long)b * SMP_G + tid) * SMP_LCAP];
        bi = ws.cand_i[((long)b * SMP_G + tid) * SMP_LCAP];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
      }
      if (tid == 0) besti = bi;
    }
    __syncthreads();
    tok = besti;
  } else if (!sp.do_sample) {  // greedy: arg-max of the processed scores, first maximum wins (torch.argmax)
    next_best(INFINITY, -1, 1.f, false);
    tok = besti;
  } else {
    // Candidate collection in three passes over the vocabulary instead of one pass per candidate: (1) the maximum and the
    // minimum (one pass), (2) a histogram of (max - score) in SMP_BINS bins over that range (integer counts: order-independent), from which the narrowest
    // threshold that keeps at least top_k scores follows, (3) everything at or above that threshold goes to an LDS list
    // (top_k + the rest of the threshold bin; insertion order does not matter, the selection below orders by
    // (value, index)).  The list is then ordered by top_k rounds of a wave-level arg-max over <= SMP_LIST entries.
    const float invT = 1.f / sp.temperature;
    const int k = min(max(sp.top_k, 1), SMP_CAP);
    bool have_list = false;  // (uniform)
    if (pre_ok) {
      // two-stage form: every slice's candidates at or above the largest of the slices' bounds (>= k scores of ONE slice lie at
      // or above its bound, so the sample's k-th largest score does too: nothing below it can be kept)
      if (tid == 0) { list_n = 0; keep_n = 0; }
      // (one slice per WAVE -- SMP_G == SMP_T / 64 -- and every load of a phase in flight together: written as a loop over the
      //  slices this was 2 x 16 dependent round trips to memory that another CU has just written, 40 us for a ~10 us kernel)
      static_assert(SMP_G == SMP_T / 64, "one wave per slice");
      float lmax = ws.lb[b * SMP_G + (lane & (SMP_G - 1))];
#pragma unroll
      for (int o = 1; o < SMP_G; o <<= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
      const int n_g = ws.cand_n[b * SMP_G + wv];
      const long s0 = ((long)b * SMP_G + wv) * SMP_LCAP;
      float cvv[SMP_LCAP / 64];
      int cii[SMP_LCAP / 64];
#pragma unroll
      for (int e = 0; e < SMP_LCAP / 64; ++e) {
        const int t_ = lane + 64 * e;
        cvv[e] = t_ < n_g ? ws.cand_v[s0 + t_] : -INFINITY;
        cii[e] = t_ < n_g ? ws.cand_i[s0 + t_] : 0;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < SMP_LCAP / 64; ++e) {
        if (lane + 64 * e < n_g && cvv[e] >= lmax) {
          const int slot = atomicAdd(&list_n, 1);
          if (slot < SMP_LIST) { list_v[slot] = cvv[e]; list_i[slot] = cii[e]; }
        }
      }
      __syncthreads();
      have_list = list_n <= SMP_LIST;  // (more: the three passes below, over the already processed logits)
      if (tid == 0 && have_list) thr_bin = 0;
      __syncthreads();
    }
    float gmax = 0.f, bscale = 0.f;
    if (!have_list) {
    next_best(INFINITY, -1, invT, true);  // (1): the maximum and, in the same pass, the smallest finite score
    gmax = bestv;
    // bin width from the spread of the finite scores: SMP_BINS bins between the maximum and the minimum
    bscale = (float)(SMP_BINS - 1) / fmaxf(gmax - minv, 1e-20f);
    __syncthreads();
    for (int i = tid; i < SMP_BINS; i += SMP_T) hist_bins[i] = 0;
    if (tid == 0) { list_n = 0; keep_n = 0; }
    __syncthreads();
    walk([&](float raw, int) {
      const float d = (gmax - raw * invT) * bscale;
      if (d < (float)SMP_BINS) atomicAdd(&hist_bins[(int)d], 1);  // (-inf scores: d = +inf, skipped)
    });
    __syncthreads();
    if (wv == 0) {
      // first bin at which the running count reaches k: every lane adds up its SMP_BINS / 64 consecutive bins, a wave scan
      // finds the lane in whose range the count crosses k, and only that lane walks its bins (one thread walking all 2048
      // bins was a chain of up to 2048 dependent LDS reads)
      constexpr int PER = SMP_BINS / 64;
      int loc = 0;
#pragma unroll 8
      for (int t_ = 0; t_ < PER; ++t_) loc += hist_bins[lane * PER + ((t_ + lane) & (PER - 1))];  // (rotated: 2-way instead of 64-way bank conflicts)
      int pre = loc;  // inclusive prefix over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(pre, o, 64);
        if (lane >= o) pre += up;
      }
      const bool hit = pre >= k && pre - loc < k;
      if (__builtin_amdgcn_ballot_w64(hit) == 0ull) {
        if (lane == 0) thr_bin = -1;  // fewer than k scores in range -> exact fallback
      } else if (hit) {
        int cum = pre - loc, tb = lane * PER;
        for (int t_ = 0; t_ < PER; ++t_) {
          cum += hist_bins[lane * PER + t_];
          if (cum >= k) { tb = lane * PER + t_; break; }
        }
        thr_bin = cum <= SMP_LIST ? tb : -1;  // -1: too many scores in the threshold bin -> exact fallback
      }
    }
    __syncthreads();
    }  // (!have_list)
    int n = 0;
    if (thr_bin >= 0) {
      if (!have_list) {
        const float lim = (float)(thr_bin + 1);
        walk([&](float raw, int i) {
          const float v = raw * invT;
          if ((gmax - v) * bscale < lim) {
            const int slot = atomicAdd(&list_n, 1);
            if (slot < SMP_LIST) { list_v[slot] = v; list_i[slot] = i; }
          }
        });
        __syncthreads();
      }
      const int ln = min(list_n, SMP_LIST);
      // order the list by RANK: entry t precedes rank(t) others in (value desc, index asc) order -- indices are distinct, so
      // the ranks are a permutation -- and goes to slot rank(t).  Every thread reads the same list entry at a time (LDS
      // broadcast).  Kept: the first k and every tie with the k-th value (as `scores < topk[-1]` keeps them), capped at
      // SMP_CAP.  (The first version picked the next entry by a wave-wide arg-max, one round per candidate.)
      float mv = 0.f;
      int mi = 0, rank = 0x7fffffff;
      if (tid < ln) {
        mv = list_v[tid];
        mi = list_i[tid];
        rank = 0;
        for (int j = 0; j < ln; ++j) rank += better(list_v[j], list_i[j], mv, mi) ? 1 : 0;
        if (rank < SMP_CAP) { cv[rank] = mv; ci[rank] = mi; }
      }
      __syncthreads();
      const int kk = min(k, ln);
      if (tid < ln && kk > 0 && (rank < kk || mv == cv[kk - 1])) atomicAdd(&keep_n, 1);
      __syncthreads();
      n = min(keep_n, SMP_CAP);
    } else {  // exact fallback: one pass over the vocabulary per candidate
      float pv = INFINITY;
      int pi = -1;
      for (;;) {
        next_best(pv, pi, invT, false);
        const float v = bestv;
        const int i = besti;
        if (i == 0x7fffffff || v == -INFINITY) break;        // nothing (finite) left
        if (n >= k && !(v == cv[k - 1])) break;               // beyond top-k and not a tie with the k-th value
        if (n >= SMP_CAP) break;
        if (tid == 0) { cv[n] = v; ci[n] = i; }
        ++n;
        pv = v;
        pi = i;
        __syncthreads();
      }
    }
    if (tid == 0) {
      // softmax over the kept candidates (descending), top-p: drop the low tail whose cumulative probability, counted
      // from the bottom and including the item, is <= 1 - top_p (keep at least one)
      const float m = cv[0];
      float tot = 0.f;
      for (int j = 0; j < n; ++j) tot += __expf(cv[j] - m);
      int keep = n;
      if (sp.top_p < 1.f) {
        float tail = 0.f;
        for (int j = n - 1; j >= 1; --j) {
          tail += __expf(cv[j] - m) / tot;
          if (tail <= 1.f - sp.top_p) keep = j;
          else break;
        }
      }
      float kt = 0.f;
      for (int j = 0; j < keep; ++j) kt += __expf(cv[j] - m);
      unsigned int r[4];
      philox4x32_10((unsigned int)step, (unsigned int)b, 0x5A3Bu, 0u, sp.seed_lo, sp.seed_hi, r);
      const float u = (float)(r[0] >> 8) * (1.0f / 16777216.0f) * kt;
      float acc = 0.f;
      int pick = keep - 1;
      for (int j = 0; j < keep; ++j) {
        acc += __expf(cv[j] - m);
        if (u < acc) { pick = j; break; }
      }
      besti = n > 0 ? ci[pick] : 0;
    }
    __syncthreads();
    tok = besti;
  }
  if (tid == 0) {
    // a finished sample keeps emitting the pad token (HF generate pads finished rows)
    const bool fin = finished[b] != 0;
    const long outt = fin ? sp.pad : tok;
    if (step < out_cap) out_tokens[(long)b * out_cap + step] = outt;
    if (!fin && sp.eos >= 0 && tok == sp.eos) finished[b] = 1;
    cur_tok[b] = outt;
    if (hl < hist_cap) {
      hist[hl] = outt;
      hist_len[b] = hl + 1;
    }
    if (advance_pos) pos[b] += 1;
  }
  if (pre && tid == 0) {
    // *step is advanced here instead of by a launch of its own: every workgroup has read it by the time it takes its ticket,
    // and the one that draws the last ticket writes step + 1 and re-arms the counter
    const int t_ = __hip_atomic_fetch_add(ws.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t_ == n_samples - 1) {
      __hip_atomic_store(ws.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *step_p = step + 1;
    }
  }
}

__global__ void step_inc_kernel(int* step) { *step += 1; }

}  // namespace tcavt

using namespace tcavt;

#define TCAVT_TRY(call)              \
  do {                               \
    const int rc_ = (call);          \
    if (rc_ != TCAVT_OK) return rc_; \
  } while (0)

extern "C" int64_t tcavt_sample_workspace_bytes(int B) {
  if (B <= 0) return 0;
  return 64 + (((int64_t)B * 4 + 63) & ~(int64_t)63) + (int64_t)B * SMP_G * 8 + (int64_t)B * SMP_G * SMP_LCAP * 8;
}

extern "C" int tcavt_sample_logits(float* logits, int B, int V, int64_t* history, int hist_cap, int32_t* hist_len,
                                   const tcavt_sample_params* sp, int32_t* step, int64_t* cur_tok, int32_t* pos,
                                   int32_t* finished, int64_t* out_tokens, int out_cap, int advance_pos, void* workspace,
                                   int64_t workspace_bytes, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(logits && history && hist_len && sp && step && cur_tok && pos && finished && out_tokens && B > 0 && V > 0 &&
                      hist_cap > 0 && out_cap > 0,
                  "sample_logits: bad args");
  TCAVT_CHECK_ARG(!sp->do_sample || (sp->temperature > 0.f && sp->top_k >= 1 && sp->top_k <= SMP_CAP && sp->top_p > 0.f && sp->top_p <= 1.f),
                  "sample_logits: sampling needs temperature > 0, 1 <= top_k <= %d, 0 < top_p <= 1", SMP_CAP);
  TCAVT_CHECK_ARG(sp->repetition_penalty > 0.f && sp->no_repeat_ngram_size >= 0, "sample_logits: bad repetition_penalty / no_repeat_ngram_size");
  SampleP p;
  p.temperature = sp->temperature; p.top_p = sp->top_p; p.rep_penalty = sp->repetition_penalty;
  p.top_k = sp->top_k; p.no_repeat_ngram = sp->no_repeat_ngram_size; p.do_sample = sp->do_sample;
  p.eos = sp->eos_token_id; p.pad = sp->pad_token_id;
  p.seed_lo = (unsigned int)(sp->seed & 0xffffffffu); p.seed_hi = (unsigned int)(sp->seed >> 32);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // two-stage form when the caller lends a workspace and the rows can be cut into 16-byte-aligned slices that fit the registers
  SliceWs ws = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const int64_t need = tcavt_sample_workspace_bytes(B);
  const int chunk4 = ((V >> 2) + SMP_G - 1) / SMP_G;
  if (workspace && workspace_bytes >= need && (V & 3) == 0 && aligned16(logits) && chunk4 <= SMP_E4 * SMP_T) {
    TCAVT_CHECK_ARG(aligned16(workspace), "sample_logits: workspace must be 16-byte aligned");
    char* w = static_cast<char*>(workspace);
    ws.ticket = reinterpret_cast<int*>(w);
    ws.ovf = reinterpret_cast<int*>(w + 64);
    size_t off = 64 + (((size_t)B * 4 + 63) & ~(size_t)63);
    ws.cand_n = reinterpret_cast<int*>(w + off); off += (size_t)B * SMP_G * 4;
    ws.lb = reinterpret_cast<float*>(w + off); off += (size_t)B * SMP_G * 4;
    ws.cand_v = reinterpret_cast<float*>(w + off); off += (size_t)B * SMP_G * SMP_LCAP * 4;
    ws.cand_i = reinterpret_cast<int*>(w + off);
    hipLaunchKernelGGL(sample_slice_kernel, dim3(B * SMP_G), dim3(SMP_T), 0, st, logits, V, reinterpret_cast<const long*>(history), hist_cap,
                       hist_len, p, ws);
  }
  hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(SMP_T), 0, st, logits, V, reinterpret_cast<long*>(history), hist_cap,
                     hist_len, p, step, reinterpret_cast<long*>(cur_tok), pos, finished, reinterpret_cast<long*>(out_tokens),
                     out_cap, advance_pos, ws, B);
  if (!ws.cand_v) hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, step);
  TCAVT_CHECK_LAUNCH("sample_logits");
  return TCAVT_OK;
}

extern "C" int tcavt_gather_last(const void* src16, const int32_t* kv_len, void* out16, int B, int L, int H,
                                 tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(src16 && kv_len && out16 && B > 0 && L > 0 && H > 0 && H % 8 == 0, "gather_last: bad args");
  hipLaunchKernelGGL(gather_last_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const bf16_t*>(src16), kv_len, static_cast<bf16_t*>(out16), L, H);
  TCAVT_CHECK_LAUNCH("gather_last");
  return TCAVT_OK;
}

extern "C" int tcavt_llama_decode_step(const tcavt_decode_args* a, tcavt_stream_t stream) {
  TCAVT_CHECK_ARG(a && a->layers && a->gamma_final && a->rope_cos && a->rope_sin && a->table && a->txt_mod && a->cur_tok && a->pos &&
                      a->h16 && a->part && a->qkv && a->att && a->act && a->k_cache && a->v_cache && a->x16 && a->logits &&
                      a->bad_id_flag,
                  "llama_decode_step: null pointer");
  TCAVT_CHECK_ARG(a->n_layers > 0 && a->B > 0 && a->H % 256 == 0 && a->I > 0 && a->nq > 0 && a->nkv > 0 && a->nq % a->nkv == 0 &&
                      a->nq / a->nkv <= 8 && a->V > 0 && a->V % 16 == 0 && a->kv_lmax > 0 && a->rope_L >= a->kv_lmax && is16(a->dtype16),
                  "llama_decode_step: bad shape (H %% 256, V %% 16, nq / nkv <= 8, rope_L >= kv_lmax)");
  const int B = a->B, H = a->H, I = a->I, nq = a->nq, nkv = a->nkv, dt = a->dtype16;
  const int nqkv = (nq + 2 * nkv) * 64;
  const int np_in = norm_out_npart(B, H, I), np_post = norm_out_npart(B, H, nq * 64);  // see csrc/stack.hip
  hipStream_t st = static_cast<hipStream_t>(stream);
  // scaled 16-bit image of the residual stream, as in tcavt_llama_stack_forward (the prefill that filled the cache ran at the same scale
  // or not: keys / values are true-scale either way)
  TCAVT_CHECK_ARG(a->stream_scale >= 0.f && a->stream_scale <= 1.f, "llama_decode_step: stream_scale must be in (0, 1] (0 means 1)");
  const float ss_ = a->stream_scale == 0.f ? 1.f : a->stream_scale;
  const float eps_s = a->rms_eps * ss_ * ss_;
  // fragment-major weight copies (tcavt_pack_weight16): the projections' weight streams read consecutive bytes
  const int wl = a->w_layout;
  TCAVT_CHECK_ARG(wl == 0 || (wl == TCAVT_W_FRAG16 && B <= 32 && I % 256 == 0 && (nq * 64) % 256 == 0),
                  "llama_decode_step: w_layout must be 0 or TCAVT_W_FRAG16 (B <= 32, I %% 256 == 0, nq * 64 %% 256 == 0)");
  // fragment-major activations: h16, att, act, x16 in the skinny GEMMs' operand order (16 or 32 whole rows each)
  const int al = a->act_layout;
  TCAVT_CHECK_ARG(al == 0 || ((al == 1 || al == 2) && a->h == nullptr && B <= (al == 2 ? 8 : 32) && I % 256 == 0 && (nq * 64) % 256 == 0),
                  "llama_decode_step: act_layout must be 0, 1 (B <= 32) or 2 (B <= 8) (16-bit residual stream: h == NULL; I %% 256 == 0, "
                  "nq * 64 %% 256 == 0)");
  const int blk8 = al == 2 ? TCAVT_ACT_BLOCK8 : 0;
  const int gA = al ? (TCAVT_ACT_A_FRAG16 | blk8) : 0, gAO = al ? (TCAVT_ACT_A_FRAG16 | TCAVT_ACT_OUT_FRAG16 | blk8) : 0;
  // h = table[cur_tok] + text modality embedding (generated tokens are text tokens: scripts/train.py:526-527); + the fused
  // norm's inputs
  // (a->h == NULL: the residual stream is the 16-bit h16 itself, as in tcavt_llama_stack_forward)
  TCAVT_TRY(embed_fuse_impl(a->table, a->cur_tok, a->txt_mod /* unused: Nq = 0 */, a->txt_mod, a->txt_mod, a->h, B, 0, 1, H, a->V,
                            a->bad_id_flag, dt, a->h16, a->part, np_in, ss_, al, stream));
  const size_t per_layer = (size_t)B * a->kv_lmax * nkv * 64;
  const int KS = std::min(16, (a->kv_lmax + 63) / 64);  // key splits (waves) per query head: one 64-key round each up to 1024 keys
  const size_t lds = ((size_t)a->kv_lmax + (size_t)KS * 66) * sizeof(float);
  TCAVT_CHECK_ARG(lds <= 64 * 1024, "llama_decode_step: kv_lmax = %d too long for the decode attention's score buffer", a->kv_lmax);
  for (int li = 0; li < a->n_layers; ++li) {
    const tcavt_llama_layer& w = a->layers[li];
    TCAVT_CHECK_ARG(w.w_qkv && w.w_o && w.w_gu && w.w_d && (!w.a_cat || (w.b_ext && a->t)), "llama_decode_step: layer %d: null weight", li);
    // LoRA down-projection: a launch of its own for layer 0 (and without a->lora_part); layers 1.. read the partial sums the
    // previous layer's down-projection GEMM wrote next to its residual epilogue
    // (a q|k|v workgroup reads the partial sums of its own tokens: all B for B <= 16, its 16-token block beyond)
    const bool lp_ok = a->lora_part && a->lora_rank > 0 && a->lora_rank <= 8 && B <= 32;
    const bool t_fused = lp_ok && li > 0 && w.a_cat && a->layers[li - 1].w_d;
    if (w.a_cat && !t_fused) {
      tcavt_gemm_args g = {};
      g.A = a->h16; g.lda = H; g.W = w.a_cat; g.ldw = H; g.C = a->t; g.ldc = 64; g.M = B; g.N = 64; g.K = H; g.act_layout = gA;
      g.out_dtype = dt; g.in_dtype = dt; g.acc_scale = a->lora_scale;  // (t is at the stream's scale, like the main term of the accumulator)
      g.splitk_ws = a->splitk_ws; g.splitk_ws_bytes = a->splitk_ws_bytes;
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
    }
    {
      tcavt_gemm_args g = {};
      g.A = a->h16; g.lda = H; g.W = w.w_qkv; g.ldw = H; g.C = a->qkv; g.ldc = nqkv; g.w_layout = wl; g.act_layout = gA;
      g.M = B; g.N = nqkv; g.K = H; g.out_dtype = dt; g.in_dtype = dt;
      if (t_fused) {
        g.W2 = w.b_ext; g.ldw2 = 64; g.lora_part = a->lora_part; g.lora_part_np = H / 16; g.lora_part_scale = a->lora_scale;
      } else if (w.a_cat) { g.A2 = a->t; g.lda2 = 64; g.W2 = w.b_ext; g.ldw2 = 64; g.K2 = 64; }
      g.epilogue = TCAVT_EPI_ROPE | TCAVT_EPI_ROWSCALE;
      g.rope_cos = a->rope_cos; g.rope_sin = a->rope_sin; g.rope_L = a->rope_L; g.rope_cols = (nq + nkv) * 64;
      g.rope_pos = a->pos;
      g.rowscale_part = a->part; g.rowscale_npart = np_in; g.rowscale_h = H; g.rowscale_eps = eps_s;
      g.splitk_ws = a->splitk_ws; g.splitk_ws_bytes = a->splitk_ws_bytes;
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
    }
    bf16_t* kc = static_cast<bf16_t*>(a->k_cache) + li * per_layer;
    bf16_t* vc = static_cast<bf16_t*>(a->v_cache) + li * per_layer;
    // (the new token's k / v rows are appended to the cache by attn_decode_kernel itself)
    {
      const bool pfv = B * nq <= 512;  // (value-row prefetch: only while the grid is about one workgroup per CU)
      auto kfn = dt == TCAVT_F16 ? (pfv ? attn_decode_kernel<true, true> : attn_decode_kernel<true, false>)
                                 : (pfv ? attn_decode_kernel<false, true> : attn_decode_kernel<false, false>);
      hipLaunchKernelGGL(kfn, dim3(B * nq), dim3(KS * 64), lds, st, static_cast<const bf16_t*>(a->qkv), kc, vc, a->pos,
                         static_cast<bf16_t*>(a->att), a->kv_lmax, nq, nkv, 0.125f, KS, al);
    }
    TCAVT_CHECK_LAUNCH("attn_decode");
    {
      tcavt_gemm_args g = {};
      g.A = a->att; g.lda = nq * 64; g.W = w.w_o; g.ldw = nq * 64; g.C = a->h; g.ldc = H; g.w_layout = wl; g.act_layout = gAO;
      g.M = B; g.N = H; g.K = nq * 64; g.out_dtype = TCAVT_F32; g.in_dtype = dt;
      g.residual = a->h; g.ldr = H; g.epilogue = TCAVT_EPI_RESIDUAL | TCAVT_EPI_NORM_OUT;
      g.norm_h16 = a->h16; g.norm_part = a->part; g.norm_scale = ss_;
      g.nonfinite_flag = a->nonfinite_flag; g.nonfinite_tag = 1 + 2 * li;
      g.splitk_ws = a->splitk_ws; g.splitk_ws_bytes = a->splitk_ws_bytes;
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
    }
    {
      tcavt_gemm_args g = {};
      g.A = a->h16; g.lda = H; g.W = w.w_gu; g.ldw = H; g.C = a->act; g.ldc = I; g.w_layout = wl; g.act_layout = gAO;
      g.M = B; g.N = 2 * I; g.K = H; g.out_dtype = dt; g.in_dtype = dt;
      g.epilogue = TCAVT_EPI_SILU_MUL | TCAVT_EPI_ROWSCALE;
      g.rowscale_part = a->part; g.rowscale_npart = np_post; g.rowscale_h = H; g.rowscale_eps = eps_s;
      g.splitk_ws = a->splitk_ws; g.splitk_ws_bytes = a->splitk_ws_bytes;
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
    }
    {
      tcavt_gemm_args g = {};
      g.A = a->act; g.lda = I; g.W = w.w_d; g.ldw = I; g.C = a->h; g.ldc = H; g.w_layout = wl; g.act_layout = gAO;
      g.M = B; g.N = H; g.K = I; g.out_dtype = TCAVT_F32; g.in_dtype = dt;
      g.residual = a->h; g.ldr = H; g.epilogue = TCAVT_EPI_RESIDUAL | TCAVT_EPI_NORM_OUT;
      g.norm_h16 = a->h16; g.norm_part = a->part; g.norm_scale = ss_;
      g.nonfinite_flag = a->nonfinite_flag; g.nonfinite_tag = 2 + 2 * li;
      if (lp_ok && li + 1 < a->n_layers && a->layers[li + 1].a_cat) {
        g.lora_part = a->lora_part; g.lora_part_a = a->layers[li + 1].a_cat; g.lora_part_lda = H;
      }
      g.splitk_ws = a->splitk_ws; g.splitk_ws_bytes = a->splitk_ws_bytes;
      TCAVT_TRY(tcavt_gemm_bf16(&g, stream));
    }
  }
  if (a->h == nullptr) TCAVT_TRY(rmsnorm16_impl(a->h16, a->gamma_final, eps_s, a->x16, nullptr, B, H, dt, al, stream));
  else TCAVT_TRY(tcavt_rmsnorm(a->h, a->gamma_final, a->rms_eps, a->x16, nullptr, B, H, nullptr, 0.f, 0, 0, dt, stream));
  // lm_head: tied to the embedding table (Llama-3.2-1B: tie_word_embeddings)
  tcavt_gemm_args g = {};
  g.A = a->x16; g.lda = H; g.W = a->table; g.ldw = H; g.C = a->logits; g.ldc = a->V;
  if (a->table_packed && B <= 32) { g.W = a->table_packed; g.w_layout = TCAVT_W_FRAG16; }
  g.act_layout = gA;
  g.M = B; g.N = a->V; g.K = H; g.out_dtype = TCAVT_F32; g.in_dtype = dt;
  g.splitk_ws = a->splitk_ws; g.splitk_ws_bytes = a->splitk_ws_bytes;
  return tcavt_gemm_bf16(&g, stream);
}
