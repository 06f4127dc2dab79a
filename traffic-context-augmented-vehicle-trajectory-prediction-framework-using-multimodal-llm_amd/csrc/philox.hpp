// Counter-based RNG for dropout (train-mode forward; the MC-dropout K-candidate protocol of
// scripts/test.py:1301-1342 runs the model in train mode under no_grad).  Philox4x32-10 (Salmon et al.,
// SC'11): key = 64-bit seed, counter = (element-octet index lo, hi, site id, 0) -> four uint32 = eight 16-bit draws per
// call, one per element of an octet (element e: call e >> 3, half-word e & 7 in the order w0.lo, w0.hi, w1.lo, ...).
// The mask of element e at dropout site s under seed k is therefore a pure function of (k, s, e): any kernel (forward
// now, backward later) regenerates it without storing it, and the CPU restatement in oracle/philox.py reproduces it bit
// for bit.   keep(e) <=> u16(e) >= ceil(p * 65536),  y = x * keep / (1 - p)   (torch.nn.functional.dropout semantics; the
// keep probability is 1 - ceil(65536 p) / 65536: 0.899994 for p = 0.1).  Sixteen bits per decision instead of a whole
// word halve the generator's VALU work where a lane masks eight consecutive elements (the LoRA down-projection spends its
// time here: two masks over all B L x 2048 activations per layer).
#pragma once
#include <stdint.h>

namespace tcavt {

struct DropoutP {
  float p;              // 0 => disabled
  float inv_keep;       // 1 / (1 - p)
  unsigned int thr16;   // ceil(p * 65536): element kept iff its 16-bit draw >= thr16
  unsigned int seed_lo, seed_hi, site;
  // optional: a device-resident 64-bit "epoch" that is ADDED to the seed when the kernel runs.  A hipGraph bakes kernel
  // arguments, so a captured training step would replay one set of masks forever; with the epoch (advanced by a node of
  // the same graph, tcavt_dropout_epoch_advance) every replay draws fresh masks, and forward and backward of one step
  // still agree.  NULL (the default) = epoch 0: masks are a function of (seed, site, element) alone.
  const unsigned long long* epoch;
};

// the library-wide epoch pointer (tcavt_set_dropout_epoch, csrc/core.hip)
const unsigned long long* dropout_epoch_ptr();

inline DropoutP make_dropout(float p, uint64_t seed, uint32_t site) {
  DropoutP d;
  d.p = p;
  d.inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  {
    const float t = p * 65536.f;
    unsigned int thr = (unsigned int)t;
    if ((float)thr < t) ++thr;  // ceil
    d.thr16 = thr > 65535u ? 65535u : thr;
  }
  d.seed_lo = (unsigned int)(seed & 0xffffffffu);
  d.seed_hi = (unsigned int)(seed >> 32);
  d.site = site;
  d.epoch = p > 0.f ? dropout_epoch_ptr() : nullptr;
  return d;
}

__device__ __forceinline__ void philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                                              unsigned int k0, unsigned int k1, unsigned int (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // (one 32 x 32 -> 64 multiply per product, v_mad_u64_u32, instead of v_mul_hi_u32 + v_mul_lo_u32: half the multiply
    //  instructions, though the wide one is slower -- tcavt_lora_down 23.0 -> 22.1 us at M = 8192, H = 2048; the generator
    //  stays bound by its 40 integer multiplies per call)
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * (unsigned long long)c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * (unsigned long long)c2;
    const unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0;
    const unsigned int hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
    const unsigned int n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void dropout_words(const DropoutP& d, unsigned long long oct, unsigned int (&r)[4]) {
  unsigned int k0 = d.seed_lo, k1 = d.seed_hi;
  if (d.epoch) {  // (uniform) seed + epoch as one 64-bit sum
    const unsigned long long s = (((unsigned long long)k1 << 32) | k0) + *d.epoch;
    k0 = (unsigned int)(s & 0xffffffffu);
    k1 = (unsigned int)(s >> 32);
  }
  philox4x32_10((unsigned int)(oct & 0xffffffffu), (unsigned int)(oct >> 32), d.site, 0u, k0, k1, r);
}

// keep-scales (0 or 1/(1-p)) of the eight elements 8o .. 8o+7 of a site's flat index space: one generator call
__device__ __forceinline__ void dropout_oct(const DropoutP& d, unsigned long long oct, float (&scale)[8]) {
  unsigned int r[4];
  dropout_words(d, oct, r);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    scale[2 * i] = ((r[i] & 0xffffu) >= d.thr16) ? d.inv_keep : 0.f;
    scale[2 * i + 1] = ((r[i] >> 16) >= d.thr16) ? d.inv_keep : 0.f;
  }
}

// keep-scales of the four elements 4q .. 4q+3: half of octet q >> 1
__device__ __forceinline__ void dropout_quad(const DropoutP& d, unsigned long long quad, float (&scale)[4]) {
  unsigned int r[4];
  dropout_words(d, quad >> 1, r);
  const unsigned int w0 = (quad & 1) ? r[2] : r[0], w1 = (quad & 1) ? r[3] : r[1];
  scale[0] = ((w0 & 0xffffu) >= d.thr16) ? d.inv_keep : 0.f;
  scale[1] = ((w0 >> 16) >= d.thr16) ? d.inv_keep : 0.f;
  scale[2] = ((w1 & 0xffffu) >= d.thr16) ? d.inv_keep : 0.f;
  scale[3] = ((w1 >> 16) >= d.thr16) ? d.inv_keep : 0.f;
}

// single element e (draw e & 7 of octet e >> 3)
__device__ __forceinline__ float dropout_one(const DropoutP& d, unsigned long long e) {
  float s[4];
  dropout_quad(d, e >> 2, s);
  return s[e & 3];
}

}  // namespace tcavt
