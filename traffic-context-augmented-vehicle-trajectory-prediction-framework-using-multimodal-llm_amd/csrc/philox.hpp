// Counter-based RNG for dropout (train-mode forward; the MC-dropout K-candidate protocol of
// scripts/test.py:1301-1342 runs the model in train mode under no_grad).  Philox4x32-10 (Salmon et al.,
// SC'11): key = 64-bit seed, counter = (element-quad index lo, hi, site id, 0) -> four uint32 per call,
// one per element of a quad.  The mask of element e at dropout site s under seed k is therefore a pure
// function of (k, s, e): any kernel (forward now, backward later) regenerates it without storing it,
// and the CPU restatement in oracle/philox.py reproduces it bit for bit.
// keep(e) <=> uniform24(e) >= p,  y = x * keep / (1 - p)     (torch.nn.functional.dropout semantics)
#pragma once
#include <stdint.h>

namespace tcavt {

struct DropoutP {
  float p;              // 0 => disabled
  float inv_keep;       // 1 / (1 - p)
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
    const unsigned int hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned int hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned int n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// keep-scales (0 or 1/(1-p)) of the four elements 4q .. 4q+3 of a site's flat index space
__device__ __forceinline__ void dropout_quad(const DropoutP& d, unsigned long long quad, float (&scale)[4]) {
  unsigned int r[4];
  unsigned int k0 = d.seed_lo, k1 = d.seed_hi;
  if (d.epoch) {  // (uniform) seed + epoch as one 64-bit sum
    const unsigned long long s = (((unsigned long long)k1 << 32) | k0) + *d.epoch;
    k0 = (unsigned int)(s & 0xffffffffu);
    k1 = (unsigned int)(s >> 32);
  }
  philox4x32_10((unsigned int)(quad & 0xffffffffu), (unsigned int)(quad >> 32), d.site, 0u, k0, k1, r);
#pragma unroll
  for (int i = 0; i < 4; ++i) scale[i] = ((float)(r[i] >> 8) * (1.0f / 16777216.0f) >= d.p) ? d.inv_keep : 0.f;
}

// single element e (uses lane e & 3 of quad e >> 2)
__device__ __forceinline__ float dropout_one(const DropoutP& d, unsigned long long e) {
  float s[4];
  dropout_quad(d, e >> 2, s);
  return s[e & 3];
}

}  // namespace tcavt
