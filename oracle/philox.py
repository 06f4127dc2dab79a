"""CPU restatement (numpy) of the dropout mask generator in csrc/philox.hpp -- TEST INFRASTRUCTURE ONLY.

Philox4x32-10 with key = 64-bit seed and counter = (octet index lo, hi, site, 0); element e of a dropout
site takes 16-bit draw e & 7 of octet e >> 3 (order w0.lo, w0.hi, w1.lo, ...); keep <=> draw >= ceil(p * 65536)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def keep_mask(n, p, seed, site):
    """Boolean keep mask of the first n elements of dropout site `site` under `seed`."""
    no = (n + 7) // 8
    o = np.arange(no, dtype=np.uint64)
    words = philox4x32_10((o & MASK).astype(np.uint32), (o >> np.uint64(32)).astype(np.uint32),
                          np.full(no, site, np.uint32), np.zeros(no, np.uint32), seed & 0xFFFFFFFF, seed >> 32)
    w = np.stack(words, axis=1)  # [octets, 4 words]
    halves = np.stack([w & np.uint32(0xFFFF), w >> np.uint32(16)], axis=2).reshape(-1)[:n]  # w0.lo, w0.hi, w1.lo, ...
    thr = int(np.ceil(np.float32(p) * np.float32(65536.0)))
    return halves >= np.uint32(min(thr, 65535))
