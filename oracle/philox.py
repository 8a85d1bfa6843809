"""TEST INFRASTRUCTURE: numpy restatement of Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random
numbers: as easy as 1, 2, 3", SC'11; the Random123 constants), the counter-based generator the library's
kernels use for device-side draws (csrc/rng.hip).  The reference draws on torch's CPU generator
(wgan_gp.py:58, 228, 291), which has no device counterpart, so this file pins the library's OWN stream:
counter = (group index, 64-bit offset), key = 64-bit seed, uniform = (word >> 8) * 2^-24."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(group, offset: int, seed: int):
    """group: integer array of counters (low 64 bits of the 128-bit counter); returns uint32 [n, 4]."""
    group = np.asarray(group, dtype=np.uint64)
    c0, c1 = group & MASK, group >> np.uint64(32)
    c2 = np.full_like(c0, offset & 0xFFFFFFFF)
    c3 = np.full_like(c0, (offset >> 32) & 0xFFFFFFFF)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=1).astype(np.uint32)


def uniforms(words):
    """uint32 words -> float32 uniforms in [0, 1) with 24 random bits."""
    return ((words >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def fill_uniform(n: int, offset: int, seed: int):
    """what s2g_rng_fill(kind 0) writes: element 4g + k = word k of group g"""
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    return uniforms(philox4x32_10(g, offset, seed)).reshape(-1)[:n]


def sampler_uniforms(rows: int, V: int, offset: int, seed: int):
    """the uniforms s2g_gumbel_softmax_fwd_rng uses: element (row, 4l + k) = word k of group row * 64 + l"""
    g = (np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(64) + np.arange(64, dtype=np.uint64)[None, :]).reshape(-1)
    u = uniforms(philox4x32_10(g, offset, seed)).reshape(rows, 256)
    return u[:, :V]
