// Counter-based random numbers for the draws of the training step (reference: torch.randn for z,
// wgan_gp.py:291; torch.rand for the Gumbel uniforms, :58, and alpha, :228; the Bernoulli masks of
// nn.LSTM(dropout=0.3), :189).  Philox4x32-10 keyed by a 64-bit seed; the 128-bit counter is
// (index of the 4-value group, 64-bit offset of the draw), so a draw is a pure function of
// (seed, offset, position): no generator state lives on the device and launches can be replayed.
// These draws replace the reference's CPU generator stream, which no device generator can reproduce;
// parity tests inject the noise instead (every consumer also accepts a tensor of draws).
#pragma once
#include "s2g_common.h"

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

struct Philox4 { unsigned x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long group, unsigned long long offset,
                                                 unsigned long long seed) {
  unsigned c0 = (unsigned)group, c1 = (unsigned)(group >> 32), c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(PHILOX_M0, c0), lo0 = PHILOX_M0 * c0;
    const unsigned hi1 = __umulhi(PHILOX_M1, c2), lo1 = PHILOX_M1 * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += PHILOX_W0; k1 += PHILOX_W1;
  }
  return Philox4{c0, c1, c2, c3};
}

// uniform in [0, 1) with 24 random bits (the float resolution torch.rand uses)
__device__ __forceinline__ float u01(unsigned r) { return (float)(r >> 8) * 5.9604644775390625e-08f; }

// out[i] = uniform [0,1) (kind 0) or standard normal (kind 1, Box-Muller on pairs of uniforms)
extern "C" __global__ __launch_bounds__(256) void s2g_k_rng_fill(float* __restrict__ out, long n, int kind,
                                                                 unsigned long long seed, unsigned long long offset) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (4 * g >= n) return;
  const Philox4 r = philox4x32_10((unsigned long long)g, offset, seed);
  float v[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
  if (kind == 1) {
    // (1 - u) lies in (0, 1]: the logarithm is finite
    const float r0 = sqrtf(-2.0f * logf(1.0f - v[0])), r1 = sqrtf(-2.0f * logf(1.0f - v[2]));
    const float a0 = 6.283185307179586f * v[1], a1 = 6.283185307179586f * v[3];
    v[0] = r0 * cosf(a0); v[1] = r0 * sinf(a0); v[2] = r1 * cosf(a1); v[3] = r1 * sinf(a1);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (4 * g + k < n) out[4 * g + k] = v[k];
}

// Dropout multipliers of an LSTM layer's outputs, mask[rows][width]: (u >= p) / (1 - p) in columns
// 0..live-1, zero in the padding columns live..width-1 (width % 4 == 0; the critic's H = 50 rows are padded
// to 52).  One Philox call per 16-byte store.
extern "C" __global__ __launch_bounds__(256) void s2g_k_dropout_mask_rng(float* __restrict__ mask, long n4, float p,
                                                                         float inv_keep, unsigned long long seed,
                                                                         unsigned long long offset, int w4, int live) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n4) return;
  const Philox4 r = philox4x32_10((unsigned long long)g, offset, seed);
  const int c0 = 4 * (int)(g % w4);
  float4 m;
  m.x = (c0 < live && u01(r.x) >= p) ? inv_keep : 0.0f;
  m.y = (c0 + 1 < live && u01(r.y) >= p) ? inv_keep : 0.0f;
  m.z = (c0 + 2 < live && u01(r.z) >= p) ? inv_keep : 0.0f;
  m.w = (c0 + 3 < live && u01(r.w) >= p) ? inv_keep : 0.0f;
  reinterpret_cast<float4*>(mask)[g] = m;
}
