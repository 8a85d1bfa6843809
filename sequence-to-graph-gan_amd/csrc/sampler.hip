// Fused Gumbel-softmax sampler over the vocabulary axis (reference: CustomSoftmax,
// wgan_gp.py:36-104).  One 64-lane wavefront per row of V logits (V = 175 -> 3 values per
// lane); max and sum are reduced with DPP inside each 16-lane row and two cross-row shuffles.
#include "s2g_common.h"
#include "rng.hip"

#define MAXPER 8   // supports V <= 512

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<DPP_QUAD(1, 0, 3, 2)>(v));
  v = fmaxf(v, dpp_mov<DPP_QUAD(2, 3, 0, 1)>(v));
  v = fmaxf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(v));
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// mode 0: gumbel (wgan_gp.py:56-86)   y = softmax((logits + g(U)) / tau)
// mode 1: relaxed (wgan_gp.py:94-104) y = softmax(logits / tau)
// mode 2: plain   (wgan_gp.py:54)     y = softmax(logits)
// hard: y_out = one_hot(argmax y) (first maximal index, as torch.argmax); y_soft keeps the soft
// sample, which is what the straight-through backward differentiates (wgan_gp.py:80-84).
// bias (optional, [V * rows_per_bias-cycle]) is added to the logits first: lets the generator's
// last Linear skip its own bias pass (bias index = (row % bias_rows) * V + j).
extern "C" __global__ __launch_bounds__(256) void s2g_k_gumbel_softmax_fwd(
    const float* __restrict__ logits, const float* __restrict__ U, const float* __restrict__ bias,
    int bias_rows, float* __restrict__ y_out, float* __restrict__ y_soft, long rows, int V,
    float inv_tau, int mode, int hard) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* lg = logits + row * V;
  const float* bs = bias ? bias + (row % bias_rows) * V : nullptr;
  float z[MAXPER];
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < MAXPER; ++e) {
    int j = lane + 64 * e;
    z[e] = -INFINITY;
    if (j < V) {
      float x = lg[j];
      if (bs) x += bs[j];
      if (mode == 0) {
        x += gumbel_noise(U[row * V + j]);
      }
      z[e] = (mode == 2) ? x : x * inv_tau;
      mx = fmaxf(mx, z[e]);
    }
  }
  mx = wave_max(mx);
  float sum = 0.0f;
#pragma unroll
  for (int e = 0; e < MAXPER; ++e) {
    int j = lane + 64 * e;
    z[e] = (j < V) ? exp_nonpos(z[e] - mx) : 0.0f;
    sum += z[e];
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  int best = 0x7fffffff;
  if (hard) {
    // soft probabilities can tie after rounding; torch takes the first index of the maximum
    float bv = -1.0f;
#pragma unroll
    for (int e = 0; e < MAXPER; ++e) {
      int j = lane + 64 * e;
      float p = z[e] * inv;
      if (j < V && p > bv) { bv = p; best = j; }
    }
    float wm = wave_max(bv);
    int cand = (bv == wm) ? best : 0x7fffffff;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) cand = min(cand, __shfl_xor(cand, o));
    best = cand;
  }
#pragma unroll
  for (int e = 0; e < MAXPER; ++e) {
    int j = lane + 64 * e;
    if (j < V) {
      float p = z[e] * inv;
      if (y_soft) y_soft[row * V + j] = p;
      y_out[row * V + j] = hard ? (j == best ? 1.0f : 0.0f) : p;
    }
  }
}

// dlogits = s * y * (dy - <dy, y>) with s = 1/tau (modes 0,1) or 1 (mode 2); y = soft sample.
extern "C" __global__ __launch_bounds__(256) void s2g_k_gumbel_softmax_bwd(
    const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dlogits, long rows,
    int V, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float yv[MAXPER], dv[MAXPER];
  float dot = 0.0f;
#pragma unroll
  for (int e = 0; e < MAXPER; ++e) {
    int j = lane + 64 * e;
    yv[e] = 0.f; dv[e] = 0.f;
    if (j < V) { yv[e] = y[row * V + j]; dv[e] = dy[row * V + j]; }
    dot = fmaf(yv[e], dv[e], dot);
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int e = 0; e < MAXPER; ++e) {
    int j = lane + 64 * e;
    if (j < V) dlogits[row * V + j] = scale * yv[e] * (dv[e] - dot);
  }
}

// ---- 16-byte form for V <= 256 (the 175-way vocabulary): lane l owns columns 4l..4l+3 of its row and
// moves them with one 16-byte access (rows of 175 floats are only 4-byte aligned; the memory pipeline
// takes 16-byte accesses at any 4-byte address), so a row is 44 requests instead of 175.  With RNG the
// Gumbel uniforms come from Philox (one call per lane = its four columns) instead of a [rows, V] tensor
// of draws: no 4 bytes per element written by a generator kernel and read back here.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void load4(const float* p, int nv, float fill, float out[4]) {
  if (nv >= 4) {
    const f4u v = *reinterpret_cast<const f4u*>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = k < nv ? p[k] : fill;
  }
}
__device__ __forceinline__ void store4(float* p, int nv, const float v[4]) {
  if (nv >= 4) {
    *reinterpret_cast<f4u*>(p) = (f4u){v[0], v[1], v[2], v[3]};
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nv) p[k] = v[k];
  }
}

// Rows per wave (all loads of a wave's rows are issued before its first reduction).  Measured at 32768 rows
// of 175, caches cold: one row per wave 23 us, four rows per wave 30 us (fewer, longer waves hide less), so
// the many-short-waves form stays.
#ifndef SAMPLER_ROWS
#define SAMPLER_ROWS 1
#endif

// lowest column holding the wave's maximum: lanes own ascending columns, so the first lane whose best value
// equals the maximum holds it (one ballot and one cross-lane read instead of a six-step shuffle tree)
__device__ __forceinline__ int first_best(float bv, int best, float wm) {
  const unsigned long long m = __ballot(bv == wm);
  const int src = m ? __builtin_ctzll(m) : 0;
  return __shfl(best, src);
}

template <bool RNG>
__global__ __launch_bounds__(256) void s2g_k_gumbel_softmax_fwd_v4(
    const float* __restrict__ logits, const float* __restrict__ U, const float* __restrict__ bias, int bias_rows,
    float* __restrict__ y_out, float* __restrict__ y_soft, long rows, int V, float inv_tau, int mode, int hard,
    unsigned long long seed, unsigned long long offset) {
  constexpr int SR = SAMPLER_ROWS;
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * SR;
  if (row0 >= rows) return;
  const int j0 = 4 * lane, nv = V - j0;          // nv >= 4: four live columns, 1..3: the row's tail, <= 0: none
  float z[SR][4], u[SR][4];
#pragma unroll
  for (int r = 0; r < SR; ++r) {
    const long row = row0 + r;
    const int n = row < rows ? nv : 0;
    load4(logits + row * V + j0, n, -INFINITY, z[r]);
    if (mode == 0 && !RNG) load4(U + row * V + j0, n, 0.5f, u[r]);
  }
  if (bias) {
#pragma unroll
    for (int r = 0; r < SR; ++r) {
      const long row = row0 + r;
      float b[4];
      load4(bias + (row % bias_rows) * V + j0, row < rows ? nv : 0, 0.0f, b);
#pragma unroll
      for (int k = 0; k < 4; ++k) z[r][k] += b[k];
    }
  }
#pragma unroll
  for (int r = 0; r < SR; ++r) {
    const long row = row0 + r;
    if (row >= rows) break;                        // wave-uniform
    const long base = row * V + j0;
    if (mode == 0) {
      if (RNG) {
        const Philox4 q = philox4x32_10((unsigned long long)row * 64 + lane, offset, seed);
        u[r][0] = u01(q.x); u[r][1] = u01(q.y); u[r][2] = u01(q.z); u[r][3] = u01(q.w);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) z[r][k] += gumbel_noise(u[r][k]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      z[r][k] = k < nv ? ((mode == 2) ? z[r][k] : z[r][k] * inv_tau) : -INFINITY;
      mx = fmaxf(mx, z[r][k]);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      z[r][k] = k < nv ? exp_nonpos(z[r][k] - mx) : 0.0f;
      sum += z[r][k];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float pr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pr[k] = z[r][k] * inv;
    if (y_soft) store4(y_soft + base, nv, pr);
    if (hard) {
      float bv = -1.0f;
      int best = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nv && pr[k] > bv) { bv = pr[k]; best = j0 + k; }
      const int cand = first_best(bv, best, wave_max(bv));
#pragma unroll
      for (int k = 0; k < 4; ++k) pr[k] = (j0 + k == cand) ? 1.0f : 0.0f;
    }
    store4(y_out + base, nv, pr);
  }
}

extern "C" __global__ __launch_bounds__(256) void s2g_k_gumbel_softmax_bwd_v4(
    const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dlogits, long rows, int V,
    float scale) {
  constexpr int SR = SAMPLER_ROWS;
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * SR;
  if (row0 >= rows) return;
  const int j0 = 4 * lane, nv = V - j0;
  float yv[SR][4], dv[SR][4];
#pragma unroll
  for (int r = 0; r < SR; ++r) {
    const long row = row0 + r;
    const int n = row < rows ? nv : 0;
    load4(y + row * V + j0, n, 0.0f, yv[r]);
    load4(dy + row * V + j0, n, 0.0f, dv[r]);
  }
#pragma unroll
  for (int r = 0; r < SR; ++r) {
    const long row = row0 + r;
    if (row >= rows) break;                        // wave-uniform
    float dot = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) dot = fmaf(yv[r][k], dv[r][k], dot);
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < 4; ++k) dv[r][k] = scale * yv[r][k] * (dv[r][k] - dot);
    store4(dlogits + row * V + j0, nv, dv[r]);
  }
}

// tokens[row] = argmax_j y[row][j]  (generate_fake_data.py:66), first maximal index.
extern "C" __global__ __launch_bounds__(256) void s2g_k_argmax_rows(const float* __restrict__ y,
                                                                    long long* __restrict__ tokens,
                                                                    long rows, int V) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float bv = -INFINITY;
  int best = 0x7fffffff;
  for (int j = lane; j < V; j += 64) {
    float p = y[row * V + j];
    if (p > bv) { bv = p; best = j; }
  }
  float wm = wave_max(bv);
  int cand = (bv == wm) ? best : 0x7fffffff;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) cand = min(cand, __shfl_xor(cand, o));
  if (lane == 0) tokens[row] = cand;
}
