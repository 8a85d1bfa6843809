// Multi-head attention core of the reference's MaskedMultiheadAttention.forward (transformer.py:243-274)
// and its autograd, on exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).  The reference's conventions are kept as
// they are written, not as attention is usually defined:
//   energy[n,h,q,k] = sum_d Q[n,q,h,d] K[n,k,h,d]
//   energy <- -1e3 where attn_mask[q,k] == 0            (a FINITE fill, transformer.py:257)
//   energy <- -1e3 where key_padding_mask[n,k] == 0      (:264; the mask helper marks PAD positions with 1,
//                                                        so this fills the NON-pad keys, transformer.py:21-34)
//   attention = softmax(energy / sqrt(embed_size), k)    (scaled by the EMBEDDING size, not the head size, :267)
//   out[n,q,h,:] = sum_k attention[n,h,q,k] V[n,k,h,:]
// Q, K, V and out are [N, T, H, D] views with a row stride (the three chunks of the fused in-projection
// [N, T, 3E], transformer.py:235-241); P = attention is kept [N, H, T, T] for the backward pass, as the
// reference's autograd keeps it.
//
// One workgroup of 4 waves per (16 queries, head, batch entry).  Scores of the 16 queries against all keys
// live in LDS (16 x T fp32: 32 KB at T = 512); wave w owns key tiles w, w+4, ... for the score MFMAs, rows
// 4w..4w+3 for the softmax, and a quarter of the keys for the P V product (partials summed through LDS).
#pragma once
#include "s2g_common.h"

typedef float f32x4a __attribute__((ext_vector_type(4)));
#define ATT_MAXD 64

struct AttnDims {
  int N, H, T, D;
  long ld;          // row stride (floats) of Q / K / V / their gradients: element (n, t, h, d) at (n T + t) ld + h D + d
  long ldo;         // row stride of out / d(out)
  float scale;      // 1 / sqrt(embed_size)
};

__device__ __forceinline__ bool attn_masked(const float* attn_mask, const float* pad_mask, int n, int q, int k, int T) {
  return (attn_mask && attn_mask[(long)q * T + k] == 0.0f) || (pad_mask && pad_mask[(long)n * T + k] == 0.0f);
}

// rows x [T] (TP = T rounded up to 16, LDS row stride TP + 4) -> 16 x D product with a [T, D] operand in
// global memory: acc over this wave's quarter of the keys, summed over the 4 waves, result to dst rows.
__device__ __forceinline__ void attn_rows_times(const float* S, int SLD, const float* Bp, long ldb, int T, int TP, int D,
                                                float* red, float* dst, long ldd, int q0, int lane, int wave) {
  const int i = lane & 15, kk = lane >> 4;
  const int nd = D >> 4;
  f32x4a acc[ATT_MAXD / 16];
#pragma unroll
  for (int dt = 0; dt < ATT_MAXD / 16; ++dt) acc[dt] = (f32x4a){0.f, 0.f, 0.f, 0.f};
  const int span = ((TP >> 2) + 3) & ~3;                   // keys per wave, a multiple of 4
  const int k_lo = wave * span, k_hi = min(TP, k_lo + span);
  for (int k = k_lo; k < k_hi; k += 4) {
    const float a = S[i * SLD + k + kk];
    const int kr = min(k + kk, T - 1);                     // rows past T meet zero probabilities
#pragma unroll
    for (int dt = 0; dt < ATT_MAXD / 16; ++dt)
      if (dt < nd) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bp[(long)kr * ldb + 16 * dt + i], acc[dt], 0, 0, 0);
  }
#pragma unroll
  for (int dt = 0; dt < ATT_MAXD / 16; ++dt)
    if (dt < nd)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * kk + r) * ATT_MAXD + 16 * dt + i] = acc[dt][r];
  __syncthreads();
  for (int e = threadIdx.x; e < 16 * D; e += 256) {
    const int r = e / D, d = e - r * D;
    if (q0 + r < T)
      dst[(long)r * ldd + d] = (red[r * ATT_MAXD + d] + red[(16 + r) * ATT_MAXD + d]) +
                               (red[(32 + r) * ATT_MAXD + d] + red[(48 + r) * ATT_MAXD + d]);
  }
}

extern "C" __global__ __launch_bounds__(256) void s2g_k_attn_fwd(const float* __restrict__ Q, const float* __restrict__ K,
                                                                 const float* __restrict__ V,
                                                                 const float* __restrict__ attn_mask,
                                                                 const float* __restrict__ pad_mask,
                                                                 float* __restrict__ out, float* __restrict__ P,
                                                                 AttnDims dm) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int T = dm.T, D = dm.D, TP = (T + 15) & ~15, SLD = TP + 4;
  float* S = lds;                           // [16][SLD]
  float* red = lds + 16 * SLD;              // [4][16][ATT_MAXD]
  const int q0 = blockIdx.x * 16, h = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const long base = (long)n * T * dm.ld + (long)h * D;
  const float* Qp = Q + base;
  const float* Kp = K + base;
  const float* Vp = V + base;
  // ---- scores: S[q][k] = Q[q] . K[k], masked, scaled
  const int qa = min(q0 + i, T - 1);
  float qf[ATT_MAXD / 4];
#pragma unroll
  for (int s = 0; s < ATT_MAXD / 4; ++s) qf[s] = (4 * s < D) ? Qp[(long)qa * dm.ld + 4 * s + kk] : 0.0f;
  for (int kt = wave; kt < (TP >> 4); kt += 4) {
    const int k0 = kt << 4;
    const int ka = min(k0 + i, T - 1);
    f32x4a acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < ATT_MAXD / 4; ++s)
      if (4 * s < D) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], Kp[(long)ka * dm.ld + 4 * s + kk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + 4 * kk + r, k = k0 + i;
      float v = acc[r];
      if (k >= T) v = -INFINITY;                                 // not a key at all
      else {
        if (q < T && attn_masked(attn_mask, pad_mask, n, q, k, T)) v = -1e3f;
        v *= dm.scale;
      }
      S[(4 * kk + r) * SLD + k] = v;
    }
  }
  __syncthreads();
  // ---- softmax over the keys, rows 4 wave .. 4 wave + 3
  for (int r = 4 * wave; r < 4 * wave + 4; ++r) {
    float* row = S + r * SLD;
    float mx = -INFINITY;
    for (int k = lane; k < TP; k += 64) mx = fmaxf(mx, row[k]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.0f;
    for (int k = lane; k < TP; k += 64) {
      const float e = expf(row[k] - mx);
      row[k] = e;
      sum += e;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    const int q = q0 + r;
    float* prow = (P && q < T) ? P + (((long)n * dm.H + h) * T + q) * T : nullptr;
    for (int k = lane; k < TP; k += 64) {
      const float p = row[k] * inv;
      row[k] = p;
      if (prow && k < T) prow[k] = p;
    }
  }
  __syncthreads();
  // ---- out[q] = sum_k P[q][k] V[k]
  attn_rows_times(S, SLD, Vp, dm.ld, T, TP, D, red, out + ((long)n * T + q0) * dm.ldo + (long)h * D, dm.ldo, q0, lane, wave);
}

// Backward, query side.  dP = dO V^T; dS = P o (dP - rowsum(dP o P)) * scale, zero where the forward pass
// replaced the energy by the fill value; dQ = dS K.  dS is kept [N, H, T, T] for the key side.
extern "C" __global__ __launch_bounds__(256) void s2g_k_attn_bwd_q(const float* __restrict__ K, const float* __restrict__ V,
                                                                   const float* __restrict__ attn_mask,
                                                                   const float* __restrict__ pad_mask,
                                                                   const float* __restrict__ P,
                                                                   const float* __restrict__ dO, float* __restrict__ dS,
                                                                   float* __restrict__ dQ, AttnDims dm) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int T = dm.T, D = dm.D, TP = (T + 15) & ~15, SLD = TP + 4;
  float* S = lds;
  float* red = lds + 16 * SLD;
  const int q0 = blockIdx.x * 16, h = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const long base = (long)n * T * dm.ld + (long)h * D;
  const float* Kp = K + base;
  const float* Vp = V + base;
  const float* dOp = dO + (long)n * T * dm.ldo + (long)h * D;
  const int qa = min(q0 + i, T - 1);
  float of[ATT_MAXD / 4];
#pragma unroll
  for (int s = 0; s < ATT_MAXD / 4; ++s) of[s] = (4 * s < D) ? dOp[(long)qa * dm.ldo + 4 * s + kk] : 0.0f;
  for (int kt = wave; kt < (TP >> 4); kt += 4) {
    const int k0 = kt << 4;
    const int ka = min(k0 + i, T - 1);
    f32x4a acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < ATT_MAXD / 4; ++s)
      if (4 * s < D) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(of[s], Vp[(long)ka * dm.ld + 4 * s + kk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(4 * kk + r) * SLD + k0 + i] = acc[r];
  }
  __syncthreads();
  for (int r = 4 * wave; r < 4 * wave + 4; ++r) {
    float* row = S + r * SLD;
    const int q = q0 + r;
    const bool live = q < T;
    const long prow = (((long)n * dm.H + h) * T + (live ? q : 0)) * T;
    float dot = 0.0f;
    for (int k = lane; k < T; k += 64) dot = fmaf(row[k], live ? P[prow + k] : 0.0f, dot);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) dot += __shfl_xor(dot, o);
    for (int k = lane; k < TP; k += 64) {
      float g = 0.0f;
      if (live && k < T) {
        g = P[prow + k] * (row[k] - dot) * dm.scale;
        if (attn_masked(attn_mask, pad_mask, n, q, k, T)) g = 0.0f;      // the fill value does not depend on Q, K
        dS[prow + k] = g;
      }
      row[k] = g;
    }
  }
  __syncthreads();
  attn_rows_times(S, SLD, Kp, dm.ld, T, TP, D, red, dQ + ((long)n * T + q0) * dm.ld + (long)h * D, dm.ld, q0, lane, wave);
}

// Backward, key side: dV[l] = sum_q P[q][l] dO[q], dK[l] = sum_q dS[q][l] Q[q], for a tile of 16 keys; the
// queries are dealt to the 4 waves and the partials summed through LDS.
extern "C" __global__ __launch_bounds__(256) void s2g_k_attn_bwd_kv(const float* __restrict__ Q, const float* __restrict__ P,
                                                                    const float* __restrict__ dS,
                                                                    const float* __restrict__ dO, float* __restrict__ dK,
                                                                    float* __restrict__ dV, AttnDims dm) {
  __shared__ float red[2][4 * 16 * ATT_MAXD];
  const int T = dm.T, D = dm.D, TP = (T + 15) & ~15;
  const int l0 = blockIdx.x * 16, h = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int nd = D >> 4;
  const long pb = ((long)n * dm.H + h) * T * T;
  const float* Qp = Q + (long)n * T * dm.ld + (long)h * D;
  const float* dOp = dO + (long)n * T * dm.ldo + (long)h * D;
  const int la = min(l0 + i, T - 1);
  f32x4a av[ATT_MAXD / 16], ak[ATT_MAXD / 16];
#pragma unroll
  for (int dt = 0; dt < ATT_MAXD / 16; ++dt) { av[dt] = (f32x4a){0.f, 0.f, 0.f, 0.f}; ak[dt] = (f32x4a){0.f, 0.f, 0.f, 0.f}; }
  const int span = ((TP >> 2) + 3) & ~3;
  const int q_lo = wave * span, q_hi = min(TP, q_lo + span);
  for (int q = q_lo; q < q_hi; q += 4) {
    const bool ok = q + kk < T;
    const int qr = ok ? q + kk : 0;
    const float p = ok ? P[pb + (long)qr * T + la] : 0.0f;           // A[i][kk] = P^T[l0 + i][q + kk]
    const float g = ok ? dS[pb + (long)qr * T + la] : 0.0f;
#pragma unroll
    for (int dt = 0; dt < ATT_MAXD / 16; ++dt)
      if (dt < nd) {
        av[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, dOp[(long)qr * dm.ldo + 16 * dt + i], av[dt], 0, 0, 0);
        ak[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(g, Qp[(long)qr * dm.ld + 16 * dt + i], ak[dt], 0, 0, 0);
      }
  }
#pragma unroll
  for (int dt = 0; dt < ATT_MAXD / 16; ++dt)
    if (dt < nd)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[0][(wave * 16 + 4 * kk + r) * ATT_MAXD + 16 * dt + i] = av[dt][r];
        red[1][(wave * 16 + 4 * kk + r) * ATT_MAXD + 16 * dt + i] = ak[dt][r];
      }
  __syncthreads();
  const long ob = ((long)n * T + l0) * dm.ld + (long)h * D;
  for (int e = tid; e < 16 * D; e += 256) {
    const int r = e / D, d = e - r * D;
    if (l0 + r < T) {
      dV[ob + (long)r * dm.ld + d] = (red[0][r * ATT_MAXD + d] + red[0][(16 + r) * ATT_MAXD + d]) +
                                     (red[0][(32 + r) * ATT_MAXD + d] + red[0][(48 + r) * ATT_MAXD + d]);
      dK[ob + (long)r * dm.ld + d] = (red[1][r * ATT_MAXD + d] + red[1][(16 + r) * ATT_MAXD + d]) +
                                     (red[1][(32 + r) * ATT_MAXD + d] + red[1][(48 + r) * ATT_MAXD + d]);
    }
  }
}

static inline size_t attn_lds_bytes(int T) { return ((size_t)16 * (((T + 15) & ~15) + 4) + 4 * 16 * ATT_MAXD) * sizeof(float); }

extern "C" int s2g_launch_attn_fwd(const float* Q, const float* K, const float* V, const float* attn_mask,
                                   const float* pad_mask, float* out, float* P, AttnDims dm, hipStream_t stream) {
  const size_t lds = attn_lds_bytes(dm.T);
  static size_t set_for = 0;
  if (lds > set_for) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&s2g_k_attn_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return -2;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&s2g_k_attn_bwd_q), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return -2;
    set_for = lds;
  }
  hipLaunchKernelGGL(s2g_k_attn_fwd, dim3((dm.T + 15) / 16, dm.H, dm.N), dim3(256), lds, stream, Q, K, V, attn_mask,
                     pad_mask, out, P, dm);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int s2g_launch_attn_bwd(const float* Q, const float* K, const float* V, const float* attn_mask,
                                   const float* pad_mask, const float* P, const float* dO, float* dS, float* dQ, float* dK,
                                   float* dV, AttnDims dm, hipStream_t stream) {
  const size_t lds = attn_lds_bytes(dm.T);
  static size_t set_for = 0;
  if (lds > set_for) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&s2g_k_attn_bwd_q), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return -2;
    set_for = lds;
  }
  const dim3 grid((dm.T + 15) / 16, dm.H, dm.N);
  hipLaunchKernelGGL(s2g_k_attn_bwd_q, grid, dim3(256), lds, stream, K, V, attn_mask, pad_mask, P, dO, dS, dQ, dm);
  hipLaunchKernelGGL(s2g_k_attn_bwd_kv, grid, dim3(256), 0, stream, Q, P, dS, dO, dK, dV, dm);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
