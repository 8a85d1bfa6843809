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

// Every global operand of an MFMA loop below is fetched in BATCHES of ~16 independent loads from clamped
// addresses before the MFMAs that consume them: a load issued right in front of its MFMA puts one memory round
// trip on every key tile (the first version: 127 us for the forward pass at N = 16, H = 4, T = 512, D = 16,
// most of it waiting).  ND = D / 16 is a template parameter so that the fragment loops have constant bounds.
#define ATT_KB 8      // key steps (of 4 keys) per batch in the rows-times-matrix products

// rows x [T] (TP = T rounded up to 16, LDS row stride TP + 4) -> 16 x D product with a [T, D] operand in
// global memory: acc over this wave's quarter of the keys, summed over the 4 waves, result to dst rows.
template <int ND>
__device__ __forceinline__ void attn_rows_times(const float* S, int SLD, const float* Bp, long ldb, int T, int TP,
                                                float* red, float* dst, long ldd, int q0, int lane, int wave) {
  constexpr int D = 16 * ND;
  const int i = lane & 15, kk = lane >> 4;
  f32x4a acc[ND];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt) acc[dt] = (f32x4a){0.f, 0.f, 0.f, 0.f};
  const int span = ((TP >> 2) + 3) & ~3;                   // keys per wave, a multiple of 4
  const int k_lo = wave * span, k_hi = min(TP, k_lo + span);
  for (int kb = k_lo; kb < k_hi; kb += 4 * ATT_KB) {
    float a[ATT_KB], b[ATT_KB][ND];
#pragma unroll
    for (int j = 0; j < ATT_KB; ++j) {
      const int k = kb + 4 * j + kk;
      const int kr = min(k, T - 1);                        // rows past T meet zero probabilities
      a[j] = (kb + 4 * j < k_hi) ? S[i * SLD + min(k, TP - 1)] : 0.0f;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) b[j][dt] = Bp[(long)kr * ldb + 16 * dt + i];
    }
#pragma unroll
    for (int j = 0; j < ATT_KB; ++j)
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j][dt], acc[dt], 0, 0, 0);
  }
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
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

// S[q][k] = sum_d A[q0 + q][d] B[k][d] for the 16 rows q0.. of A and all keys, key tiles dealt to the 4 waves;
// TC key tiles (16 / ND-ish loads) are fetched together.  MASKED: the forward pass's fill and scale.
template <int ND, bool MASKED>
__device__ __forceinline__ void attn_scores(const float* Ap, long lda, const float* Bp, long ldb,
                                            const float* attn_mask, const float* pad_mask, float scale, int n, int q0,
                                            int T, int TP, float* S, int SLD, int lane, int wave) {
  constexpr int TC = ND == 1 ? 4 : (ND == 2 ? 2 : 1);
  const int i = lane & 15, kk = lane >> 4;
  const int qa = min(q0 + i, T - 1);
  float af[4 * ND];
#pragma unroll
  for (int s = 0; s < 4 * ND; ++s) af[s] = Ap[(long)qa * lda + 4 * s + kk];
  const int ntile = TP >> 4;
  // absent masks read the A operand instead (any finite value: the result is not used)
  const bool has_am = MASKED && attn_mask != nullptr, has_pm = MASKED && pad_mask != nullptr;
  for (int kt0 = wave; kt0 < ntile; kt0 += 4 * TC) {
    float bf[TC][4 * ND], am[TC][4], pm[TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      const int k0 = min(kt0 + 4 * c, ntile - 1) << 4;
      const int ka = min(k0 + i, T - 1);
#pragma unroll
      for (int s = 0; s < 4 * ND; ++s) bf[c][s] = Bp[(long)ka * ldb + 4 * s + kk];
      if (MASKED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = min(q0 + 4 * kk + r, T - 1);
          am[c][r] = has_am ? attn_mask[(long)q * T + ka] : 1.0f;
        }
        pm[c] = has_pm ? pad_mask[(long)n * T + ka] : 1.0f;
      }
    }
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      const int kt = kt0 + 4 * c;
      f32x4a acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4 * ND; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[c][s], acc, 0, 0, 0);
      if (kt < ntile) {
        const int k = (kt << 4) + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[r];
          if (MASKED) {
            if (k >= T) v = -INFINITY;                               // not a key at all
            else {
              if (q0 + 4 * kk + r < T && (am[c][r] == 0.0f || pm[c] == 0.0f)) v = -1e3f;
              v *= scale;
            }
          }
          S[(4 * kk + r) * SLD + k] = v;
        }
      }
    }
  }
}

template <int ND>
__global__ __launch_bounds__(256) void s2g_k_attn_fwd(const float* __restrict__ Q, const float* __restrict__ K,
                                                      const float* __restrict__ V, const float* __restrict__ attn_mask,
                                                      const float* __restrict__ pad_mask, float* __restrict__ out,
                                                      float* __restrict__ P, AttnDims dm) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int D = 16 * ND;
  const int T = dm.T, TP = (T + 15) & ~15, SLD = TP + 4;
  float* S = lds;                           // [16][SLD]
  float* red = lds + 16 * SLD;              // [4][16][ATT_MAXD]
  const int q0 = blockIdx.x * 16, h = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long base = (long)n * T * dm.ld + (long)h * D;
  // ---- scores: S[q][k] = Q[q] . K[k], masked, scaled
  attn_scores<ND, true>(Q + base, dm.ld, K + base, dm.ld, attn_mask, pad_mask, dm.scale, n, q0, T, TP, S, SLD, lane, wave);
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
  attn_rows_times<ND>(S, SLD, V + base, dm.ld, T, TP, red, out + ((long)n * T + q0) * dm.ldo + (long)h * D, dm.ldo, q0,
                      lane, wave);
}

// Backward, query side.  dP = dO V^T; dS = P o (dP - rowsum(dP o P)) * scale, zero where the forward pass
// replaced the energy by the fill value; dQ = dS K.  dS is kept [N, H, T, T] for the key side.
template <int ND>
__global__ __launch_bounds__(256) void s2g_k_attn_bwd_q(const float* __restrict__ K, const float* __restrict__ V,
                                                        const float* __restrict__ attn_mask,
                                                        const float* __restrict__ pad_mask, const float* __restrict__ P,
                                                        const float* __restrict__ dO, float* __restrict__ dS,
                                                        float* __restrict__ dQ, AttnDims dm) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int D = 16 * ND;
  const int T = dm.T, TP = (T + 15) & ~15, SLD = TP + 4;
  float* S = lds;
  float* red = lds + 16 * SLD;
  const int q0 = blockIdx.x * 16, h = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long base = (long)n * T * dm.ld + (long)h * D;
  attn_scores<ND, false>(dO + (long)n * T * dm.ldo + (long)h * D, dm.ldo, V + base, dm.ld, nullptr, nullptr, 1.0f, n, q0,
                         T, TP, S, SLD, lane, wave);
  __syncthreads();
  for (int r = 4 * wave; r < 4 * wave + 4; ++r) {
    float* row = S + r * SLD;
    const int q = q0 + r;
    const bool live = q < T;
    const long prow = (((long)n * dm.H + h) * T + (live ? q : 0)) * T;
    // two passes over the row; each fetches its probabilities (and mask values) eight keys per lane at a time
    float dot = 0.0f;
    for (int jb = 0; 64 * jb < T; jb += 8) {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = P[prow + min(lane + 64 * (jb + j), T - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = lane + 64 * (jb + j);
        if (k < T) dot = fmaf(row[k], live ? pv[j] : 0.0f, dot);
      }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) dot += __shfl_xor(dot, o);
    const long arow = (long)(live ? q : 0) * T;
    for (int jb = 0; 64 * jb < TP; jb += 8) {
      float pv[8], am[8], pm[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = min(lane + 64 * (jb + j), T - 1);
        pv[j] = P[prow + k];
        am[j] = attn_mask ? attn_mask[arow + k] : 1.0f;
        pm[j] = pad_mask ? pad_mask[(long)n * T + k] : 1.0f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = lane + 64 * (jb + j);
        if (k < TP) {
          float g = 0.0f;
          if (live && k < T) {
            g = pv[j] * (row[k] - dot) * dm.scale;
            if (am[j] == 0.0f || pm[j] == 0.0f) g = 0.0f;      // the fill value does not depend on Q, K
            dS[prow + k] = g;
          }
          row[k] = g;
        }
      }
    }
  }
  __syncthreads();
  attn_rows_times<ND>(S, SLD, K + base, dm.ld, T, TP, red, dQ + ((long)n * T + q0) * dm.ld + (long)h * D, dm.ld, q0, lane,
                      wave);
}

// Backward, key side: dV[l] = sum_q P[q][l] dO[q], dK[l] = sum_q dS[q][l] Q[q], for a tile of 16 keys; the
// queries are dealt to the 4 waves and the partials summed through LDS.
template <int ND>
__global__ __launch_bounds__(256) void s2g_k_attn_bwd_kv(const float* __restrict__ Q, const float* __restrict__ P,
                                                         const float* __restrict__ dS, const float* __restrict__ dO,
                                                         float* __restrict__ dK, float* __restrict__ dV, AttnDims dm) {
  __shared__ float red[2][4 * 16 * ATT_MAXD];
  constexpr int D = 16 * ND;
  constexpr int QB = ND == 1 ? 4 : (ND == 2 ? 2 : 1);      // query steps (of 4) fetched together
  const int T = dm.T, TP = (T + 15) & ~15;
  const int l0 = blockIdx.x * 16, h = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const long pb = ((long)n * dm.H + h) * T * T;
  const float* Qp = Q + (long)n * T * dm.ld + (long)h * D;
  const float* dOp = dO + (long)n * T * dm.ldo + (long)h * D;
  const int la = min(l0 + i, T - 1);
  f32x4a av[ND], ak[ND];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt) { av[dt] = (f32x4a){0.f, 0.f, 0.f, 0.f}; ak[dt] = (f32x4a){0.f, 0.f, 0.f, 0.f}; }
  const int span = ((TP >> 2) + 3) & ~3;
  const int q_lo = wave * span, q_hi = min(TP, q_lo + span);
  for (int qb = q_lo; qb < q_hi; qb += 4 * QB) {
    float p[QB], g[QB], bo[QB][ND], bq[QB][ND];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      const int q = qb + 4 * j + kk;
      const bool ok = qb + 4 * j < q_hi && q < T;
      const int qr = min(q, T - 1);
      const float pl = P[pb + (long)qr * T + la];           // A[i][kk] = P^T[l0 + i][q + kk]
      const float gl = dS[pb + (long)qr * T + la];
      p[j] = ok ? pl : 0.0f;
      g[j] = ok ? gl : 0.0f;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        bo[j][dt] = dOp[(long)qr * dm.ldo + 16 * dt + i];
        bq[j][dt] = Qp[(long)qr * dm.ld + 16 * dt + i];
      }
    }
#pragma unroll
    for (int j = 0; j < QB; ++j)
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        av[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[j], bo[j][dt], av[dt], 0, 0, 0);
        ak[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[j], bq[j][dt], ak[dt], 0, 0, 0);
      }
  }
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
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

template <int ND>
static int attn_launch_fwd(const float* Q, const float* K, const float* V, const float* attn_mask, const float* pad_mask,
                           float* out, float* P, AttnDims dm, hipStream_t stream) {
  const size_t lds = attn_lds_bytes(dm.T);
  static S2GLdsCap lds_cap;
  if (s2g_lds_cap(lds_cap, reinterpret_cast<const void*>(&s2g_k_attn_fwd<ND>), lds) != 0) return -2;
  hipLaunchKernelGGL((s2g_k_attn_fwd<ND>), dim3((dm.T + 15) / 16, dm.H, dm.N), dim3(256), lds, stream, Q, K, V, attn_mask,
                     pad_mask, out, P, dm);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int ND>
static int attn_launch_bwd(const float* Q, const float* K, const float* V, const float* attn_mask, const float* pad_mask,
                           const float* P, const float* dO, float* dS, float* dQ, float* dK, float* dV, AttnDims dm,
                           hipStream_t stream) {
  const size_t lds = attn_lds_bytes(dm.T);
  static S2GLdsCap lds_cap;
  if (s2g_lds_cap(lds_cap, reinterpret_cast<const void*>(&s2g_k_attn_bwd_q<ND>), lds) != 0) return -2;
  const dim3 grid((dm.T + 15) / 16, dm.H, dm.N);
  hipLaunchKernelGGL((s2g_k_attn_bwd_q<ND>), grid, dim3(256), lds, stream, K, V, attn_mask, pad_mask, P, dO, dS, dQ, dm);
  hipLaunchKernelGGL((s2g_k_attn_bwd_kv<ND>), grid, dim3(256), 0, stream, Q, P, dS, dO, dK, dV, dm);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int s2g_launch_attn_fwd(const float* Q, const float* K, const float* V, const float* attn_mask,
                                   const float* pad_mask, float* out, float* P, AttnDims dm, hipStream_t stream) {
  switch (dm.D >> 4) {
    case 1: return attn_launch_fwd<1>(Q, K, V, attn_mask, pad_mask, out, P, dm, stream);
    case 2: return attn_launch_fwd<2>(Q, K, V, attn_mask, pad_mask, out, P, dm, stream);
    case 3: return attn_launch_fwd<3>(Q, K, V, attn_mask, pad_mask, out, P, dm, stream);
    case 4: return attn_launch_fwd<4>(Q, K, V, attn_mask, pad_mask, out, P, dm, stream);
  }
  return -1;
}

extern "C" int s2g_launch_attn_bwd(const float* Q, const float* K, const float* V, const float* attn_mask,
                                   const float* pad_mask, const float* P, const float* dO, float* dS, float* dQ, float* dK,
                                   float* dV, AttnDims dm, hipStream_t stream) {
  switch (dm.D >> 4) {
    case 1: return attn_launch_bwd<1>(Q, K, V, attn_mask, pad_mask, P, dO, dS, dQ, dK, dV, dm, stream);
    case 2: return attn_launch_bwd<2>(Q, K, V, attn_mask, pad_mask, P, dO, dS, dQ, dK, dV, dm, stream);
    case 3: return attn_launch_bwd<3>(Q, K, V, attn_mask, pad_mask, P, dO, dS, dQ, dK, dV, dm, stream);
    case 4: return attn_launch_bwd<4>(Q, K, V, attn_mask, pad_mask, P, dO, dS, dQ, dK, dV, dm, stream);
  }
  return -1;
}
