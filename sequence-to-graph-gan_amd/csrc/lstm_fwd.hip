// Forward-in-time sweeps of the critic's 2-layer LSTM (reference: nn.LSTM(175, 50, 2) inside
// Discriminator.forward, wgan_gp.py:189,217): sweep A (primal states) and sweep C (tangent
// states along an input direction, used by the gradient penalty's second-order term,
// wgan_gp.py:247-256).
//
// Decomposition: ONE workgroup per sequence, both layers resident, exact fp32 on the VALU.
//   waves 0-3: layer 0, step tau          lane = 4*unit + s, s = K-slice of 13 hidden inputs
//   waves 4-7: layer 1, step tau-1        (layer 1 trails layer 0 by one step, so the two
//                                          layers' steps overlap and one barrier per step suffices)
// A lane keeps its 4 gate rows x 13 (26 for layer 1) recurrent weights in registers, fetches
// its 13-value slice of h from LDS (16-byte reads), reduces the 4 partial sums over the quad
// with DPP, applies the gate nonlinearity of gate `s`, and the quad shares the four gates by
// DPP broadcasts for the cell update.  Per-step inputs (input projection, dropout mask, saved
// gates) are prefetched PF steps ahead into registers so no step waits on HBM.
#include "s2g_common.h"

#define PF 4

__device__ __forceinline__ void load_slice13(const float* buf, int s, float hv[13]) {
  const float4* p = reinterpret_cast<const float4*>(buf + 16 * s);
  float4 a = p[0], b = p[1], c = p[2];
  float d = buf[16 * s + 12];
  hv[0] = a.x; hv[1] = a.y; hv[2] = a.z; hv[3] = a.w;
  hv[4] = b.x; hv[5] = b.y; hv[6] = b.z; hv[7] = b.w;
  hv[8] = c.x; hv[9] = c.y; hv[10] = c.z; hv[11] = c.w;
  hv[12] = d;
}

__device__ __forceinline__ float sel4(int s, float a0, float a1, float a2, float a3) {
  float lo = (s & 1) ? a1 : a0;
  float hi = (s & 1) ? a3 : a2;
  return (s & 2) ? hi : lo;
}

// Loads the 4 gate rows (torch layout rows q*H+u) x 13-column slice `s` of a [4H, H] matrix.
__device__ __forceinline__ void load_w_slice(const float* W, int u, int s, bool valid, float w[4][13]) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      int k = 13 * s + j;
      w[q][j] = (valid && k < S2G_H) ? W[(q * S2G_H + u) * S2G_H + k] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// Sweep A.  Saves per (row = ws_row + t): gates{0,1}[row][4u+q] = activated gate q of unit u;
// cell0[row][4u+{0,1,2,3}] = (c, tanh c, h, mask*h); cell1 likewise with the 4th slot 0;
// hT[n][u] = final hidden state of layer 1.
// ------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(512) void s2g_k_lstm2_fwd(
    const S2GSeq* __restrict__ seqs, const int* __restrict__ tokens, const float* __restrict__ WT0,
    const float* __restrict__ P, const float* __restrict__ mask, const float* __restrict__ b0,
    const float* __restrict__ b1, const float* __restrict__ whh0, const float* __restrict__ wih1,
    const float* __restrict__ whh1, float* __restrict__ gates0, float* __restrict__ cell0,
    float* __restrict__ gates1, float* __restrict__ cell1, float* __restrict__ hT) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* h0buf = reinterpret_cast<float*>(smem);   // [2][64]
  float* ybuf = h0buf + 128;                        // [2][64]
  float* h1buf = ybuf + 128;                        // [2][64]
  int* tok_lds = reinterpret_cast<int*>(h1buf + 128);

  const int n = blockIdx.x;
  const S2GSeq sq = seqs[n];
  const int tid = threadIdx.x;
  const int grp = tid >> 8;            // 0: layer 0, 1: layer 1 (wave-uniform)
  const int l = tid & 255;
  const int u = l >> 2, s = l & 3;
  const bool valid = u < S2G_H;
  const int uc = valid ? u : S2G_H - 1;
  const int len = sq.len;
  const bool has_tok = sq.tok_off >= 0 && sq.ca != 0.0f;
  const bool has_p = sq.p_off >= 0 && sq.cp != 0.0f;
  const bool has_mask = sq.mask_off >= 0;

  for (int i = tid; i < 384; i += 512) h0buf[i] = 0.0f;
  if (has_tok)
    for (int i = tid; i < len; i += 512) tok_lds[i] = tokens[sq.tok_off + i];

  // per-lane activation constants: gate 2 (g) is tanh = 2*sigma(2x)-1, the others sigma
  const float ksc = (s == 2) ? -2.8853900817779268f : -1.4426950408889634f;
  const float ka = (s == 2) ? 2.0f : 1.0f;
  const float kb = (s == 2) ? -1.0f : 0.0f;

  float wA[4][13], wB[4][13];
  float bias;
  if (grp == 0) {
    load_w_slice(whh0, uc, s, valid, wA);
    bias = b0[4 * uc + s];
  } else {
    load_w_slice(wih1, uc, s, valid, wA);
    load_w_slice(whh1, uc, s, valid, wB);
    bias = b1[4 * uc + s];
  }
  __syncthreads();

  const int col = 4 * uc + s;   // this lane's column in every [rows, 200] buffer
  float xa[PF], xp[PF], mk[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) { xa[j] = 0.f; xp[j] = 0.f; mk[j] = 1.f; }
  auto issue = [&](int j, int t) {
    if (grp == 0 && t < len) {
      if (has_tok) xa[j] = WT0[(size_t)tok_lds[t] * S2G_G + col];
      if (has_p) xp[j] = (t < sq.p_len) ? P[(size_t)(sq.p_off + t) * S2G_G + col] : 0.0f;
      if (has_mask) mk[j] = mask[(size_t)(sq.mask_off + t) * S2G_MASKW + uc];
    }
  };
#pragma unroll
  for (int j = 0; j < PF; ++j) issue(j, j);

  float c = 0.0f;
  for (int base = 0; base <= len; base += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int tau = base + j;
      if (tau <= len) {
        if (grp == 0) {
          if (tau < len) {
            const int t = tau;
            float xv = sq.ca * xa[j] + sq.cp * xp[j] + bias;
            float m = mk[j];
            issue(j, t + PF);
            float hv[13];
            load_slice13(h0buf + ((t + 1) & 1) * 64, s, hv);
            float a0 = (s == 0) ? xv : 0.f, a1 = (s == 1) ? xv : 0.f, a2 = (s == 2) ? xv : 0.f,
                  a3 = (s == 3) ? xv : 0.f;
#pragma unroll
            for (int k = 0; k < 13; ++k) {
              a0 = fmaf(wA[0][k], hv[k], a0); a1 = fmaf(wA[1][k], hv[k], a1);
              a2 = fmaf(wA[2][k], hv[k], a2); a3 = fmaf(wA[3][k], hv[k], a3);
            }
            a0 = quad_sum(a0); a1 = quad_sum(a1); a2 = quad_sum(a2); a3 = quad_sum(a3);
            float pre = sel4(s, a0, a1, a2, a3);
            float val = fmaf(ka, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ksc * pre)), kb);
            float gi = quad_bcast<0>(val), gf = quad_bcast<1>(val), gg = quad_bcast<2>(val),
                  go = quad_bcast<3>(val);
            c = fmaf(gf, c, gi * gg);
            float tc = fast_tanh(c);
            float h = go * tc;
            float y = m * h;
            if (valid) {
              const size_t row = (size_t)(sq.ws_row + t) * S2G_G;
              gates0[row + col] = val;
              cell0[row + col] = sel4(s, c, tc, h, y);
              if (s == 0) h0buf[(t & 1) * 64 + pos13(u)] = h;
              if (s == 1) ybuf[(t & 1) * 64 + pos13(u)] = y;
            }
          }
        } else {
          if (tau >= 1) {
            const int t = tau - 1;
            float hv[13], yv[13];
            load_slice13(ybuf + (t & 1) * 64, s, yv);
            load_slice13(h1buf + ((t + 1) & 1) * 64, s, hv);
            float a0 = (s == 0) ? bias : 0.f, a1 = (s == 1) ? bias : 0.f, a2 = (s == 2) ? bias : 0.f,
                  a3 = (s == 3) ? bias : 0.f;
#pragma unroll
            for (int k = 0; k < 13; ++k) {
              a0 = fmaf(wA[0][k], yv[k], a0); a1 = fmaf(wA[1][k], yv[k], a1);
              a2 = fmaf(wA[2][k], yv[k], a2); a3 = fmaf(wA[3][k], yv[k], a3);
            }
#pragma unroll
            for (int k = 0; k < 13; ++k) {
              a0 = fmaf(wB[0][k], hv[k], a0); a1 = fmaf(wB[1][k], hv[k], a1);
              a2 = fmaf(wB[2][k], hv[k], a2); a3 = fmaf(wB[3][k], hv[k], a3);
            }
            a0 = quad_sum(a0); a1 = quad_sum(a1); a2 = quad_sum(a2); a3 = quad_sum(a3);
            float pre = sel4(s, a0, a1, a2, a3);
            float val = fmaf(ka, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ksc * pre)), kb);
            float gi = quad_bcast<0>(val), gf = quad_bcast<1>(val), gg = quad_bcast<2>(val),
                  go = quad_bcast<3>(val);
            c = fmaf(gf, c, gi * gg);
            float tc = fast_tanh(c);
            float h = go * tc;
            if (valid) {
              const size_t row = (size_t)(sq.ws_row + t) * S2G_G;
              gates1[row + col] = val;
              cell1[row + col] = sel4(s, c, tc, h, 0.0f);
              if (s == 0) {
                h1buf[(t & 1) * 64 + pos13(u)] = h;
                if (t == len - 1) hT[(size_t)n * S2G_H + u] = h;
              }
            }
          }
        }
        __syncthreads();
      }
    }
  }
  if (len == 0 && grp == 1 && valid && s == 0) hT[(size_t)n * S2G_H + u] = 0.0f;
}

// ------------------------------------------------------------------------------------------
// Sweep C: tangent states.  adot_t = xdot_t + W_hh hdot_{t-1}, gate tangents from the saved
// primal gates, cdot_t = fdot c_{t-1} + f cdot_{t-1} + idot g + i gdot,
// hdot_t = odot tanh(c) + o (1 - tanh^2 c) cdot.  Saves adot{0,1}[row][4u+q],
// celld0[row][4u+{0,1,2,3}] = (cdot, hdot, mask*hdot, 0), celld1 = (cdot, hdot, 0, 0), hdotT.
// Xdot holds the projected input direction rows [ws_row + t][200] (no bias).
// ------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(512) void s2g_k_lstm2_tangent(
    const S2GSeq* __restrict__ seqs, const float* __restrict__ Xdot, const float* __restrict__ mask,
    const float* __restrict__ whh0, const float* __restrict__ wih1, const float* __restrict__ whh1,
    const float* __restrict__ gates0, const float* __restrict__ cell0, const float* __restrict__ gates1,
    const float* __restrict__ cell1, float* __restrict__ adot0, float* __restrict__ celld0,
    float* __restrict__ adot1, float* __restrict__ celld1, float* __restrict__ hdotT) {
  __shared__ __attribute__((aligned(16))) float lds[384];
  float* h0buf = lds;          // hdot of layer 0 [2][64]
  float* ybuf = lds + 128;     // mask * hdot0    [2][64]
  float* h1buf = lds + 256;    // hdot of layer 1 [2][64]

  const int n = blockIdx.x;
  const S2GSeq sq = seqs[n];
  const int tid = threadIdx.x;
  const int grp = tid >> 8;
  const int l = tid & 255;
  const int u = l >> 2, s = l & 3;
  const bool valid = u < S2G_H;
  const int uc = valid ? u : S2G_H - 1;
  const int len = sq.len;
  const bool has_mask = sq.mask_off >= 0;

  for (int i = tid; i < 384; i += 512) lds[i] = 0.0f;

  float wA[4][13], wB[4][13];
  if (grp == 0) {
    load_w_slice(whh0, uc, s, valid, wA);
  } else {
    load_w_slice(wih1, uc, s, valid, wA);
    load_w_slice(whh1, uc, s, valid, wB);
  }
  __syncthreads();

  const int col = 4 * uc + s;
  const float* gates = grp == 0 ? gates0 : gates1;
  const float* cell = grp == 0 ? cell0 : cell1;
  // ring: xd (layer 0 only), gate value of (u, s), and one cell component per lane:
  // lane s=0 fetches c_{t-1}, lane s=1 fetches tanh(c_t)
  float xd[PF], gv[PF], cv[PF], mk[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) { xd[j] = 0.f; gv[j] = 0.f; cv[j] = 0.f; mk[j] = 1.f; }
  auto issue = [&](int j, int t) {
    if (t < len) {
      const size_t row = (size_t)(sq.ws_row + t) * S2G_G;
      gv[j] = gates[row + col];
      if (s == 0) cv[j] = (t > 0) ? cell[row - S2G_G + col] : 0.0f;
      if (s == 1) cv[j] = cell[row + col];
      if (grp == 0) {
        xd[j] = Xdot[row + col];
        if (has_mask) mk[j] = mask[(size_t)(sq.mask_off + t) * S2G_MASKW + uc];
      }
    }
  };
#pragma unroll
  for (int j = 0; j < PF; ++j) issue(j, j);

  float cd = 0.0f;   // cdot carried
  // One step of this lane's layer, consuming ring slot jj (a compile-time constant at every
  // call site after unrolling, so the ring stays in registers).
  auto step = [&](const int jj, const int t) __attribute__((always_inline)) {
    float g_s = gv[jj], c_s = cv[jj], m = mk[jj];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (grp == 0) {
      float xv = xd[jj];
      a0 = (s == 0) ? xv : 0.f; a1 = (s == 1) ? xv : 0.f; a2 = (s == 2) ? xv : 0.f;
      a3 = (s == 3) ? xv : 0.f;
    }
    issue(jj, t + PF);
    if (grp == 0) {
      float hv[13];
      load_slice13(h0buf + ((t + 1) & 1) * 64, s, hv);
#pragma unroll
      for (int k = 0; k < 13; ++k) {
        a0 = fmaf(wA[0][k], hv[k], a0); a1 = fmaf(wA[1][k], hv[k], a1);
        a2 = fmaf(wA[2][k], hv[k], a2); a3 = fmaf(wA[3][k], hv[k], a3);
      }
    } else {
      float hv[13], yv[13];
      load_slice13(ybuf + (t & 1) * 64, s, yv);
      load_slice13(h1buf + ((t + 1) & 1) * 64, s, hv);
#pragma unroll
      for (int k = 0; k < 13; ++k) {
        a0 = fmaf(wA[0][k], yv[k], a0); a1 = fmaf(wA[1][k], yv[k], a1);
        a2 = fmaf(wA[2][k], yv[k], a2); a3 = fmaf(wA[3][k], yv[k], a3);
      }
#pragma unroll
      for (int k = 0; k < 13; ++k) {
        a0 = fmaf(wB[0][k], hv[k], a0); a1 = fmaf(wB[1][k], hv[k], a1);
        a2 = fmaf(wB[2][k], hv[k], a2); a3 = fmaf(wB[3][k], hv[k], a3);
      }
    }
    a0 = quad_sum(a0); a1 = quad_sum(a1); a2 = quad_sum(a2); a3 = quad_sum(a3);
    float ad = sel4(s, a0, a1, a2, a3);
    float dsig = (s == 2) ? (1.0f - g_s * g_s) : g_s * (1.0f - g_s);
    float gd_s = dsig * ad;                       // tangent of gate s
    float gi = quad_bcast<0>(g_s), gf = quad_bcast<1>(g_s), gg = quad_bcast<2>(g_s),
          go = quad_bcast<3>(g_s);
    float di = quad_bcast<0>(gd_s), df = quad_bcast<1>(gd_s), dg = quad_bcast<2>(gd_s),
          dO = quad_bcast<3>(gd_s);
    float cprev = quad_bcast<0>(c_s), tc = quad_bcast<1>(c_s);
    cd = df * cprev + gf * cd + di * gg + gi * dg;
    float hd = dO * tc + go * (1.0f - tc * tc) * cd;
    float yd = m * hd;
    if (valid) {
      const size_t row = (size_t)(sq.ws_row + t) * S2G_G;
      if (grp == 0) {
        adot0[row + col] = ad;
        celld0[row + col] = sel4(s, cd, hd, yd, 0.0f);
        if (s == 0) h0buf[(t & 1) * 64 + pos13(u)] = hd;
        if (s == 1) ybuf[(t & 1) * 64 + pos13(u)] = yd;
      } else {
        adot1[row + col] = ad;
        celld1[row + col] = sel4(s, cd, hd, 0.0f, 0.0f);
        if (s == 0) {
          h1buf[(t & 1) * 64 + pos13(u)] = hd;
          if (t == len - 1) hdotT[(size_t)n * S2G_H + u] = hd;
        }
      }
    }
  };
  for (int base = 0; base <= len; base += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int tau = base + j;
      if (tau <= len) {
        if (grp == 0) {
          if (tau < len) step(j, tau);
        } else {
          if (tau >= 1) step((j + PF - 1) % PF, tau - 1);
        }
        __syncthreads();
      }
    }
  }
  if (len == 0 && grp == 1 && valid && s == 0) hdotT[(size_t)n * S2G_H + u] = 0.0f;
}
