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
// gates) are prefetched PF steps ahead into a register ring.  Every prefetch is issued
// UNCONDITIONALLY from a clamped address (lanes/steps that do not need the value read a valid
// dummy location with stride 0): a load under a branch makes the compiler drain vmcnt to 0 at
// the join, which would put a full memory round trip on every one of the 512 sequential steps.
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
      bool ok = valid && k < S2G_H;
      float v = W[(q * S2G_H + u) * S2G_H + (k < S2G_H ? k : S2G_H - 1)];
      w[q][j] = ok ? v : 0.0f;
    }
}

#define FMA13(acc, w, v)                                                    \
  _Pragma("unroll") for (int k_ = 0; k_ < 13; ++k_) {                      \
    acc##0 = fmaf(w[0][k_], v[k_], acc##0); acc##1 = fmaf(w[1][k_], v[k_], acc##1); \
    acc##2 = fmaf(w[2][k_], v[k_], acc##2); acc##3 = fmaf(w[3][k_], v[k_], acc##3); \
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
  const bool has_p = sq.p_off >= 0 && sq.cp != 0.0f && sq.p_len > 0;
  const bool has_mask = sq.mask_off >= 0 && mask != nullptr;

  if (len <= 0) {
    if (grp == 1 && valid && s == 0) hT[(size_t)n * S2G_H + u] = 0.0f;
    return;
  }
  for (int i = tid; i < 384; i += 512) h0buf[i] = 0.0f;
  for (int i = tid; i < len; i += 512) tok_lds[i] = has_tok ? tokens[sq.tok_off + i] : 0;

  // per-lane activation constants: gate 2 (g) is tanh = 2*sigma(2x)-1, the others sigma
  const float ksc = (s == 2) ? -2.8853900817779268f : -1.4426950408889634f;
  const float ka = (s == 2) ? 2.0f : 1.0f;
  const float kb = (s == 2) ? -1.0f : 0.0f;

  float wA[4][13], wB[4][13];
  const int col = 4 * uc + s;   // this lane's column in every [rows, 200] buffer
  float bias;
  if (grp == 0) {
    load_w_slice(whh0, uc, s, valid, wA);
    bias = b0[col];
  } else {
    load_w_slice(wih1, uc, s, valid, wA);
    load_w_slice(whh1, uc, s, valid, wB);
    bias = b1[col];
  }
  __syncthreads();

  // prefetch sources: real pointer + stride for lanes that need the value, (b0, 0) otherwise
  const bool l0 = grp == 0;
  const float* pa = (l0 && has_tok) ? WT0 + col : b0;
  const int sa = (l0 && has_tok) ? S2G_G : 0;
  const float* pp = (l0 && has_p) ? P + (size_t)sq.p_off * S2G_G + col : b0;
  const int sp = (l0 && has_p) ? S2G_G : 0;
  const float* pm = (l0 && has_mask) ? mask + (size_t)sq.mask_off * S2G_MASKW + uc : b0;
  const int sm = (l0 && has_mask) ? S2G_MASKW : 0;
  const float ca = has_tok ? sq.ca : 0.0f;
  const float cp = has_p ? sq.cp : 0.0f;
  const int plast = sq.p_len > 0 ? sq.p_len - 1 : 0;

  float xa[PF], xp[PF], mk[PF];
#define ISSUE_A(j, t)                                                        \
  {                                                                          \
    int tc_ = min((t), len - 1);                                             \
    xa[j] = pa[(size_t)tok_lds[tc_] * sa];                                   \
    xp[j] = pp[(size_t)min(tc_, plast) * sp];                                \
    mk[j] = pm[(size_t)tc_ * sm];                                            \
  }
#pragma unroll
  for (int j = 0; j < PF; ++j) ISSUE_A(j, j)

  float c = 0.0f;
  for (int base = 0; base <= len; base += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int tau = base + j;
      // consume ring slot j (step tau) and refill it for step tau+PF, unconditionally
      const float xa_c = xa[j], xp_c = xp[j], mk_c = mk[j];
      ISSUE_A(j, tau + PF)
      if (tau <= len) {
        if (grp == 0) {
          if (tau < len) {
            const int t = tau;
            float xv = fmaf(ca, xa_c, bias);
            xv = fmaf((t < sq.p_len) ? cp : 0.0f, xp_c, xv);
            const float m = has_mask ? mk_c : 1.0f;
            float hv[13];
            load_slice13(h0buf + ((t + 1) & 1) * 64, s, hv);
            float a0 = (s == 0) ? xv : 0.f, a1 = (s == 1) ? xv : 0.f, a2 = (s == 2) ? xv : 0.f,
                  a3 = (s == 3) ? xv : 0.f;
            FMA13(a, wA, hv)
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
            FMA13(a, wA, yv)
            FMA13(a, wB, hv)
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
  const bool has_mask = sq.mask_off >= 0 && mask != nullptr;

  if (len <= 0) {
    if (grp == 1 && valid && s == 0) hdotT[(size_t)n * S2G_H + u] = 0.0f;
    return;
  }
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
  const bool l0 = grp == 0;
  const size_t rbase = (size_t)sq.ws_row * S2G_G + col;
  const float* pg = (l0 ? gates0 : gates1) + rbase;
  // one cell component per lane: lane s=0 needs c_{t-1}, lane s=1 needs tanh(c_t); others unused
  const float* pc = (l0 ? cell0 : cell1) + rbase;
  const int cshift = (s == 0) ? 1 : 0;
  const float* px = l0 ? Xdot + rbase : whh0;
  const int sx = l0 ? S2G_G : 0;
  const float* pm = (l0 && has_mask) ? mask + (size_t)sq.mask_off * S2G_MASKW + uc : whh0;
  const int sm = (l0 && has_mask) ? S2G_MASKW : 0;

  float xd[PF], gv[PF], cv[PF], mk[PF];
  // Layer 1 runs one step behind layer 0: at unrolled position tau both groups use ring slot
  // tau % PF, which holds step (tau - toff) of the lane's own layer (toff = 0 / 1).
  const int toff = grp;
#define ISSUE_C(j, t)                                                        \
  {                                                                          \
    int tc_ = max(min((t), len - 1), 0);                                     \
    gv[j] = pg[(size_t)tc_ * S2G_G];                                         \
    cv[j] = pc[(size_t)max(tc_ - cshift, 0) * S2G_G];                        \
    xd[j] = px[(size_t)tc_ * sx];                                            \
    mk[j] = pm[(size_t)tc_ * sm];                                            \
  }
#pragma unroll
  for (int j = 0; j < PF; ++j) ISSUE_C(j, j - toff)

  float cd = 0.0f;   // cdot carried
  for (int base = 0; base <= len; base += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int tau = base + j;
      // consume ring slot j and refill it for position tau+PF, unconditionally
      float g_s = gv[j], c_s = cv[j];
      const float x_s = l0 ? xd[j] : 0.0f, m_s = mk[j];
      ISSUE_C(j, tau + PF - toff)
      if (tau <= len) {
        const bool act = l0 ? (tau < len) : (tau >= 1);
        if (act) {
          const int t = l0 ? tau : tau - 1;
          if (s == 0 && t == 0) c_s = 0.0f;        // c_{-1} = 0
          const float m = has_mask ? m_s : 1.0f;
          float a0 = (s == 0) ? x_s : 0.f, a1 = (s == 1) ? x_s : 0.f, a2 = (s == 2) ? x_s : 0.f,
                a3 = (s == 3) ? x_s : 0.f;
          if (l0) {
            float hv[13];
            load_slice13(h0buf + ((t + 1) & 1) * 64, s, hv);
            FMA13(a, wA, hv)
          } else {
            float hv[13], yv[13];
            load_slice13(ybuf + (t & 1) * 64, s, yv);
            load_slice13(h1buf + ((t + 1) & 1) * 64, s, hv);
            FMA13(a, wA, yv)
            FMA13(a, wB, hv)
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
            if (l0) {
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
        }
        __syncthreads();
      }
    }
  }
}
