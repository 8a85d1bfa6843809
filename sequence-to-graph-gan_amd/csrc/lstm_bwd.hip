// Reverse-in-time sweeps of the critic's 2-layer LSTM: sweep B (back-propagation through time
// for a given d(out); what autograd does for d_loss.backward(), wgan_gp.py:317, and for the
// input gradient of the penalty, wgan_gp.py:247-254) and sweep D (adjoints of the primal states
// for the loss <v, dD/dx>, i.e. the second-order term that `create_graph=True` makes autograd
// differentiate, wgan_gp.py:251,256,317).
//
// Decomposition: ONE workgroup per sequence, exact fp32 on the VALU.
//   waves 0-3: layer 1, step t = len-1-i        waves 4-7: layer 0, step t = len-i
// Lane = 16*kgroup + s: it produces 4 outputs k = 4*kgroup..+3 of the transposed mat-vec
// dh[k] = sum_j W[j][k] delta[j] over its slice s of 13 gate rows j (weights in registers,
// delta slice from LDS), the 16 lanes of a DPP row reduce the 4 partial sums, and lanes s<4 of
// each row then run the cell backward for unit k = 4*kgroup+s and publish its 4 delta values
// to LDS for the next step.  Only W^T delta recurrences live here; every weight gradient is a
// batched GEMM over the saved delta afterwards.  Saved activations are prefetched PF steps
// ahead by UNCONDITIONAL loads from clamped addresses (a load under a branch makes the compiler
// drain vmcnt to 0 at the join: one full memory round trip per sequential step).
#include "s2g_common.h"

#define PF 4

__device__ __forceinline__ void load_slice13b(const float* buf, int s, float hv[13]) {
  const float4* p = reinterpret_cast<const float4*>(buf + 16 * s);
  float4 a = p[0], b = p[1], c = p[2];
  float d = buf[16 * s + 12];
  hv[0] = a.x; hv[1] = a.y; hv[2] = a.z; hv[3] = a.w;
  hv[4] = b.x; hv[5] = b.y; hv[6] = b.z; hv[7] = b.w;
  hv[8] = c.x; hv[9] = c.y; hv[10] = c.z; hv[11] = c.w;
  hv[12] = d;
}

__device__ __forceinline__ float sel4b(int s, float a0, float a1, float a2, float a3) {
  float lo = (s & 1) ? a1 : a0;
  float hi = (s & 1) ? a3 : a2;
  return (s & 2) ? hi : lo;
}

// w[r][j] = W[torch row of internal gate row 13*s+j][4*g+r]  (W is [4H, H] torch layout)
__device__ __forceinline__ void load_wT_slice(const float* W, int g, int s, float w[4][13]) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      int ji = 13 * s + j, k = 4 * g + r;
      bool ok = ji < S2G_G && k < S2G_H;
      float v = W[pt_row(ok ? ji : 0) * S2G_H + (ok ? k : 0)];
      w[r][j] = ok ? v : 0.0f;
    }
}

__device__ __forceinline__ void zero_tail(float* buf, int ws_row, int len, int T, int width) {
  // rows [len, T) of this sequence <- 0 so that batched GEMMs over all rows see no stale data
  const size_t base = (size_t)(ws_row + len) * width;
  const size_t cnt = (size_t)(T - len) * width;
  for (size_t i = threadIdx.x; i < cnt; i += blockDim.x) buf[base + i] = 0.0f;
}

#define FMA13B(w, v)                                                        \
  _Pragma("unroll") for (int k_ = 0; k_ < 13; ++k_) {                      \
    a0 = fmaf(w[0][k_], v[k_], a0); a1 = fmaf(w[1][k_], v[k_], a1);        \
    a2 = fmaf(w[2][k_], v[k_], a2); a3 = fmaf(w[3][k_], v[k_], a3);        \
  }

// ------------------------------------------------------------------------------------------
// Sweep B.  dhT[n][k] = d loss / d h1_final.  Writes delta{0,1}[row][4k+q] (gradient wrt the
// gate pre-activations) and, for sequences n >= adj_seq0 (if adj0 != nullptr),
// adj{0,1}[row - adj_row0][2k+{0,1}] = (dh_t, dc_t), the total adjoints of h_t and c_t, which
// sweep D reuses as the adjoints of the tangent states.
// ------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(512) void s2g_k_lstm2_bwd(
    const S2GSeq* __restrict__ seqs, const float* __restrict__ mask, const float* __restrict__ whh0,
    const float* __restrict__ wih1, const float* __restrict__ whh1, const float* __restrict__ gates0,
    const float* __restrict__ cell0, const float* __restrict__ gates1, const float* __restrict__ cell1,
    const float* __restrict__ dhT, float* __restrict__ delta0, float* __restrict__ delta1,
    float* __restrict__ adj0, float* __restrict__ adj1, int T, int adj_seq0, int adj_row0) {
  __shared__ __attribute__((aligned(16))) float lds[1024];
  float* d1buf = lds;         // delta of layer 1 [2][256] (slices of 13 padded to 16)
  float* d0buf = lds + 512;   // delta of layer 0 [2][256]

  const int n = blockIdx.x;
  const S2GSeq sq = seqs[n];
  const int tid = threadIdx.x;
  const int grp = tid >> 8;            // 0: layer 1, 1: layer 0 (one iteration behind)
  const int l = tid & 255;
  const int g = l >> 4, s = l & 15;
  const int len = sq.len;
  const bool has_mask = sq.mask_off >= 0 && mask != nullptr;
  const int k = 4 * g + s;                       // unit handled by the element-wise lanes
  const bool ew = (s < 4) && (k < S2G_H);        // this lane runs the cell backward of unit k
  const int kc = ew ? k : 0;

  if (len <= 0) {
    zero_tail(delta0, sq.ws_row, 0, T, S2G_G);
    zero_tail(delta1, sq.ws_row, 0, T, S2G_G);
    return;
  }
  for (int i = tid; i < 1024; i += 512) lds[i] = 0.0f;

  float wA[4][13], wB[4][13];
  if (grp == 0) {
    load_wT_slice(whh1, g, s, wA);
  } else {
    load_wT_slice(wih1, g, s, wA);
    load_wT_slice(whh0, g, s, wB);
  }
  const size_t rb = (size_t)sq.ws_row * S2G_G + 4 * kc;
  const float* pg = (grp == 0 ? gates1 : gates0) + rb;
  const float* pc = (grp == 0 ? cell1 : cell0) + rb;
  float* delta = grp == 0 ? delta1 : delta0;
  float* adj = (adj0 != nullptr && n >= adj_seq0) ? (grp == 0 ? adj1 : adj0) : nullptr;
  const bool l0m = grp == 1 && has_mask && g < 13;
  const float* pm = l0m ? mask + (size_t)sq.mask_off * S2G_MASKW + 4 * g : whh0;
  const int sm = l0m ? S2G_MASKW : 0;
  float dh_final = 0.0f;
  if (grp == 0 && ew) dh_final = dhT[(size_t)n * S2G_H + k];
  int dpos[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) dpos[q] = pos13(4 * kc + q);
  __syncthreads();

  // At iteration i both groups use ring slot i % PF; it holds step t = len-1-(i-toff) of the
  // lane's own layer (toff = 0 for layer 1, 1 for layer 0).
  const int toff = grp;
  float4 rg[PF], rc[PF], rm[PF];
#define ISSUE_B(j, i_)                                                       \
  {                                                                          \
    int tc_ = max(min(len - 1 - ((i_) - toff), len - 1), 0);                 \
    rg[j] = *reinterpret_cast<const float4*>(pg + (size_t)tc_ * S2G_G);      \
    rc[j] = *reinterpret_cast<const float4*>(pc + (size_t)tc_ * S2G_G);      \
    rm[j] = *reinterpret_cast<const float4*>(pm + (size_t)tc_ * sm);         \
  }
#pragma unroll
  for (int j = 0; j < PF; ++j) ISSUE_B(j, j)

  float dc = 0.0f;
  for (int base = 0; base <= len; base += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = base + j;
      const float4 G = rg[j], C = rc[j], Mk = rm[j];
      const float cprev_raw = rc[(j + 1) % PF].x;            // slot of the next iteration = step t-1
      ISSUE_B(j, i + PF)
      if (i <= len) {
        const bool act = grp == 0 ? (i < len) : (i >= 1);
        if (act) {
          const int t = len - 1 - (i - toff);
          float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
          float dv[13];
          load_slice13b(d1buf + ((i + 1) & 1) * 256, s, dv);   // layer 1: delta1_{t+1}; layer 0: delta1_t
          FMA13B(wA, dv)
          if (grp == 1) {
            if (has_mask) { a0 *= Mk.x; a1 *= Mk.y; a2 *= Mk.z; a3 *= Mk.w; }   // dropout on layer-0 output
            load_slice13b(d0buf + (i & 1) * 256, s, dv);       // delta0_{t+1}
            FMA13B(wB, dv)
          }
          a0 = row16_sum(a0); a1 = row16_sum(a1); a2 = row16_sum(a2); a3 = row16_sum(a3);
          float dh = sel4b(s, a0, a1, a2, a3);
          if (grp == 0 && t == len - 1) dh += dh_final;
          const float cprev = (t > 0) ? cprev_raw : 0.0f;
          if (ew) {
            float gi = G.x, gf = G.y, gg = G.z, go = G.w, tc = C.y;
            float dct = fmaf(dh * go, 1.0f - tc * tc, dc);
            float4 d;
            d.x = dct * gg * gi * (1.0f - gi);
            d.y = dct * cprev * gf * (1.0f - gf);
            d.z = dct * gi * (1.0f - gg * gg);
            d.w = dh * tc * go * (1.0f - go);
            dc = dct * gf;
            const size_t row = (size_t)(sq.ws_row + t);
            *reinterpret_cast<float4*>(delta + row * S2G_G + 4 * k) = d;
            if (adj) *reinterpret_cast<float2*>(adj + (row - adj_row0) * (2 * S2G_H) + 2 * k) = make_float2(dh, dct);
            float* db = (grp == 0 ? d1buf : d0buf) + ((grp == 0 ? i : i + 1) & 1) * 256;
            db[dpos[0]] = d.x; db[dpos[1]] = d.y; db[dpos[2]] = d.z; db[dpos[3]] = d.w;
          }
        }
        __syncthreads();
      }
    }
  }
  zero_tail(delta0, sq.ws_row, len, T, S2G_G);
  zero_tail(delta1, sq.ws_row, len, T, S2G_G);
}

// ------------------------------------------------------------------------------------------
// Sweep D.  Inputs: primal gates/cell (sweep A), tangent adot/celld (sweep C), adj = (dh, dc)
// of sweep B run with d(out) = e_0.  Output abar{0,1}[row][4k+q]: adjoint of the primal gate
// pre-activation for the loss sum_n fc.weight[0] . hdot1_final(n).
// ------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(512) void s2g_k_lstm2_bwd2(
    const S2GSeq* __restrict__ seqs, const float* __restrict__ mask, const float* __restrict__ whh0,
    const float* __restrict__ wih1, const float* __restrict__ whh1, const float* __restrict__ gates0,
    const float* __restrict__ cell0, const float* __restrict__ gates1, const float* __restrict__ cell1,
    const float* __restrict__ adot0, const float* __restrict__ celld0, const float* __restrict__ adot1,
    const float* __restrict__ celld1, const float* __restrict__ adj0, const float* __restrict__ adj1,
    float* __restrict__ abar0, float* __restrict__ abar1, int T) {
  __shared__ __attribute__((aligned(16))) float lds[1024];
  float* d1buf = lds;
  float* d0buf = lds + 512;

  const int n = blockIdx.x;
  const S2GSeq sq = seqs[n];
  const int tid = threadIdx.x;
  const int grp = tid >> 8;            // 0: layer 1, 1: layer 0
  const int l = tid & 255;
  const int g = l >> 4, s = l & 15;
  const int len = sq.len;
  const bool has_mask = sq.mask_off >= 0 && mask != nullptr;
  const int k = 4 * g + s;
  const bool ew = (s < 4) && (k < S2G_H);
  const int kc = ew ? k : 0;

  if (len <= 0) {
    zero_tail(abar0, sq.ws_row, 0, T, S2G_G);
    zero_tail(abar1, sq.ws_row, 0, T, S2G_G);
    return;
  }
  for (int i = tid; i < 1024; i += 512) lds[i] = 0.0f;

  float wA[4][13], wB[4][13];
  if (grp == 0) {
    load_wT_slice(whh1, g, s, wA);
  } else {
    load_wT_slice(wih1, g, s, wA);
    load_wT_slice(whh0, g, s, wB);
  }
  const size_t rb = (size_t)sq.ws_row * S2G_G + 4 * kc;
  const float* pg = (grp == 0 ? gates1 : gates0) + rb;
  const float* pc = (grp == 0 ? cell1 : cell0) + rb;
  const float* pa = (grp == 0 ? adot1 : adot0) + rb;
  const float* pcd = (grp == 0 ? celld1 : celld0) + rb;
  const float* pj = (grp == 0 ? adj1 : adj0) + (size_t)sq.ws_row * (2 * S2G_H) + 2 * kc;
  float* abar = grp == 0 ? abar1 : abar0;
  const bool l0m = grp == 1 && has_mask && g < 13;
  const float* pm = l0m ? mask + (size_t)sq.mask_off * S2G_MASKW + 4 * g : whh0;
  const int sm = l0m ? S2G_MASKW : 0;
  int dpos[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) dpos[q] = pos13(4 * kc + q);
  __syncthreads();

  const int toff = grp;
  float4 rg[PF], rc[PF], ra[PF], rm[PF];
  float2 rcd[PF], rj[PF];
#define ISSUE_D(j, i_)                                                       \
  {                                                                          \
    int tc_ = max(min(len - 1 - ((i_) - toff), len - 1), 0);                 \
    rg[j] = *reinterpret_cast<const float4*>(pg + (size_t)tc_ * S2G_G);      \
    rc[j] = *reinterpret_cast<const float4*>(pc + (size_t)tc_ * S2G_G);      \
    ra[j] = *reinterpret_cast<const float4*>(pa + (size_t)tc_ * S2G_G);      \
    rcd[j] = *reinterpret_cast<const float2*>(pcd + (size_t)tc_ * S2G_G);    \
    rj[j] = *reinterpret_cast<const float2*>(pj + (size_t)tc_ * (2 * S2G_H)); \
    rm[j] = *reinterpret_cast<const float4*>(pm + (size_t)tc_ * sm);         \
  }
#pragma unroll
  for (int j = 0; j < PF; ++j) ISSUE_D(j, j)

  float cb = 0.0f;   // adjoint of the primal cell state, carried
  for (int base = 0; base <= len; base += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = base + j;
      const float4 G = rg[j], C = rc[j], A = ra[j], Mk = rm[j];
      const float2 CD = rcd[j], J = rj[j];
      const float cprev_raw = rc[(j + 1) % PF].x, cdprev_raw = rcd[(j + 1) % PF].x;
      ISSUE_D(j, i + PF)
      if (i <= len) {
        const bool act = grp == 0 ? (i < len) : (i >= 1);
        if (act) {
          const int t = len - 1 - (i - toff);
          float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
          float dv[13];
          load_slice13b(d1buf + ((i + 1) & 1) * 256, s, dv);
          FMA13B(wA, dv)
          if (grp == 1) {
            if (has_mask) { a0 *= Mk.x; a1 *= Mk.y; a2 *= Mk.z; a3 *= Mk.w; }
            load_slice13b(d0buf + (i & 1) * 256, s, dv);
            FMA13B(wB, dv)
          }
          a0 = row16_sum(a0); a1 = row16_sum(a1); a2 = row16_sum(a2); a3 = row16_sum(a3);
          const float hb = sel4b(s, a0, a1, a2, a3);
          const float cprev = (t > 0) ? cprev_raw : 0.0f;
          const float cdprev = (t > 0) ? cdprev_raw : 0.0f;
          if (ew) {
            float gi = G.x, gf = G.y, gg = G.z, go = G.w, tc = C.y;
            float ai = A.x, af = A.y, ag = A.z, ao = A.w;
            float cd = CD.x, Hd = J.x, Cd = J.y;
            float si = gi * (1.0f - gi), sf = gf * (1.0f - gf), sg = 1.0f - gg * gg, so = go * (1.0f - go);
            float id_ = si * ai, fd = sf * af, gd = sg * ag, od = so * ao;
            float uu = 1.0f - tc * tc;
            float tcb = Hd * (od - 2.0f * go * tc * cd) + hb * go;
            float ob = Hd * uu * cd + hb * tc + Hd * tc * (1.0f - 2.0f * go) * ao;
            float cbt = cb + tcb * uu;
            float fb = Cd * cdprev + cbt * cprev + Cd * cprev * (1.0f - 2.0f * gf) * af;
            float ib = Cd * gd + cbt * gg + Cd * gg * (1.0f - 2.0f * gi) * ai;
            float gb = Cd * id_ + cbt * gi + Cd * gi * (-2.0f * gg) * ag;
            cb = cbt * gf + Cd * fd;
            float4 d = make_float4(ib * si, fb * sf, gb * sg, ob * so);
            const size_t row = (size_t)(sq.ws_row + t);
            *reinterpret_cast<float4*>(abar + row * S2G_G + 4 * k) = d;
            float* db = (grp == 0 ? d1buf : d0buf) + ((grp == 0 ? i : i + 1) & 1) * 256;
            db[dpos[0]] = d.x; db[dpos[1]] = d.y; db[dpos[2]] = d.z; db[dpos[3]] = d.w;
          }
        }
        __syncthreads();
      }
    }
  }
  zero_tail(abar0, sq.ws_row, len, T, S2G_G);
  zero_tail(abar1, sq.ws_row, len, T, S2G_G);
}
