// LSTM sweeps for the weight-clipping WGAN variant (reference wgan.py: nn.LSTM(64, 64, num_layers) in
// Generator :58 and Discriminator :100, hidden_dim = 64) and its attention pooling (wgan.py:75-89).
// Stacks of two or three layers run ALL layers in one sweep per direction (s2g_k_lstm64_stack_fwd / _bwd below:
// one group of 256 threads per layer and, for two layers, a third group for the upper layer's input-side product).
// Other depths: one layer per launch over all its time steps, one workgroup of 256 threads per sequence: a thread
// owns one gate row (forward: the four gates of a unit in one quad) / one gate block's slice of a column (reverse)
// with its 64 recurrent weights in registers; the input part
// x_t W_ih^T + b_ih + b_hh of every step is a batched GEMM done beforehand (it does not depend on the
// recurrence), and the weight gradients are batched GEMMs over the saved gate gradients afterwards.
// Rows: sequence n owns rows [n * rows_per_seq, n * rows_per_seq + len_n) of every [rows, .] buffer; rows at or
// beyond its length are written as zeros where a later batched product reads them (pad_packed_sequence
// semantics, wgan.py:112-114).
#pragma once
#include "s2g_common.h"

#define X_H 64
#define X_G 256
#define X_PF 4         // steps of saved / precomputed rows in flight ahead of the recurrence

// gates [rows, 256] (activated i, f, g, o), cst [rows, 64] (cell state), hout [rows, 64].
// out_mask (optional, [rows, 64]): hout is ALSO written multiplied by it to hdrop (the next layer's input under
// nn.LSTM's inter-layer dropout); the recurrence itself uses the undropped h.
// Thread j = (unit j / 4, gate j % 4): the four gates of a unit sit in one quad, so the cell update needs no LDS
// (DPP quad broadcasts; every lane of the quad advances the same c) and a step has ONE barrier: h is double
// buffered in LDS.
extern "C" __global__ __launch_bounds__(X_G) void s2g_k_lstm64_fwd(const float* __restrict__ pin,
                                                                   const float* __restrict__ whh,
                                                                   const int* __restrict__ lengths, int T,
                                                                   float* __restrict__ gates, float* __restrict__ cst,
                                                                   float* __restrict__ hout,
                                                                   const float* __restrict__ out_mask,
                                                                   float* __restrict__ hdrop) {
  __shared__ __attribute__((aligned(16))) float hs[2][X_H];
  const int n = blockIdx.x, j = threadIdx.x;
  const int unit = j >> 2, q = j & 3;          // gate block q: 0 i, 1 f, 2 g (tanh), 3 o
  const int row = q * X_H + unit;              // torch gate row
  const int len = lengths ? min(max(lengths[n], 0), T) : T;
  float w[X_H];
#pragma unroll
  for (int k = 0; k < X_H; ++k) w[k] = whh[row * X_H + k];
  if (j < 2 * X_H) hs[0][j] = 0.0f;
  const size_t r0 = (size_t)n * T;
  // sigma(a) and tanh(a) = 2 sigma(2a) - 1 share one exp / rcp: per-lane constants instead of a branch
  const float kact = (q == 2) ? -2.8853900817779268f : -1.4426950408889634f;
  const float actA = (q == 2) ? 2.0f : 1.0f, actB = (q == 2) ? -1.0f : 0.0f;
  float c = 0.0f;
  // the input part of the next X_PF steps sits in a register ring, refilled by UNCONDITIONAL loads from clamped
  // rows after a slot's last use (it keeps its registers: no copies, no waits); the dropout multiplier travels
  // in the same ring (without a mask the loads read the input part instead and the value is not used)
  float nx[X_PF], mk[X_PF];
  const int lastrow = max(len - 1, 0);
  const bool has_mask = hdrop != nullptr && out_mask != nullptr;
  const float* pm = has_mask ? out_mask + r0 * X_H + unit : pin + r0 * X_G + row;
  const int sm = has_mask ? X_H : X_G;
#pragma unroll
  for (int i = 0; i < X_PF; ++i) {
    nx[i] = pin[(r0 + min(i, lastrow)) * X_G + row];
    mk[i] = pm[(size_t)min(i, lastrow) * sm];
  }
  __syncthreads();
  for (int base = 0; base < len; base += X_PF) {
#pragma unroll
    for (int i = 0; i < X_PF; ++i) {
      const int t = base + i;
      // the up to X_PF - 1 steps past the end run on clamped inputs and store nothing: a branch around the step
      // (with its barrier) makes the compiler wait for every load in flight at the join
      const bool live = t < len;
      float a0 = nx[i], a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;         // four independent chains: issue-bound, not latency-bound
      const float4* h4 = reinterpret_cast<const float4*>(hs[t & 1]);
#pragma unroll
      for (int k = 0; k < X_H / 4; ++k) {
        const float4 hv = h4[k];
        a0 = fmaf(w[4 * k], hv.x, a0); a1 = fmaf(w[4 * k + 1], hv.y, a1);
        a2 = fmaf(w[4 * k + 2], hv.z, a2); a3 = fmaf(w[4 * k + 3], hv.w, a3);
      }
      const float a = (a0 + a1) + (a2 + a3);
      const float g = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(kact * a)), actA, actB);
      if (live) gates[(r0 + t) * X_G + row] = g;
      const float gi = quad_bcast<0>(g), gf = quad_bcast<1>(g), gg = quad_bcast<2>(g), go = quad_bcast<3>(g);
      c = fmaf(gf, c, gi * gg);
      const float h = go * fast_tanh(c);
      if (q == 0) {
        hs[(t + 1) & 1][unit] = h;
        if (live) {
          cst[(r0 + t) * X_H + unit] = c;
          hout[(r0 + t) * X_H + unit] = h;
          if (hdrop) hdrop[(r0 + t) * X_H + unit] = has_mask ? h * mk[i] : h;
        }
      }
      nx[i] = pin[(r0 + min(t + X_PF, lastrow)) * X_G + row];
      mk[i] = pm[(size_t)min(t + X_PF, lastrow) * sm];
      __syncthreads();
    }
  }
  // rows beyond the sequence: zeros (what pad_packed_sequence returns, and what the batched products must see)
  for (size_t e = (size_t)len * X_H + j; e < (size_t)T * X_H; e += X_G) {
    hout[r0 * X_H + e] = 0.0f;
    if (hdrop) hdrop[r0 * X_H + e] = 0.0f;
  }
}

// Lane q of a quad holds slice [16 q, 16 q + 16) of the vector and, as pairs, the matching 16 weights of each of the
// unit's four gate rows: four 16-byte LDS reads per product instead of sixteen (every lane reading the whole
// vector makes a step LDS-bandwidth bound: a 16-byte read of a full wave occupies the LDS for 8 cycles whatever
// the addresses).
__device__ __forceinline__ void x_slice_fma(const f2 (&w)[4][8], const float* slice, f2 (&p)[4]) {
  const f4* v4 = reinterpret_cast<const f4*>(slice);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f4 v = v4[k];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      p[g] = __builtin_elementwise_fma(w[g][2 * k], (f2){v.x, v.y}, p[g]);
      p[g] = __builtin_elementwise_fma(w[g][2 * k + 1], (f2){v.z, v.w}, p[g]);
    }
  }
}
// p0..p3: this lane's partial sums of gates 0..3; returns to lane q the sum of gate q over the quad's four lanes
__device__ __forceinline__ float quad_gate_sum(float p0, float p1, float p2, float p3, int q) {
  const bool b0 = q & 1, b1 = q & 2;
  const float a = (b0 ? p1 : p0) + dpp_mov<DPP_QUAD(1, 0, 3, 2)>(b0 ? p0 : p1);
  const float b = (b0 ? p3 : p2) + dpp_mov<DPP_QUAD(1, 0, 3, 2)>(b0 ? p2 : p3);
  return (b1 ? b : a) + dpp_mov<DPP_QUAD(2, 3, 0, 1)>(b1 ? a : b);
}

// All layers of the stack in one sweep (2 or 3 layers).  One workgroup per sequence, one group of 256 threads per
// layer; at tick s group l runs time step s - l, so layer l + 1 follows layer l one step behind and the sweep
// takes len + layers - 1 ticks instead of layers * len steps.  Group 0 is the single-layer kernel above (its
// input part comes from the batched product); a thread of an upper group also holds its row of W_ih (128 weights
// in registers) and adds W_ih x_t to the recurrent product, x_t = mask * h_t of the layer below, handed over
// through LDS: the batched input products of the upper layers and their round trips through HBM disappear.
// Every group executes the same number of barriers (ticks rounded up to a multiple of X_PF for all of them).
struct S2GLstm64Stack {    // mirrors include/s2g.h
  const float* pin0;        // [rows, 256] input part of layer 0 (x W_ih0^T + b_ih0 + b_hh0)
  const float* whh[3];      // [256, 64] per layer
  const float* wih[3];      // [256, 64], layers >= 1
  const float* bsum[3];     // [256] b_ih + b_hh, layers >= 1
  float* gates[3];          // [rows, 256] saved activated gates (optional: null = not saved)
  float* cst[3];            // [rows, 64] saved cell states (optional)
  float* hout[3];           // [rows, 64] outputs (optional below the top layer)
  const float* mask[3];     // [rows, 64] dropout multiplier between layer l and l + 1 (optional)
  float* hdrop[3];          // [rows, 64] mask * h: the saved input of layer l + 1 (optional)
  // token mode (tokens != null, pin0 unused): layer 0's input part is looked up, tab[token_t] + tabl[label_n]
  const int* tokens;        // [rows]
  const float* tab;         // [V, 256] = emb W_ih0[:, :32]^T + b_ih0 + b_hh0
  const long long* labels;  // [sequences]
  const float* tabl;        // [n_classes, 256] = label_emb W_ih0[:, 32:]^T
  int V, n_classes;
};

// SPLIT (two layers only): a third group of 256 threads takes layer 1's input product W_ih1 x_t + b off the cell
// group's hands one tick ahead (through LDS), and the cell group runs two ticks behind layer 0: three waves per SIMD
// with 32 packed FMAs each per tick instead of two with 32 and 64.
template <int NL, bool TOK, bool SPLIT>
__global__ __launch_bounds__(X_G * (NL + (SPLIT ? 1 : 0))) void s2g_k_lstm64_stack_fwd(S2GLstm64Stack a,
                                                                                       const int* __restrict__ lengths,
                                                                                       int T) {
  static_assert(!SPLIT || NL == 2, "the split arrangement is written for two layers");
  __shared__ __attribute__((aligned(16))) float hs[NL][2][X_H];   // h of layer l, double buffered
  __shared__ __attribute__((aligned(16))) float xs[NL][2][X_H];   // xs[l]: input of layer l + 1
  __shared__ float xp[2][X_G];                                    // SPLIT: W_ih1 x_t + b, double buffered
  const int n = blockIdx.x;
  const int wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), j = threadIdx.x & (X_G - 1);
  const bool producer = SPLIT && wg == 2;
  const int grp = producer ? 1 : wg;                              // the layer this group works for
  const int unit = j >> 2, q = j & 3;
  const int row = q * X_H + unit;
  const int len = lengths ? min(max(lengths[n], 0), T) : T;
  const int nticks = len > 0 ? ((len + NL - 1 + (SPLIT ? 1 : 0) + X_PF - 1) / X_PF) * X_PF : 0;
  const size_t r0 = (size_t)n * T;
  const float kact = (q == 2) ? -2.8853900817779268f : -1.4426950408889634f;
  const float actA = (q == 2) ? 2.0f : 1.0f, actB = (q == 2) ? -1.0f : 0.0f;
  const int lastrow = max(len - 1, 0);
  for (int e = threadIdx.x; e < NL * 2 * X_H; e += blockDim.x) { (&hs[0][0][0])[e] = 0.0f; (&xs[0][0][0])[e] = 0.0f; }
  for (int e = threadIdx.x; e < 2 * X_G; e += blockDim.x) (&xp[0][0])[e] = 0.0f;
  float* __restrict__ gates = producer ? nullptr : a.gates[grp];
  float* __restrict__ cst = producer ? nullptr : a.cst[grp];
  float* __restrict__ hout = producer ? nullptr : a.hout[grp];
  float* __restrict__ hdrop = producer ? nullptr : a.hdrop[grp];
  const float* whh = a.whh[grp];
  const bool feeds = grp + 1 < NL;                                // a layer above reads this group's output
  const bool has_mask = feeds && a.mask[grp] != nullptr;
  // the dropout multiplier of the next X_PF steps travels in a register ring (unconditional loads from clamped
  // rows; without a mask the loads read a weight instead and the value is not used)
  const float* pm = has_mask ? a.mask[grp] + r0 * X_H + unit : whh + row;
  const int sm = has_mask ? X_H : 0;
  float mk[X_PF];
  float c = 0.0f;
  if (producer) {
    // x_t of tick s - 1 (layer 0 wrote it during that tick) -> xp for the cell group's next tick
    f2 wi[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < 8; ++k) wi[g][k] = *reinterpret_cast<const f2*>(a.wih[1] + (g * X_H + unit) * X_H + 16 * q + 2 * k);
    const float bias = a.bsum[1][row];
    __syncthreads();
    for (int tick = 0; tick < nticks; ++tick) {
      const int t = tick - 1;
      f2 px[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
      x_slice_fma(wi, xs[0][t & 1] + 16 * q, px);
      xp[t & 1][row] = bias + quad_gate_sum(px[0].x + px[0].y, px[1].x + px[1].y, px[2].x + px[2].y, px[3].x + px[3].y, q);
      __syncthreads();
    }
  } else if (grp == 0) {
    f2 w[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < 8; ++k) w[g][k] = *reinterpret_cast<const f2*>(whh + (g * X_H + unit) * X_H + 16 * q + 2 * k);
    // token mode: the input part of step t is row token_t of a [V, 256] table (L2 resident) plus a per-sequence
    // constant; the tokens travel in a ring of their own, X_PF steps ahead of the table rows they select
    const float* __restrict__ pin = TOK ? a.tab : a.pin0;
    const int* __restrict__ tokens = a.tokens;
    const int vmax = a.V - 1;
    float lbias = 0.0f;
    if (TOK) lbias = a.tabl[(size_t)min(max((int)a.labels[n], 0), a.n_classes - 1) * X_G + row];
    float nx[X_PF];
    int tk[X_PF];
#pragma unroll
    for (int i = 0; i < X_PF; ++i) {
      if (TOK) {
        nx[i] = pin[(size_t)min(max(tokens[r0 + min(i, lastrow)], 0), vmax) * X_G + row];
        tk[i] = tokens[r0 + min(i + X_PF, lastrow)];
      } else {
        nx[i] = pin[(r0 + min(i, lastrow)) * X_G + row];
      }
      mk[i] = pm[(size_t)min(i, lastrow) * sm];
    }
    __syncthreads();
    for (int base = 0; base < nticks; base += X_PF) {
#pragma unroll
      for (int i = 0; i < X_PF; ++i) {
        const int t = base + i;
        const bool live = t < len;
        f2 p[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
        x_slice_fma(w, hs[0][t & 1] + 16 * q, p);
        const float s = (nx[i] + lbias) + quad_gate_sum(p[0].x + p[0].y, p[1].x + p[1].y, p[2].x + p[2].y, p[3].x + p[3].y, q);
        const float g = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(kact * s)), actA, actB);
        if (live && gates) gates[(r0 + t) * X_G + row] = g;
        const float gi = quad_bcast<0>(g), gf = quad_bcast<1>(g), gg = quad_bcast<2>(g), go = quad_bcast<3>(g);
        c = fmaf(gf, c, gi * gg);
        const float h = go * fast_tanh(c);
        if (q == 0) {
          const float hx = has_mask ? h * mk[i] : h;
          hs[0][(t + 1) & 1][unit] = h;
          xs[0][t & 1][unit] = hx;
          if (live) {
            if (cst) cst[(r0 + t) * X_H + unit] = c;
            if (hout) hout[(r0 + t) * X_H + unit] = h;
            if (hdrop) hdrop[(r0 + t) * X_H + unit] = hx;
          }
        }
        if (TOK) {
          nx[i] = pin[(size_t)min(max(tk[i], 0), vmax) * X_G + row];
          tk[i] = tokens[r0 + min(t + 2 * X_PF, lastrow)];
        } else {
          nx[i] = pin[(r0 + min(t + X_PF, lastrow)) * X_G + row];
        }
        mk[i] = pm[(size_t)min(t + X_PF, lastrow) * sm];
        __syncthreads();
      }
    }
  } else {
    f2 wh[4][8], wi[4][8];
    const float* wih = a.wih[grp];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        wh[g][k] = *reinterpret_cast<const f2*>(whh + (g * X_H + unit) * X_H + 16 * q + 2 * k);
        if (!SPLIT) wi[g][k] = *reinterpret_cast<const f2*>(wih + (g * X_H + unit) * X_H + 16 * q + 2 * k);
      }
    const float bias = a.bsum[grp][row];
    const int lag = SPLIT ? 2 : grp;                   // ticks behind layer 0
#pragma unroll
    for (int i = 0; i < X_PF; ++i) mk[i] = pm[(size_t)min(max(i - lag, 0), lastrow) * sm];
    __syncthreads();
    for (int base = 0; base < nticks; base += X_PF) {
#pragma unroll
      for (int i = 0; i < X_PF; ++i) {
        const int t = base + i - lag;                  // this layer's time step at this tick
        const bool live = t >= 0 && t < len;
        f2 ph[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
        f2 px[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
        x_slice_fma(wh, hs[grp][t & 1] + 16 * q, ph);
        float s;
        if (SPLIT) {
          s = xp[t & 1][row] + quad_gate_sum(ph[0].x + ph[0].y, ph[1].x + ph[1].y, ph[2].x + ph[2].y, ph[3].x + ph[3].y, q);
        } else {
          x_slice_fma(wi, xs[grp - 1][t & 1] + 16 * q, px);
          s = bias + quad_gate_sum((ph[0].x + ph[0].y) + (px[0].x + px[0].y), (ph[1].x + ph[1].y) + (px[1].x + px[1].y),
                                   (ph[2].x + ph[2].y) + (px[2].x + px[2].y), (ph[3].x + ph[3].y) + (px[3].x + px[3].y), q);
        }
        const float g = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(kact * s)), actA, actB);
        if (live && gates) gates[(r0 + t) * X_G + row] = g;
        const float gi = quad_bcast<0>(g), gf = quad_bcast<1>(g), gg = quad_bcast<2>(g), go = quad_bcast<3>(g);
        const float cn = fmaf(gf, c, gi * gg);
        c = live ? cn : c;                             // before its first step the layer stays at c = 0, h = 0
        const float h = live ? go * fast_tanh(cn) : 0.0f;
        if (q == 0) {
          const float hx = has_mask ? h * mk[i] : h;
          hs[grp][(t + 1) & 1][unit] = h;
          if (feeds) xs[grp][t & 1][unit] = hx;
          if (live) {
            if (cst) cst[(r0 + t) * X_H + unit] = cn;
            if (hout) hout[(r0 + t) * X_H + unit] = h;
            if (hdrop) hdrop[(r0 + t) * X_H + unit] = hx;
          }
        }
        mk[i] = pm[(size_t)min(max(t + X_PF, 0), lastrow) * sm];
        __syncthreads();
      }
    }
  }
  // rows beyond the sequence: zeros (what pad_packed_sequence returns, and what the batched products must see)
  for (size_t e = (size_t)len * X_H + j; e < (size_t)T * X_H; e += X_G) {
    if (hout) hout[r0 * X_H + e] = 0.0f;
    if (hdrop) hdrop[r0 * X_H + e] = 0.0f;
  }
}

// Reverse sweep of one layer.  dh_in [rows, 64]: gradient reaching h_t from above (the attention pooling or
// the next layer's input gradient; already multiplied by that layer's dropout mask where one applies).
// delta [rows, 256]: gradient wrt the gate pre-activations (rows beyond the length zeroed).
extern "C" __global__ __launch_bounds__(X_G) void s2g_k_lstm64_bwd(const float* __restrict__ dh_in,
                                                                   const float* __restrict__ whh,
                                                                   const int* __restrict__ lengths, int T,
                                                                   const float* __restrict__ gates,
                                                                   const float* __restrict__ cst,
                                                                   float* __restrict__ delta) {
  __shared__ __attribute__((aligned(16))) float ds[X_G];
  __shared__ float part[4][X_H];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int len = lengths ? min(max(lengths[n], 0), T) : T;
  const int k = tid & 63, blk = tid >> 6;
  // W_hh^T slice: thread (k, blk) sums over gate rows blk*64 .. blk*64+63 of column k
  float w[X_H];
#pragma unroll
  for (int u = 0; u < X_H; ++u) w[u] = whh[(blk * X_H + u) * X_H + k];
  ds[tid] = 0.0f;
  const size_t r0 = (size_t)n * T;
  float dc = 0.0f;
  // saved values of the step in hand, fetched one step ahead by UNCONDITIONAL loads from clamped rows (a load
  // inside the cell section would put a memory round trip on every step's critical path); lanes >= 64 of the
  // first wave's siblings load too (harmless, same addresses modulo 64) so that no load sits under a branch
  const int u0 = tid & 63;
  const int lastrow = max(len - 1, 0);
  float n_dh[X_PF], n_gi[X_PF], n_gf[X_PF], n_gg[X_PF], n_go[X_PF], n_c[X_PF], n_cp[X_PF];
#define X_FETCH(i, t_)                                                        \
  {                                                                           \
    const size_t r = r0 + min(max((t_), 0), lastrow);                         \
    n_dh[i] = dh_in[r * X_H + u0];                                            \
    n_gi[i] = gates[r * X_G + u0]; n_gf[i] = gates[r * X_G + X_H + u0];       \
    n_gg[i] = gates[r * X_G + 2 * X_H + u0]; n_go[i] = gates[r * X_G + 3 * X_H + u0]; \
    n_c[i] = cst[r * X_H + u0];                                               \
    n_cp[i] = cst[(r0 + min(max((t_) - 1, 0), lastrow)) * X_H + u0];          \
  }
#pragma unroll
  for (int i = 0; i < X_PF; ++i) X_FETCH(i, len - 1 - i)
  __syncthreads();
  for (int base = 0; base < len; base += X_PF) {
#pragma unroll
    for (int i = 0; i < X_PF; ++i) {
      const int t = len - 1 - (base + i);
      const float dhi = n_dh[i], gi = n_gi[i], gf = n_gf[i], gg = n_gg[i], go = n_go[i], ct = n_c[i];
      const float cprev = t > 0 ? n_cp[i] : 0.0f;
      const bool live = t >= 0;                      // steps past the start run on clamped rows and store nothing
      const float tc = fast_tanh(ct);                // off the recurrent chain: it depends on saved values only
      float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
      const float4* d4 = reinterpret_cast<const float4*>(ds + blk * X_H);
#pragma unroll
      for (int u = 0; u < X_H / 4; ++u) {
        const float4 dv = d4[u];
        s0 = fmaf(w[4 * u], dv.x, s0); s1 = fmaf(w[4 * u + 1], dv.y, s1);
        s2 = fmaf(w[4 * u + 2], dv.z, s2); s3 = fmaf(w[4 * u + 3], dv.w, s3);
      }
      part[blk][k] = (s0 + s1) + (s2 + s3);
      __syncthreads();
      if (tid < X_H) {
        const float dh = dhi + ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
        const float dct = fmaf(dh * go, 1.0f - tc * tc, dc);
        const float di = dct * gg * gi * (1.0f - gi);
        const float df = dct * cprev * gf * (1.0f - gf);
        const float dg = dct * gi * (1.0f - gg * gg);
        const float dO = dh * tc * go * (1.0f - go);
        dc = dct * gf;
        if (live) {
          float* dr = delta + (r0 + t) * X_G;
          dr[tid] = di; dr[X_H + tid] = df; dr[2 * X_H + tid] = dg; dr[3 * X_H + tid] = dO;
        }
        ds[tid] = di; ds[X_H + tid] = df; ds[2 * X_H + tid] = dg; ds[3 * X_H + tid] = dO;
      }
      X_FETCH(i, t - X_PF)      // after the slot's last use: it keeps its registers across iterations
      __syncthreads();
    }
  }
#undef X_FETCH
  for (size_t e = (size_t)len * X_G + tid; e < (size_t)T * X_G; e += X_G) delta[r0 * X_G + e] = 0.0f;
}

// Reverse sweep of a two-layer stack in one launch: group 0 (256 threads) runs the top layer at step len - 1 - s at
// tick s, group 1 the layer below one tick behind; the gradient reaching the lower layer's h_t,
// mask_t * (W_ih1^T delta1_t), never leaves the chip (no batched dX product, no multiply by the mask afterwards).
// Thread (row R of 16 lanes, lane m): partial sums of outputs 4R..4R+3 over gate rows [16 m, 16 m + 16) -- four
// 16-byte LDS reads per product -- then a transpose-reduce over the 16 lanes (quad exchange + two row rotations)
// leaves lane m with the sum for unit 4R + (m & 3), and the lane computes the gradient of gate m >> 2 of that
// unit: all 256 lanes share the cell section, and a tick has one barrier (the gate gradients are double buffered).
struct S2GLstm64StackBwd {   // mirrors include/s2g.h
  const float* dh_top;       // [rows, 64] gradient reaching the top layer's h
  const float* whh[2];       // [256, 64], layer 0 and 1
  const float* wih1;         // [256, 64] W_ih of layer 1
  const float* gates[2];     // saved by the forward sweep
  const float* cst[2];
  const float* mask0;        // [rows, 64] dropout multiplier between the layers (optional)
  float* delta[2];           // [rows, 256] gradients of the gate pre-activations (zeros beyond the length)
  float* bias_part[2];       // [sequences, 256] per-sequence sums of delta over time (optional): the bias gradient
};

// sum over the 16 lanes of a DPP row of their partial sums p0..p3 of four outputs: lane m gets output m & 3
__device__ __forceinline__ float row16_out_sum(f2 pa, f2 pb, int m) {
  float v = quad_gate_sum(pa.x, pa.y, pb.x, pb.y, m & 3);
  v += dpp_mov<DPP_ROW_ROR(4)>(v);
  v += dpp_mov<DPP_ROW_ROR(8)>(v);
  return v;
}
// (pa, pb) += W^T slice (as output pairs 0|1 and 2|3) times the 16 gate gradients of this lane's slice
// The vector is stored so that the k-th 16-byte piece of lane m's slice sits at piece 16 k + m: the 16 lanes of a
// row then read 256 consecutive bytes (pieces 64 bytes apart would put lanes m, m + 4, m + 8, m + 12 on the same
// banks: a four-way conflict on every read).
__device__ __forceinline__ int x_tpos(int e) { return ((((e >> 2) & 3) * 16 + (e >> 4)) << 2) + (e & 3); }
__device__ __forceinline__ void x_slice_fma_t(const f2 (&w01)[16], const f2 (&w23)[16], const float* vec, int m, f2& pa,
                                              f2& pb) {
  const f4* v4 = reinterpret_cast<const f4*>(vec) + m;
  f2 qa = {0.0f, 0.0f}, qb = {0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f4 v = v4[16 * k];
    pa = __builtin_elementwise_fma(w01[4 * k], (f2){v.x, v.x}, pa);
    pb = __builtin_elementwise_fma(w23[4 * k], (f2){v.x, v.x}, pb);
    qa = __builtin_elementwise_fma(w01[4 * k + 1], (f2){v.y, v.y}, qa);
    qb = __builtin_elementwise_fma(w23[4 * k + 1], (f2){v.y, v.y}, qb);
    pa = __builtin_elementwise_fma(w01[4 * k + 2], (f2){v.z, v.z}, pa);
    pb = __builtin_elementwise_fma(w23[4 * k + 2], (f2){v.z, v.z}, pb);
    qa = __builtin_elementwise_fma(w01[4 * k + 3], (f2){v.w, v.w}, qa);
    qb = __builtin_elementwise_fma(w23[4 * k + 3], (f2){v.w, v.w}, qb);
  }
  pa += qa;
  pb += qb;
}

// Three groups of 256 threads: the top layer; a producer that turns the top layer's gate gradients of the previous
// tick into W_ih1^T delta1_t (64 values through LDS); the lower layer's cell group two ticks behind the top layer
// (three waves per SIMD with 32 packed FMAs each per tick instead of two with 32 and 64).
extern "C" __global__ __launch_bounds__(3 * X_G) void s2g_k_lstm64_stack_bwd(S2GLstm64StackBwd a,
                                                                             const int* __restrict__ lengths, int T) {
  __shared__ __attribute__((aligned(16))) float ds[2][2][X_G];     // [group][buffer][gate row]
  __shared__ float xq[2][X_H];                                     // W_ih1^T delta1_t, double buffered
  const int n = blockIdx.x;
  const int wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), j = threadIdx.x & (X_G - 1);
  const bool producer = wg == 2;
  const int grp = producer ? 1 : wg;
  const int R = j >> 4, m = j & 15;
  const int u = 4 * R + (m & 3), gate = m >> 2;
  const int wpos = x_tpos(gate * X_H + u);
  const int len = lengths ? min(max(lengths[n], 0), T) : T;
  const int nticks = len > 0 ? ((len + 2 + X_PF - 1) / X_PF) * X_PF : 0;
  const int lastrow = max(len - 1, 0);
  const size_t r0 = (size_t)n * T;
  for (int e = threadIdx.x; e < 2 * 2 * X_G; e += blockDim.x) (&ds[0][0][0])[e] = 0.0f;
  if (threadIdx.x < 2 * X_H) (&xq[0][0])[threadIdx.x] = 0.0f;
  const int layer = 1 - grp;
  const float* __restrict__ gates = a.gates[layer];
  const float* __restrict__ cst = a.cst[layer];
  float* __restrict__ delta = producer ? nullptr : a.delta[layer];
  const float* whh = producer ? a.wih1 : a.whh[layer];
  // own weights, W[16 m + rr][4R + o]: W_hh of the group's layer; the producer holds W_ih of the top layer
  f2 w01[16], w23[16];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    w01[rr] = *reinterpret_cast<const f2*>(whh + (16 * m + rr) * X_H + 4 * R);
    w23[rr] = *reinterpret_cast<const f2*>(whh + (16 * m + rr) * X_H + 4 * R + 2);
  }
  float dc = 0.0f, bsum = 0.0f;
  float n_x[X_PF], n_gi[X_PF], n_gf[X_PF], n_gg[X_PF], n_go[X_PF], n_c[X_PF], n_cp[X_PF];
  if (producer) {
    __syncthreads();
    for (int tick = 0; tick < nticks; ++tick) {
      // the top layer wrote delta1 of step len - tick into ds[0][tick & 1] during the previous tick
      f2 xa = {0.0f, 0.0f}, xb = {0.0f, 0.0f};
      x_slice_fma_t(w01, w23, ds[0][tick & 1], m, xa, xb);
      const float above = row16_out_sum(xa, xb, m);
      if (gate == 0) xq[tick & 1][u] = above;
      __syncthreads();
    }
  } else if (grp == 0) {
    const float* __restrict__ dh_in = a.dh_top;
#define XB_FETCH(i, t_)                                                       \
  {                                                                           \
    const size_t r = r0 + min(max((t_), 0), lastrow);                         \
    n_x[i] = px[r * sx];                                                      \
    n_gi[i] = gates[r * X_G + u]; n_gf[i] = gates[r * X_G + X_H + u];         \
    n_gg[i] = gates[r * X_G + 2 * X_H + u]; n_go[i] = gates[r * X_G + 3 * X_H + u]; \
    n_c[i] = cst[r * X_H + u];                                                \
    n_cp[i] = cst[(r0 + min(max((t_) - 1, 0), lastrow)) * X_H + u];           \
  }
#define XB_CELL(dh_)                                                          \
  const float gi = n_gi[i], gf = n_gf[i], gg = n_gg[i], go = n_go[i];         \
  const float cprev = t > 0 ? n_cp[i] : 0.0f;                                 \
  const float tc = fast_tanh(n_c[i]);                                         \
  const float dh = (dh_);                                                     \
  const float dct = fmaf(dh * go, 1.0f - tc * tc, dc);                        \
  const float di = dct * gg * gi * (1.0f - gi);                               \
  const float df = dct * cprev * gf * (1.0f - gf);                            \
  const float dg = dct * gi * (1.0f - gg * gg);                               \
  const float dO = dh * tc * go * (1.0f - go);                                \
  dc = live ? dct * gf : dc;                                                  \
  const float mine0 = gate == 0 ? di : gate == 1 ? df : gate == 2 ? dg : dO;  \
  const float mine = live ? mine0 : 0.0f;                                     \
  ds[grp][(tick + 1) & 1][wpos] = mine;                                       \
  bsum += mine;                                                               \
  if (live) delta[(r0 + t) * X_G + gate * X_H + u] = mine;
    const float* px = dh_in + u;
    const size_t sx = X_H;
#pragma unroll
    for (int i = 0; i < X_PF; ++i) XB_FETCH(i, len - 1 - i)
    __syncthreads();
    for (int base = 0; base < nticks; base += X_PF) {
#pragma unroll
      for (int i = 0; i < X_PF; ++i) {
        const int tick = base + i, t = len - 1 - tick;
        const bool live = t >= 0;
        f2 pa = {0.0f, 0.0f}, pb = {0.0f, 0.0f};
        x_slice_fma_t(w01, w23, ds[0][tick & 1], m, pa, pb);
        const float rec = row16_out_sum(pa, pb, m);
        XB_CELL(n_x[i] + rec)
        XB_FETCH(i, t - X_PF)
        __syncthreads();
      }
    }
  } else {
    const bool has_mask = a.mask0 != nullptr;
    const float* px = has_mask ? a.mask0 + u : whh;        // without a mask: a harmless load, the value is not used
    const size_t sx = has_mask ? X_H : 0;
#pragma unroll
    for (int i = 0; i < X_PF; ++i) XB_FETCH(i, len + 1 - i)
    __syncthreads();
    for (int base = 0; base < nticks; base += X_PF) {
#pragma unroll
      for (int i = 0; i < X_PF; ++i) {
        const int tick = base + i, t = len + 1 - tick;     // two ticks behind the layer above (one behind the producer)
        const bool live = t >= 0 && t < len;
        f2 pa = {0.0f, 0.0f}, pb = {0.0f, 0.0f};
        x_slice_fma_t(w01, w23, ds[1][tick & 1], m, pa, pb);
        const float rec = row16_out_sum(pa, pb, m);
        const float above = xq[(tick - 1) & 1][u];         // written by the producer during the previous tick
        XB_CELL(fmaf(has_mask ? n_x[i] : 1.0f, above, rec))
        XB_FETCH(i, t - X_PF)
        __syncthreads();
      }
    }
  }
#undef XB_FETCH
#undef XB_CELL
  if (producer) return;
  if (a.bias_part[layer]) a.bias_part[layer][(size_t)n * X_G + gate * X_H + u] = bsum;
  for (size_t e = (size_t)len * X_G + j; e < (size_t)T * X_G; e += X_G) delta[r0 * X_G + e] = 0.0f;
}

// x[row] = (emb[token[row]] | label_emb[label[n]])  (wgan.py:107-111 with the lookup branch of RelaxedEmbedding,
// :39-42); token ids clamped to the table.  soft != 0: the first 32 columns already hold y @ emb (the matmul
// branch, :45) and only the label half is written.
extern "C" __global__ void s2g_k_embed_concat(const int* __restrict__ tokens, const float* __restrict__ emb, int V,
                                              const long long* __restrict__ labels, const float* __restrict__ lab_emb,
                                              int n_classes, int T, long rows, int soft, float* __restrict__ x) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * 64) return;
  const long r = e >> 6;
  const int c = (int)(e & 63);
  if (c < 32) {
    if (!soft) x[e] = emb[(long)min(max(tokens[r], 0), V - 1) * 32 + c];
  } else {
    const long long l = min(max(labels[r / T], 0LL), (long long)n_classes - 1);
    x[e] = lab_emb[l * 32 + (c - 32)];
  }
}

// gradient of the lookup branch: demb[token[row]] += dx[row][0:32] for live rows (t < len), in two launches with a
// fixed order of additions (no float atomics).  Stage 1: a workgroup owns EG_CHUNK consecutive rows; each of its four
// 32-thread groups walks a quarter of them in ascending order, adding row r's 32 columns into its own [V][32] table in
// LDS at the row's token (the 32 threads of a group own distinct columns); the four tables are added in order and the
// chunk's table goes to part[chunk][V * 32].  Stage 2 (s2g_k_embed_grad_final) adds the chunks in ascending order.
#define EG_CHUNK 512
#define EG_GROUPS 4
#define EG_BATCH 32
extern "C" __global__ __launch_bounds__(32 * EG_GROUPS) void s2g_k_embed_grad_tokens_part(
    const int* __restrict__ tokens, const int* __restrict__ lengths, int T, long rows, int V,
    const float* __restrict__ dx, int ld, int chunk, float* __restrict__ part, int width, int x_seg) {
  // blockIdx.y: a slice of 32 columns of dx (row stride ld); part [chunk][slice][V * 32].
  // width: live columns of dx (the last slice may be partial: its dead columns re-read the last live one and are
  // dropped by the final stage); x_seg: rows a sequence owns in dx (>= T; tokens and lengths count T per sequence).
  extern __shared__ float eg_tab[];      // [EG_GROUPS][V * 32]
  const int g = threadIdx.x >> 5;
  const int c = min((int)(threadIdx.x & 31), width - 1 - 32 * (int)blockIdx.y);
  dx += blockIdx.y * 32;
  const int n = V * 32;
  for (int i = threadIdx.x; i < EG_GROUPS * n; i += 32 * EG_GROUPS) eg_tab[i] = 0.0f;
  __syncthreads();
  float* tab = eg_tab + g * n;
  const long r0 = (long)blockIdx.x * chunk + (long)g * (chunk / EG_GROUPS);     // chunk: a multiple of EG_BATCH EG_GROUPS
  const long r1 = min(r0 + chunk / EG_GROUPS, rows);
  // (sequence, step) of the row in hand, advanced by increments: a 64-bit division per row cost more than
  // everything else in this loop together
  const int nseq = (int)(rows / T);
  int sq = (int)(min(r0, rows - 1) / T), st = (int)(min(r0, rows - 1) % T);
  for (long rb = r0; rb < r1; rb += EG_BATCH) {
    // a batch of rows' loads first (independent: the loop is bound by bytes in flight, two waves per CU), then
    // their additions in ascending row order
    float v[EG_BATCH];
    int tk[EG_BATCH];
#pragma unroll
    for (int k = 0; k < EG_BATCH; ++k) {
      // every load unconditional (a load in the arm of a select or behind && becomes a branch, and the join
      // waits for all loads in flight: one memory round trip per row)
      const long r = min(rb + k, rows - 1);
      const int sqc = min(sq, nseq - 1);
      const int len = min(max(lengths[sqc], 0), T);
      const int tv = tokens[r];
      v[k] = dx[((long)sqc * x_seg + st) * ld + c];
      const bool live = (rb + k < r1) & (st < len);
      tk[k] = live ? min(max(tv, 0), V - 1) : -1;
      const bool wrap = st + 1 == T;
      st = wrap ? 0 : st + 1;
      sq += wrap ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < EG_BATCH; ++k)
      if (tk[k] >= 0) tab[tk[k] * 32 + (threadIdx.x & 31)] += v[k];
  }
  __syncthreads();
  float* out = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * n;
  for (int i = threadIdx.x; i < n; i += 32 * EG_GROUPS)
    out[i] = ((eg_tab[i] + eg_tab[n + i]) + eg_tab[2 * n + i]) + eg_tab[3 * n + i];
}

// out[i] (+)= part[0][i] + part[1][i] + ... (ascending), i < n
// (blockIdx.y = column slice; out row v = i / 32 has stride ldo)
extern "C" __global__ __launch_bounds__(256) void s2g_k_embed_grad_final(const float* __restrict__ part, int nchunks,
                                                                         int n, float* __restrict__ out, int ldo,
                                                                         int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int nsl = gridDim.y, sl = blockIdx.y;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int k = 0;
  for (; k + 3 < nchunks; k += 4) {      // four chunks' loads in flight; one fixed order of additions
    s0 += part[((size_t)k * nsl + sl) * n + i];
    s1 += part[((size_t)(k + 1) * nsl + sl) * n + i];
    s2 += part[((size_t)(k + 2) * nsl + sl) * n + i];
    s3 += part[((size_t)(k + 3) * nsl + sl) * n + i];
  }
  for (; k < nchunks; ++k) s0 += part[((size_t)k * nsl + sl) * n + i];
  const float s = (s0 + s1) + (s2 + s3);
  float* o = out + (size_t)(i >> 5) * ldo + sl * 32 + (i & 31);
  *o = accumulate ? *o + s : s;
}

// the same final stage written TRANSPOSED: out[(sl * 32 + c) * ldo + v] for columns below `width` (the bucketed sums
// of gate gradients enter a weight gradient stored [gate row][token])
extern "C" __global__ __launch_bounds__(256) void s2g_k_embed_grad_final_t(const float* __restrict__ part, int nchunks,
                                                                           int n, int width, float* __restrict__ out,
                                                                           int ldo) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int nsl = gridDim.y, sl = blockIdx.y;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int k = 0;
  for (; k + 3 < nchunks; k += 4) {
    s0 += part[((size_t)k * nsl + sl) * n + i];
    s1 += part[((size_t)(k + 1) * nsl + sl) * n + i];
    s2 += part[((size_t)(k + 2) * nsl + sl) * n + i];
    s3 += part[((size_t)(k + 3) * nsl + sl) * n + i];
  }
  for (; k < nchunks; ++k) s0 += part[((size_t)k * nsl + sl) * n + i];
  const int col = sl * 32 + (i & 31);
  if (col < width) out[(size_t)col * ldo + (i >> 5)] = (s0 + s1) + (s2 + s3);
}

// out[l][c] = sum over sequences n with label l (ascending) of x[n][c]; one workgroup per class, thread = column
extern "C" __global__ void s2g_k_bucket_rows_by_label(const long long* __restrict__ labels, int B,
                                                      const float* __restrict__ x, int W, float* __restrict__ out) {
  const int l = blockIdx.x, c = threadIdx.x;
  if (c >= W) return;
  float s = 0.0f;
#pragma unroll 8
  for (int b = 0; b < B; ++b) {          // unconditional loads: eight rows in flight
    const float v = x[(size_t)b * W + c];
    s += labels[b] == l ? v : 0.0f;
  }
  out[(size_t)l * W + c] = s;
}

// gradient of the label half: dlab[l] (+)= sum over sequences with label l and live rows of dx[row][32:64].
// Stage 1: one workgroup per sequence sums its live rows (8 groups of 32 columns, fixed order) into part[b][32];
// stage 2: one workgroup per class adds the sequences of that class in ascending order.
extern "C" __global__ __launch_bounds__(256) void s2g_k_embed_grad_labels_part(const int* __restrict__ lengths, int T,
                                                                               const float* __restrict__ dx,
                                                                               float* __restrict__ part) {
  __shared__ float red[8][32];
  const int b = blockIdx.x, c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  const float* p = dx + (size_t)b * T * 64 + 32 + c;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int t = g;
  for (; t + 24 < len; t += 32) {
    s0 += p[(size_t)t * 64]; s1 += p[(size_t)(t + 8) * 64]; s2 += p[(size_t)(t + 16) * 64]; s3 += p[(size_t)(t + 24) * 64];
  }
  for (; t < len; t += 8) s0 += p[(size_t)t * 64];
  red[g][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0) {
    float r = 0.0f;
    for (int k = 0; k < 8; ++k) r += red[k][c];
    part[b * 32 + c] = r;
  }
}

extern "C" __global__ __launch_bounds__(32) void s2g_k_embed_grad_labels(const long long* __restrict__ labels, int B,
                                                                         const float* __restrict__ part,
                                                                         float* __restrict__ dlab, int accumulate) {
  const int l = blockIdx.x, c = threadIdx.x;
  float s = 0.0f;
  for (int b = 0; b < B; ++b)
    if (labels[b] == l) s += part[b * 32 + c];
  dlab[l * 32 + c] = accumulate ? dlab[l * 32 + c] + s : s;
}

// ---- attention pooling (wgan.py:82-89, 116-118) over Tp = the longest sequence of the batch (what
// pad_packed_sequence returns): scores_t = wc . tanh(pre_t), pre = out Wa^T + ba (a batched GEMM);
// weights = softmax over ALL Tp positions (padded rows hold out = 0, so their score is wc . tanh(ba): they
// take part in the softmax, as in the reference); pooled = sum_t weights_t out_t; D = wl . pooled + bl.
// One workgroup per sequence.  Saves tanh(pre) in place, the weights and pooled for the backward pass.
#define AP_NW 16        // waves per workgroup: the row loops are latency bound, so a sequence gets a whole CU's worth
extern "C" __global__ __launch_bounds__(64 * AP_NW) void s2g_k_attnpool_fwd(float* __restrict__ pre, const float* __restrict__ out,
                                                                     const float* __restrict__ wc,
                                                                     const float* __restrict__ wl,
                                                                     const float* __restrict__ bl, int T, int Tp,
                                                                     float* __restrict__ weights,
                                                                     float* __restrict__ pooled, float* __restrict__ dval) {
  extern __shared__ float sc[];             // [Tp] scores, then weights
  __shared__ float red[AP_NW][X_H];
  __shared__ float stat[2];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t r0 = (size_t)n * T;
  const float wk = wc[lane];
  // (the row loops are unrolled eight deep so that eight rows' loads are in flight together: one row at a time,
  // every iteration paid a memory round trip)
  for (int tb = wave; tb < Tp; tb += 8 * AP_NW) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = pre[(r0 + min(tb + AP_NW * j, Tp - 1)) * X_H + lane];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = tb + AP_NW * j;
      if (t < Tp) {
        const float a = tanhf(v[j]);
        pre[(r0 + t) * X_H + lane] = a;
        float s = wk * a;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
        if (lane == 0) sc[t] = s;
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    float mx = -INFINITY;
    for (int t = lane; t < Tp; t += 64) mx = fmaxf(mx, sc[t]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.0f;
    for (int t = lane; t < Tp; t += 64) sum += expf(sc[t] - mx);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) { stat[0] = mx; stat[1] = 1.0f / sum; }
  }
  __syncthreads();
  float acc = 0.0f;
#pragma unroll 8
  for (int t = wave; t < Tp; t += AP_NW) {
    const float wgt = expf(sc[t] - stat[0]) * stat[1];
    if (lane == 0) weights[(size_t)n * T + t] = wgt;
    acc = fmaf(wgt, out[(r0 + t) * X_H + lane], acc);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    float p = 0.0f;
#pragma unroll
    for (int w = 0; w < AP_NW; ++w) p += red[w][lane];
    pooled[(size_t)n * X_H + lane] = p;
    float s = wl[lane] * p;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    if (lane == 0) dval[n] = s + bl[0];
  }
}

// backward of the pooling for d(loss)/d(D_n) = dD[n]: d(out) (direct part: weights_t * dpooled), dpre =
// ds_t * wc * (1 - a^2) written over a, and this sequence's share of d(wc); d(wl), d(bl) follow from pooled.
// dpooled [nseq, 64] (optional): the upstream gradient of the pooled vector itself, instead of dD[n] * wl (the
// module-level Attention.forward returns the pooled vector); dweights [nseq, T] (optional): an upstream gradient
// on the attention weights (its second return value), added to what flows back through the pooled vector.
extern "C" __global__ __launch_bounds__(64 * AP_NW) void s2g_k_attnpool_bwd(float* __restrict__ a_dpre, const float* __restrict__ out,
                                                                     const float* __restrict__ wc,
                                                                     const float* __restrict__ wl,
                                                                     const float* __restrict__ weights,
                                                                     const float* __restrict__ dD, int T, int Tp,
                                                                     float* __restrict__ dout, float* __restrict__ dwc_part,
                                                                     const float* __restrict__ dpooled,
                                                                     const float* __restrict__ dweights) {
  extern __shared__ float dw[];             // [Tp] d(weights), then d(scores)
  __shared__ float red[AP_NW][X_H];
  __shared__ float dot_s;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t r0 = (size_t)n * T;
  const float dp = dpooled ? dpooled[(size_t)n * X_H + lane] : dD[n] * wl[lane];      // d(pooled)
  for (int tb = wave; tb < Tp; tb += 8 * AP_NW) {
    float ov[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = out[(r0 + min(tb + AP_NW * j, Tp - 1)) * X_H + lane];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = tb + AP_NW * j;
      if (t < Tp) {
        float s = dp * ov[j];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
        if (lane == 0) dw[t] = dweights ? s + dweights[(size_t)n * T + t] : s;
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    float d = 0.0f;
    for (int t = lane; t < Tp; t += 64) d = fmaf(weights[(size_t)n * T + t], dw[t], d);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) d += __shfl_xor(d, o);
    if (lane == 0) dot_s = d;
  }
  __syncthreads();
  const float wk = wc[lane];
  float acc = 0.0f;
  for (int tb = wave; tb < Tp; tb += 8 * AP_NW) {
    float av[8], wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tc = min(tb + AP_NW * j, Tp - 1);
      av[j] = a_dpre[(r0 + tc) * X_H + lane];
      wv[j] = weights[(size_t)n * T + tc];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = tb + AP_NW * j;
      if (t < Tp) {
        const float dsc = wv[j] * (dw[t] - dot_s);             // softmax backward
        acc = fmaf(dsc, av[j], acc);                           // d(wc)
        a_dpre[(r0 + t) * X_H + lane] = dsc * wk * (1.0f - av[j] * av[j]);
        dout[(r0 + t) * X_H + lane] = wv[j] * dp;
      }
    }
  }
  for (int t = Tp + wave; t < T; t += AP_NW) {                 // rows the pooling never saw
    a_dpre[(r0 + t) * X_H + lane] = 0.0f;
    dout[(r0 + t) * X_H + lane] = 0.0f;
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    float p = 0.0f;
#pragma unroll
    for (int w = 0; w < AP_NW; ++w) p += red[w][lane];
    dwc_part[(size_t)n * X_H + lane] = p;
  }
}

// critic loss of the weight-clipping WGAN (wgan.py:145-147, 189): values = [real B | fake B];
// loss = -(mean(real) - mean(fake)); dD = (-1/B, +1/B).  One workgroup.
extern "C" __global__ __launch_bounds__(256) void s2g_k_wasserstein_loss(const float* __restrict__ dval, int B,
                                                                         float* __restrict__ dD, float* __restrict__ loss) {
  __shared__ float red[2][256];
  float sr = 0.0f, sf = 0.0f;
  for (int b = threadIdx.x; b < B; b += 256) {
    sr += dval[b]; sf += dval[B + b];
    dD[b] = -1.0f / B; dD[B + b] = 1.0f / B;
  }
  red[0][threadIdx.x] = sr; red[1][threadIdx.x] = sf;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, c = 0.0f;
    for (int k = 0; k < 256; ++k) { a += red[0][k]; c += red[1][k]; }
    loss[0] = -(a / B - c / B);
  }
}

// p <- clamp(p, -clip, clip): the weight clipping after the critic's optimizer step (wgan.py:155-158, 198)
extern "C" __global__ void s2g_k_clamp(float* __restrict__ p, long n, float clip) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = fminf(fmaxf(p[i], -clip), clip);
}

// a[i] *= b[i]  (an input gradient times the dropout mask of the layer it came through)
extern "C" __global__ void s2g_k_mul_inplace(float* __restrict__ a, const float* __restrict__ b, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] *= b[i];
}
extern "C" __global__ void s2g_k_add_inplace(float* __restrict__ a, const float* __restrict__ b, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}
