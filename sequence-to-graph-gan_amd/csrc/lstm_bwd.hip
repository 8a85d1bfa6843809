// Reverse-in-time sweeps of the critic's 2-layer LSTM: sweep B (back-propagation through time
// for a given d(out); what autograd does for d_loss.backward(), wgan_gp.py:317, and for the
// input gradient of the penalty, wgan_gp.py:247-254) and sweep D (adjoints of the primal states
// for the loss <v, dD/dx>, i.e. the second-order term that `create_graph=True` makes autograd
// differentiate, wgan_gp.py:251,256,317).
//
// Same shape as the forward sweeps (lstm_fwd.hip): ONE workgroup of EIGHT waves per sequence,
// the three transposed 50x200 mat-vecs of a step running concurrently on different time steps,
// one barrier per step, one cell wave and one projection wave on every SIMD:
//   waves 0,1  t = len-1-i   dh1 = W_hh1^T delta1_{t+1} (+ dhT at the last step) -> cell backward -> delta1_t
//   waves 4..7 t = len-i     dy  = mask * (W_ih1^T delta1_t)                      -> LDS
//   waves 2,3  t = len+1-i   dh0 = dy + W_hh0^T delta0_{t+1}                      -> cell backward -> delta0_t
// Cell waves: lane = (unit group g of 4 units, slice s of 8): it accumulates the 4 outputs
// k = 4g + ((s&3) ^ q), q = 0..3, over gate rows 25s..25s+24 (100 weights in registers, packed
// pairs -> 50 v_pk_fma_f32 per step; the delta slice comes from LDS in 7 16-byte reads).  A quad
// reduce-scatter (3 DPP adds, no selects thanks to the per-lane slot order) and one row shift
// leave the total of unit 4g+s in lanes s < 4, which run the cell backward of that unit.
// Projection waves: lane = (unit pair p, slice s of 8), 2 outputs x 25 rows -> 25 v_pk_fma_f32
// and 3 DPP adds.  Every weight gradient is a batched GEMM over the saved delta afterwards.
// Saved activations are prefetched by UNCONDITIONAL loads from clamped addresses (lstm_fwd.hip).
#include "s2g_common.h"

#define PF 6
#ifndef PFD
#define PFD 6   // ring depth of sweep D's cell waves (5 saved tensors per step)
#endif
#define BWD_THREADS 512
#define DVEC 224   // LDS delta vector: 8 slices of 25 gate rows, each padded to 28 floats
#define DPP_ROW_SHL(n) (0x100 + (n))

// position of internal gate row j' in the padded LDS delta vector
__device__ __forceinline__ int dpos(int j) { return 28 * (j / 25) + (j % 25); }

__device__ __forceinline__ void load_dslice(const float* slice, float v[28]) {
  const float4* p = reinterpret_cast<const float4*>(slice);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const float4 a = p[i];
    v[4 * i] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w;
  }
}

// Cell-wave weights of lane (g, s): slot q = output k = 4g + ((s&3) ^ q); gate rows
// j' = 25 s .. 25 s + 24 (internal order); W is a torch-layout [4H, H] matrix.
__device__ __forceinline__ void load_wT_oct(const float* W, int g, int s, f2 w01[25], f2 w23[25]) {
#pragma unroll
  for (int j = 0; j < 25; ++j) {
    const int row = pt_row(25 * s + j) * S2G_H;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 4 * g + ((s & 3) ^ q);
      const float x = W[row + (k < S2G_H ? k : 0)];
      v[q] = (k < S2G_H) ? x : 0.0f;
    }
    w01[j] = (f2){v[0], v[1]};
    w23[j] = (f2){v[2], v[3]};
  }
}

// (W^T d)[4g + s], complete in lanes s < 4; slice = this lane's 28-float slice of the delta vector
__device__ __forceinline__ float matvecT_oct(const f2 w01[25], const f2 w23[25], const float* slice) {
  float v[28];
  load_dslice(slice, v);
  f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 25; ++j) {
    const f2 vv = {v[j], v[j]};
    a01 = __builtin_elementwise_fma(w01[j], vv, a01);
    a23 = __builtin_elementwise_fma(w23[j], vv, a23);
  }
  float r = quad_reduce_scatter(a01.x, a01.y, a23.x, a23.y);
  r += dpp_mov<DPP_ROW_SHL(4)>(r);      // lanes s < 4 add the partial of lane s + 4
  return r;
}

// Projection-wave weights of lane (p, s): slot q = output k = 2p + ((s&1) ^ q), rows 25s..25s+24
__device__ __forceinline__ void load_wT_pair(const float* W, int p, int s, f2 w[25]) {
#pragma unroll
  for (int j = 0; j < 25; ++j) {
    const int row = pt_row(25 * s + j) * S2G_H;
    w[j] = (f2){W[row + 2 * p + (s & 1)], W[row + 2 * p + ((s & 1) ^ 1)]};
  }
}

// (W^T d)[2p + s], complete in lanes s < 2
__device__ __forceinline__ float matvecT_pair(const f2 w[25], const float* slice) {
  float v[28];
  load_dslice(slice, v);
  f2 a = {0.f, 0.f}, b = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 24; j += 2) {
    a = __builtin_elementwise_fma(w[j], (f2){v[j], v[j]}, a);
    b = __builtin_elementwise_fma(w[j + 1], (f2){v[j + 1], v[j + 1]}, b);
  }
  a = __builtin_elementwise_fma(w[24], (f2){v[24], v[24]}, a);
  a += b;
  float r = a.x + dpp_mov<DPP_QUAD(1, 0, 3, 2)>(a.y);
  r += dpp_mov<DPP_QUAD(2, 3, 0, 1)>(r);
  r += dpp_mov<DPP_ROW_SHL(4)>(r);
  return r;
}

__device__ __forceinline__ void zero_tail(float* buf, int ws_row, int len, int T, int width) {
  // rows [len, T) of this sequence <- 0 so that batched GEMMs over all rows see no stale data
  if (len >= T) return;
  const size_t base = (size_t)(ws_row + len) * width;
  const size_t cnt = (size_t)(T - len) * width;
  for (size_t i = threadIdx.x; i < cnt; i += blockDim.x) buf[base + i] = 0.0f;
}

#define ISSUE_M(j, i_) { mk[j] = (pm + (unsigned)(max(min(len - (i_), len - 1), 0) * sm))[unit]; }

// projection waves of both reverse sweeps: dy_t = mask_t * (W_ih1^T d1_t) -> dybuf.
// With STORES (sweep D, whose cell waves have the shallower prefetch ring) waves 6 and 7 also write
// the gate gradients of layer 1 / layer 0 to global memory, one step
// behind the cell waves and out of the LDS vectors those publish anyway (out1/out0 = row 0 of this
// sequence).  Loads and stores share one in-order completion counter per wave, so a store issued by
// a cell wave would sit in front of every wait for its prefetched activations: measured, the
// 16-byte store per step cost the second-order sweep a quarter of its time.  Here 50 lanes write
// one full 800-byte row per step.
// sum1/sum0 (optional): the same lanes also add up the gate gradients over the sequence's steps and
// leave the 200 sums of layer 1 / layer 0 there (this sequence's share of the bias gradients, which
// would otherwise cost a pass over all saved gate gradients).
template <bool STORES>
__device__ __forceinline__ void reverse_projection(const S2GSeq& sq, const float* mask, const float* wih1,
                                                   const float* dummy, const float* d1buf, const float* d0buf,
                                                   float* dybuf, float* out1, float* out0, float* sum1,
                                                   float* sum0, int wv, int lane, int nit) {
  const int len = sq.len;
  const bool has_mask = sq.mask_off >= 0 && mask != nullptr;
  const int p = 8 * (wv - 4) + (lane >> 3), s = lane & 7;
  const bool owner = p < 25 && s < 2;
  const int pc = p < 25 ? p : 24;
  const int unit = 2 * pc + (s & 1);
  f2 w[25];
  load_wT_pair(wih1, pc, s, w);
  const float* pm = has_mask ? mask + (size_t)sq.mask_off * S2G_MASKW : dummy;
  const int sm = has_mask ? S2G_MASKW : 0;
  // store duty: wave 6 -> layer 1 (its cell waves handled t = len-i one iteration ago), wave 7 ->
  // layer 0 (t = len+2-i); lane = unit
  const bool summer = sum1 != nullptr && wv >= 6 && lane < S2G_H;
  const bool storer = STORES && wv >= 6 && lane < S2G_H;
  const bool reader = storer || summer;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int su = lane < S2G_H ? lane : 0;
  const float* sbuf = (wv == 6) ? d1buf : d0buf;
  float* sout = ((wv == 6) ? out1 : out0) + 4 * su;
  const int slag = (wv == 6) ? 0 : 2;
  const int q0 = dpos(4 * su), q1 = dpos(4 * su + 1), q2 = dpos(4 * su + 2), q3 = dpos(4 * su + 3);
  __syncthreads();
  float mk[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) ISSUE_M(j, j)
  for (int base = 0; base < nit; base += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = base + j;
      if (i < nit) {
        const int t = len - i;
        const int ts = t + slag;
        float4 dv;
        const bool rd = reader && ts >= 0 && ts < len;
        const bool st = STORES && rd;
        if (rd) {
          const float* b = sbuf + (ts & 1) * DVEC;
          dv = make_float4(b[q0], b[q1], b[q2], b[q3]);
          acc.x += dv.x; acc.y += dv.y; acc.z += dv.z; acc.w += dv.w;
        }
        if (t >= 0 && t < len) {
          const float dy = matvecT_pair(w, d1buf + (t & 1) * DVEC + 28 * s);
          if (owner) dybuf[(t & 1) * 64 + unit] = has_mask ? dy * mk[j] : dy;
        }
        if (st) store_stream(sout + (unsigned)(ts * S2G_G), dv);
        __syncthreads();
      }
      ISSUE_M(j, i + PF)
    }
  }
  // layer 0's last step (t = 0) was published in the last iteration
  if (reader && wv == 7 && len >= 1) {
    const float* b = d0buf;
    const float4 dv = make_float4(b[q0], b[q1], b[q2], b[q3]);
    acc.x += dv.x; acc.y += dv.y; acc.z += dv.z; acc.w += dv.w;
    if (STORES) store_stream(sout, dv);
  }
  if (summer) *reinterpret_cast<float4*>(((wv == 6) ? sum1 : sum0) + 4 * su) = acc;
}

// ------------------------------------------------------------------------------------------
// Sweep B.  dhT[n][k] = d loss / d h1_final.  Writes delta{0,1}[row][4k+q] (gradient wrt the
// gate pre-activations) and, for sequences n >= adj_seq0 (if adj0 != nullptr),
// adj{0,1}[row - adj_row0][2k+{0,1}] = (dh_t, dc_t), the total adjoints of h_t and c_t, which
// sweep D reuses as the adjoints of the tangent states.
// ------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(BWD_THREADS) void s2g_k_lstm2_bwd(
    const S2GSeq* __restrict__ seqs, const float* __restrict__ mask, const float* __restrict__ whh0,
    const float* __restrict__ wih1, const float* __restrict__ whh1, const float* __restrict__ gates0,
    const float* __restrict__ cell0, const float* __restrict__ gates1, const float* __restrict__ cell1,
    const float* __restrict__ dhT, float* __restrict__ delta0, float* __restrict__ delta1,
    float* __restrict__ adj0, float* __restrict__ adj1, int T, int adj_seq0, int adj_row0, S2GHead head,
    float* __restrict__ bsum) {
  __shared__ __attribute__((aligned(16))) float lds[4 * DVEC + 128];
  float* d1buf = lds;                // delta of layer 1 [2][DVEC]
  float* d0buf = lds + 2 * DVEC;     // delta of layer 0 [2][DVEC]
  float* dybuf = lds + 4 * DVEC;     // mask * W_ih1^T delta1 [2][64]

  const int n = blockIdx.x;
  S2GSeq sq = seqs[n];
  sq.len = min(sq.len, T);      // a sequence owns T workspace rows
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: row pointers stay scalar
  const int lane = tid & 63;
  const int len = sq.len;

  // bsum (optional) [2][gridDim.x][200]: per-sequence sums over time of delta0 / delta1
  float* bs0 = bsum ? bsum + (size_t)n * S2G_G : nullptr;
  float* bs1 = bsum ? bsum + ((size_t)gridDim.x + n) * S2G_G : nullptr;
  if (len <= 0) {
    zero_tail(delta0, sq.ws_row, 0, T, S2G_G);
    zero_tail(delta1, sq.ws_row, 0, T, S2G_G);
    if (bsum && tid < S2G_G) { bs0[tid] = 0.0f; bs1[tid] = 0.0f; }
    return;
  }
  for (int i = tid; i < 4 * DVEC + 128; i += BWD_THREADS) lds[i] = 0.0f;
  const int nit = len + 2;

  if (wv >= 4) {
    reverse_projection<false>(sq, mask, wih1, whh0, d1buf, d0buf, dybuf, nullptr, nullptr, bs1, bs0, wv, lane, nit);
  } else {
    // ---------------- cell waves: 0,1 = layer 1 (lag 0), 2,3 = layer 0 (lag 2) ----------------
    const bool top = wv < 2;
    const int lag = top ? 0 : 2;
    const int g = 8 * (wv & 1) + (lane >> 3), s = lane & 7;
    const int u = 4 * g + (s & 3);
    const bool valid = u < S2G_H;
    const bool owner = valid && s < 4;
    const int uc = valid ? u : 0;
    f2 w01[25], w23[25];
    load_wT_oct(top ? whh1 : whh0, g, s, w01, w23);
    // wave-uniform row pointers + constant per-lane offsets: no vector address arithmetic per step
    const unsigned lo4 = 4 * uc, lo2 = 2 * uc;
    const size_t rb = (size_t)sq.ws_row * S2G_G;
    const float* pg = (top ? gates1 : gates0) + rb;
    const float* pc = (top ? cell1 : cell0) + rb;
    float* delta = (top ? delta1 : delta0) + rb;
    float* adj = (adj0 != nullptr && n >= adj_seq0)
                     ? (top ? adj1 : adj0) + ((size_t)sq.ws_row - adj_row0) * (2 * S2G_H) : nullptr;
    float* dbuf = top ? d1buf : d0buf;
    float dh_final = 0.0f;
    if (top) {
      if (head.hT != nullptr) {
        // fc head + loss derivative of this sequence, so that nothing sits between sweeps A and B
        // (all addresses are wave-uniform; same arithmetic as s2g_k_critic_head)
        const float* hrow = head.hT + (size_t)n * S2G_H;
        float o0 = head.bias[0], o1 = head.bias[1], o2 = head.bias[2];
        for (int k = 0; k < S2G_H; ++k) {
          const float hk = hrow[k];
          o0 = fmaf(head.W[k], hk, o0);
          o1 = fmaf(head.W[S2G_H + k], hk, o1);
          o2 = fmaf(head.W[2 * S2G_H + k], hk, o2);
        }
        const int kind = head.mode == 1 ? 3 : n / head.B;      // 0 real, 1 fake, 2 interp, 3 generator pass
        float e0 = 1.0f, e1 = 0.0f, e2 = 0.0f;
        if (kind != 2) {
          float l, c0, c1;
          ce_on_probs(o1, o2, (int)head.labels[n % head.B], head.w0, head.w1, head.inv_wsum, l, c0, c1);
          const float sc = kind == 3 ? 0.25f : 0.5f;
          e0 = kind == 1 ? head.inv_n : -head.inv_n;
          e1 = sc * c0;
          e2 = sc * c1;
        }
        dh_final = valid ? fmaf(e2, head.W[2 * S2G_H + u], fmaf(e1, head.W[S2G_H + u], e0 * head.W[u])) : 0.0f;
      } else {
        dh_final = valid ? dhT[(size_t)n * S2G_H + u] : 0.0f;
      }
    }
    const int p0 = dpos(4 * uc), p1 = dpos(4 * uc + 1), p2 = dpos(4 * uc + 2), p3 = dpos(4 * uc + 3);
    __syncthreads();

    // ring slot i % PF holds step t = len-1-(i-lag)
    float4 rg[PF];
    float2 rc[PF];
#define ISSUE_B(j, i_)                                                       \
    {                                                                        \
      const unsigned o_ = (unsigned)(max(min(len - 1 - ((i_) - lag), len - 1), 0) * S2G_G); \
      rg[j] = *reinterpret_cast<const float4*>(pg + o_ + lo4);               \
      rc[j] = *reinterpret_cast<const float2*>(pc + o_ + lo4);               \
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) ISSUE_B(j, j)

    float dc = 0.0f;
    for (int base = 0; base < nit; base += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int i = base + j;
        if (i < nit) {
          const int t = len - 1 - (i - lag);
          if (t >= 0 && t < len) {
            // delta_{t+1} of this layer (zeros at t = len-1)
            float dh = matvecT_oct(w01, w23, dbuf + ((t + 1) & 1) * DVEC + 28 * s);
            if (top) { if (t == len - 1) dh += dh_final; }
            else dh += dybuf[(t & 1) * 64 + uc];
            const float cprev = (t > 0) ? rc[(j + 1) % PF].x : 0.0f;   // slot of the next iteration = step t-1
            const float gi = rg[j].x, gf = rg[j].y, gg = rg[j].z, go = rg[j].w, tc = rc[j].y;
            const float dct = fmaf(dh * go, 1.0f - tc * tc, dc);
            float4 d;
            d.x = dct * gg * gi * (1.0f - gi);
            d.y = dct * cprev * gf * (1.0f - gf);
            d.z = dct * gi * (1.0f - gg * gg);
            d.w = dh * tc * go * (1.0f - go);
            dc = dct * gf;
            if (owner) {
              *reinterpret_cast<float4*>(delta + (unsigned)(t * S2G_G) + lo4) = d;
              if (adj) *reinterpret_cast<float2*>(adj + (unsigned)(t * 2 * S2G_H) + lo2) = make_float2(dh, dct);
              float* db = dbuf + (t & 1) * DVEC;
              db[p0] = d.x; db[p1] = d.y; db[p2] = d.z; db[p3] = d.w;
            }
          }
          __syncthreads();
        }
        ISSUE_B(j, i + PF)     // refill after the slot's last use: it keeps its registers
      }
    }
  }
  zero_tail(delta0, sq.ws_row, len, T, S2G_G);
  zero_tail(delta1, sq.ws_row, len, T, S2G_G);
}

// ------------------------------------------------------------------------------------------
// Sweep D.  Inputs: primal gates/cell (sweep A), tangent adot/celld (sweep C), adj = (dh, dc)
// of sweep B run with d(out) = e_0.  Output abar{0,1}[row][4k+q]: adjoint of the primal gate
// pre-activation for the loss sum_n fc.weight[0] . hdot1_final(n).  Same pipeline as sweep B
// with abar in the role of delta.
// ------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(BWD_THREADS) void s2g_k_lstm2_bwd2(
    const S2GSeq* __restrict__ seqs, const float* __restrict__ mask, const float* __restrict__ whh0,
    const float* __restrict__ wih1, const float* __restrict__ whh1, const float* __restrict__ gates0,
    const float* __restrict__ cell0, const float* __restrict__ gates1, const float* __restrict__ cell1,
    const float* __restrict__ adot0, const float* __restrict__ celld0, const float* __restrict__ adot1,
    const float* __restrict__ celld1, const float* __restrict__ adj0, const float* __restrict__ adj1,
    float* __restrict__ abar0, float* __restrict__ abar1, int T, float* __restrict__ bsum) {
  __shared__ __attribute__((aligned(16))) float lds[4 * DVEC + 128];
  float* d1buf = lds;
  float* d0buf = lds + 2 * DVEC;
  float* dybuf = lds + 4 * DVEC;

  const int n = blockIdx.x;
  S2GSeq sq = seqs[n];
  sq.len = min(sq.len, T);      // a sequence owns T workspace rows
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: row pointers stay scalar
  const int lane = tid & 63;
  const int len = sq.len;

  float* bs0 = bsum ? bsum + (size_t)n * S2G_G : nullptr;       // as in sweep B: sums over time of abar0 / abar1
  float* bs1 = bsum ? bsum + ((size_t)gridDim.x + n) * S2G_G : nullptr;
  if (len <= 0) {
    zero_tail(abar0, sq.ws_row, 0, T, S2G_G);
    zero_tail(abar1, sq.ws_row, 0, T, S2G_G);
    if (bsum && tid < S2G_G) { bs0[tid] = 0.0f; bs1[tid] = 0.0f; }
    return;
  }
  for (int i = tid; i < 4 * DVEC + 128; i += BWD_THREADS) lds[i] = 0.0f;
  const int nit = len + 2;

  if (wv >= 4) {
    const size_t rb = (size_t)sq.ws_row * S2G_G;
    reverse_projection<true>(sq, mask, wih1, whh0, d1buf, d0buf, dybuf, abar1 + rb, abar0 + rb, bs1, bs0, wv, lane,
                             nit);
  } else {
    const bool top = wv < 2;
    const int lag = top ? 0 : 2;
    const int g = 8 * (wv & 1) + (lane >> 3), s = lane & 7;
    const int u = 4 * g + (s & 3);
    const bool valid = u < S2G_H;
    const bool owner = valid && s < 4;
    const int uc = valid ? u : 0;
    f2 w01[25], w23[25];
    load_wT_oct(top ? whh1 : whh0, g, s, w01, w23);
    const unsigned lo4 = 4 * uc, lo2 = 2 * uc;
    const size_t rb = (size_t)sq.ws_row * S2G_G;
    const float* pg = (top ? gates1 : gates0) + rb;
    const float* pc = (top ? cell1 : cell0) + rb;
    const float* pa = (top ? adot1 : adot0) + rb;
    const float* pcd = (top ? celld1 : celld0) + rb;
    const float* pj = (top ? adj1 : adj0) + (size_t)sq.ws_row * (2 * S2G_H);
    float* dbuf = top ? d1buf : d0buf;
    const int p0 = dpos(4 * uc), p1 = dpos(4 * uc + 1), p2 = dpos(4 * uc + 2), p3 = dpos(4 * uc + 3);
    __syncthreads();

    float4 rg[PFD], ra[PFD];
    float2 rc[PFD], rj[PFD];
    float rcd[PFD];
#define ISSUE_D(j, i_)                                                       \
    {                                                                        \
      const int tc_ = max(min(len - 1 - ((i_) - lag), len - 1), 0);          \
      const unsigned o_ = (unsigned)(tc_ * S2G_G);                           \
      rg[j] = *reinterpret_cast<const float4*>(pg + o_ + lo4);               \
      rc[j] = *reinterpret_cast<const float2*>(pc + o_ + lo4);               \
      ra[j] = *reinterpret_cast<const float4*>(pa + o_ + lo4);               \
      rcd[j] = (pcd + o_)[lo4];                                              \
      rj[j] = *reinterpret_cast<const float2*>(pj + (unsigned)(tc_ * 2 * S2G_H) + lo2); \
    }
#pragma unroll
    for (int j = 0; j < PFD; ++j) ISSUE_D(j, j)

    float cb = 0.0f;   // adjoint of the primal cell state, carried
    for (int base = 0; base < nit; base += PFD) {
#pragma unroll
      for (int j = 0; j < PFD; ++j) {
        const int i = base + j;
        if (i < nit) {
          const int t = len - 1 - (i - lag);
          if (t >= 0 && t < len) {
            float hb = matvecT_oct(w01, w23, dbuf + ((t + 1) & 1) * DVEC + 28 * s);
            if (!top) hb += dybuf[(t & 1) * 64 + uc];
            const float cprev = (t > 0) ? rc[(j + 1) % PFD].x : 0.0f;   // slot of the next iteration = step t-1
            const float cdprev = (t > 0) ? rcd[(j + 1) % PFD] : 0.0f;
            // Second-order cell backward, two gates per packed instruction.  With s = gate derivative
            // factors, X' = gate tangents, (Hd, Cd) = adjoints of the tangent states:
            //   tcb = Hd (od - 2 go tc cd) + hb go                  (adjoint of tanh c)
            //   cbt = cb + tcb (1 - tc^2)
            //   ib  = Cd gd     + cbt gg    + Cd gg    (1 - 2 gi) ai
            //   fb  = Cd cdprev + cbt cprev + Cd cprev (1 - 2 gf) af
            //   gb  = Cd id     + cbt gi    + Cd gi    (-2 gg)    ag
            //   ob  = Hd uu cd  + hb  tc    + Hd tc    (1 - 2 go) ao
            // i.e. r = K (X + Y M A) + S Y for the pairs (ib, fb) and (gb, ob); everything but the S Y
            // term is independent of this step's mat-vec.
            const float tc = rc[j].y, cd = rcd[j], Hd = rj[j].x, Cd = rj[j].y;
            const f2 G01 = {rg[j].x, rg[j].y}, G23 = {rg[j].z, rg[j].w};      // (gi, gf), (gg, go)
            const f2 A01 = {ra[j].x, ra[j].y}, A23 = {ra[j].z, ra[j].w};      // (ai, af), (ag, ao)
            const f2 S01 = __builtin_elementwise_fma(-G01, G01, G01);          // (si, sf) = g (1 - g)
            const f2 S23 = __builtin_elementwise_fma(-G23, G23, (f2){1.0f, G23.y});   // (1 - gg^2, go (1 - go))
            const f2 D01 = S01 * A01, D23 = S23 * A23;                         // (id, fd), (gd, od)
            const float uu = fmaf(-tc, tc, 1.0f);
            const float tcb_pre = Hd * fmaf(-2.0f * (G23.y * tc), cd, D23.y);
            const f2 Y1 = {G23.x, cprev}, Y2 = {G01.x, tc};
            const f2 M1 = __builtin_elementwise_fma((f2){-2.0f, -2.0f}, G01, (f2){1.0f, 1.0f});
            const f2 M2 = __builtin_elementwise_fma((f2){-2.0f, -2.0f}, G23, (f2){0.0f, 1.0f});
            const f2 pre1 = (f2){Cd, Cd} * __builtin_elementwise_fma(Y1 * M1, A01, (f2){D23.x, cdprev});
            const f2 pre2 = (f2){Cd, Hd} * __builtin_elementwise_fma(Y2 * M2, A23, (f2){D01.x, uu * cd});
            const float cb_pre = Cd * D01.y;
            // ---- the part behind the mat-vec
            const float tcb = fmaf(hb, G23.y, tcb_pre);
            const float cbt = fmaf(tcb, uu, cb);
            const f2 r1 = __builtin_elementwise_fma((f2){cbt, cbt}, Y1, pre1);   // (ib, fb)
            const f2 r2 = __builtin_elementwise_fma((f2){cbt, hb}, Y2, pre2);    // (gb, ob)
            cb = fmaf(cbt, G01.y, cb_pre);
            const f2 d01 = r1 * S01, d23 = r2 * S23;
            const float4 d = make_float4(d01.x, d01.y, d23.x, d23.y);
            if (owner) {
              float* db = dbuf + (t & 1) * DVEC;
              db[p0] = d.x; db[p1] = d.y; db[p2] = d.z; db[p3] = d.w;
            }
          }
          __syncthreads();
        }
        ISSUE_D(j, i + PFD)     // refill after the slot's last use: it keeps its registers
      }
    }
  }
  zero_tail(abar0, sq.ws_row, len, T, S2G_G);
  zero_tail(abar1, sq.ws_row, len, T, S2G_G);
}
