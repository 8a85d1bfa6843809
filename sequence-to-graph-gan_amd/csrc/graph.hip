// Node features of the transition graph of a token sequence (reference: GraphEncoder.encode,
// graph_encoder.py:57-133).  The graph has one node per distinct symbol of the sequence and a directed
// edge u -> v weighted by the number of times v follows u; the reference computes four centralities per
// node with networkx (graph_encoder.py:86-91):
//   betweenness  Brandes' algorithm on the unweighted digraph, scaled by 1 / ((n-1)(n-2)) for n > 2
//   closeness    (r-1)/sum of incoming distances * (r-1)/(n-1), r = nodes that reach the node
//   pagerank     x <- 0.85 (x A + dangling mass / n) + 0.15 / n on the row-normalised weights, until the
//                L1 change is below n * 1e-6 (at most 100 iterations)
//   katz         solve (I - 0.1 W^T) c = 1 (dense LU, partial pivoting), c / (sign(sum c) ||c||_2); n > 2 only
// ONE workgroup of 16 waves per sequence, everything in float64 like the reference; every phase is a chain of
// short dependent steps (LDS round trips, barriers), so the design is about waves in flight, not arithmetic:
//   build    transition counts as packed 16-bit pairs in LDS (atomics), then in/out neighbour bitmasks and the
//            in-edge lists of the pagerank product
//   traverse waves 1..15 run Brandes' algorithm, one source per wave at a time (sources s = w-1, w+14, ...):
//            a level-synchronous breadth-first search over the bitmasks (a lane owns nodes lane, lane+64, ...;
//            path counts and dependencies of the wave's source in LDS), which also yields the closeness
//            distances; wave 0 runs the whole pagerank iteration beside them with wave-level reductions only
//   solve    the Katz system by Gauss-Jordan elimination with partial pivoting on a matrix held in REGISTERS:
//            thread (ty, tx) of a 32 x 32 grid owns rows 32 i + ty, columns 32 j + tx (6 x 6 values for
//            n <= 192).  Rows are never moved: a finished-rows mask and a position table reproduce LAPACK's
//            pivot order.  The owners of column k are one half-wave, so the pivot search is four DPP steps and
//            one shuffle; multipliers and the pivot row go through LDS, and the rank-1 update trails two steps
//            behind the search (gr_gj_steps): one barrier per step.  Graphs with more than 192 nodes use LU on
//            a matrix in a caller-provided global workspace (L2-resident) instead.
// Node numbering: ascending symbol id (every quantity above is independent of the numbering except for
// the order of floating-point sums, which differs from networkx's in the last bits).
#pragma once
#include "s2g_common.h"

#define GR_THREADS 1024
#define GR_WAVES 16
#define GR_NMAX 256               // nodes per graph
#define GR_T 6                    // LU register tile
#define GR_LU_NMAX (32 * GR_T)    // largest graph of the register path
#define GR_LDS_LIMIT 163840       // 160 KB per workgroup on gfx950

struct GraphArgs {
  const int* tokens;      // [n_seq][L] symbol ids in [0, V) (clamped into the range)
  const int* lengths;     // [n_seq] or nullptr (every sequence has L symbols)
  int L, V;
  double* feat;           // [n_seq][V][4]: katz, betweenness, closeness, pagerank at the symbol's id; 0 if absent
  int* info;              // [n_seq][4]: nodes, status (bit 0: pagerank did not converge, bit 1: Katz matrix
                          //             singular), pagerank iterations, distinct edges
  double* work;           // [n_seq][V*V] Katz matrices of graphs with more than GR_LU_NMAX nodes
  int ecap;               // capacity of the in-edge lists (distinct edges <= L - 1)
  int state_off;          // offset of the traversal / solver state inside the shared region: behind the count
                          // matrix (V <= GR_LU_NMAX: the solver reads the counts) or 0 (it overlays the counts)
};

// byte offsets of the LDS regions (shared by the host, which sizes the allocation, and the kernel)
struct GraphLds {
  int idx, sym, misc, red, inmask, outmask, invs, colptr, eval, esrc, uni, total;
};
__host__ __device__ inline int graph_cnt_bytes(int V) { return (2 * V * V + 4 + 15) & ~15; }
// sigma[16][V], coeff[16][V], x[V] doubles, reach[V], tot[V] ints; solver arrays of S = max(V, GR_LU_NMAX) entries:
// bvec, lmul[3], prow[2], sol doubles, pos, rowat, perm ints
__host__ __device__ inline int graph_solver_stride(int V) { return V > GR_LU_NMAX ? V : GR_LU_NMAX; }
__host__ __device__ inline int graph_state_bytes(int V) {
  const int S = graph_solver_stride(V);
  return (8 * (33 * V + 7 * S) + 4 * (2 * V + 3 * S) + 15) & ~15;
}
__host__ __device__ inline GraphLds graph_lds_layout(int V, int ecap, int state_off) {
  GraphLds o;
  int p = 0;
  o.idx = p; p += 4 * GR_NMAX;
  o.sym = p; p += 4 * GR_NMAX;
  o.misc = p; p += 4 * 32;
  o.red = p; p += 8 * 32;
  o.inmask = p; p += 32 * V;
  o.outmask = p; p += 32 * V;
  o.invs = p; p += 8 * V;
  o.eval = p; p += 8 * ecap;
  o.colptr = p; p += 4 * (V + 2);
  o.esrc = p; p += 2 * ecap;
  p = (p + 15) & ~15;
  o.uni = p;
  const int a = graph_cnt_bytes(V), b = state_off + graph_state_bytes(V);
  p += a > b ? a : b;
  o.total = p;
  return o;
}

__device__ __forceinline__ double gr_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// sum over the workgroup, the same value (bit for bit) in every thread
__device__ __forceinline__ double gr_block_sum(double v, double* red) {
  v = gr_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < GR_WAVES; ++w) s += red[w];
  return s;
}

__device__ __forceinline__ unsigned gr_cnt(const unsigned* cnt32, int e) { return (cnt32[e >> 1] >> (16 * (e & 1))) & 0xffffu; }

// element `which` of a register array as a sum of masked terms (a chain of selects on the index gets rewritten
// by the compiler into a dynamically indexed array, i.e. the whole tile in scratch memory)
__device__ __forceinline__ double gr_pick(const double (&v)[GR_T], int which) {
  double r = 0.0;
#pragma unroll
  for (int j = 0; j < GR_T; ++j) r += (which == j) ? v[j] : 0.0;
  return r;
}

#ifdef GR_STAMPS      // development aid: shader-clock stamps into the (otherwise unused) workspace of graph 0
#define GR_STAMP(slot) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) gr_stamp_buf[16 * (threadIdx.x >> 6) + (slot)] = (double)clock64(); } while (0)
#define GR_ACC(slot, v) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) gr_stamp_buf[16 * (threadIdx.x >> 6) + (slot)] += (double)(v); } while (0)
__device__ double* gr_stamp_buf;
#else
#define GR_STAMP(slot) do {} while (0)
#define GR_ACC(slot, v) do {} while (0)
#endif

struct GrSolve {      // LDS state of the Katz solver
  double *bvec, *lmul, *prow, *sol;
  int *pos, *rowat;
  volatile int* misc;
  int n, S;      // nodes; entries per solver array (>= GR_LU_NMAX)
};

template <int CTRL>
__device__ __forceinline__ int gr_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ double gr_dpp_d(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)gr_dpp_i<CTRL>((int)(unsigned)b), hi = (unsigned)gr_dpp_i<CTRL>((int)(unsigned)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
// keep the better of two pivot candidates: larger magnitude, then the smaller key (position in LAPACK's row order)
__device__ __forceinline__ void gr_better(double& bv, int& key, double ov, int okey) {
  if (ov > bv || (ov == bv && okey < key)) { bv = ov; key = okey; }
}

// Gauss-Jordan steps with partial pivoting for the columns ks = 32 KJ .. 32 KJ + 31 of the register-tiled
// matrix, ONE workgroup barrier per step.  Step s publishes (before barrier s) its pivot row, the
// multipliers l_s[r] = M[r][s] / pivot of every other row and the updated right-hand side; the rank-1 update of
// the rest of the matrix trails behind: iteration ks applies update ks-2 (its pivot row was put in LDS during
// iteration ks-1), while the half-wave that owns column ks brings that one column up to date by itself (update
// ks-2 from LDS, update ks-1 with the pivot-row element it holds) and searches pivot ks.  Eliminating above the
// pivot as well costs nothing here (every thread updates its whole tile anyway) and leaves a diagonal system, so
// there is no back substitution.  Returns true when a zero pivot column was met.
template <int KJ>
__device__ __forceinline__ bool gr_gj_steps(double (&m)[GR_T][GR_T], unsigned& fin, const GrSolve& z, int tx, int ty,
                                            int lane) {
  // fused multiply-adds are allowed here (LAPACK's kernels use them too); the other phases keep separate roundings
  const int n = z.n, V = z.S;      // stride of the multiplier / pivot-row buffers
  for (int kc = 0; kc < 32; ++kc) {
    const int ks = 32 * KJ + kc;
    if (ks >= n) break;
#ifdef GR_STAMPS
    const long long st0 = clock64();
    long long st1 = st0;
#endif
    int pkp = 0;      // pivot row of step ks - 1
    if (ks >= 1) {
      pkp = __builtin_amdgcn_readfirstlane(z.misc[13 + ((ks - 1) & 1)]);
      if (ty == (pkp & 31)) fin |= 1u << (pkp >> 5);
    }
    const double* lm2 = z.lmul + ((ks + 1) % 3) * V;      // multipliers of step ks - 2
    const double* lm1 = z.lmul + ((ks + 2) % 3) * V;      // multipliers of step ks - 1
    double* lm0 = z.lmul + (ks % 3) * V;                   // multipliers of step ks (written here)
    const double* pr2 = z.prow + (ks & 1) * V;             // pivot row of step ks - 2
    double* pr1 = z.prow + ((ks + 1) & 1) * V;             // pivot row of step ks - 1 (written here)
    // LDS loads below are UNCONDITIONAL and the conditions are applied to the loaded values: a load under a
    // per-lane condition becomes a branch with its own wait for LDS inside, i.e. one full LDS round trip per use
    // instead of one per batch (this section was 5 000 cycles per step that way).  The solver's arrays hold
    // GR_LU_NMAX entries whatever n is, so row 32 i + ty is always inside them and its address is one base
    // register per array plus an immediate offset.
    bool rok[GR_T];
#pragma unroll
    for (int i = 0; i < GR_T; ++i) rok[i] = 32 * i + ty < n;
    if (tx == kc) {      // a half-wave: the owners of column ks; the other waves wait for what it publishes
      __builtin_amdgcn_s_setprio(3);
      double l2[GR_T], l1[GR_T];
      int ps[GR_T];
#pragma unroll
      for (int i = 0; i < GR_T; ++i) { l2[i] = lm2[32 * i + ty]; l1[i] = lm1[32 * i + ty]; ps[i] = z.pos[32 * i + ty]; }
      double p2 = pr2[ks];
      const int rk = z.rowat[ks];
      p2 = ks >= 2 ? p2 : 0.0;
#pragma unroll
      for (int i = 0; i < GR_T; ++i) {
        l2[i] = (ks >= 2 && rok[i]) ? l2[i] : 0.0;      // (a stale buffer may hold anything, NaN included)
        l1[i] = (ks >= 1 && rok[i]) ? l1[i] : 0.0;
        m[i][KJ] -= l2[i] * p2;
      }
      {
        double col[GR_T];
#pragma unroll
        for (int i = 0; i < GR_T; ++i) col[i] = m[i][KJ];
        double p1 = __shfl(gr_pick(col, pkp >> 5), (lane & 32) + (pkp & 31));
        p1 = ks >= 1 ? p1 : 0.0;
#pragma unroll
        for (int i = 0; i < GR_T; ++i) m[i][KJ] -= l1[i] * p1;
      }
      // pivot: largest magnitude among the unfinished rows, the row LAPACK's row order lists first on ties
      double gv = -1.0, bs = 0.0;
      int gk = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < GR_T; ++i) {
        const double av = ((fin >> i) & 1u) ? -2.0 : fabs(m[i][KJ]);
        const int kk = (ps[i] << 8) | (32 * i + ty);
        const bool take = av > gv || (av == gv && kk < gk);
        gv = take ? av : gv;
        gk = take ? kk : gk;
        bs = take ? m[i][KJ] : bs;
      }
      gr_better(gv, gk, gr_dpp_d<DPP_QUAD(1, 0, 3, 2)>(gv), gr_dpp_i<DPP_QUAD(1, 0, 3, 2)>(gk));
      gr_better(gv, gk, gr_dpp_d<DPP_QUAD(2, 3, 0, 1)>(gv), gr_dpp_i<DPP_QUAD(2, 3, 0, 1)>(gk));
      gr_better(gv, gk, gr_dpp_d<DPP_ROW_HALF_MIRROR>(gv), gr_dpp_i<DPP_ROW_HALF_MIRROR>(gk));
      gr_better(gv, gk, gr_dpp_d<DPP_ROW_MIRROR>(gv), gr_dpp_i<DPP_ROW_MIRROR>(gk));
      gr_better(gv, gk, __shfl_xor(gv, 16), __shfl_xor(gk, 16));
      if (!(gv > 0.0)) {      // uniform over the half-wave
        if (ty == 0) z.misc[12] = 1;
      } else {
        const int br = gk & 255;
        const int wl = (lane & 32) + (br & 31);      // the winner: its own best candidate is the pivot
        const double piv = __shfl(bs, wl);
        double bo[GR_T];      // right-hand side of this thread's rows: all reads first, then all writes
#pragma unroll
        for (int i = 0; i < GR_T; ++i) bo[i] = z.bvec[32 * i + ty];
        const double bk = z.bvec[br];
        const double rinv = 1.0 / piv;
#pragma unroll
        for (int i = 0; i < GR_T; ++i) {
          const int r = 32 * i + ty;
          const double l = (r != br) ? m[i][KJ] * rinv : 0.0;
          if (rok[i]) {      // stores only: nothing waits inside these branches
            lm0[r] = l;
            z.bvec[r] = bo[i] - l * bk;      // (l = 0 on the pivot row: its entry is rewritten unchanged)
          }
        }
        if (ty == 0) {
          const int p = gk >> 8;      // the pivot row's place in LAPACK's row order; it swaps with the row at place ks
          z.misc[13 + (ks & 1)] = br;
          z.sol[ks] = piv;
          z.pos[br] = ks; z.pos[rk] = p; z.rowat[ks] = br; z.rowat[p] = rk;
        }
      }
      __builtin_amdgcn_s_setprio(0);
#ifdef GR_STAMPS
      st1 = clock64();
#endif
    }
    if (ks >= 2) {      // trailing update ks - 2 on the columns beyond ks: tile columns below KJ are finished
      double l2[GR_T], p[GR_T];      // (read here, not before the search: the tile leaves few registers to hold them)
#pragma unroll
      for (int i = 0; i < GR_T; ++i) l2[i] = lm2[32 * i + ty];
#pragma unroll
      for (int j = KJ; j < GR_T; ++j) p[j] = pr2[32 * j + tx];
#pragma unroll
      for (int i = 0; i < GR_T; ++i) l2[i] = rok[i] ? l2[i] : 0.0;
#pragma unroll
      for (int j = KJ; j < GR_T; ++j) { const int c = 32 * j + tx; p[j] = (c < n && c > ks) ? p[j] : 0.0; }
#pragma unroll
      for (int i = 0; i < GR_T; ++i)
#pragma unroll
        for (int j = KJ; j < GR_T; ++j) m[i][j] -= l2[i] * p[j];
    }
    if (ks >= 1 && ty == (pkp & 31)) {      // pivot row of step ks - 1, now current through update ks - 2
      const int ip = pkp >> 5;
#pragma unroll
      for (int i = 0; i < GR_T; ++i) {
        if (ip == i) {      // scalar test: a branch around plain stores of row i
#pragma unroll
          for (int j = KJ; j < GR_T; ++j) {
            const int c = 32 * j + tx;
            if (c < n) pr1[c] = m[i][j];
          }
        }
      }
    }
#ifdef GR_STAMPS
    const long long st2 = clock64();
#endif
    __syncthreads();
#ifdef GR_STAMPS
    GR_ACC(8, st1 - st0);                 // owner chain (only the owner wave's lane-0 half adds a nonzero value)
    GR_ACC(9, st2 - st1);                 // trailing update + pivot-row store
    GR_ACC(10, clock64() - st2);          // barrier wait
#endif
    if (z.misc[12]) return true;      // uniform
  }
  return false;
}

template <int NS>
__device__ __forceinline__ bool gr_any(const unsigned long long (&F)[NS]) {
  unsigned long long a = 0ull;
#pragma unroll
  for (int k = 0; k < NS; ++k) a |= F[k];
  return a != 0ull;
}

// Brandes' algorithm for the sources of wave wv (1..15) of a graph of at most 64 NS nodes: a lane owns nodes
// lane + 64 k, k < NS.  Leaves the wave's betweenness sums in its row of sigma and adds its reach / distance
// sums (closeness) into reach / tot.
template <int NS>
__device__ __forceinline__ void gr_brandes(int n, int V, int lane, int wv, const unsigned long long* inmask,
                                           const unsigned long long* outmask, double* sigma, double* coeff, int* reach,
                                           int* tot) {
#pragma clang fp contract(off)
    volatile double* sg = sigma + wv * V;
    volatile double* cf = coeff + wv * V;
    // the neighbour masks of this lane's nodes do not depend on the source: registers, not LDS reads per level
    // (up to 192 nodes; the 32 mask words of a 256-node graph would not fit beside the rest and stay in LDS)
    constexpr bool REGS = NS <= 3;
    constexpr int NM = REGS ? NS : 1;
    unsigned long long inm[NM][NM], outm[NM][NM];
    if (REGS) {
#pragma unroll
      for (int k = 0; k < NM; ++k) {
        const int v = lane + 64 * k;
#pragma unroll
        for (int q = 0; q < NM; ++q) {
          inm[k][q] = v < n ? inmask[4 * v + q] : 0ull;
          outm[k][q] = v < n ? outmask[4 * v + q] : 0ull;
        }
      }
    }
    double bet[NS];
    int rch[NS], tt[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) { bet[k] = 0.0; rch[k] = 0; tt[k] = 0; }
#ifdef GR_SKIP_BRANDES      // timing builds only: no traversal (zero betweenness / closeness)
    for (int s = wv - 1; s < 0; s += GR_WAVES - 1) {
#else
    for (int s = wv - 1; s < n; s += GR_WAVES - 1) {
#endif
      int dist[NS];
      unsigned long long F[NS];
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        dist[k] = (lane + 64 * k == s) ? 0 : -1;
        F[k] = (s >> 6) == k ? 1ull << (s & 63) : 0ull;
      }
      if (lane == (s & 63)) sg[s] = 1.0;
      int d = 0;
      while (gr_any<NS>(F)) {      // at most n levels: every level marks at least one new node
        unsigned long long nf[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          {
            const int v = lane + 64 * k;
            bool hit = false;
            double acc = 0.0;
            if (dist[k] < 0) {      // nodes beyond n have empty masks
#pragma unroll
              for (int q = 0; q < NS; ++q) {
                unsigned long long m = (REGS ? inm[k % NM][q % NM] : (v < n ? inmask[4 * v + q] : 0ull)) & F[q];
                hit = hit || m != 0ull;
                while (m) {      // two path counts in flight per trip
                  const int b0 = __builtin_ctzll(m);
                  m &= m - 1ull;
                  const double s0 = sg[64 * q + b0];
                  double s1 = 0.0;
                  if (m) {
                    const int b1 = __builtin_ctzll(m);
                    m &= m - 1ull;
                    s1 = sg[64 * q + b1];
                  }
                  acc += s0;
                  acc += s1;
                }
              }
            }
            if (hit) { dist[k] = d + 1; sg[v] = acc; }
            nf[k] = __ballot(hit);
          }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) F[k] = nf[k];
        ++d;
      }
      const int dmax = d - 1;      // deepest level reached (the loop ran once more to find it empty)
#pragma unroll
      for (int k = 0; k < NS; ++k)
        if (dist[k] >= 0) { rch[k] += 1; tt[k] += dist[k]; }
      for (int lev = dmax; lev >= 0; --lev) {
        unsigned long long nxt[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) nxt[k] = __ballot(dist[k] == lev + 1);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          const int v = lane + 64 * k;
          if (dist[k] == lev) {
            const double sv = sg[v];
            double dl = 0.0;
#pragma unroll
            for (int q = NS - 1; q >= 0; --q) {
              unsigned long long m = (REGS ? outm[k % NM][q % NM] : (v < n ? outmask[4 * v + q] : 0ull)) & nxt[q];
              while (m) {
                const int b0 = 63 - __builtin_clzll(m);
                m &= ~(1ull << b0);
                const double c0 = cf[64 * q + b0];
                double c1 = 0.0;
                if (m) {
                  const int b1 = 63 - __builtin_clzll(m);
                  m &= ~(1ull << b1);
                  c1 = cf[64 * q + b1];
                }
                dl += sv * c0;
                dl += sv * c1;
              }
            }
            cf[v] = (1.0 + dl) / sv;
            if (v != s) bet[k] += dl;
          }
        }
      }
    }
    // the wave's sums: betweenness into its own (now idle) path-count row, the integer sums by atomics
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int v = lane + 64 * k;
      if (v < n) {
        sg[v] = bet[k];
        if (rch[k]) { atomicAdd(&reach[v], rch[k]); atomicAdd(&tot[v], tt[k]); }
      }
    }
}

extern "C" __global__ __launch_bounds__(GR_THREADS) void s2g_k_graph_features(GraphArgs a) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int V = a.V;
  const GraphLds lo = graph_lds_layout(V, a.ecap, a.state_off);
  int* idx = reinterpret_cast<int*>(smem + lo.idx);
  int* sym = reinterpret_cast<int*>(smem + lo.sym);
  volatile int* misc = reinterpret_cast<int*>(smem + lo.misc);
  double* red = reinterpret_cast<double*>(smem + lo.red);
  unsigned long long* inmask = reinterpret_cast<unsigned long long*>(smem + lo.inmask);
  unsigned long long* outmask = reinterpret_cast<unsigned long long*>(smem + lo.outmask);
  double* invS = reinterpret_cast<double*>(smem + lo.invs);
  int* colptr = reinterpret_cast<int*>(smem + lo.colptr);
  double* eval = reinterpret_cast<double*>(smem + lo.eval);
  unsigned short* esrc = reinterpret_cast<unsigned short*>(smem + lo.esrc);
  char* uni = smem + lo.uni;

  const int len = a.lengths ? min(max(a.lengths[g], 0), a.L) : a.L;
  const int* tk = a.tokens + (size_t)g * a.L;
  double* feat = a.feat + (size_t)g * V * 4;
  int* info = a.info + 4 * g;
  for (int i = tid; i < 4 * V; i += GR_THREADS) feat[i] = 0.0;
#ifdef GR_STAMPS
  if (tid == 0) gr_stamp_buf = a.work;
  __syncthreads();
#endif
  GR_STAMP(0);

  // ---- nodes: the distinct symbols, numbered in ascending order (threads 0..255 = waves 0..3 hold one symbol each)
  if (tid < GR_NMAX) idx[tid] = 0;
  __syncthreads();
  for (int i = tid; i < len; i += GR_THREADS) idx[min(max(tk[i], 0), V - 1)] = 1;
  __syncthreads();
  const int present = tid < GR_NMAX ? idx[tid] : 0;
  const unsigned long long bal = __ballot(present);
  const int within = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0 && wv < 4) misc[wv] = __popcll(bal);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wv && w < 4; ++w) base += misc[w];
  const int n = misc[0] + misc[1] + misc[2] + misc[3];
  if (tid < GR_NMAX) idx[tid] = present ? base + within : -1;
  if (present) sym[base + within] = tid;
  __syncthreads();
  if (n < 2) {      // no node or a single one: the reference assigns zeros (graph_encoder.py:112-120)
    if (tid == 0) { info[0] = n; info[1] = 0; info[2] = 0; info[3] = (n == 1 && len > 1) ? 1 : 0; }
    return;
  }

  // ---- transition counts, packed pairs of 16 bits (a count is at most len - 1 <= 65535)
  unsigned* cnt32 = reinterpret_cast<unsigned*>(uni);
  const int nn = n * n;
  for (int i = tid; i < (nn + 1) / 2; i += GR_THREADS) cnt32[i] = 0u;
  __syncthreads();
  for (int i = tid; i + 1 < len; i += GR_THREADS) {
    const int s = idx[min(max(tk[i], 0), V - 1)], d = idx[min(max(tk[i + 1], 0), V - 1)];
    const int e = s * n + d;
    atomicAdd(&cnt32[e >> 1], 1u << (16 * (e & 1)));
  }
  __syncthreads();

  // ---- neighbour bitmasks (bit j of word q = node 64 q + j), row sums, in-edge lists
  for (int t = tid; t < 4 * n; t += GR_THREADS) {
    const int i = t >> 2, q = t & 3;
    unsigned long long mo = 0ull, mi = 0ull;
    const int j1 = min(64 * q + 64, n);
    for (int j = 64 * q; j < j1; ++j) {
      if (gr_cnt(cnt32, i * n + j)) mo |= 1ull << (j & 63);
      if (gr_cnt(cnt32, j * n + i)) mi |= 1ull << (j & 63);
    }
    outmask[t] = mo;
    inmask[t] = mi;
  }
  if (tid < n) {
    unsigned s = 0;
    for (int j = 0; j < n; ++j) s += gr_cnt(cnt32, tid * n + j);
    invS[tid] = s ? 1.0 / (double)s : 0.0;      // 0 marks a dangling node (no out-edge)
  }
  __syncthreads();
  {
    int deg = 0;
    if (tid < n)
      for (int q = 0; q < 4; ++q) deg += __popcll(inmask[4 * tid + q]);
    int inc = deg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o);
      if (lane >= o) inc += up;
    }
    if (lane == 63 && wv < 4) misc[4 + wv] = inc;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wv && w < 4; ++w) off += misc[4 + w];
    const int start = off + inc - deg;
    if (tid < n) colptr[tid] = start;
    if (tid == n - 1) colptr[n] = start + deg;
    if (tid < n) {
      int p = start;
      for (int q = 0; q < 4; ++q) {
        unsigned long long m = inmask[4 * tid + q];
        while (m) {
          const int i = 64 * q + __builtin_ctzll(m);
          m &= m - 1ull;
          esrc[p] = (unsigned short)i;
          eval[p] = invS[i] * (double)gr_cnt(cnt32, i * n + tid);
          ++p;
        }
      }
    }
  }
#ifdef GR_SKIP_KATZ         // timing builds only: no solve (zero Katz column)
  const bool want_katz = false;
#else
  const bool want_katz = n > 2;
#endif
  const bool lu_regs = n <= GR_LU_NMAX && a.state_off > 0;      // the register path reads the counts later
  double* Mg = a.work + (size_t)g * V * V;
  if (want_katz && !lu_regs) {
    // M = I - 0.1 W^T, row-major with row stride n: M[r][c] = [r == c] - 0.1 * count(c -> r)
    for (int r = wv; r < n; r += GR_WAVES)
      for (int c = lane; c < n; c += 64) Mg[r * n + c] = (r == c ? 1.0 : 0.0) - 0.1 * (double)gr_cnt(cnt32, c * n + r);
  }
  __syncthreads();      // with state_off == 0 the count matrix is dead from here on: its bytes become the state below

  char* st = uni + a.state_off;
  double* sigma = reinterpret_cast<double*>(st);        // [16][V]
  double* coeff = sigma + GR_WAVES * V;                 // [16][V]
  const int S = graph_solver_stride(V);
  double* xv = coeff + GR_WAVES * V;                    // [V]
  double* bvec = xv + V;                                // [S]
  double* lmul = bvec + S;                              // [3][S]
  double* prow = lmul + 3 * S;                          // [2][S]
  double* sol = prow + 2 * S;                           // [S]
  int* reach = reinterpret_cast<int*>(sol + S);         // [V]
  int* tot = reach + V;
  int* pos = tot + V;                                   // [S]
  int* rowat = pos + S;
  int* perm = rowat + S;
  if (tid < n) { reach[tid] = 0; tot[tid] = 0; }
  __syncthreads();
  GR_STAMP(1);

  int status = 0, iters = 0;
  if (wv == 0) {
    // ---- pagerank on one wave: lane owns nodes lane + 64 k; no workgroup barrier inside
    volatile double* x = xv;
    const double p = 1.0 / (double)n;
    double xo[4];
    bool mine[4], dang[4];
    int e0[4], e1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = lane + 64 * k;
      mine[k] = j < n;
      dang[k] = mine[k] && invS[j] == 0.0;
      e0[k] = mine[k] ? colptr[j] : 0;
      e1[k] = mine[k] ? colptr[j + 1] : 0;
      xo[k] = mine[k] ? p : 0.0;
      if (mine[k]) x[j] = p;
    }
    bool done = false;
    for (int it = 1; it <= 100 && !done; ++it) {
      double dpart = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) dpart += dang[k] ? xo[k] : 0.0;
      const double dsum = gr_wave_sum(dpart);
      double xn[4], epart = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double t1 = 0.0;
        for (int e = e0[k]; e < e1[k]; ++e) t1 += x[esrc[e]] * eval[e];
        xn[k] = mine[k] ? 0.85 * (t1 + dsum * p) + (1.0 - 0.85) * p : 0.0;
        epart += fabs(xn[k] - xo[k]);
      }
      const double err = gr_wave_sum(epart);      // every lane's gathers of x are complete before the stores below
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xo[k] = xn[k];
        if (mine[k]) x[lane + 64 * k] = xn[k];
      }
      iters = it;
      done = err < (double)n * 1.0e-6;
    }
    if (!done) status |= 1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (mine[k]) feat[4 * sym[lane + 64 * k] + 3] = xo[k];
  } else {
    // ---- Brandes betweenness + closeness: one breadth-first search per source, a wave per source
    // (one node slot and one mask word per 64 nodes: small graphs do not pay for empty slots)
    if (n <= 64) gr_brandes<1>(n, V, lane, wv, inmask, outmask, sigma, coeff, reach, tot);
    else if (n <= 128) gr_brandes<2>(n, V, lane, wv, inmask, outmask, sigma, coeff, reach, tot);
    else if (n <= 192) gr_brandes<3>(n, V, lane, wv, inmask, outmask, sigma, coeff, reach, tot);
    else gr_brandes<4>(n, V, lane, wv, inmask, outmask, sigma, coeff, reach, tot);
  }
  GR_STAMP(2);
  __syncthreads();
  GR_STAMP(3);
  if (tid < n) {
    double b = 0.0;
#pragma unroll
    for (int w = 1; w < GR_WAVES; ++w) b += sigma[w * V + tid];
    if (n > 2) b *= 1.0 / (double)((n - 1) * (n - 2));
    const int r = reach[tid], t = tot[tid];
    double c = 0.0;
    if (t > 0) {
      c = ((double)r - 1.0) / (double)t;
      c *= ((double)r - 1.0) / (double)(n - 1);
    }
    feat[4 * sym[tid] + 1] = b;
    feat[4 * sym[tid] + 2] = c;
  }
  if (tid == 0) misc[12] = 0;      // singular flag of the solver
  if (tid < n) { bvec[tid] = 1.0; pos[tid] = tid; rowat[tid] = tid; perm[tid] = tid; }
  __syncthreads();

  bool singular = false;
  if (want_katz && lu_regs) {
    // ---- Katz, matrix in registers: M[r][c] = [r == c] - 0.1 count(c -> r), r = 32 i + ty, c = 32 j + tx
    const int tx = tid >> 5, ty = tid & 31;
    double m[GR_T][GR_T];
    unsigned fin = 0;      // bit i: row 32 i + ty is a finished pivot row (or lies beyond n)
#pragma unroll
    for (int i = 0; i < GR_T; ++i) {
      const int r = 32 * i + ty;
      if (r >= n) fin |= 1u << i;
#pragma unroll
      for (int j = 0; j < GR_T; ++j) {
        const int c = 32 * j + tx;
        const double e = (r == c) ? 1.0 : 0.0;
        m[i][j] = (r < n && c < n) ? e - 0.1 * (double)gr_cnt(cnt32, c * n + r) : e;
      }
    }
    static_assert(GR_T == 6, "the calls below enumerate the six tile columns");
    GR_STAMP(4);
    GrSolve z;
    z.bvec = bvec; z.lmul = lmul; z.prow = prow; z.sol = sol; z.pos = pos; z.rowat = rowat; z.misc = misc; z.n = n; z.S = S;
    singular = gr_gj_steps<0>(m, fin, z, tx, ty, lane) || gr_gj_steps<1>(m, fin, z, tx, ty, lane) ||
               gr_gj_steps<2>(m, fin, z, tx, ty, lane) || gr_gj_steps<3>(m, fin, z, tx, ty, lane) ||
               gr_gj_steps<4>(m, fin, z, tx, ty, lane) || gr_gj_steps<5>(m, fin, z, tx, ty, lane);
    // the system is diagonal now: c_k = b[pivot row of step k] / pivot k (kept in sol by the searchers)
    GR_STAMP(5);
    if (!singular && tid < n) sol[tid] = bvec[rowat[tid]] / sol[tid];
    __syncthreads();
  } else if (want_katz) {
    // ---- Katz, matrix in the global workspace (rows addressed through perm), the right-hand side carried along
    double* M = Mg;
    for (int k = 0; k < n && !singular; ++k) {
      double best = -1.0;
      int bi = 0x7fffffff;
      if (tid >= k && tid < n) { best = fabs(M[perm[tid] * n + k]); bi = tid; }
#pragma unroll
      for (int o = 32; o; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0 && wv < 4) { red[16 + wv] = best; misc[16 + wv] = bi; }
      __syncthreads();
      best = red[16]; bi = misc[16];
#pragma unroll
      for (int w = 1; w < 4; ++w)
        if (red[16 + w] > best || (red[16 + w] == best && misc[16 + w] < bi)) { best = red[16 + w]; bi = misc[16 + w]; }
      if (!(best > 0.0)) { singular = true; break; }      // uniform: every thread read the same four candidates
      const int pk = perm[bi], pold = perm[k];
      __syncthreads();
      if (tid == 0) { perm[k] = pk; perm[bi] = pold; }
      __syncthreads();
      const double rinv = 1.0 / M[pk * n + k];
      const double bk = bvec[pk];
      if (tid > k && tid < n) {
        const int pr = perm[tid];
        const double l = M[pr * n + k] * rinv;
        lmul[tid] = l;
        bvec[pr] -= l * bk;
        prow[tid] = M[pk * n + tid];
      }
      __syncthreads();
      for (int r = k + 1 + (tid >> 5); r < n; r += 32) {
        const int pr = perm[r];
        const double l = lmul[r];
        const int c0 = k + 1 + (tid & 31);
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int c = c0 + 32 * q; v[q] = c < n ? M[pr * n + c] : 0.0; }
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int c = c0 + 32 * q; if (c < n) M[pr * n + c] = v[q] - l * prow[c]; }
      }
      __syncthreads();
    }
    if (!singular) {
      for (int k = n - 1; k >= 0; --k) {
        const int pk = perm[k];
        const double xk = bvec[pk] / M[pk * n + k];
        __syncthreads();
        if (tid < k) bvec[perm[tid]] -= M[perm[tid] * n + k] * xk;
        if (tid == k) sol[k] = xk;
        __syncthreads();
      }
    }
  }
  if (want_katz) {
    if (singular) {
      status |= 2;
    } else {
      // c / (sign(sum c) * ||c||_2)   (katz_centrality_numpy's normalisation)
      const double c = tid < n ? sol[tid] : 0.0;
      const double s = gr_block_sum(c, red);
      const double ss = gr_block_sum(c * c, red);
      const double sgn = s > 0.0 ? 1.0 : (s < 0.0 ? -1.0 : 0.0);
      const double norm = sgn * sqrt(ss);
      if (tid < n) feat[4 * sym[tid] + 0] = c / norm;
    }
  }
  if (tid == 0) { info[0] = n; info[1] = status; info[2] = iters; info[3] = colptr[n]; }
  GR_STAMP(6);
}

// offset of the solver state inside the shared LDS region: behind the count matrix when both fit (the register
// solver reads the counts after the traversal), else 0; < 0: the launch does not fit in LDS at all
static inline int graph_state_off(int V, int ecap) {
  if (V <= GR_LU_NMAX && graph_lds_layout(V, ecap, graph_cnt_bytes(V)).total <= GR_LDS_LIMIT) return graph_cnt_bytes(V);
  return graph_lds_layout(V, ecap, 0).total <= GR_LDS_LIMIT ? 0 : -1;
}
