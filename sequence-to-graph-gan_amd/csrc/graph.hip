// Node features of the transition graph of a token sequence (reference: GraphEncoder.encode,
// graph_encoder.py:57-133).  The graph has one node per distinct symbol of the sequence and a directed
// edge u -> v weighted by the number of times v follows u; the reference computes four centralities per
// node with networkx (graph_encoder.py:86-91):
//   betweenness  Brandes' algorithm on the unweighted digraph, scaled by 1 / ((n-1)(n-2)) for n > 2
//   closeness    (r-1)/sum of incoming distances * (r-1)/(n-1), r = nodes that reach the node
//   pagerank     x <- 0.85 (x A + dangling mass / n) + 0.15 / n on the row-normalised weights, until the
//                L1 change is below n * 1e-6 (at most 100 iterations)
//   katz         solve (I - 0.1 W^T) c = 1 (dense LU, partial pivoting), c / (sign(sum c) ||c||_2); n > 2 only
// ONE workgroup of 256 threads per sequence, everything in float64 like the reference.  The graph lives in
// LDS: transition counts as packed 16-bit pairs while the structures are derived (in/out neighbour
// bitmasks, the in-edge lists of the pagerank product), then the same bytes hold the traversal state.
// Brandes and the closeness distances share one breadth-first search per source; the four waves take
// sources s = wave, wave + 4, ... and keep their sums in registers (a lane owns nodes lane, lane + 64, ...).
// The Katz matrix is n x n float64 (245 KB at n = 175): it is factorised in LDS when it fits beside the
// rest and in a caller-provided global workspace (L2-resident) otherwise, through the same code.
// Node numbering: ascending symbol id (every quantity above is independent of the numbering except for
// the order of floating-point sums, which differs from networkx's in the last bits).
#pragma once
#include "s2g_common.h"

#define GR_THREADS 256
#define GR_NMAX 256               // nodes per graph
#define GR_LDS_LIMIT 163840       // 160 KB per workgroup on gfx950

struct GraphArgs {
  const int* tokens;      // [n_seq][L] symbol ids in [0, V) (clamped into the range)
  const int* lengths;     // [n_seq] or nullptr (every sequence has L symbols)
  int L, V;
  double* feat;           // [n_seq][V][4]: katz, betweenness, closeness, pagerank at the symbol's id; 0 if absent
  int* info;              // [n_seq][4]: nodes, status (bit 0: pagerank did not converge, bit 1: Katz matrix
                          //             singular), pagerank iterations, distinct edges
  double* work;           // [n_seq][V*V] Katz matrices that do not fit in LDS
  int ecap;               // capacity of the in-edge lists (distinct edges <= L - 1)
  int uni_bytes;          // size of the LDS region shared by the count matrix and the later phases
};

// byte offsets of the LDS regions (shared by the host, which sizes the allocation, and the kernel)
struct GraphLds {
  int idx, sym, misc, red, inmask, outmask, invs, colptr, eval, esrc, uni, total;
};
__host__ __device__ inline GraphLds graph_lds_layout(int V, int ecap, int uni_bytes) {
  GraphLds o;
  int p = 0;
  o.idx = p; p += 4 * GR_NMAX;
  o.sym = p; p += 4 * GR_NMAX;
  o.misc = p; p += 4 * 16;
  o.red = p; p += 8 * 16;
  o.inmask = p; p += 32 * V;
  o.outmask = p; p += 32 * V;
  o.invs = p; p += 8 * V;
  o.eval = p; p += 8 * ecap;
  o.colptr = p; p += 4 * (V + 2);
  o.esrc = p; p += 2 * ecap;
  p = (p + 15) & ~15;
  o.uni = p; p += uni_bytes;
  o.total = p;
  return o;
}
// bytes of the phase-two state at the head of the shared region (the Katz matrix may follow it)
__host__ __device__ inline int graph_state_bytes(int V) {
  // sigma[4][V], coeff[4][V], bet[4][V] doubles; reach[4][V], tot[4][V] ints; x[V], bvec[V], lmul[V], prow[V],
  // sol[V] doubles; perm[V] ints
  return ((8 * (12 * V + 5 * V) + 4 * (8 * V + V)) + 15) & ~15;
}

__device__ __forceinline__ double gr_block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ unsigned gr_cnt(const unsigned* cnt32, int e) { return (cnt32[e >> 1] >> (16 * (e & 1))) & 0xffffu; }

extern "C" __global__ __launch_bounds__(GR_THREADS) void s2g_k_graph_features(GraphArgs a) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int V = a.V;
  const GraphLds lo = graph_lds_layout(V, a.ecap, a.uni_bytes);
  int* idx = reinterpret_cast<int*>(smem + lo.idx);
  int* sym = reinterpret_cast<int*>(smem + lo.sym);
  int* misc = reinterpret_cast<int*>(smem + lo.misc);
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

  // ---- nodes: the distinct symbols, numbered in ascending order
  idx[tid] = 0;
  __syncthreads();
  for (int i = tid; i < len; i += GR_THREADS) idx[min(max(tk[i], 0), V - 1)] = 1;
  __syncthreads();
  const int present = idx[tid];
  const unsigned long long bal = __ballot(present);
  const int within = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) misc[wv] = __popcll(bal);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wv; ++w) base += misc[w];
  const int n = misc[0] + misc[1] + misc[2] + misc[3];
  idx[tid] = present ? base + within : -1;
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

  // ---- neighbour bitmasks (bit j of word q = node 64 q + j), row sums, in-edge lists, Katz matrix
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
    if (lane == 63) misc[4 + wv] = inc;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wv; ++w) off += misc[4 + w];
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
  const bool want_katz = n > 2;
  const int state_bytes = graph_state_bytes(V);
  const bool mat_in_lds = (long)state_bytes + 8L * nn <= (long)a.uni_bytes;
  double* Mg = a.work + (size_t)g * V * V;
  if (want_katz) {
    // M = I - 0.1 W^T, row-major with row stride n: M[r][c] = [r == c] - 0.1 * count(c -> r)
    for (int r = wv; r < n; r += 4)
      for (int c = lane; c < n; c += 64) Mg[r * n + c] = (r == c ? 1.0 : 0.0) - 0.1 * (double)gr_cnt(cnt32, c * n + r);
  }
  __syncthreads();      // the count matrix is dead from here on: its bytes become the state below

  double* sigma = reinterpret_cast<double*>(uni);      // [4][V]
  double* coeff = sigma + 4 * V;                        // [4][V]
  double* betp = coeff + 4 * V;                         // [4][V]
  double* xv = betp + 4 * V;                            // [V]
  double* bvec = xv + V;
  double* lmul = bvec + V;
  double* prow = lmul + V;
  double* sol = prow + V;
  int* reachp = reinterpret_cast<int*>(sol + V);        // [4][V]
  int* totp = reachp + 4 * V;                           // [4][V]
  int* perm = totp + 4 * V;                             // [V]
  double* Ml = reinterpret_cast<double*>(uni + state_bytes);
  double* M = mat_in_lds ? Ml : Mg;
  if (want_katz && mat_in_lds) {
    for (int i = tid; i < nn; i += GR_THREADS) Ml[i] = Mg[i];
  }

  // ---- Brandes betweenness + closeness: one breadth-first search per source, a wave per source
  {
    volatile double* sg = sigma + wv * V;
    volatile double* cf = coeff + wv * V;
    double bet[4] = {0.0, 0.0, 0.0, 0.0};
    int reach[4] = {0, 0, 0, 0}, tot[4] = {0, 0, 0, 0};
    for (int s = wv; s < n; s += 4) {
      int dist[4];
      unsigned long long F[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int v = lane + 64 * k;
        dist[k] = (v == s) ? 0 : -1;
        F[k] = (s >> 6) == k ? 1ull << (s & 63) : 0ull;
      }
      if (lane == (s & 63)) sg[s] = 1.0;
      int d = 0;
      while (F[0] | F[1] | F[2] | F[3]) {      // at most n levels: every level marks at least one new node
        unsigned long long nf[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int v = lane + 64 * k;
          bool hit = false;
          double acc = 0.0;
          if (v < n && dist[k] < 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              unsigned long long m = inmask[4 * v + q] & F[q];
              while (m) {
                acc += sg[64 * q + __builtin_ctzll(m)];
                m &= m - 1ull;
                hit = true;
              }
            }
          }
          if (hit) { dist[k] = d + 1; sg[v] = acc; }
          nf[k] = __ballot(hit);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) F[k] = nf[k];
        ++d;
      }
      const int dmax = d - 1;      // deepest level reached (the loop ran once more to find it empty)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (dist[k] >= 0) { reach[k] += 1; tot[k] += dist[k]; }
      for (int lev = dmax; lev >= 0; --lev) {
        unsigned long long nxt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) nxt[k] = __ballot(dist[k] == lev + 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int v = lane + 64 * k;
          if (v < n && dist[k] == lev) {
            const double sv = sg[v];
            double dl = 0.0;
#pragma unroll
            for (int q = 3; q >= 0; --q) {
              unsigned long long m = outmask[4 * v + q] & nxt[q];
              while (m) {
                const int b = 63 - __builtin_clzll(m);
                dl += sv * cf[64 * q + b];
                m &= ~(1ull << b);
              }
            }
            cf[v] = (1.0 + dl) / sv;
            if (v != s) bet[k] += dl;
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = lane + 64 * k;
      if (v < n) { betp[wv * V + v] = bet[k]; reachp[wv * V + v] = reach[k]; totp[wv * V + v] = tot[k]; }
    }
  }
  __syncthreads();
  if (tid < n) {
    double b = ((betp[tid] + betp[V + tid]) + betp[2 * V + tid]) + betp[3 * V + tid];
    if (n > 2) b *= 1.0 / (double)((n - 1) * (n - 2));
    const int r = reachp[tid] + reachp[V + tid] + reachp[2 * V + tid] + reachp[3 * V + tid];
    const int t = totp[tid] + totp[V + tid] + totp[2 * V + tid] + totp[3 * V + tid];
    double c = 0.0;
    if (t > 0) {
      c = ((double)r - 1.0) / (double)t;
      c *= ((double)r - 1.0) / (double)(n - 1);
    }
    feat[4 * sym[tid] + 1] = b;
    feat[4 * sym[tid] + 2] = c;
  }

  // ---- pagerank: power iteration on the row-normalised weights; thread j owns node j
  int status = 0, iters = 0;
  {
    const double p = 1.0 / (double)n;
    const bool mine = tid < n;
    const bool dang = mine && invS[tid] == 0.0;
    const int e0 = mine ? colptr[tid] : 0, e1 = mine ? colptr[tid + 1] : 0;
    if (mine) xv[tid] = p;
    __syncthreads();
    bool done = false;
    for (int it = 1; it <= 100 && !done; ++it) {
      const double xo = mine ? xv[tid] : 0.0;
      const double dsum = gr_block_sum(dang ? xo : 0.0, red);
      double t1 = 0.0;
      for (int e = e0; e < e1; ++e) t1 += xv[esrc[e]] * eval[e];
      const double xn = mine ? 0.85 * (t1 + dsum * p) + (1.0 - 0.85) * p : 0.0;
      const double err = gr_block_sum(fabs(xn - xo), red);      // its barriers order the reads above before the write
      if (mine) xv[tid] = xn;
      __syncthreads();
      iters = it;
      done = err < (double)n * 1.0e-6;
    }
    if (!done) status |= 1;
    if (mine) feat[4 * sym[tid] + 3] = xv[tid];
  }

  // ---- Katz: LU with partial pivoting on M (rows addressed through perm), the right-hand side carried along
  if (want_katz) {
    if (tid < n) { perm[tid] = tid; bvec[tid] = 1.0; }
    __syncthreads();
    bool singular = false;
    for (int k = 0; k < n && !singular; ++k) {
      // pivot: largest |M[r][k]|, r >= k, first such row (LAPACK's choice)
      double best = -1.0;
      int bi = 0x7fffffff;
      if (tid >= k && tid < n) { best = fabs(M[perm[tid] * n + k]); bi = tid; }
#pragma unroll
      for (int o = 32; o; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) { red[8 + wv] = best; misc[8 + wv] = bi; }
      __syncthreads();
      best = red[8]; bi = misc[8];
#pragma unroll
      for (int w = 1; w < 4; ++w)
        if (red[8 + w] > best || (red[8 + w] == best && misc[8 + w] < bi)) { best = red[8 + w]; bi = misc[8 + w]; }
      if (best == 0.0) { singular = true; break; }      // uniform: every thread read the same four candidates
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
      for (int r = k + 1 + (tid >> 4); r < n; r += 16) {
        const int pr = perm[r];
        const double l = lmul[r];
        for (int c = k + 1 + (tid & 15); c < n; c += 16) M[pr * n + c] -= l * prow[c];
      }
      __syncthreads();
    }
    if (singular) {
      status |= 2;
    } else {
      // back substitution, column by column: c_k = b_k / U_kk, then b_r -= U_rk c_k for r < k
      for (int k = n - 1; k >= 0; --k) {
        const int pk = perm[k];
        const double xk = bvec[pk] / M[pk * n + k];
        __syncthreads();
        if (tid < k) bvec[perm[tid]] -= M[perm[tid] * n + k] * xk;
        if (tid == k) sol[k] = xk;
        __syncthreads();
      }
      const double c = tid < n ? sol[tid] : 0.0;
      const double s = gr_block_sum(c, red);
      const double ss = gr_block_sum(c * c, red);
      const double sgn = s > 0.0 ? 1.0 : (s < 0.0 ? -1.0 : 0.0);
      const double norm = sgn * sqrt(ss);
      if (tid < n) feat[4 * sym[tid] + 0] = c / norm;
    }
  }
  if (tid == 0) { info[0] = n; info[1] = status; info[2] = iters; info[3] = colptr[n]; }
}

// dynamic LDS bytes of a launch, the shared region as large as the limit allows (more of the Katz matrices fit)
static inline int graph_uni_bytes(int V, int ecap) {
  const int fixed = graph_lds_layout(V, ecap, 0).total;
  int need = 2 * V * V + 16;                              // count matrix
  if (graph_state_bytes(V) > need) need = graph_state_bytes(V);
  int room = GR_LDS_LIMIT - fixed;
  room &= ~15;
  return room >= need ? room : -1;
}
