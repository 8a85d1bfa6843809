// Exact-fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32), LDS-staged 64x64x32 tiles.
// Serves every dense product of the step: the critic's input projection [B*T,175]x[175,200]
// and its transposes (reference: the W_ih part of nn.LSTM, wgan_gp.py:189,217), the batched
// weight-gradient products over saved gate gradients, and the generator's Linear layers
// (wgan_gp.py:154-160).
//
//   C[m][n] = act( alpha * sum_k A(m,k) * B(k,n) + bias[n] )          (+ C if accumulate)
//   A(m,k) = A[m*sam + k*sak]      B(k,n) = B[k*sbk + n*sbn]
// Arbitrary strides cover NN / NT / TN without separate kernels; the global->LDS copy picks a
// thread mapping whose fastest index is the one contiguous in memory.  gridDim.z > 1 splits K:
// slab z covers k in [z*kchunk, (z+1)*kchunk) and is written to C + z*slab (reduced afterwards
// by s2g_k_reduce_slabs), which keeps the sum order fixed (no float atomics).
// segT > 0: the reduction index k is a flattened (sequence b, step t) pair, k = b*segT + t, and
// the two operands may keep a different number of rows per sequence: A reads row b*a_seg + t,
// B reads row b*b_seg + t (in units of sak / sbk).  With `shift` set, B reads step t-1 and is
// zero at t == 0 (the "previous hidden state" operand of dW_hh).
#include "s2g_common.h"

#define BM 64
#define BN 64
#define BK 32
#define LDS_LD 68   // k-major tiles [BK][64 + 4]: 16-byte rows, float4 stores stay aligned

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Operand pointers reach the kernels inside a struct (for the grouped launch, a struct read from
// memory), where the compiler cannot prove they are global: it would emit flat loads, which also
// count in lgkmcnt, so the first wait for an LDS read in the MFMA loop would also wait for every
// prefetch in flight.  Loads go through explicit global-address-space pointers instead.
typedef float nf4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) float* gfp;
typedef const __attribute__((address_space(1))) nf4* gfp4;
__device__ __forceinline__ float gload(const float* p) { return *(gfp)p; }
__device__ __forceinline__ float4 gload4(const float* p) {
  const nf4 v = *(gfp4)p;
  return make_float4(v.x, v.y, v.z, v.w);
}

struct GemmArgs {
  const float* A; long sam, sak;
  const float* B; long sbk, sbn;
  float* C; long ldc; long slab;
  const float* bias;
  int M, N, K, kchunk, segT, a_seg, b_seg, shift, accumulate, act;
  float alpha;
  const float* bscale;      // grouped kernels, segT > 0: B's row k times bscale[k / segT] (per-sequence factor) or nullptr
};

// (sequence, step) position of a reduction index under the segT mapping, advanced incrementally so
// the inner loop carries no integer division.
struct SegPos {
  int b, t;
  __device__ __forceinline__ void init(int k, int segT) {
    if (segT > 0) { b = k / segT; t = k - b * segT; } else { b = 0; t = k; }
  }
  __device__ __forceinline__ void advance(int dk, int segT) {
    t += dk;
    if (segT > 0) {
      const bool wrap = t >= segT;             // the common single wrap stays a select, not a branch
      t -= wrap ? segT : 0;
      b += wrap ? 1 : 0;
      if (t >= segT) { const int w = t / segT; b += w; t -= w * segT; }   // segments shorter than the stride
    }
  }
};

// Loader of one operand tile [BK][64] into k-major LDS.
//   MODE 0: generic strides, scalar loads; lanes run along the non-k index (m or n).
//   MODE 1: non-k index contiguous in memory (stride 1) and 16-byte aligned rows: float4 along it.
//   MODE 2: k contiguous in memory (stride 1), rows 16-byte aligned, no segT mapping: float4 along k.
//   MODE 3: k contiguous in memory but rows unaligned (e.g. 175-wide rows): scalar loads with the
//           lanes running along k, so a wave still reads 128-byte segments.
template <int MODE>
struct TileLoader {
  float r[8];
  unsigned live;   // bit e: r[e] (MODE 0/3) or float4 e (MODE 1/2) is inside the matrix
  SegPos pos;      // position of this thread's first k of the current K-block
  int x, kq;       // x: non-k index inside the tile; kq: first k inside the K-block

  __device__ __forceinline__ void setup(int tid, int kbeg, int segT) {
    if (MODE == 1) { x = (tid & 15) * 4; kq = tid >> 4; }        // k = kq, kq + 16
    else if (MODE == 2) { x = tid & 63; kq = (tid >> 6) * 8; }   // k = kq .. kq + 7
    else if (MODE == 3) { x = tid >> 5; kq = tid & 31; }         // x = x + 8 e, k = kq
    else { x = tid & 63; kq = tid >> 6; }                        // k = kq + 4 e
    pos.init(kbeg + kq, segT);
  }
  // P: base pointer; sx / sk: strides of the non-k / k index; seg: rows per sequence of this operand;
  // x0: tile origin; X: bound of the non-k index; shift: read step t-1, zero at t == 0.
  // Every load is unconditional: an out-of-range element reads P[0] instead (offset selected, not
  // branched) and is zeroed when the tile is written to LDS.  A load under a branch would force a
  // wait for it at the join and expose the memory latency once per element.
  __device__ __forceinline__ void load(const float* P, long sx, long sk, int seg, int segT, bool shift, int x0,
                                       int X, int k0, int kend) {
    const int xi = x0 + x;
    live = 0;
    if (MODE == 1) {
      SegPos q = pos;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = k0 + kq + 16 * e;
        long kr = segT > 0 ? (long)q.b * seg + q.t : (long)k;
        bool in = k < kend;
        if (shift) { in = in && q.t != 0; kr -= 1; }
        const bool full = in && (xi + 3 < X);
        const float* src = P + kr * sk + xi;
        float4 v = gload4(full ? src : P);
        if (in && !full && xi < X) {       // ragged edge: element-wise (tiles on the matrix border only)
          v.x = gload(src);
          v.y = (xi + 1 < X) ? gload(src + 1) : 0.0f;
          v.z = (xi + 2 < X) ? gload(src + 2) : 0.0f;
          v.w = 0.0f;
          live |= 1u << e;
        }
        live |= full ? (1u << e) : 0u;
        r[4 * e] = v.x; r[4 * e + 1] = v.y; r[4 * e + 2] = v.z; r[4 * e + 3] = v.w;
        q.advance(16, segT);
      }
    } else if (MODE == 2) {
      const int k = k0 + kq;
      const float* src = P + (long)xi * sx + k;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool full = (xi < X) && (k + 4 * e + 3 < kend);
        float4 v = gload4(full ? src + 4 * e : P);
        if (!full && xi < X && k + 4 * e < kend) {
          v.x = gload(src + 4 * e);
          v.y = (k + 4 * e + 1 < kend) ? gload(src + 4 * e + 1) : 0.0f;
          v.z = (k + 4 * e + 2 < kend) ? gload(src + 4 * e + 2) : 0.0f;
          v.w = 0.0f;
          live |= 1u << e;
        }
        live |= full ? (1u << e) : 0u;
        r[4 * e] = v.x; r[4 * e + 1] = v.y; r[4 * e + 2] = v.z; r[4 * e + 3] = v.w;
      }
    } else if (MODE == 3) {
      const int k = k0 + kq;
      long kr = segT > 0 ? (long)pos.b * seg + pos.t : (long)k;
      bool okk = k < kend;
      if (shift) { okk = okk && pos.t != 0; kr -= 1; }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int xe = xi + 8 * e;
        const bool ok = okk && xe < X;
        r[e] = gload(P + (ok ? kr * sk + (long)xe * sx : 0));
        live |= ok ? (1u << e) : 0u;
      }
    } else {
      SegPos q = pos;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + kq + 4 * e;
        long kr = segT > 0 ? (long)q.b * seg + q.t : (long)k;
        bool ok = (xi < X) && (k < kend);
        if (shift) { ok = ok && q.t != 0; kr -= 1; }
        r[e] = gload(P + (ok ? kr * sk + (long)xi * sx : 0));
        live |= ok ? (1u << e) : 0u;
        q.advance(4, segT);
      }
    }
    pos.advance(BK, segT);
  }
  __device__ __forceinline__ float val(int e) const { return (live >> e) & 1u ? r[e] : 0.0f; }
  __device__ __forceinline__ float4 val4(int e) const {
    return (live >> e) & 1u ? make_float4(r[4 * e], r[4 * e + 1], r[4 * e + 2], r[4 * e + 3])
                            : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ void store(float* S) const {
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < 2; ++e) *reinterpret_cast<float4*>(S + (kq + 16 * e) * LDS_LD + x) = val4(e);
    } else if (MODE == 2) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float4 v = val4(e);
        S[(kq + 4 * e) * LDS_LD + x] = v.x; S[(kq + 4 * e + 1) * LDS_LD + x] = v.y;
        S[(kq + 4 * e + 2) * LDS_LD + x] = v.z; S[(kq + 4 * e + 3) * LDS_LD + x] = v.w;
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int e = 0; e < 8; ++e) S[kq * LDS_LD + x + 8 * e] = val(e);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) S[(kq + 4 * e) * LDS_LD + x] = val(e);
    }
  }
};

template <int AMODE, int BMODE>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int bx, const int by, const int bz, float* As,
                                          float* Bs) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);

  TileLoader<AMODE> la;
  TileLoader<BMODE> lb;
  la.setup(tid, kbeg, g.segT);
  lb.setup(tid, kbeg, g.segT);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

  const int aoff = (lane >> 5) * LDS_LD + wr * 32 + (lane & 31);
  const int boff = (lane >> 5) * LDS_LD + wc * 32 + (lane & 31);

  if (kbeg < kend) {
    la.load(g.A, g.sam, g.sak, g.a_seg, g.segT, false, m0, g.M, kbeg, kend);
    lb.load(g.B, g.sbn, g.sbk, g.b_seg, g.segT, g.shift != 0, n0, g.N, kbeg, kend);
    la.store(As); lb.store(Bs);
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = (k0 + BK) < kend;
      // the wave issues in order: the next K-block's loads (and their address arithmetic) go between
      // the MFMAs, not in front of them with the matrix pipe idle
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        if (kk == 2 && more) la.load(g.A, g.sam, g.sak, g.a_seg, g.segT, false, m0, g.M, k0 + BK, kend);
        if (kk == 10 && more) lb.load(g.B, g.sbn, g.sbk, g.b_seg, g.segT, g.shift != 0, n0, g.N, k0 + BK, kend);
        float a = As[aoff + kk * LDS_LD];
        float b = Bs[boff + kk * LDS_LD];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      __syncthreads();
      if (more) {
        la.store(As); lb.store(Bs);
        __syncthreads();
      }
    }
  }

  float* C = g.C + (long)bz * g.slab;
  const int col = n0 + wc * 32 + (lane & 31);
  if (col < g.N) {
    const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < g.M) {
        float v = g.alpha * acc[r] + bv;
        long off = (long)row * g.ldc + col;
        if (g.accumulate) v += C[off];
        if (g.act == 1) v = v > 0.0f ? v : 0.2f * v;
        else if (g.act == 2) v = fmaxf(v, 0.0f);
        C[off] = v;
      }
    }
  }
}

template <int AMODE, int BMODE>
__global__ __launch_bounds__(256) void s2g_k_gemm_f32(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[BK * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDS_LD];
  gemm_tile<AMODE, BMODE>(g, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// Weight-gradient tile (TN shape, K = all saved rows): C[0:M, n0:n0+64] over one K-slab, M <= 208.
// A = [rows, M] row-major with sam == 1 and M % 4 == 0, B generic strides.
// Loader: wave w fetches rows k0 + w + 4e (e = 0..7) of both operands, one row per load
// instruction with the lanes along m (float4) / n, so the row index -- and with it the whole
// (sequence, step) bookkeeping of the segT mapping and the row address -- is wave-uniform and
// lives in scalar registers; the vector unit only adds a constant lane offset.
// MFMA (v_mfma_f32_16x16x4_f32): wave w owns columns [16w, 16w+16) x all 13 row tiles, so the 200 rows
// are padded to 208 (not 256) and the four waves carry the same load.
#define WG_MT 13     // 16-row tiles: M <= 208
#define WLD_A 272   // k-major A tile [BK][256 + 16]: = 16 mod 64 floats, conflict-free 16x4 fragment reads
#define WLD_B 80    // k-major B tile [BK][64 + 16]
// SCALED: B's row k is multiplied by bscale[sequence of k] (a per-sequence factor, segT > 0).  The factor travels
// with the row as one more VECTOR load from a wave-uniform address: a scalar load would be counted in lgkmcnt, and
// the wait for it would also wait for every LDS fragment read in flight.  A template parameter, so that the
// products without factors keep their exact instruction stream.
template <bool SCALED>
__device__ __forceinline__ void wgrad_tile(const GemmArgs& g, const int bx, const int bz, float* As, float* Bs) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int n0 = bx * BN;
  const int kbeg = bz * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int segT = g.segT;

  // lane-constant parts
  const bool a_in = 4 * lane + 3 < g.M;
  const unsigned a_lane = a_in ? 4 * lane : 0;
  const int ncol = n0 + lane;
  const bool b_in = ncol < g.N;
  const long b_lane = b_in ? (long)ncol * g.sbn : 0;

  // Scalar row stepper over rows kbeg + wave, +4, +8, ...: (sequence rb, step rt) of the segT
  // mapping and the two row pointers.  Inside a sequence a step is two pointer increments; the
  // pointers are recomputed only when a row crosses into the next sequence.
  int rk = kbeg + wave, rb = 0, rt = rk;
  if (segT > 0) { rb = rk / segT; rt = rk - rb * segT; }
  const long shiftB = g.shift ? 1 : 0;
  const float *pa, *pb;
  const int nseq_k = segT > 0 ? (g.K + segT - 1) / segT : 1;
  auto rowptrs = [&]() {
    const long rowA = segT > 0 ? (long)rb * g.a_seg + rt : (long)rk;
    const long rowB = (segT > 0 ? (long)rb * g.b_seg + rt : (long)rk) - shiftB;
    pa = g.A + rowA * g.sak;
    pb = g.B + rowB * g.sbk;      // with shift at rt == 0 this is never dereferenced
  };
  rowptrs();
  const long stepA = 4 * g.sak, stepB = 4 * g.sbk;

  float4 ra[8];
  float rbv[8], rsc[8];
  unsigned okA = 0, okB = 0;        // scalar masks: row e exists for A / B
  auto fetch = [&](int e) {         // row rk -> slot e, then step to row rk + 4
    const bool in = rk < kend;
    const bool inB = in && !(g.shift && rt == 0);
    ra[e] = gload4((in ? pa : g.A) + a_lane);
    rbv[e] = gload((inB ? pb : g.B) + b_lane);
    if (SCALED) rsc[e] = gload(g.bscale + min(rb, nseq_k - 1));
    if (e == 0) { okA = 0; okB = 0; }
    okA |= in ? (1u << e) : 0u;
    okB |= inB ? (1u << e) : 0u;
    rk += 4; rt += 4;
    if (segT > 0 && rt >= segT) {
      do { rt -= segT; ++rb; } while (rt >= segT);
      rowptrs();
    } else {
      pa += stepA; pb += stepB;
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kr = wave + 4 * e;
      const bool la = a_in && ((okA >> e) & 1u), lb = b_in && ((okB >> e) & 1u);
      float4 v;
      v.x = la ? ra[e].x : 0.0f; v.y = la ? ra[e].y : 0.0f; v.z = la ? ra[e].z : 0.0f; v.w = la ? ra[e].w : 0.0f;
      *reinterpret_cast<float4*>(As + kr * WLD_A + 4 * lane) = v;
      Bs[kr * WLD_B + lane] = lb ? (SCALED ? rbv[e] * rsc[e] : rbv[e]) : 0.0f;
    }
  };

  // wave w owns column tile w (16 columns) x all 13 row tiles (208 rows): per k-quad 13 A fragments and one
  // B fragment feed 13 MFMAs; fragments are double-buffered in registers (reads of quad q+1 before the
  // MFMAs of quad q)
  f32x4 acc[WG_MT];
#pragma unroll
  for (int i = 0; i < WG_MT; ++i) acc[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  const int aoff = (lane >> 4) * WLD_A + (lane & 15);
  const int boff = (lane >> 4) * WLD_B + 16 * wave + (lane & 15);
  const bool wave_live = n0 + 16 * wave < g.N;      // a column tile past N: nothing to multiply or store

  if (kbeg < kend) {
#pragma unroll
    for (int e = 0; e < 8; ++e) fetch(e);
    store();
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      // The wave issues in order, so the next K-block's 8 row fetches (and their scalar
      // bookkeeping) are spread between the MFMAs instead of running before them with the matrix
      // pipe idle.  Past the last K-block they read element 0 and are discarded.
      float fa[2][WG_MT], fb[2];
      auto frag = [&](int q, float (&a)[WG_MT], float& bq) {
#pragma unroll
        for (int i = 0; i < WG_MT; ++i) a[i] = As[aoff + 4 * q * WLD_A + 16 * i];
        bq = Bs[boff + 4 * q * WLD_B];
      };
      auto mma = [&](const float (&a)[WG_MT], float bq) {
#pragma unroll
        for (int i = 0; i < WG_MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bq, acc[i], 0, 0, 0);
      };
      if (wave_live) frag(0, fa[0], fb[0]);
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        fetch(q); fetch(q + 1);
        if (wave_live) {
          frag(q + 1, fa[1], fb[1]);
          __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead of the MFMAs they do not feed
          mma(fa[0], fb[0]);
          __builtin_amdgcn_sched_barrier(0);
          if (q + 2 < 8) frag(q + 2, fa[0], fb[0]);
          __builtin_amdgcn_sched_barrier(0);
          mma(fa[1], fb[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
      if (k0 + BK < kend) {
        store();
        __syncthreads();
      }
    }
  }

  if (!wave_live) return;
  float* C = g.C + (long)bz * g.slab;
  const int col = n0 + 16 * wave + (lane & 15);
  if (col < g.N) {
#pragma unroll
    for (int i = 0; i < WG_MT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + 4 * (lane >> 4) + r;
        if (row < g.M) C[(long)row * g.ldc + col] = g.alpha * acc[i][r];
      }
    }
  }
}

// Grouped launch: `table` holds nprob independent problems, all of the TN weight-gradient shape:
// A = [rows, M] row-major with sam == 1, sak % 4 == 0 and a 16-byte aligned base (loaded with
// float4 along m), M % 4 == 0 and M <= 208, B with arbitrary strides, no bias / activation /
// accumulate; problem p owns blocks [prefix[p], prefix[p+1]) of (all M) x 64 output tiles, ordered
// (z, x) with x fastest.  One launch fills the chip with work that would otherwise be a dozen
// small back-to-back launches.
extern "C" __global__ __launch_bounds__(256) void s2g_k_gemm_grouped(const GemmArgs* __restrict__ table,
                                                                     const int* __restrict__ prefix, int nprob,
                                                                     int nz) {
  __shared__ __attribute__((aligned(16))) float As[BK * WLD_A];
  __shared__ __attribute__((aligned(16))) float Bs[BK * WLD_B];
  // Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  Remap so that
  // XCD x works through one contiguous range of logical blocks: the output tiles of one K-slab
  // (which share that slab's rows of A and B) then run back to back on one L2.
  const int total = gridDim.x, q = total >> 3, r = total & 7;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  // every thread walks the (uniform) prefix table, so the problem descriptor arrives by scalar
  // loads and its fields stay in scalar registers
  int p = 0;
  if (nz > 0) {
    // Interleaved order (every problem has nz K-chunks; prefix counts OUTPUT TILES): all tiles of all
    // problems for K-chunk z are neighbours, so products that share an operand (dW_ih0 and dW_hh0 read
    // the same gate-gradient rows, dW_ih1 and dW_hh1 likewise) meet those rows in L2 instead of
    // fetching them from HBM once per product.
    const int tiles = prefix[nprob];
    const int bz = lid / tiles, t = lid - bz * tiles;
    while (p + 1 < nprob && t >= prefix[p + 1]) ++p;
    const GemmArgs gs = table[p];
    if (gs.bscale != nullptr && gs.segT > 0) wgrad_tile<true>(gs, t - prefix[p], bz, As, Bs);
    else wgrad_tile<false>(gs, t - prefix[p], bz, As, Bs);
    return;
  }
  while (p + 1 < nprob && lid >= prefix[p + 1]) ++p;
  const GemmArgs gs = table[p];
  const int local = lid - prefix[p];
  const int nx = (gs.N + BN - 1) / BN;
  const int bx = local % nx, bz = local / nx;
  if (gs.bscale != nullptr && gs.segT > 0) wgrad_tile<true>(gs, bx, bz, As, Bs);
  else wgrad_tile<false>(gs, bx, bz, As, Bs);
}

// out[perm(r)][c] (+)= scale * sum_z slabs[z][r][c];   perm = internal gate row -> torch row when
// permute_rows, identity otherwise.  A block of 4 waves owns 64 consecutive output elements; wave w
// sums slabs w, w+4, w+8, ... (four loads in flight per thread), the partial sums meet in LDS.  The
// summation order is fixed by (nslab), so results are reproducible run to run.
__device__ __forceinline__ void reduce_slabs_block(const float* __restrict__ slabs, long slab, int nslab, int rows,
                                                   int cols, float* __restrict__ out, int permute_rows,
                                                   int accumulate, float scale, int block) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long i = (long)block * 64 + lane;
  const bool ok = i < (long)rows * cols;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (ok) {
    int z = w;
    for (; z + 12 < nslab; z += 16) {
      s0 += slabs[(long)z * slab + i]; s1 += slabs[(long)(z + 4) * slab + i];
      s2 += slabs[(long)(z + 8) * slab + i]; s3 += slabs[(long)(z + 12) * slab + i];
    }
    for (; z < nslab; z += 4) s0 += slabs[(long)z * slab + i];
  }
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w != 0 || !ok) return;
  float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  const int r = (int)(i / cols), c = (int)(i % cols);
  int ro = permute_rows ? pt_row(r) : r;
  long o = (long)ro * cols + c;
  s *= scale;
  out[o] = accumulate ? out[o] + s : s;
}

extern "C" __global__ __launch_bounds__(256) void s2g_k_reduce_slabs(const float* __restrict__ slabs, long slab,
                                                                     int nslab, int rows, int cols,
                                                                     float* __restrict__ out, int permute_rows,
                                                                     int accumulate, float scale) {
  reduce_slabs_block(slabs, slab, nslab, rows, cols, out, permute_rows, accumulate, scale, blockIdx.x);
}

// out[i] = sum_z w[z] * slabs[z][i] (ascending z, i < n): a weighted slab sum.  With one slab per sequence (K-chunk =
// a sequence's rows) the factor of a per-sequence weighted sum of outer products is applied here, after the
// product, so the product itself can run before the weights are known.
extern "C" __global__ __launch_bounds__(256) void s2g_k_reduce_slabs_weighted(const float* __restrict__ slabs, long slab,
                                                                              int nslab, const float* __restrict__ w,
                                                                              float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int z = 0;
  for (; z + 3 < nslab; z += 4) {       // four slabs' loads in flight; one fixed order of additions
    s0 = fmaf(w[z], slabs[(long)z * slab + i], s0);
    s1 = fmaf(w[z + 1], slabs[(long)(z + 1) * slab + i], s1);
    s2 = fmaf(w[z + 2], slabs[(long)(z + 2) * slab + i], s2);
    s3 = fmaf(w[z + 3], slabs[(long)(z + 3) * slab + i], s3);
  }
  for (; z < nslab; ++z) s0 = fmaf(w[z], slabs[(long)z * slab + i], s0);
  out[i] = (s0 + s1) + (s2 + s3);
}

// Several independent reductions in one launch (the four weight gradients of a critic iteration):
// job j owns blocks [first_block[j], first_block[j+1]).
#define S2G_MAX_REDUCE_JOBS 8
struct ReduceJob {
  const float* slabs; long slab; float* out;
  int nslab, rows, cols, first_block;
};
struct ReduceJobs {
  ReduceJob j[S2G_MAX_REDUCE_JOBS];
  int njobs;
};
extern "C" __global__ __launch_bounds__(256) void s2g_k_reduce_slabs_multi(ReduceJobs a, int permute_rows,
                                                                           int accumulate, float scale) {
  int k = 0;
  while (k + 1 < a.njobs && (int)blockIdx.x >= a.j[k + 1].first_block) ++k;
  const ReduceJob& q = a.j[k];
  reduce_slabs_block(q.slabs, q.slab, q.nslab, q.rows, q.cols, q.out, permute_rows, accumulate, scale,
                     blockIdx.x - q.first_block);
}

// The critic's whole parameter update in one launch (one process: no all-reduce between the gradient
// and Adam): the final slab reduction of the four LSTM weight gradients (same blocks and order of
// additions as s2g_k_reduce_slabs_multi, gradient written in torch layout as before), Adam on the
// element just reduced (the expression of s2g_k_adam), and the packed copies the sweeps read
// (s2g_k_pack_critic: WT0 / W0p for W_ih0, b0 / b1 = b_ih + b_hh).  Blocks from tail_block on update the
// parameters whose gradients other kernels left in g: one thread per gate row does b_ih and b_hh of a
// layer and their sum, the rest take the fc head.
struct CriticUpdate {
  ReduceJob j[4];            // W_ih0, W_hh0, W_ih1, W_hh1 (out = gradient view, first_block as in ReduceJobs)
  long poff[4];              // offset of the job's parameter in the flat buffers
  float *p, *g, *m, *v;      // flat parameter / gradient / Adam moment buffers
  long o_bih0, o_bhh0, o_bih1, o_bhh1, o_fcw, o_fcb;
  int n_fcw, n_fcb, tail_block, V, Vp;
  float *WT0, *W0p, *b0, *b1;
  float b1c, b2c, eps, step_size, inv_sqrt_bc2;
};
__device__ __forceinline__ float adam_one(const CriticUpdate& a, long o, float gg) {
  const float mm = a.b1c * a.m[o] + (1.0f - a.b1c) * gg;
  const float vv = a.b2c * a.v[o] + (1.0f - a.b2c) * gg * gg;
  const float pn = a.p[o] - a.step_size * (mm / (sqrtf(vv) * a.inv_sqrt_bc2 + a.eps));
  a.m[o] = mm; a.v[o] = vv; a.p[o] = pn;
  return pn;
}
extern "C" __global__ __launch_bounds__(256) void s2g_k_critic_update(CriticUpdate a) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if ((int)blockIdx.x >= a.tail_block) {
    const int t = ((int)blockIdx.x - a.tail_block) * 256 + threadIdx.x;
    if (t < 2 * S2G_G) {
      const int layer = t / S2G_G, jr = t % S2G_G, o = pt_row(jr);
      const long oi = (layer ? a.o_bih1 : a.o_bih0) + o, oh = (layer ? a.o_bhh1 : a.o_bhh0) + o;
      const float bi = adam_one(a, oi, a.g[oi]);
      const float bh = adam_one(a, oh, a.g[oh]);
      (layer ? a.b1 : a.b0)[jr] = bi + bh;
    } else if (t < 2 * S2G_G + a.n_fcw) {
      const long o = a.o_fcw + (t - 2 * S2G_G);
      adam_one(a, o, a.g[o]);
    } else if (t < 2 * S2G_G + a.n_fcw + a.n_fcb) {
      const long o = a.o_fcb + (t - 2 * S2G_G - a.n_fcw);
      adam_one(a, o, a.g[o]);
    }
    return;
  }
  int k = 0;
  while (k + 1 < 4 && (int)blockIdx.x >= a.j[k + 1].first_block) ++k;
  const ReduceJob& q = a.j[k];
  const long i = (long)((int)blockIdx.x - q.first_block) * 64 + lane;
  const bool ok = i < (long)q.rows * q.cols;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (ok) {
    int z = w;
    for (; z + 12 < q.nslab; z += 16) {
      s0 += q.slabs[(long)z * q.slab + i]; s1 += q.slabs[(long)(z + 4) * q.slab + i];
      s2 += q.slabs[(long)(z + 8) * q.slab + i]; s3 += q.slabs[(long)(z + 12) * q.slab + i];
    }
    for (; z < q.nslab; z += 4) s0 += q.slabs[(long)z * q.slab + i];
  }
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w != 0 || !ok) return;
  const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  const int r = (int)(i / q.cols), c = (int)(i % q.cols);
  const long o = (long)pt_row(r) * q.cols + c;
  q.out[o] = s;
  const float pn = adam_one(a, a.poff[k] + o, s);
  if (k == 0) {
    a.W0p[(long)r * a.Vp + c] = pn;
    a.WT0[(long)c * S2G_G + r] = pn;
  }
}

// column sums: out[perm(c)] (+)= scale * sum_r X[r][c], X row-major [rows, cols] with row stride ld.
// grid = (ceil(cols/64), nchunk); partial sums land in `part[chunk][cols]`, reduced by
// s2g_k_reduce_slabs with rows = 1.
extern "C" __global__ __launch_bounds__(256) void s2g_k_colsum(const float* __restrict__ X, long ld,
                                                               long rows, int cols,
                                                               float* __restrict__ part, long rows_per_chunk) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  const long r1 = min(rows, r0 + rows_per_chunk);
  float s = 0.0f;
  if (c < cols)
    for (long r = r0 + w; r < r1; r += 4) s += X[r * ld + c];
  red[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < cols)
    part[(long)blockIdx.y * cols + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                                        red[3][threadIdx.x];
}

// Bias gradients of both LSTM layers in two launches: the bias of a gate row is the column sum of the
// gate gradients over all sequence-steps (X0 = layer 0, X1 = layer 1, [rows, 200] each).
// Pass 1, grid (4, nchunk, 2): partial sums over BIAS_CHUNK rows, four loads in flight per thread.
#define BIAS_CHUNK 512
extern "C" __global__ __launch_bounds__(256) void s2g_k_bias_sums_part(const float* __restrict__ X0,
                                                                       const float* __restrict__ X1, long rows,
                                                                       float* __restrict__ part) {
  __shared__ float red[4][64];
  const float* X = blockIdx.z ? X1 : X0;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  const long r0 = (long)blockIdx.y * BIAS_CHUNK;
  const long r1 = min(rows, r0 + BIAS_CHUNK);
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (c < S2G_G) {
    long r = r0 + w;
    for (; r + 12 < r1; r += 16) {
      s0 += X[r * S2G_G + c]; s1 += X[(r + 4) * S2G_G + c];
      s2 += X[(r + 8) * S2G_G + c]; s3 += X[(r + 12) * S2G_G + c];
    }
    for (; r < r1; r += 4) s0 += X[r * S2G_G + c];
  }
  red[w][l] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && c < S2G_G)
    part[((long)blockIdx.z * gridDim.y + blockIdx.y) * S2G_G + c] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
}
// Pass 2, grid 8 x 256 threads: block b < 7 owns 64 of the 2 x 200 sums (4 waves split the nchunk partial
// rows of a layer, which start layer_stride floats apart; fixed order) and writes bias_ih and bias_hh of that layer (the two share one gradient) in torch row order;
// block 7 adds sum_b hdotT[b][k] to fcw0[k] (the penalty's part of fc.weight[0]) when hdotT is given.
extern "C" __global__ __launch_bounds__(256) void s2g_k_bias_sums_final(
    const float* __restrict__ part, int nchunk, long layer_stride, const float* __restrict__ hdotT, int B,
    float* __restrict__ bih0,
    float* __restrict__ bhh0, float* __restrict__ bih1, float* __restrict__ bhh1, float* __restrict__ fcw0,
    int accumulate) {
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (blockIdx.x == 7) {
    if (hdotT == nullptr) return;
    float s = 0.0f;
    if (l < S2G_H)
      for (int b = w; b < B; b += 4) s += hdotT[(long)b * S2G_H + l];
    red[w][l] = s;
    __syncthreads();
    if (w == 0 && l < S2G_H) fcw0[l] += (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    return;
  }
  const int i = blockIdx.x * 64 + l;            // 0 .. 399: layer = i / 200, internal gate row = i % 200
  const bool ok = i < 2 * S2G_G;
  const int layer = ok ? i / S2G_G : 0, j = ok ? i % S2G_G : 0;
  float s = 0.0f;
  if (ok) {
    // eight partial rows' loads in flight, added in the same ascending order (one load per iteration paid a cache
    // round trip each: nchunk / 4 of them in a row, in a kernel that sits between two sweeps)
    const float* pp = part + layer * layer_stride + j;
    for (int zb = w; zb < nchunk; zb += 32) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = pp[(long)min(zb + 4 * k, nchunk - 1) * S2G_G];
#pragma unroll
      for (int k = 0; k < 8; ++k) s = zb + 4 * k < nchunk ? s + v[k] : s;
    }
  }
  red[w][l] = s;
  __syncthreads();
  if (w != 0 || !ok) return;
  s = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
  float* bi = layer ? bih1 : bih0;
  float* bh = layer ? bhh1 : bhh0;
  const int o = pt_row(j);
  const float v = accumulate ? bi[o] + s : s;
  bi[o] = v;
  bh[o] = v;
}

// Tall product C[M,N] = A[M,K] B[K,N] with M = all sequence-steps and K, N <= 224 (the input
// projection x W_ih0^T, its transpose delta W_ih0 and delta (W_ih0 W_ih0^T)): a block owns 128 rows and
// ALL N columns, so A is read exactly once and B (<= 160 KB, L2-resident) once per block.  Eight waves
// per block (two per SIMD, so one wave's LDS reads, chunk hand-over and stores run under the other's
// MFMAs): wave w owns rows [16w, 16w+16) as NT16 16x16 accumulators (N is padded to 16).  Per k-quad
// one A fragment feeds NT16 MFMAs (v_mfma_f32_16x16x4_f32: lane l supplies A[l%16][l/16] and
// B[l/16][l%16], holds D[4(l/16)+r][l%16]).  K is streamed in chunks of 32 through two LDS stages (one
// barrier per chunk); the next chunk's global loads are issued between the MFMAs of the current one.
// LDS rows: A stride 34 floats (banks 2*row + k: no conflicts among the 32 lanes of a half wave), B
// stride = 16 mod 32 floats (two consecutive k rows land on disjoint halves of the banks).
// A: row-major, k contiguous (any lda, rows may be unaligned: lanes run along k).  B: row-major with
// ldb % 4 == 0, 16-byte aligned, readable up to column roundup(N, 4) (pad columns must hold zeros).
// Register-only MFMA loops reach 156 TFLOP/s on this part (tools/mfma_peak.hip); this kernel's MFMA
// phase is 15-17 us of its 33-36 us at M = 32768, the rest being launch, first-chunk latency, the chunk
// hand-overs and the C stores, none of which another block hides (one block per CU, one round).
#define TALL_ROWS 128
#define TALL16_ALD 34
template <int NT16>
__global__ __launch_bounds__(512) void s2g_k_gemm_tall16(const float* __restrict__ A, long lda,
                                                         const float* __restrict__ B, long ldb,
                                                         float* __restrict__ C, long ldc, int M, int N, int K) {
  constexpr int BLD = NT16 * 16 + ((NT16 & 1) ? 0 : 16);   // B stage row length (floats)
  constexpr int NC4 = NT16 * 4;                             // float4 per B stage row
  constexpr int NB = (32 * NC4 + 511) / 512;                // B float4 loads per thread per chunk
  constexpr int STAGE = TALL_ROWS * TALL16_ALD + 32 * BLD;
  extern __shared__ __attribute__((aligned(16))) float tall_lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int m0 = blockIdx.x * TALL_ROWS;
  const int ar = tid >> 5, ak = tid & 31;     // A loader: rows ar + 16e (e < 8), column k0 + ak

  float ra[8];
  float4 rb[NB];
  int kf = 0;                                 // first k of the chunk being fetched
  // Out-of-range elements read element 0 and are zeroed only when the chunk is written to LDS (bit
  // masks): nothing touches a loaded value before then, so no wait for the loads lands inside the
  // MFMA loop.
  unsigned liveA = 0, liveB = 0;
  auto fetch_a = [&](int e) {
    const int row = m0 + ar + 16 * e, k = kf + ak;
    const bool in = row < M && k < K;
    ra[e] = gload(A + (in ? (long)row * lda + k : 0));
    if (e == 0) liveA = 0;
    liveA |= in ? (1u << e) : 0u;
  };
  auto fetch_b = [&](int e) {
    const int f = tid + 512 * e;
    const int kr = f / NC4, c4 = f - kr * NC4;
    const bool in = f < 32 * NC4 && kf + kr < K && 4 * c4 < N;
    rb[e] = gload4(B + (in ? (long)(kf + kr) * ldb + 4 * c4 : 0));
    if (e == 0) liveB = 0;
    liveB |= in ? (1u << e) : 0u;
  };
  auto store = [&](float* S) {
#pragma unroll
    for (int e = 0; e < 8; ++e) S[(ar + 16 * e) * TALL16_ALD + ak] = (liveA >> e) & 1u ? ra[e] : 0.0f;
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      const int f = tid + 512 * e;
      const int kr = f / NC4, c4 = f - kr * NC4;
      const bool l = (liveB >> e) & 1u;
      float4 v;
      v.x = l ? rb[e].x : 0.0f; v.y = l ? rb[e].y : 0.0f; v.z = l ? rb[e].z : 0.0f; v.w = l ? rb[e].w : 0.0f;
      if (f < 32 * NC4) *reinterpret_cast<float4*>(S + TALL_ROWS * TALL16_ALD + kr * BLD + 4 * c4) = v;
    }
  };

  f32x4 acc[NT16];
#pragma unroll
  for (int j = 0; j < NT16; ++j) acc[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  const int aoff = (16 * wave + (lane & 15)) * TALL16_ALD + (lane >> 4);
  const int boff = TALL_ROWS * TALL16_ALD + (lane >> 4) * BLD + (lane & 15);

#pragma unroll
  for (int e = 0; e < 8; ++e) fetch_a(e);
#pragma unroll
  for (int e = 0; e < NB; ++e) fetch_b(e);
  store(tall_lds);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < K; k0 += 32) {
    const float* S = tall_lds + cur * STAGE;
    kf = k0 + 32;                             // past the last chunk every fetch is masked
    // Fragments are double-buffered in registers: the LDS reads of quad q+1 are issued before the MFMAs
    // of quad q.  Dead k (>= K) holds zeros in both operands; a chunk whose second half is dead skips it.
    float a0, a1, b0[NT16], b1[NT16];
    auto frag = [&](int q, float& a, float (&b)[NT16]) {
      a = S[aoff + 4 * q];
#pragma unroll
      for (int j = 0; j < NT16; ++j) b[j] = S[boff + 4 * q * BLD + 16 * j];
    };
    auto mma = [&](float a, const float (&b)[NT16]) {
#pragma unroll
      for (int j = 0; j < NT16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], acc[j], 0, 0, 0);
    };
    frag(0, a0, b0);
#pragma unroll
    for (int q = 0; q < 4; q += 2) {          // all fetches in the first half: they land before the store
#pragma unroll
      for (int e = 0; e < 4; ++e) fetch_a(2 * q + e);
      if (q < NB) fetch_b(q);
      if (q + 1 < NB) fetch_b(q + 1);
      frag(q + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead of the MFMAs they do not feed
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      frag(q + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (K - k0 > 16) {
#pragma unroll
      for (int q = 4; q < 8; q += 2) {
        frag(q + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 2 < 8) frag(q + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    store(tall_lds + (cur ^ 1) * STAGE);      // the other stage: nobody reads it during this chunk
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int j = 0; j < NT16; ++j) {
    const int col = 16 * j + (lane & 15);
    if (col < N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * wave + 4 * (lane >> 4) + r;
        if (row < M) C[(long)row * ldc + col] = acc[j][r];
      }
    }
  }
}

extern "C" int s2g_launch_gemm_tall(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M,
                                    int N, int K, hipStream_t stream) {
  const int nt16 = (N + 15) / 16;
  const dim3 grid((M + TALL_ROWS - 1) / TALL_ROWS), block(512);
#define S2G_TALL16_CASE(NT_)                                                                                   \
  if (nt16 <= NT_) {                                                                                           \
    const size_t lds = 2 * (TALL_ROWS * TALL16_ALD + 32 * (NT_ * 16 + ((NT_ & 1) ? 0 : 16))) * sizeof(float);  \
    static S2GLdsCap lds_cap;                                                                                      \
    if (s2g_lds_cap(lds_cap, reinterpret_cast<const void*>(&s2g_k_gemm_tall16<NT_>), lds) != 0) return -2;         \
    hipLaunchKernelGGL((s2g_k_gemm_tall16<NT_>), grid, block, lds, stream, A, lda, B, ldb, C, ldc, M, N, K);   \
    return hipGetLastError() == hipSuccess ? 0 : -2;                                                           \
  }
  S2G_TALL16_CASE(11) S2G_TALL16_CASE(13) S2G_TALL16_CASE(14)
#undef S2G_TALL16_CASE
  return -1;
}

// y[M, N] = x[M, K] W[N, K]^T + bias for a small batch M <= 64 and a large N (the generator's 512 -> L*V
// output layer: 183 MB of weights against 23 MB of outputs): bound by streaming W once from HBM, so W never
// passes through LDS.  x^T sits in LDS (Xs[k][m], the MFMA B operand, once per workgroup); a wave owns 16
// output features at a time and loads their W rows straight into the A-operand layout of
// v_mfma_f32_16x16x4_f32: lane (i = l % 16, kk = l / 16) fetches the 32 bytes W[n0 + i][32 g + 8 kk .. + 7]
// (the four lanes of a row cover one full 128-byte line), and MFMA call c of group g multiplies component
// c of every lane's eight floats, i.e. k = 32 g + 8 kk + c, with
// Xs rows of the same k (the reduction index may be visited in any order).  Four groups (8 KB per wave,
// 64 KB per CU) are in flight ahead of the MFMAs.  Workgroups of 8 waves, one per CU (x^T takes 135 KB of
// LDS), stride over the N / 16 feature tiles.  Measured 70-77 us at N = 89600 (generic 64x64 tiles: 80 us;
// with 1/16 of the MFMAs 45 us, without the in-loop loads 71 us: the MFMA + LDS-read loop is what is left).
#define WS_LD 66          // Xs row: 8 * 66 = 16 mod 64 floats between the k rows one fragment read touches
#define WS_WAVES 8
#define WS_PF 4           // ring depth in groups of 32 k (two float4 per lane each)
extern "C" __global__ __launch_bounds__(64 * WS_WAVES) void s2g_k_linear_wstream(
    const float* __restrict__ W, const float* __restrict__ X, const float* __restrict__ bias, float* __restrict__ Y,
    int N, int M, int K) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // x^T into LDS: four independent 16-byte loads in flight per thread (one at a time, this loop alone cost
  // tens of microseconds of exposed latency)
  const int k4n = K >> 2, total4 = 64 * k4n;
  for (int base = tid; base < total4; base += 4 * 64 * WS_WAVES) {
    float4 v[4];
    int mm[4], kq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 64 * WS_WAVES;
      mm[u] = idx / k4n;
      kq[u] = 4 * (idx - mm[u] * k4n);
      v[u] = gload4(X + ((idx < total4 && mm[u] < M) ? (long)mm[u] * K + kq[u] : 0));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 64 * WS_WAVES;
      if (idx < total4) {
        const bool live = mm[u] < M;
        float* d = Xs + kq[u] * WS_LD + mm[u];
        d[0] = live ? v[u].x : 0.0f; d[WS_LD] = live ? v[u].y : 0.0f;
        d[2 * WS_LD] = live ? v[u].z : 0.0f; d[3 * WS_LD] = live ? v[u].w : 0.0f;
      }
    }
  }
  __syncthreads();
  const int ntile = N >> 4, ngrp = K >> 5;
  const int i = lane & 15, kk = lane >> 4;
  const int boff = 8 * kk * WS_LD + i;
  for (int tile = blockIdx.x * WS_WAVES + wave; tile < ntile; tile += gridDim.x * WS_WAVES) {
    const int n0 = tile << 4;
    const float* wp = W + (long)(n0 + i) * K + 8 * kk;
    float4 ring[WS_PF][2];
#pragma unroll
    for (int e = 0; e < WS_PF; ++e) {
      ring[e][0] = gload4(wp + 32 * e);
      ring[e][1] = gload4(wp + 32 * e + 4);
    }
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    // Straight-line body (K % 128 == 0, all four batch tiles: rows m >= M of Xs are zero): a branch around
    // a load or an LDS read makes the compiler drain the counters at the join.  The scheduling barriers
    // keep each refill of the ring and the LDS reads of the NEXT half group in front of this one's MFMAs
    // (left alone the compiler sinks all refills behind the MFMAs and then waits for them at once).
    float bq[2][16];
    auto bfrag = [&](int g, int h, float (&b)[16]) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) b[4 * c + j] = Xs[boff + (32 * g + 4 * h + c) * WS_LD + 16 * j];
    };
    bfrag(0, 0, bq[0]);
    for (int g0 = 0; g0 < ngrp; g0 += WS_PF) {
#pragma unroll
      for (int e = 0; e < WS_PF; ++e) {
        const int g = g0 + e;
        const float4 w0 = ring[e][0], w1 = ring[e][1];
        const int gn = g + WS_PF < ngrp ? g + WS_PF : 0;         // unconditional, clamped refill
        ring[e][0] = gload4(wp + 32 * gn);
        ring[e][1] = gload4(wp + 32 * gn + 4);
        bfrag(g, 1, bq[1]);
        __builtin_amdgcn_sched_barrier(0);
        const float wa[4] = {w0.x, w0.y, w0.z, w0.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c], bq[0][4 * c + j], acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        bfrag(g + 1 < ngrp ? g + 1 : 0, 0, bq[0]);
        __builtin_amdgcn_sched_barrier(0);
        const float wb[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[c], bq[1][4 * c + j], acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // D[4 kk + r][i]: feature n0 + 4 kk + r, batch row 16 j + i
    const int n = n0 + 4 * kk;
    const float4 bv = bias ? gload4(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = 16 * j + i;
      if (m < M)
        *reinterpret_cast<float4*>(Y + (long)m * N + n) =
            make_float4(acc[j][0] + bv.x, acc[j][1] + bv.y, acc[j][2] + bv.z, acc[j][3] + bv.w);
    }
  }
}

// The same weight stream fused with sampling and decoding (generate_fake_data.py:60-66: generator forward,
// Gumbel-softmax, argmax(-1)): tokens[m][t] = argmax_v (logits[m][t][v] + gumbel(u[m][t][v])) without ever
// writing the [M, L, V] logits or samples.  Softmax and the division by tau are monotone, so the argmax of
// the sample (soft or straight-through one-hot) is the argmax of the perturbed logits; the first index wins
// a tie, as torch.argmax does.  A workgroup owns TS consecutive time steps: their ceil(V/16) * TS feature
// tiles are dealt to its 8 waves; a wave reduces its 16 features per batch row in registers (4 in-lane
// values, then the 4 lane groups), leaves (max, index) per batch row and tile in LDS, and after a barrier
// one wave per time step combines the tiles.  The accumulation order is the plain kernel's, so the logits
// are bit-identical to s2g_k_linear_wstream's.  Uniforms: U [M, L, V] if given, else Philox with the
// in-kernel sampler's numbering (counter (m * L + t) * 64 + v / 4, word v % 4).
#include "rng.hip"
#define GT_MAXTILES 48
__device__ __forceinline__ float gumbel_of(float u) { return gumbel_noise(u); }
extern "C" __global__ __launch_bounds__(64 * WS_WAVES) void s2g_k_generate_tokens(
    const float* __restrict__ W, const float* __restrict__ X, const float* __restrict__ bias,
    const float* __restrict__ U, long long* __restrict__ tokens, int L, int V, int M, int K, int TS,
    unsigned long long seed, unsigned long long offset) {
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  float* pmax = Xs + (size_t)K * WS_LD;                    // [tiles of the group][64]
  int* pidx = reinterpret_cast<int*>(pmax + GT_MAXTILES * 64);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k4n = K >> 2, total4 = 64 * k4n;
  for (int idx = tid; idx < total4; idx += 64 * WS_WAVES) {
    const int mm = idx / k4n, kq = 4 * (idx - mm * k4n);
    const bool live = mm < M;
    const float4 v = gload4(X + (live ? (long)mm * K + kq : 0));
    float* d = Xs + kq * WS_LD + mm;
    d[0] = live ? v.x : 0.0f; d[WS_LD] = live ? v.y : 0.0f; d[2 * WS_LD] = live ? v.z : 0.0f; d[3 * WS_LD] = live ? v.w : 0.0f;
  }
  __syncthreads();
  const int ngrp = K >> 5, tps = (V + 15) >> 4;            // feature tiles per time step
  const int i = lane & 15, kk = lane >> 4;
  const int boff = 8 * kk * WS_LD + i;
  const long N = (long)L * V;
  for (int t0 = blockIdx.x * TS; t0 < L; t0 += gridDim.x * TS) {
    const int nts = min(TS, L - t0);
    for (int q = wave; q < tps * nts; q += WS_WAVES) {
      const int ts = q / tps, tile = q - ts * tps, t = t0 + ts;
      const long n0 = (long)t * V + 16 * tile;
      const float* wp = W + min(n0 + i, N - 1) * K + 8 * kk;        // rows past the last feature: clamped, masked below
      float4 ring[WS_PF][2];
#pragma unroll
      for (int e = 0; e < WS_PF; ++e) {
        ring[e][0] = gload4(wp + 32 * e);
        ring[e][1] = gload4(wp + 32 * e + 4);
      }
      f32x4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
      float bq[2][16];
      auto bfrag = [&](int g, int h, float (&b)[16]) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j) b[4 * c + j] = Xs[boff + (32 * g + 4 * h + c) * WS_LD + 16 * j];
      };
      bfrag(0, 0, bq[0]);
      for (int g0 = 0; g0 < ngrp; g0 += WS_PF) {
#pragma unroll
        for (int e = 0; e < WS_PF; ++e) {
          const int g = g0 + e;
          const float4 w0 = ring[e][0], w1 = ring[e][1];
          const int gn = g + WS_PF < ngrp ? g + WS_PF : 0;
          ring[e][0] = gload4(wp + 32 * gn);
          ring[e][1] = gload4(wp + 32 * gn + 4);
          bfrag(g, 1, bq[1]);
          __builtin_amdgcn_sched_barrier(0);
          const float wa[4] = {w0.x, w0.y, w0.z, w0.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c], bq[0][4 * c + j], acc[j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          bfrag(g + 1 < ngrp ? g + 1 : 0, 0, bq[0]);
          __builtin_amdgcn_sched_barrier(0);
          const float wb[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[c], bq[1][4 * c + j], acc[j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // D[4 kk + r][i]: column v = 16 tile + 4 kk + r of time step t, batch row 16 j + i
      const int v0 = 16 * tile + 4 * kk;
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = (bias && v0 + r < V) ? bias[(long)t * V + v0 + r] : 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = 16 * j + i;
        const long row = (long)min(m, M - 1) * L + t;
        float u[4];
        if (U) {
#pragma unroll
          for (int r = 0; r < 4; ++r) u[r] = (v0 + r < V) ? U[row * V + v0 + r] : 0.5f;
        } else {
          const Philox4 ph = philox4x32_10((unsigned long long)row * 64 + (v0 >> 2), offset, seed);
          u[0] = u01(ph.x); u[1] = u01(ph.y); u[2] = u01(ph.z); u[3] = u01(ph.w);
        }
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float z = (acc[j][r] + bv[r]) + gumbel_of(u[r]);
          if (v0 + r < V && z > best) { best = z; bi = v0 + r; }
        }
        // the four lane groups kk hold the other 12 columns of this tile for the same batch row
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          const float ob = __shfl_xor(best, o);
          const int oi = __shfl_xor(bi, o);
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (kk == 0) { pmax[q * 64 + m] = best; pidx[q * 64 + m] = bi; }
      }
    }
    __syncthreads();
    if (wave < nts && lane < M) {
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int tile = 0; tile < tps; ++tile) {            // ascending columns: a later tile wins only if larger
        const float ob = pmax[(wave * tps + tile) * 64 + lane];
        const int oi = pidx[(wave * tps + tile) * 64 + lane];
        if (ob > best) { best = ob; bi = oi; }
      }
      tokens[(long)lane * L + t0 + wave] = bi;
    }
    __syncthreads();
  }
}

extern "C" int s2g_launch_generate_tokens(const float* W, const float* X, const float* bias, const float* U,
                                          long long* tokens, int L, int V, int M, int K, unsigned long long seed,
                                          unsigned long long offset, hipStream_t stream) {
  static S2GLdsCap lds_cap;
  const int cus = s2g_device_cus();
  if (cus <= 0 || s2g_lds_cap(lds_cap, reinterpret_cast<const void*>(&s2g_k_generate_tokens), 160 * 1024) != 0) return -2;
  const int tps = (V + 15) >> 4;
  // time steps per workgroup: as many as fill the 8 waves evenly and still give every CU a group
  int TS = 1;
  while (TS < 8 && tps * (TS + 1) <= GT_MAXTILES && (L + TS) / (TS + 1) >= cus) ++TS;
  if (TS == 1 && tps * 2 <= GT_MAXTILES && L >= 2) TS = 2;      // 11 tiles on 8 waves: two steps share the rounds
  const size_t lds = ((size_t)K * WS_LD + 2 * GT_MAXTILES * 64) * sizeof(float);
  int blocks = (L + TS - 1) / TS;
  if (blocks > cus) blocks = cus;
  hipLaunchKernelGGL(s2g_k_generate_tokens, dim3(blocks), dim3(64 * WS_WAVES), lds, stream, W, X, bias, U, tokens, L,
                     V, M, K, TS, seed, offset);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// max_workgroups: CUs the stream can reach (a CU-masked stream: fewer than the device has; a workgroup takes
// a whole CU, so a grid larger than that would run in two rounds); 0 = all of the device's.
extern "C" int s2g_launch_linear_wstream(const float* W, const float* X, const float* bias, float* Y, int N, int M,
                                         int K, int max_workgroups, hipStream_t stream) {
  const size_t lds = (size_t)K * WS_LD * sizeof(float);
  static S2GLdsCap lds_cap;
  const int cus = s2g_device_cus();
  if (cus <= 0 || s2g_lds_cap(lds_cap, reinterpret_cast<const void*>(&s2g_k_linear_wstream), 160 * 1024) != 0) return -2;
  int blocks = ((N >> 4) + WS_WAVES - 1) / WS_WAVES;
  const int cap = (max_workgroups > 0 && max_workgroups < cus) ? max_workgroups : cus;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(s2g_k_linear_wstream, dim3(blocks), dim3(64 * WS_WAVES), lds, stream, W, X, bias, Y, N, M, K);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int s2g_launch_gemm(const GemmArgs* ga, int nsplit, hipStream_t stream) {
  GemmArgs g = *ga;
  if (g.M <= 0 || g.N <= 0 || g.K < 0 || nsplit < 1) return -1;
  if (g.kchunk <= 0) g.kchunk = ((g.K + nsplit - 1) / nsplit + BK - 1) / BK * BK;
  if (g.kchunk <= 0) g.kchunk = BK;
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, nsplit);
  auto al16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  // operand load modes (see TileLoader): 1 = float4 along the non-k index, 2 = float4 along k, 0 = scalar
  int am = 0, bm = 0;
  if (g.sam == 1 && (g.sak % 4) == 0 && al16(g.A)) am = 1;
  else if (g.sak == 1 && (g.sam % 4) == 0 && al16(g.A) && g.segT == 0 && (g.kchunk % 4) == 0) am = 2;
  else if (g.sak == 1) am = 3;
  if (g.sbn == 1 && (g.sbk % 4) == 0 && al16(g.B)) bm = 1;
  else if (g.sbk == 1 && (g.sbn % 4) == 0 && al16(g.B) && g.segT == 0 && (g.kchunk % 4) == 0) bm = 2;
  else if (g.sbk == 1) bm = 3;
#define S2G_GEMM_CASE(AM, BM_)                                                                     \
  if (am == AM && bm == BM_) hipLaunchKernelGGL((s2g_k_gemm_f32<AM, BM_>), grid, dim3(256), 0, stream, g);
  S2G_GEMM_CASE(0, 0) S2G_GEMM_CASE(0, 1) S2G_GEMM_CASE(0, 2)
  S2G_GEMM_CASE(1, 0) S2G_GEMM_CASE(1, 1) S2G_GEMM_CASE(1, 2)
  S2G_GEMM_CASE(2, 0) S2G_GEMM_CASE(2, 1) S2G_GEMM_CASE(2, 2)
  S2G_GEMM_CASE(3, 0) S2G_GEMM_CASE(3, 1) S2G_GEMM_CASE(3, 2) S2G_GEMM_CASE(3, 3)
  S2G_GEMM_CASE(0, 3) S2G_GEMM_CASE(1, 3) S2G_GEMM_CASE(2, 3)
#undef S2G_GEMM_CASE
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int s2g_launch_gemm_grouped(const GemmArgs* table_dev, const int* prefix_dev, int nprob, int total_blocks,
                                       int nz, hipStream_t stream) {
  hipLaunchKernelGGL(s2g_k_gemm_grouped, dim3(total_blocks), dim3(256), 0, stream, table_dev, prefix_dev, nprob, nz);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int s2g_launch_reduce_slabs_weighted(const float* slabs, long slab, int nslab, const float* w, float* out,
                                               long n, hipStream_t stream) {
  hipLaunchKernelGGL(s2g_k_reduce_slabs_weighted, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, slabs, slab,
                     nslab, w, out, n);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int s2g_launch_critic_update(const CriticUpdate* a, int blocks, hipStream_t stream) {
  hipLaunchKernelGGL(s2g_k_critic_update, dim3(blocks), dim3(256), 0, stream, *a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
