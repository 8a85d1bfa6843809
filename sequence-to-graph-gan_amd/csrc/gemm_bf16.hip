// bf16-operand form of the tall products of the critic's input layer (x W_ih0^T, delta (W_ih0 W_ih0^T),
// delta W_ih0; the W_ih part of nn.LSTM and its autograd, wgan_gp.py:189, 217, 247-254):
// C[M,N] = A[M,K] B[K,N] with both operands rounded to bf16 (round to nearest even) and fp32 accumulation
// on v_mfma_f32_16x16x32_bf16, 16x the matrix rate of the exact-fp32 16x16x4 form.  The optional
// reduced-precision path of the step (2e-2 relative against the fp32 reference); everything recurrent
// (cell state, gate pre-activations, reductions) stays fp32.
//
// M is all sequence-steps (32 768), K and N <= 224.  A block owns 128 rows and ALL columns: B (<= 224 x 224)
// is converted once per block into LDS as bf16, transposed to Bt[n][k] so that a lane's B fragment (8
// consecutive k of one column) is one 16-byte read; a wave owns 16 rows and loads its whole A strip
// (16 x K fp32, 8 consecutive k per lane = the A fragment layout, two 16-byte loads per 32-wide k step)
// straight from global memory into registers, all loads in flight at once: A never passes through LDS and
// there is no pipeline to drain.  HBM floor: M (K + N) 4 bytes, ~10 us at M = 32768.
#pragma once
#include "s2g_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4b __attribute__((ext_vector_type(4)));
typedef float f4ua __attribute__((ext_vector_type(4), aligned(4)));

#define TB_ROWS 128
#define TB_KMAX 224
#define TB_LD (TB_KMAX + 8)        // Bt row stride in bf16: 464 bytes = 116 dwords = 52 mod 64 banks

template <int NT>                  // 16-column tiles: N <= 16 NT
__global__ __launch_bounds__(512) void s2g_k_gemm_tall_bf16(const float* __restrict__ A, long lda,
                                                            const float* __restrict__ B, long ldb,
                                                            float* __restrict__ C, long ldc, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) __bf16 Bt[];      // [16 NT][TB_LD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Kp = (K + 31) & ~31;
  // ---- A strip of this wave: rows m0 + (lane & 15), k = 32 s + 8 (lane >> 4) + 0..7
  const int i = lane & 15, kg = lane >> 4;
  const long m0 = (long)blockIdx.x * TB_ROWS + 16 * wave;
  const long row = min(m0 + i, (long)M - 1);                       // rows past M: clamped, never stored
  const float* ap = A + row * lda + 8 * kg;
  f32x8 a[TB_KMAX / 32];
#pragma unroll
  for (int s = 0; s < TB_KMAX / 32; ++s) {
    const int k0 = 32 * s + 8 * kg;
    if (32 * s < Kp) {
      if (k0 + 8 <= K) {
        const f4ua lo = *reinterpret_cast<const f4ua*>(ap + 32 * s);
        const f4ua hi = *reinterpret_cast<const f4ua*>(ap + 32 * s + 4);
        a[s] = (f32x8){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[s][j] = (k0 + j < K) ? ap[32 * s + j] : 0.0f;      // the row's tail
      }
    }
  }
  // ---- B -> LDS, bf16, transposed; k rows K..Kp-1 and columns N..16 NT-1 are zero
  const int n4 = (16 * NT) >> 2;
  for (int idx = tid; idx < Kp * n4; idx += 512) {
    const int k = idx / n4, n = 4 * (idx - k * n4);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {
      if (n + 3 < N && !(ldb & 3)) {
        const float4 q = *reinterpret_cast<const float4*>(B + (long)k * ldb + n);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (n + j < N) ? B[(long)k * ldb + n + j] : 0.0f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) Bt[(n + j) * TB_LD + k] = (__bf16)v[j];
  }
  __syncthreads();
  f32x4b acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4b){0.f, 0.f, 0.f, 0.f};
  const __bf16* bp = Bt + i * TB_LD + 8 * kg;
#pragma unroll
  for (int s = 0; s < TB_KMAX / 32; ++s) {
    if (32 * s < Kp) {
      const bf16x8 af = __builtin_convertvector(a[s], bf16x8);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bp + 16 * t * TB_LD + 32 * s);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[t], 0, 0, 0);
      }
    }
  }
  // D[4 kg + r][i]: row m0 + 4 kg + r, column 16 t + i
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = 16 * t + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = m0 + 4 * kg + r;
      if (m < M && col < N) C[m * ldc + col] = acc[t][r];
    }
  }
}

extern "C" int s2g_launch_gemm_tall_bf16(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M,
                                         int N, int K, hipStream_t stream) {
  const int nt = (N + 15) / 16;
  const dim3 grid((M + TB_ROWS - 1) / TB_ROWS), block(512);
#define S2G_TALLB_CASE(NT_)                                                                              \
  if (nt <= NT_) {                                                                                       \
    const size_t lds = (size_t)16 * NT_ * TB_LD * sizeof(__bf16);                                        \
    static S2GLdsCap lds_cap;                                                                                \
    if (s2g_lds_cap(lds_cap, reinterpret_cast<const void*>(&s2g_k_gemm_tall_bf16<NT_>), lds) != 0) return -2; \
    hipLaunchKernelGGL((s2g_k_gemm_tall_bf16<NT_>), grid, block, lds, stream, A, lda, B, ldb, C, ldc, M, N, K); \
    return hipGetLastError() == hipSuccess ? 0 : -2;                                                     \
  }
  S2G_TALLB_CASE(11) S2G_TALLB_CASE(13) S2G_TALLB_CASE(14)
#undef S2G_TALLB_CASE
  return -1;
}

// ------------------------------------------------------------------------------------------------------------
// bf16-operand form of the grouped weight-gradient product (s2g_k_gemm_grouped, gemm.hip): the same problems,
// block order and split-K slabs, C[0:M, n0:n0+64] += sum_k A[k][m] B[k][n] over one K-chunk with M <= 208, but the
// operands are rounded to bf16 (fp32 accumulation) and multiplied on v_mfma_f32_16x16x32_bf16.  Both MFMA
// operands want 8 CONSECUTIVE k per lane while memory is k-major (a saved row per k), so the tiles are transposed on
// their way into LDS: wave w fetches rows k0 + 8w .. 8w + 7 of the chunk (row index wave-uniform: the scalar row
// stepper of the fp32 kernel), lane = m-quad (A, 16-byte loads) / column (B); the eight values a lane holds for one
// m (or n) are exactly one fragment piece = one 16-byte LDS write at At[m][8 slot].  slot = (w + (m >> 2)) & 3
// spreads the fragment READS of 16 consecutive m over all 64 banks (the writes of A take a 4-way conflict: all
// lanes of an instruction share m mod 4).  With 1/16 of the fp32 matrix time the kernel is bound by its loads.
#define GB_MT 13                     // 16-row tiles: M <= 208
template <bool SCALED>              // per-sequence factor on B's rows, as in wgrad_tile
__device__ __forceinline__ void wgrad_tile_bf16(const GemmArgs& g, const int bx, const int bz, __bf16* At, __bf16* Bt) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int n0 = bx * 64;
  const int kbeg = bz * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int segT = g.segT;

  const bool a_in = 4 * lane + 3 < g.M;                  // M % 4 == 0: a quad is inside or outside as a whole
  const unsigned a_lane = a_in ? 4 * lane : 0;
  const int ncol = n0 + lane;
  const bool b_in = ncol < g.N;
  const long b_lane = b_in ? (long)ncol * g.sbn : 0;

  // scalar row stepper over the rows this wave fetches: kbeg + 8 w + j (j = 0..7), then + 32 per chunk
  int rk = kbeg + 8 * wave, rb = 0, rt = rk;
  if (segT > 0) { rb = rk / segT; rt = rk - rb * segT; }
  const long shiftB = g.shift ? 1 : 0;
  const float *pa, *pb;
  const int nseq_k = segT > 0 ? (g.K + segT - 1) / segT : 1;
  auto rowptrs = [&]() {
    const long rowA = segT > 0 ? (long)rb * g.a_seg + rt : (long)rk;
    const long rowB = (segT > 0 ? (long)rb * g.b_seg + rt : (long)rk) - shiftB;
    pa = g.A + rowA * g.sak;
    pb = g.B + rowB * g.sbk;          // with shift at rt == 0 this is never dereferenced
  };
  rowptrs();
  auto advance = [&](int d) {
    rk += d; rt += d;
    if (segT > 0 && rt >= segT) {
      do { rt -= segT; ++rb; } while (rt >= segT);
      rowptrs();
    } else {
      pa += d * g.sak; pb += d * g.sbk;
    }
  };

  float4 ra[8];
  float rbv[8], rsc[8];
  unsigned okA = 0, okB = 0;          // scalar masks: row j exists for A / B
  auto fetch = [&](int j) {           // row rk -> slot j, then on to the next row of this wave
    const bool in = rk < kend;
    const bool inB = in && !(g.shift && rt == 0);
    ra[j] = gload4((in ? pa : g.A) + a_lane);
    rbv[j] = gload((inB ? pb : g.B) + b_lane);
    if (SCALED) rsc[j] = gload(g.bscale + min(rb, nseq_k - 1));
    if (j == 0) { okA = 0; okB = 0; }
    okA |= in ? (1u << j) : 0u;
    okB |= inB ? (1u << j) : 0u;
    advance(j == 7 ? 25 : 1);
  };
  const int aslot = (wave + lane) & 3;                   // m = 4 lane + c: (m >> 2) = lane
  const int bslot = (wave + (lane >> 2)) & 3;
  auto store = [&]() {
    f32x8 va[4], vb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool la = a_in && ((okA >> j) & 1u), lb = b_in && ((okB >> j) & 1u);
      va[0][j] = la ? ra[j].x : 0.0f; va[1][j] = la ? ra[j].y : 0.0f;
      va[2][j] = la ? ra[j].z : 0.0f; va[3][j] = la ? ra[j].w : 0.0f;
      vb[j] = lb ? (SCALED ? rbv[j] * rsc[j] : rbv[j]) : 0.0f;
    }
    if (lane < GB_MT * 4) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<bf16x8*>(At + (4 * lane + c) * 32 + 8 * aslot) = __builtin_convertvector(va[c], bf16x8);
    }
    *reinterpret_cast<bf16x8*>(Bt + lane * 32 + 8 * bslot) = __builtin_convertvector(vb, bf16x8);
  };

  f32x4b acc[GB_MT];
#pragma unroll
  for (int i = 0; i < GB_MT; ++i) acc[i] = (f32x4b){0.0f, 0.0f, 0.0f, 0.0f};
  const int fi = lane & 15, kg = lane >> 4;
  const bool wave_live = n0 + 16 * wave < g.N;           // a column tile past N: nothing to multiply or store
  const int bn = 16 * wave + fi;
  const __bf16* bfrag = Bt + bn * 32 + 8 * ((kg + (bn >> 2)) & 3);

  if (kbeg < kend) {
#pragma unroll
    for (int j = 0; j < 8; ++j) fetch(j);
    store();
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
      // the next chunk's row fetches are issued in front of this chunk's MFMAs (past the last chunk they read
      // element 0 and are discarded)
#pragma unroll
      for (int j = 0; j < 8; ++j) fetch(j);
      if (wave_live) {
        const bf16x8 bq = *reinterpret_cast<const bf16x8*>(bfrag);
#pragma unroll
        for (int i = 0; i < GB_MT; ++i) {
          const int m = 16 * i + fi;
          const bf16x8 aq = *reinterpret_cast<const bf16x8*>(At + m * 32 + 8 * ((kg + (m >> 2)) & 3));
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bq, acc[i], 0, 0, 0);
        }
      }
      __syncthreads();
      if (k0 + 32 < kend) {
        store();
        __syncthreads();
      }
    }
  }

  if (!wave_live) return;
  float* C = g.C + (long)bz * g.slab;
  const int col = n0 + 16 * wave + fi;
  if (col < g.N) {
#pragma unroll
    for (int i = 0; i < GB_MT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + 4 * kg + r;
        if (row < g.M) C[(long)row * g.ldc + col] = g.alpha * acc[i][r];
      }
    }
  }
}

extern "C" __global__ __launch_bounds__(256) void s2g_k_gemm_grouped_bf16(const GemmArgs* __restrict__ table,
                                                                          const int* __restrict__ prefix, int nprob,
                                                                          int nz) {
  __shared__ __attribute__((aligned(16))) __bf16 At[16 * GB_MT * 32];
  __shared__ __attribute__((aligned(16))) __bf16 Bt[64 * 32];
  // block -> (problem, tile, K-chunk): the mapping of s2g_k_gemm_grouped (one XCD works through a contiguous range
  // of logical blocks; K-chunk first across all problems when nz > 0)
  const int total = gridDim.x, q = total >> 3, r = total & 7;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  int p = 0;
  if (nz > 0) {
    const int tiles = prefix[nprob];
    const int bz = lid / tiles, t = lid - bz * tiles;
    while (p + 1 < nprob && t >= prefix[p + 1]) ++p;
    const GemmArgs gs = table[p];
    if (gs.bscale != nullptr && gs.segT > 0) wgrad_tile_bf16<true>(gs, t - prefix[p], bz, At, Bt);
    else wgrad_tile_bf16<false>(gs, t - prefix[p], bz, At, Bt);
    return;
  }
  while (p + 1 < nprob && lid >= prefix[p + 1]) ++p;
  const GemmArgs gs = table[p];
  const int local = lid - prefix[p];
  const int nx = (gs.N + 63) / 64;
  if (gs.bscale != nullptr && gs.segT > 0) wgrad_tile_bf16<true>(gs, local % nx, local / nx, At, Bt);
  else wgrad_tile_bf16<false>(gs, local % nx, local / nx, At, Bt);
}

extern "C" int s2g_launch_gemm_grouped_bf16(const GemmArgs* table_dev, const int* prefix_dev, int nprob,
                                            int total_blocks, int nz, hipStream_t stream) {
  hipLaunchKernelGGL(s2g_k_gemm_grouped_bf16, dim3(total_blocks), dim3(256), 0, stream, table_dev, prefix_dev, nprob,
                     nz);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
