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
#define LDT (BK + 1)

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A; long sam, sak;
  const float* B; long sbk, sbn;
  float* C; long ldc; long slab;
  const float* bias;
  int M, N, K, kchunk, segT, a_seg, b_seg, shift, accumulate, act;
  float alpha;
};

template <bool A_KFAST, bool B_KFAST>
__global__ __launch_bounds__(256) void s2g_k_gemm_f32(GemmArgs g) {
  __shared__ float As[BM * LDT];
  __shared__ float Bs[BN * LDT];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);

  float ra[8], rb[8];
  auto gload = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int idx = tid + 256 * e;
      int kk = A_KFAST ? (idx & (BK - 1)) : (idx >> 6);
      int mm = A_KFAST ? (idx >> 5) : (idx & (BM - 1));
      int m = m0 + mm, k = k0 + kk;
      long ka = k;
      if (g.segT > 0) ka = (long)(k / g.segT) * g.a_seg + (k % g.segT);
      ra[e] = (m < g.M && k < kend) ? g.A[(long)m * g.sam + ka * g.sak] : 0.0f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int idx = tid + 256 * e;
      int kk = B_KFAST ? (idx & (BK - 1)) : (idx >> 6);
      int nn = B_KFAST ? (idx >> 5) : (idx & (BN - 1));
      int n = n0 + nn, k = k0 + kk;
      bool ok = (n < g.N && k < kend);
      long kr = k;
      if (g.segT > 0) {
        int t = k % g.segT;
        kr = (long)(k / g.segT) * g.b_seg + t;
        if (g.shift) { ok = ok && t != 0; kr -= 1; }
      }
      rb[e] = ok ? g.B[kr * g.sbk + (long)n * g.sbn] : 0.0f;
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int idx = tid + 256 * e;
      int kk = A_KFAST ? (idx & (BK - 1)) : (idx >> 6);
      int mm = A_KFAST ? (idx >> 5) : (idx & (BM - 1));
      As[mm * LDT + kk] = ra[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int idx = tid + 256 * e;
      int kk = B_KFAST ? (idx & (BK - 1)) : (idx >> 6);
      int nn = B_KFAST ? (idx >> 5) : (idx & (BN - 1));
      Bs[nn * LDT + kk] = rb[e];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

  const int arow = (wr * 32 + (lane & 31)) * LDT + (lane >> 5);
  const int brow = (wc * 32 + (lane & 31)) * LDT + (lane >> 5);

  if (kbeg < kend) {
    gload(kbeg);
    sstore();
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = (k0 + BK) < kend;
      if (more) gload(k0 + BK);
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float a = As[arow + kk];
        float b = Bs[brow + kk];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      __syncthreads();
      if (more) {
        sstore();
        __syncthreads();
      }
    }
  }

  float* C = g.C + (long)blockIdx.z * g.slab;
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
        C[off] = v;
      }
    }
  }
}

// out[perm(r)][c] (+)= scale * sum_z slabs[z][r][c];   perm = internal gate row -> torch row when
// permute_rows, identity otherwise.  One thread per output element.
extern "C" __global__ void s2g_k_reduce_slabs(const float* __restrict__ slabs, long slab, int nslab,
                                              int rows, int cols, float* __restrict__ out,
                                              int permute_rows, int accumulate, float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  int r = (int)(i / cols), c = (int)(i % cols);
  float s = 0.0f;
  for (int z = 0; z < nslab; ++z) s += slabs[(long)z * slab + i];
  int ro = permute_rows ? pt_row(r) : r;
  long o = (long)ro * cols + c;
  s *= scale;
  out[o] = accumulate ? out[o] + s : s;
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

extern "C" int s2g_launch_gemm(const GemmArgs* ga, int nsplit, hipStream_t stream) {
  GemmArgs g = *ga;
  if (g.M <= 0 || g.N <= 0 || g.K < 0 || nsplit < 1) return -1;
  if (g.kchunk <= 0) g.kchunk = ((g.K + nsplit - 1) / nsplit + BK - 1) / BK * BK;
  if (g.kchunk <= 0) g.kchunk = BK;
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, nsplit);
  const bool ak = (g.sak == 1), bk = (g.sbk == 1);
  if (ak && bk) hipLaunchKernelGGL((s2g_k_gemm_f32<true, true>), grid, dim3(256), 0, stream, g);
  else if (ak) hipLaunchKernelGGL((s2g_k_gemm_f32<true, false>), grid, dim3(256), 0, stream, g);
  else if (bk) hipLaunchKernelGGL((s2g_k_gemm_f32<false, true>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((s2g_k_gemm_f32<false, false>), grid, dim3(256), 0, stream, g);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
