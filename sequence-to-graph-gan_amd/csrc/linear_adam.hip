// Weight gradient of a wide Linear layer fused with its Adam update (the generator's output layer,
// wgan_gp.py:159 `nn.Linear(512, seq_len * vocab)`: 99.6 % of the generator's parameters; its
// gradient dW = dY^T a comes from loss.backward() (wgan_gp.py:341) and is consumed by
// optimizer_G.step() (wgan_gp.py:342) and nothing else).
//
// Unfused, the layer costs three passes over 184 MB for the gradient (write, read, and the split-K
// slabs before it) and seven for Adam.  Here a workgroup forms a block of dW = sum_b dY[b][n] a[b][k]
// (32 output features x 256 columns) in LDS with v_mfma_f32_16x16x4_f32 (exact fp32, the 64 batch
// entries in 16 instructions per 16 x 16 tile) and applies Adam to it at once: W, m and v are read
// once and written once, the gradient never reaches memory.  The layer is HBM-bound; the matrix work
// is a fraction of the memory time and runs while the next block's W / m / v are in flight.
//
// Why the block goes through LDS: an MFMA tile leaves a lane with 16 bytes of one of 16 rows, so a
// wave-wide access of W in tile layout is 16 pieces of 64 bytes, 2 KB apart -- every piece in the same
// L2 channel; measured 4.3 TB/s against the 6 TB/s of the flat Adam kernel.  Out of LDS every wave
// instruction of the update touches 1 KB of ONE row, as the flat kernel does.
//
// Layout.  a^T of the workgroup's half of the columns sits in LDS ([256][64 batch entries], zero
// beyond the batch).  A tile is (reduction-side index k) x (output feature n): the MFMA result leaves
// lane (c, g) with dW[n0 + c][k0 + 4g .. k0 + 4g + 3].  MFMA q covers batch entries {q, q + 16,
// q + 32, q + 48}: lane group g supplies entry 16 g + q, so a lane's 16 A-fragments are 64
// contiguous bytes of LDS and its 16 B-fragments (dY) are registers shared by the wave's two tiles.
#include "s2g_common.h"

#define LA_WAVES 8
#define LA_LD 68          // floats per a^T row in LDS: 64 batch entries + 4 (16-lane groups hit 64 distinct banks)
#define LA_KH 256         // W columns per workgroup
#define LA_SLD 260        // floats per row of the staged gradient block
#ifndef LA_NB
#define LA_NB 2           // 16-row blocks of W per workgroup pass
#endif
#define LA_ROWS (16 * LA_NB)
#define LA_UG (2 * LA_NB)  // 16-byte groups per thread and pass
#define LA_STAGE (LA_ROWS * LA_SLD)

__device__ __forceinline__ f4 la_adam(f4 g, f4& m, f4& v, f4 p, float b1, float b2, float eps, float step_size,
                                      float inv_sqrt_bc2) {
  // s2g_k_adam's update with the hardware reciprocal and square root (1 ulp each)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float gg = g[r];
    m[r] = b1 * m[r] + (1.0f - b1) * gg;
    v[r] = b2 * v[r] + (1.0f - b2) * gg * gg;
    p[r] -= step_size * (m[r] * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v[r]) * inv_sqrt_bc2 + eps));
  }
  return p;
}

// grid: even blocks own columns [0, 256), odd blocks [256, 512) (K = 256: every block owns all of them);
// the blocks of one parity share the 16-row blocks of W round-robin.
extern "C" __global__ __launch_bounds__(64 * LA_WAVES) void s2g_k_linear_wgrad_adam(
    const float* __restrict__ dY, const float* __restrict__ A, float* __restrict__ W, float* __restrict__ Mo,
    float* __restrict__ Vo, float* __restrict__ gradW, int N, int B, int K, float b1, float b2, float eps,
    float step_size, float inv_sqrt_bc2) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* aT = lds;                          // [LA_KH][LA_LD]
  float* stage = lds + LA_KH * LA_LD;       // [2][LA_ROWS][LA_SLD]
  const int tid = threadIdx.x;
  const int nkh = K / LA_KH;                // 1 or 2 column halves
  const int kh = blockIdx.x % nkh;
  const int kbase = kh * LA_KH;
  for (int i = tid; i < LA_KH * 64; i += 64 * LA_WAVES) {
    const int b = i / LA_KH, k = i - b * LA_KH;                   // coalesced along k
    aT[k * LA_LD + b] = b < B ? A[(size_t)b * K + kbase + k] : 0.0f;
  }
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int nblk = (N + LA_ROWS - 1) / LA_ROWS;
  const int first = blockIdx.x / nkh, stride = gridDim.x / nkh;
  // update phase: thread -> LA_UG 16-byte groups of the LA_ROWS x 256 block; a wave covers 1 KB of one row
  const int urow = tid >> 6;               // + 8 h
  const int ucol = 4 * (tid & 63);

  f4 w4[LA_UG], m4[LA_UG], v4[LA_UG];
  float bf[LA_NB][16];
#define LA_FETCH(nb_)                                                                          \
  {                                                                                            \
    const int n0_ = (nb_) * LA_ROWS;                                                           \
    _Pragma("unroll") for (int h = 0; h < LA_UG; ++h) {                                        \
      const size_t off = (size_t)min(n0_ + urow + 8 * h, N - 1) * K + kbase + ucol;            \
      w4[h] = *reinterpret_cast<const f4*>(W + off);                                           \
      m4[h] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Mo + off));               \
      v4[h] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Vo + off));               \
    }                                                                                          \
    /* rows beyond the batch meet zeros of a^T: any finite value will do, so the row is clamped */ \
    _Pragma("unroll") for (int j = 0; j < LA_NB; ++j)                                          \
      _Pragma("unroll") for (int q = 0; q < 16; ++q)                                           \
        bf[j][q] = dY[(size_t)min(16 * g + q, B - 1) * N + min(n0_ + 16 * j + c, N - 1)];      \
  }
  if (first < nblk) LA_FETCH(first)
  int buf = 0;
  for (int nb = first; nb < nblk; nb += stride, buf ^= 1) {
    // this block's operands move to fresh registers, the next block's loads start before any arithmetic
    f4 cw[LA_UG], cm[LA_UG], cv[LA_UG];
    float cb[LA_NB][16];
#pragma unroll
    for (int h = 0; h < LA_UG; ++h) { cw[h] = w4[h]; cm[h] = m4[h]; cv[h] = v4[h]; }
#pragma unroll
    for (int j = 0; j < LA_NB; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) cb[j][q] = bf[j][q];
    LA_FETCH(min(nb + stride, nblk - 1))          // unconditional: the last block fetches a valid block again
    float* st = stage + buf * LA_STAGE;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int k0 = 16 * (2 * wv + t);
      const f4* ap = reinterpret_cast<const f4*>(aT + (k0 + c) * LA_LD + 16 * g);
      const f4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
#pragma unroll
      for (int j = 0; j < LA_NB; ++j) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], cb[j][q], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], cb[j][4 + q], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[q], cb[j][8 + q], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[q], cb[j][12 + q], acc, 0, 0, 0);
        *reinterpret_cast<f4*>(st + (16 * j + c) * LA_SLD + k0 + 4 * g) = acc;
      }
    }
    __syncthreads();      // one barrier per block: the staging buffers alternate
#pragma unroll
    for (int h = 0; h < LA_UG; ++h) {
      const int r = urow + 8 * h;
      const int n = nb * LA_ROWS + r;
      const f4 gr = *reinterpret_cast<const f4*>(st + r * LA_SLD + ucol);
      const f4 p = la_adam(gr, cm[h], cv[h], cw[h], b1, b2, eps, step_size, inv_sqrt_bc2);
      if (n < N) {
        const size_t off = (size_t)n * K + kbase + ucol;
        *reinterpret_cast<f4*>(W + off) = p;
        __builtin_nontemporal_store(cm[h], reinterpret_cast<f4*>(Mo + off));
        __builtin_nontemporal_store(cv[h], reinterpret_cast<f4*>(Vo + off));
        if (gradW) __builtin_nontemporal_store(gr, reinterpret_cast<f4*>(gradW + off));
      }
    }
  }
}

// max_workgroups: CUs the stream can reach (a workgroup needs ~100 KB of LDS: one per CU); 0 = all.
extern "C" int s2g_launch_linear_wgrad_adam(const float* dY, const float* A, float* W, float* Mo, float* Vo,
                                            float* gradW, int N, int B, int K, float b1, float b2, float eps,
                                            float step_size, float inv_sqrt_bc2, int max_workgroups,
                                            hipStream_t stream) {
  static S2GLdsCap lds_cap;
  const int cus = s2g_device_cus();
  if (cus <= 0 || s2g_lds_cap(lds_cap, reinterpret_cast<const void*>(&s2g_k_linear_wgrad_adam), 160 * 1024) != 0) return -2;
  const size_t lds = ((size_t)LA_KH * LA_LD + 2 * LA_STAGE) * sizeof(float);
  const int nkh = K / LA_KH;
  const long items = (long)((N + LA_ROWS - 1) / LA_ROWS) * nkh;
  long blocks = (max_workgroups > 0 && max_workgroups < cus) ? max_workgroups : cus;
  if (blocks > items) blocks = items;
  blocks -= blocks % nkh;                   // the same number of workgroups for every column half
  if (blocks < nkh) blocks = nkh;
  hipLaunchKernelGGL(s2g_k_linear_wgrad_adam, dim3((unsigned)blocks), dim3(64 * LA_WAVES), lds, stream, dY, A, W, Mo,
                     Vo, gradW, N, B, K, b1, b2, eps, step_size, inv_sqrt_bc2);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------
// Weight and bias gradients of up to four SMALL Linear layers in one launch (the generator's
// 100 -> 128 -> 256 -> 512 stack, wgan_gp.py:147-156, after loss.backward(), wgan_gp.py:341):
// dW[i][j] = sum_b dy[b][i] x[b][j], db[i] = sum_b dy[b][i] for a batch of at most a few hundred rows.
// Each product is a handful of microseconds of arithmetic; as separate split-K GEMM + slab reduction +
// column sum launches the three layers cost nine launches on a launch-bound chain.  Here a thread owns
// four consecutive j of one output row i and walks the batch (operands are L2-resident: dy is a
// broadcast within the wave, x a coalesced 16-byte load).
// ------------------------------------------------------------------------------------------
struct S2GSmallDw {
  const float* dy[4];   // [batch, n_out]
  const float* x[4];    // [batch, n_in], n_in % 4 == 0
  float* dW[4];         // [n_out, n_in]
  float* db[4];         // [n_out] or NULL
  int n_out[4], n_in[4];
  int first[5];         // first 16-byte group of each job in the launch's flat index, first[n_jobs] = total
  int n_jobs, batch;
};

extern "C" __global__ __launch_bounds__(256) void s2g_k_small_dw(S2GSmallDw a) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= a.first[a.n_jobs]) return;
  int j = 0;
  while (j + 1 < a.n_jobs && gid >= a.first[j + 1]) ++j;
  const int q = a.n_in[j] >> 2;
  const int e = gid - a.first[j];
  const int i = e / q, c4 = (e - i * q) * 4;
  const float* dy = a.dy[j] + i;
  const float* x = a.x[j] + c4;
  const int ldy = a.n_out[j], ldx = a.n_in[j];
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  float sb = 0.0f;
#pragma unroll 32
  for (int b = 0; b < a.batch; ++b) {
    const float d = dy[(size_t)b * ldy];
    const f4 xv = *reinterpret_cast<const f4*>(x + (size_t)b * ldx);
    acc += d * xv;
    sb += d;
  }
  *reinterpret_cast<f4*>(a.dW[j] + (size_t)i * ldx + c4) = acc;
  if (c4 == 0 && a.db[j]) a.db[j][i] = sb;
}

extern "C" int s2g_launch_small_dw(const S2GSmallDw* args, hipStream_t stream) {
  S2GSmallDw a = *args;
  int total = 0;
  for (int j = 0; j < a.n_jobs; ++j) {
    a.first[j] = total;
    total += a.n_out[j] * (a.n_in[j] >> 2);
  }
  a.first[a.n_jobs] = total;
  hipLaunchKernelGGL(s2g_k_small_dw, dim3((total + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
