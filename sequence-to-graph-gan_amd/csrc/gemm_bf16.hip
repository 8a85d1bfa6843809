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
