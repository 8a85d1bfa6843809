// Measures the sustained rate of the fp32 MFMA instructions from registers only (no memory traffic):
// the practical ceiling for the exact-fp32 products of this repository.  Development aid.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 12345.0f) out[threadIdx.x] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0;
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.0f) out[threadIdx.x] = s;
}

// the same 13 MFMAs per step, but every step's 14 operand values come out of LDS (one ds_read_b32 per
// value, fragments double-buffered in registers as in the product kernels): the feed rate the
// LDS-staged kernels can hope for
__global__ __launch_bounds__(512) void k16_lds(float* out, int iters) {
  __shared__ float S[32 * 224];
  for (int i = threadIdx.x; i < 32 * 224; i += blockDim.x) S[i] = 1.0f / (1 + i);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int aoff = (lane >> 4) * 208 + 16 * wave + (lane & 15);
  f32x4 acc[13];
  for (int j = 0; j < 13; ++j) acc[j] = (f32x4){0, 0, 0, 0};
  float a0, a1, b0[13], b1[13];
  auto frag = [&](int q, float& a, float (&b)[13]) {
    a = S[aoff + 4 * q * 208 + 13];
#pragma unroll
    for (int j = 0; j < 13; ++j) b[j] = S[aoff + 4 * q * 208 + 16 * j - 16 * wave];
  };
  frag(0, a0, b0);
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      frag((q + 1) & 7, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 13; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      frag((q + 2) & 7, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 13; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int j = 0; j < 13; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 12345.0f) out[threadIdx.x] = s;
}

template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  const int iters = 4000;
  for (int threads : {256, 512, 1024}) {
    for (int blocks : {256, 512}) {
      float ms = timeit([&] { hipLaunchKernelGGL(k16<13>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f); });
      double fl = (double)blocks * (threads / 64) * iters * 13 * 2048.0;
      printf("16x16x4  blocks=%d threads=%d: %.3f ms  %.1f TFLOP/s\n", blocks, threads, ms, fl / ms * 1e-9);
      ms = timeit([&] { hipLaunchKernelGGL(k32<7>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f); });
      fl = (double)blocks * (threads / 64) * iters * 7 * 4096.0;
      printf("32x32x2  blocks=%d threads=%d: %.3f ms  %.1f TFLOP/s\n", blocks, threads, ms, fl / ms * 1e-9);
    }
  }
  {
    float ms = timeit([&] { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 2.0f); });
    printf("16x16x4  4 accumulators (dependent every 4th MFMA) blocks=256 threads=512: %.1f TFLOP/s\n",
           256.0 * 8 * iters * 4 * 2048.0 / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(k16<2>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 2.0f); });
    printf("16x16x4  2 accumulators (dependent every 2nd MFMA) blocks=256 threads=512: %.1f TFLOP/s\n",
           256.0 * 8 * iters * 2 * 2048.0 / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
    printf("16x16x4  4 accumulators, one wave per SIMD: %.1f TFLOP/s\n", 256.0 * 4 * iters * 4 * 2048.0 / ms * 1e-9);
  }
  for (int blocks : {256, 512}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k16_lds, dim3(blocks), dim3(512), 0, 0, out, iters); });
    double fl = (double)blocks * 8 * iters * 13 * 2048.0;
    printf("16x16x4 fed from LDS  blocks=%d threads=512: %.3f ms  %.1f TFLOP/s\n", blocks, ms, fl / ms * 1e-9);
  }
  return 0;
}
