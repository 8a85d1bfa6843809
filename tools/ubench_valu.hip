// Development aid: issue cost of v_fma_f32 vs v_pk_fma_f32 for one wave per SIMD and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float w = 1.0001f, v = 0.5f;
  const f2 w2 = {1.0001f, 0.9999f};
  __syncthreads();
  unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        a0 = fmaf(a0, w, v); a1 = fmaf(a1, w, v); a2 = fmaf(a2, w, v); a3 = fmaf(a3, w, v);
        a4 = fmaf(a4, w, v); a5 = fmaf(a5, w, v); a6 = fmaf(a6, w, v); a7 = fmaf(a7, w, v);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const f2 vv = {a0, a0};     // broadcast operand like the mat-vec (op_sel)
        p0 = __builtin_elementwise_fma(w2, vv, p0); p1 = __builtin_elementwise_fma(w2, vv, p1);
        p2 = __builtin_elementwise_fma(w2, vv, p2); p3 = __builtin_elementwise_fma(w2, vv, p3);
        p0 = __builtin_elementwise_fma(w2, vv, p0); p1 = __builtin_elementwise_fma(w2, vv, p1);
        p2 = __builtin_elementwise_fma(w2, vv, p2); p3 = __builtin_elementwise_fma(w2, vv, p3);
      }
    }
  }
  unsigned long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int threads : {256, 512, 1024}) {
    for (int mode = 0; mode < 2; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
      else hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
      else hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("%s waves/SIMD=%d: %.2f ticks per instr, %.2f ns per instr (kernel %.1f us)\n", mode ? "v_pk_fma_f32" : "v_fma_f32   ",
             threads / 256, (double)c / (iters * 64.0), ms * 1e6 / (iters * 64.0), ms * 1e3);
    }
  }
  return 0;
}
