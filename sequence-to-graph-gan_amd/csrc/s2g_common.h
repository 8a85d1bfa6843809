// Shared device helpers for the gfx950 kernels of the WGAN-GP step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

#define S2G_H 50          // critic hidden size (wgan_gp.py:394)
#define S2G_G 200         // 4*H gate rows
#define S2G_MASKW 52      // dropout-mask row width: H padded to a multiple of 4 (16-byte rows)

// Internal gate order: row j' = 4*unit + q with q = 0:i 1:f 2:g 3:o.  torch.nn.LSTM stores
// row j = q*H + unit (gate blocks i|f|g|o); pt_row() maps internal -> torch.
__host__ __device__ inline int pt_row(int jint) { return (jint & 3) * S2G_H + (jint >> 2); }

// One sequence of a critic batch.  xproj_t = ca * WT[token_t] + cp * P[p_off + t] (t < p_len) + b0
struct S2GSeq {
  int len;        // live steps
  int tok_off;    // offset of this sequence's tokens (int32) or -1
  int p_off;      // first row of this sequence in the projection buffer P, or -1
  int p_len;      // rows of P that exist for it
  float ca;       // coefficient on the gathered row of W_ih0^T
  float cp;       // coefficient on the projection row
  int mask_off;   // first row in the dropout-mask buffer [rows, 52], or -1
  int ws_row;     // first row of this sequence in every workspace
};

// The fc head of the critic and the loss derivative for one sequence, evaluated by sweep B itself
// (hT == nullptr: sweep B takes d loss / d h_T from its dhT argument instead).
// mode 0: critic batch [real B | fake B | interp B]: d(out) = (-inv_n, .5 dCE), (+inv_n, .5 dCE), (1, 0, 0);
// mode 1: generator pass on fakes: d(out) = (-inv_n, .25 dCE).
struct S2GHead {
  const float* hT;            // [nseq, 50] final hidden states of sweep A
  const float* W;             // fc.weight [3, 50]
  const float* bias;          // fc.bias [3]
  const long long* labels;    // [B]
  int B;
  int mode;
  float inv_n, inv_wsum, w0, w1;
};

// CrossEntropyLoss(weight=[w0,w1]) applied to p = softmax(z) (probabilities fed as logits, the
// reference's double softmax, wgan_gp.py:268,299,310): per-sample value and d/dz.
__device__ __forceinline__ void ce_on_probs(float z0, float z1, int y, float w0, float w1, float inv_wsum,
                                            float& loss, float& dz0, float& dz1) {
  float m = fmaxf(z0, z1);
  float e0 = expf(z0 - m), e1 = expf(z1 - m);
  float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
  float m2 = fmaxf(p0, p1);
  float f0 = expf(p0 - m2), f1 = expf(p1 - m2);
  float lse = m2 + logf(f0 + f1);
  float s0 = f0 / (f0 + f1), s1 = f1 / (f0 + f1);        // softmax(p)
  float wy = (y == 0 ? w0 : w1) * inv_wsum;
  loss = wy * (lse - (y == 0 ? p0 : p1));
  float dp0 = wy * (s0 - (y == 0 ? 1.0f : 0.0f)), dp1 = wy * (s1 - (y == 1 ? 1.0f : 0.0f));
  float dot = dp0 * p0 + dp1 * p1;
  dz0 = p0 * (dp0 - dot);
  dz1 = p1 * (dp1 - dot);
}

// Gradient-penalty scalars of one sample from its four partial sums of |g|^2 (s2g_k_gp_dots):
// n = |g| and coef = upstream * 2 (n - 1) / (n_batch * n), the factor on g in d(penalty)/d(...)
// (wgan_gp.py:255-256).  One fixed order of additions wherever these are evaluated.
__device__ __forceinline__ float gp_norm_from_partials(const float* p) {
  return sqrtf(fmaxf((p[0] + p[1]) + (p[2] + p[3]), 0.0f));
}
__device__ __forceinline__ float gp_coef_from_partials(const float* p, float upstream, float n_batch) {
  const float nb = gp_norm_from_partials(p);
  return nb > 0.0f ? upstream * 2.0f * (nb - 1.0f) / (n_batch * nb) : 0.0f;
}

// ---- DPP cross-lane helpers (64-wide wave, 16-lane DPP rows) ---------------------------------
#define DPP_QUAD(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_HALF_MIRROR 0x141

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(float, r);
}
// sum over the 4 lanes of a quad; every lane gets the total
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_mov<DPP_QUAD(1, 0, 3, 2)>(v);
  v += dpp_mov<DPP_QUAD(2, 3, 0, 1)>(v);
  return v;
}
// sum over the 16 lanes of a DPP row; every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
  v = quad_sum(v);
  v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_mov<DPP_ROW_MIRROR>(v);
  return v;
}
// reduce-scatter over a quad.  Lane s (= lane & 3) holds four partial sums ordered so that slot q
// belongs to item (s ^ q); on return lane s has the quad total of item s.  Three DPP adds and no
// selects: at each butterfly stage the partner's "send" slot is exactly this lane's "keep" slot.
__device__ __forceinline__ float quad_reduce_scatter(float s0, float s1, float s2, float s3) {
  float r0 = s0 + dpp_mov<DPP_QUAD(1, 0, 3, 2)>(s1);
  float r2 = s2 + dpp_mov<DPP_QUAD(1, 0, 3, 2)>(s3);
  return r0 + dpp_mov<DPP_QUAD(2, 3, 0, 1)>(r2);
}
// sum of the 4 lanes {l, l^4, l^8, l^12} of a 16-lane DPP row (rotate-and-add); all 4 get the total
#define DPP_ROW_ROR(n) (0x120 + (n))
__device__ __forceinline__ float row_quads_sum(float v) {
  v += dpp_mov<DPP_ROW_ROR(4)>(v);
  v += dpp_mov<DPP_ROW_ROR(8)>(v);
  return v;
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Streaming (non-temporal) global stores for rows a sweep saves for OTHER kernels.  Stores and loads
// share one in-order completion counter per wave on this ISA, so every wait for a prefetched load
// also waits for the stores issued before it; the streaming form is acknowledged sooner.
__device__ __forceinline__ void store_stream(float* p, float4 v) {
  __builtin_nontemporal_store((f4){v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(p));
}
__device__ __forceinline__ void store_stream(float* p, float2 v) {
  __builtin_nontemporal_store((f2){v.x, v.y}, reinterpret_cast<f2*>(p));
}

template <int L>
__device__ __forceinline__ float quad_bcast(float v) {
  return dpp_mov<DPP_QUAD(L, L, L, L)>(v);
}

// ---- activations -----------------------------------------------------------------------------
// sigma(x) = 1/(1+2^(-x*log2 e)); tanh(x) = 2*sigma(2x)-1.  v_exp_f32 / v_rcp_f32 are ~1 ulp.
__device__ __forceinline__ float fast_sigmoid(float x) {
  float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x);
  return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float fast_tanh(float x) {
  float e = __builtin_amdgcn_exp2f(-2.8853900817779268f * x);
  return 2.0f * __builtin_amdgcn_rcpf(1.0f + e) - 1.0f;
}

// Gumbel noise of one uniform draw, as CustomSoftmax.sample_gumbel computes it in fp32 (wgan_gp.py:58-61):
// u' = 1e-10 + (1 - 1e-10) u;  g = -log(-log(u' + 1e-20) + 1e-20).  Both logarithms on the hardware's v_log_f32
// (log2, ~1 ulp) times ln 2: u' >= 1e-10 and -log u' >= 6e-8 are normal numbers, so no range handling is needed,
// and the result differs from the library logf by a few ulp of a value that is ADDED to a logit (the library
// form costs ~20 instructions per logarithm, which made the sampler VALU-bound).  Every kernel that perturbs
// logits uses this one function, so the fused decode and the sampler agree on every argmax.
__device__ __forceinline__ float gumbel_noise(float u) {
  const float uu = 1e-10f + (1.0f - 1e-10f) * u;
  const float inner = -0.6931471805599453f * __builtin_amdgcn_logf(uu + 1e-20f) + 1e-20f;
  return -0.6931471805599453f * __builtin_amdgcn_logf(inner);
}
// exp(x) for x <= 0 (softmax numerators after the row maximum is subtracted): v_exp_f32 (2^x, ~1 ulp)
__device__ __forceinline__ float exp_nonpos(float x) { return __builtin_amdgcn_exp2f(1.4426950408889634f * x); }

// storage position of hidden unit k (or gate row j) in an LDS vector cut into slices of 13
// values, each slice padded to 16 floats so a lane fetches its slice with 16-byte reads
__device__ __forceinline__ int pos13(int k) { return ((k / 13) << 4) + (k % 13); }

#define S2G_CHECK(expr)                                   \
  do {                                                    \
    hipError_t _e = (expr);                               \
    if (_e != hipSuccess) return -(int)_e - 1000;         \
  } while (0)

// ---- host helpers ----------------------------------------------------------------------------
// A kernel's cap on dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize) belongs to the DEVICE it was
// set on: a process that drives several GPUs must raise it on each.  One S2GLdsCap per kernel remembers what
// every device has been given; lds_cap() raises the cap of the current device when `bytes` exceeds that.
struct S2GLdsCap {
  std::mutex mu;
  int set[64] = {0};
};
static inline int s2g_lds_cap(S2GLdsCap& c, const void* kernel, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -2;
  std::lock_guard<std::mutex> lk(c.mu);
  if (c.set[dev] < (int)bytes) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
      (void)hipGetLastError();
      return -2;
    }
    c.set[dev] = (int)bytes;
  }
  return 0;
}
// Compute units of the current device (cached per device).
static inline int s2g_device_cus() {
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  std::lock_guard<std::mutex> lk(mu);
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    cus[dev] = prop.multiProcessorCount;
  }
  return cus[dev];
}
