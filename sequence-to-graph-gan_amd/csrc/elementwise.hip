// HBM-bound element-wise / small-reduction kernels of the WGAN-GP step: generator glue
// (label-embedding product, BatchNorm(batch stats, eps=0.8)+LeakyReLU(0.2), wgan_gp.py:147-171),
// fused Adam (torch.optim.Adam as constructed at wgan_gp.py:405-406), critic weight packing,
// one-hot encoding of real tokens (wgan_gp.py:277), the critic head + losses
// (wgan_gp.py:220-222, 298-315, 333-340) and the gradient-penalty pieces (wgan_gp.py:228-256).
#include "s2g_common.h"

// ------------------------------------------------------------------ generator glue
// out[b][k] = z[b][k] * emb[labels[b]][k]                                   (wgan_gp.py:166-167)
extern "C" __global__ void s2g_k_embed_mul(const float* __restrict__ z, const long long* __restrict__ labels,
                                           const float* __restrict__ emb, float* __restrict__ out, int B, int D,
                                           int n_classes) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  int b = i / D, k = i % D;
  const long long c = min(max(labels[b], 0LL), (long long)n_classes - 1);   // never outside the table
  out[i] = z[i] * emb[c * D + k];
}
// demb[c][k] = sum_{b: labels[b]==c} z[b][k] * dout[b][k]   (16 lanes = one DPP row per (c,k), each
// lane taking every 16th sample; blockDim.x must be a multiple of 16)
extern "C" __global__ void s2g_k_embed_mul_bwd(const float* __restrict__ z, const long long* __restrict__ labels,
                                               const float* __restrict__ dout, float* __restrict__ demb, int B,
                                               int D, int n_classes) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 4, r = t & 15;
  const bool live = i < n_classes * D;
  const int ic = live ? i : 0;
  const int c = ic / D, k = ic % D;
  float s = 0.0f;
  for (int b = r; b < B; b += 16) {
    const float p = z[b * D + k] * dout[b * D + k];
    s += (labels[b] == c) ? p : 0.0f;
  }
  s = row16_sum(s);
  if (live && r == 0) demb[i] = s;
}

// y = x > 0 ? x : 0.2 x  applied in place is fused into the GEMM epilogue; its backward:
extern "C" __global__ void s2g_k_lrelu_bwd(const float* __restrict__ y, float* __restrict__ dy, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dy[i] = y[i] > 0.0f ? dy[i] : 0.2f * dy[i];
}

// backward of ReLU (fused into the GEMM epilogue as act = 2): dy <- y > 0 ? dy : 0
extern "C" __global__ void s2g_k_relu_bwd(const float* __restrict__ y, float* __restrict__ dy, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dy[i] = y[i] > 0.0f ? dy[i] : 0.0f;
}

// ---- residual + LayerNorm of a transformer encoder layer (reference transformer.py:305-312:
// src = norm(src + dropout(src2))).  z = x + r (* rmask, the dropout multiplier, when given); y = LayerNorm(z) =
// (z - mean) rstd * gamma + beta with the biased variance over the E features of a row (nn.LayerNorm, eps
// inside the root).  One wavefront per row, lane l owns features l, l + 64, ...; two-pass mean / variance.
// Saves xhat = (z - mean) rstd and rstd for the backward.
#define LN_MAXPER 16      // E <= 1024
extern "C" __global__ __launch_bounds__(256) void s2g_k_add_layernorm_fwd(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ rmask,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ y,
    float* __restrict__ xhat, float* __restrict__ rstd, long rows, int E) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long base = row * E;
  float z[LN_MAXPER];
  float sum = 0.0f;
#pragma unroll
  for (int e = 0; e < LN_MAXPER; ++e) {
    const int j = lane + 64 * e;
    z[e] = 0.0f;
    if (j < E) {
      float rv = r ? r[base + j] : 0.0f;
      if (rmask) rv *= rmask[base + j];
      z[e] = x[base + j] + rv;
    }
    sum += z[e];
  }
  const float mean = wave_sum(sum) / (float)E;
  float sq = 0.0f;
#pragma unroll
  for (int e = 0; e < LN_MAXPER; ++e) {
    const int j = lane + 64 * e;
    const float d = j < E ? z[e] - mean : 0.0f;
    sq = fmaf(d, d, sq);
  }
  const float rs = 1.0f / sqrtf(wave_sum(sq) / (float)E + eps);
#pragma unroll
  for (int e = 0; e < LN_MAXPER; ++e) {
    const int j = lane + 64 * e;
    if (j < E) {
      const float xh = (z[e] - mean) * rs;
      if (xhat) xhat[base + j] = xh;
      y[base + j] = fmaf(xh, gamma[j], beta[j]);
    }
  }
  if (rstd && lane == 0) rstd[row] = rs;
}

// its backward: g = dy gamma; dz = rstd (g - mean(g) - xhat mean(g xhat)) (= d/dx and, times rmask, d/dr);
// dyxh = dy xhat, whose column sums are d(gamma) (those of dy are d(beta)).
extern "C" __global__ __launch_bounds__(256) void s2g_k_add_layernorm_bwd(
    const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
    const float* __restrict__ gamma, float* __restrict__ dz, float* __restrict__ dyxh, long rows, int E) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long base = row * E;
  float g[LN_MAXPER], xh[LN_MAXPER];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int e = 0; e < LN_MAXPER; ++e) {
    const int j = lane + 64 * e;
    g[e] = 0.0f; xh[e] = 0.0f;
    if (j < E) {
      const float d = dy[base + j];
      xh[e] = xhat[base + j];
      g[e] = d * gamma[j];
      dyxh[base + j] = d * xh[e];
    }
    s1 += g[e];
    s2 = fmaf(g[e], xh[e], s2);
  }
  const float m1 = wave_sum(s1) / (float)E, m2 = wave_sum(s2) / (float)E;
  const float rs = rstd[row];
#pragma unroll
  for (int e = 0; e < LN_MAXPER; ++e) {
    const int j = lane + 64 * e;
    if (j < E) dz[base + j] = rs * (g[e] - m1 - xh[e] * m2);
  }
}

// Inter-layer dropout multiplier of nn.LSTM(dropout=p) in train mode (wgan_gp.py:189): from uniform
// draws u[rows][52] to m = (u >= p) / (1 - p) in place; columns 50, 51 are row padding and become 0.
// One float4 (= 4 columns of one row, 52 % 4 == 0) per thread.
extern "C" __global__ void s2g_k_dropout_mask(float* __restrict__ u, long n4, float p, float keep_scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = reinterpret_cast<float4*>(u)[i];
  const int c = (int)(i % (S2G_MASKW / 4)) * 4;
  v.x = (v.x >= p) ? keep_scale : 0.0f;
  v.y = (v.y >= p) ? keep_scale : 0.0f;
  v.z = (c + 2 < S2G_H && v.z >= p) ? keep_scale : 0.0f;
  v.w = (c + 3 < S2G_H && v.w >= p) ? keep_scale : 0.0f;
  reinterpret_cast<float4*>(u)[i] = v;
}

// BatchNorm1d over the batch axis, split in two so data-parallel ranks can all-reduce the
// per-feature sums in between (SyncBN):   stats[0][f] = sum_b x, stats[1][f] = sum_b x^2.
extern "C" __global__ void s2g_k_bn_stats(const float* __restrict__ x, float* __restrict__ stats, int B, int F) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  float s = 0.0f, q = 0.0f;
  for (int b = 0; b < B; ++b) { float v = x[(long)b * F + f]; s += v; q = fmaf(v, v, q); }
  stats[f] = s; stats[F + f] = q;
}
// training: mean/var from the (global) sums over n_global rows; updates the running stats with
// momentum and the unbiased variance exactly once (update_running != 0).  eval: stats == nullptr,
// normalise with the running stats.  Saves xhat and invstd for the backward.  y = lrelu(bn(x)).
extern "C" __global__ void s2g_k_bn_lrelu_apply(const float* __restrict__ x, const float* __restrict__ stats,
                                                float n_global, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float* __restrict__ run_mean,
                                                float* __restrict__ run_var, float eps, float momentum,
                                                int update_running, float* __restrict__ y,
                                                float* __restrict__ xhat, float* __restrict__ invstd_out, int B,
                                                int F) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  float mean, var;
  if (stats) {
    mean = stats[f] / n_global;
    var = fmaxf(stats[F + f] / n_global - mean * mean, 0.0f);
    if (update_running) {
      float unb = n_global > 1.0f ? var * n_global / (n_global - 1.0f) : var;
      run_mean[f] = (1.0f - momentum) * run_mean[f] + momentum * mean;
      run_var[f] = (1.0f - momentum) * run_var[f] + momentum * unb;
    }
  } else {
    mean = run_mean[f]; var = run_var[f];
  }
  float inv = 1.0f / sqrtf(var + eps);
  if (invstd_out) invstd_out[f] = inv;
  float ga = gamma[f], be = beta[f];
  for (int b = 0; b < B; ++b) {
    long i = (long)b * F + f;
    float xh = (x[i] - mean) * inv;
    if (xhat) xhat[i] = xh;
    float v = fmaf(xh, ga, be);
    y[i] = v > 0.0f ? v : 0.2f * v;
  }
}
// backward, part 1: dyh = dy * lrelu'(y) (written back to dy); sums[0][f] = sum dyh (= dbeta),
// sums[1][f] = sum dyh * xhat (= dgamma)
extern "C" __global__ void s2g_k_bn_lrelu_bwd_stats(const float* __restrict__ y, float* __restrict__ dy,
                                                    const float* __restrict__ xhat, float* __restrict__ sums,
                                                    int B, int F) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  float s = 0.0f, q = 0.0f;
  for (int b = 0; b < B; ++b) {
    long i = (long)b * F + f;
    float d = y[i] > 0.0f ? dy[i] : 0.2f * dy[i];
    dy[i] = d; s += d; q = fmaf(d, xhat[i], q);
  }
  sums[f] = s; sums[F + f] = q;
}
// part 2: dx = gamma * invstd * (dyh - sums0/n - xhat * sums1/n), sums being global over n rows
extern "C" __global__ void s2g_k_bn_bwd_apply(const float* __restrict__ dyh, const float* __restrict__ xhat,
                                              const float* __restrict__ sums, float n_global,
                                              const float* __restrict__ gamma, const float* __restrict__ invstd,
                                              float* __restrict__ dx, int B, int F) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  float a = sums[f] / n_global, c = sums[F + f] / n_global, gi = gamma[f] * invstd[f];
  for (int b = 0; b < B; ++b) {
    long i = (long)b * F + f;
    dx[i] = gi * (dyh[i] - a - xhat[i] * c);
  }
}

// Single-process forms of the two BatchNorm halves (no all-reduce in between): one launch per layer.
// Block = 32 features x 8 row lanes; a thread sums rows rl, rl+8, ... of its feature, the 8 partials
// meet in LDS and every thread adds them in the same fixed order (deterministic), then normalises
// the rows it summed.  Same outputs as s2g_k_bn_stats + s2g_k_bn_lrelu_apply (training mode).
#define BN_FX 32
#define BN_RL 8
extern "C" __global__ __launch_bounds__(BN_FX * BN_RL) void s2g_k_bn_lrelu_fwd_fused(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ run_mean, float* __restrict__ run_var, float eps, float momentum, int update_running,
    float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ invstd_out, int B, int F) {
  __shared__ float red[2][BN_RL][BN_FX];
  const int fx = threadIdx.x % BN_FX, rl = threadIdx.x / BN_FX;
  const int f = blockIdx.x * BN_FX + fx;
  const bool live = f < F;
  const int fc = live ? f : F - 1;
  float s = 0.0f, q = 0.0f;
  for (int b = rl; b < B; b += BN_RL) { const float v = x[(long)b * F + fc]; s += v; q = fmaf(v, v, q); }
  red[0][rl][fx] = s; red[1][rl][fx] = q;
  __syncthreads();
  s = 0.0f; q = 0.0f;
#pragma unroll
  for (int r = 0; r < BN_RL; ++r) { s += red[0][r][fx]; q += red[1][r][fx]; }
  const float n = (float)B;
  const float mean = s / n;
  const float var = fmaxf(q / n - mean * mean, 0.0f);
  const float inv = 1.0f / sqrtf(var + eps);
  if (live && rl == 0) {
    if (update_running) {
      const float unb = n > 1.0f ? var * n / (n - 1.0f) : var;
      run_mean[f] = (1.0f - momentum) * run_mean[f] + momentum * mean;
      run_var[f] = (1.0f - momentum) * run_var[f] + momentum * unb;
    }
    invstd_out[f] = inv;
  }
  if (!live) return;
  const float ga = gamma[f], be = beta[f];
  for (int b = rl; b < B; b += BN_RL) {
    const long i = (long)b * F + f;
    const float xh = (x[i] - mean) * inv;
    xhat[i] = xh;
    const float v = fmaf(xh, ga, be);
    y[i] = v > 0.0f ? v : 0.2f * v;
  }
}
// backward in one launch: dyh = dy * lrelu'(y); dbeta = sum dyh; dgamma = sum dyh * xhat;
// dx = gamma * invstd * (dyh - dbeta/B - xhat * dgamma/B)
extern "C" __global__ __launch_bounds__(BN_FX * BN_RL) void s2g_k_bn_lrelu_bwd_fused(
    const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ xhat,
    const float* __restrict__ gamma, const float* __restrict__ invstd, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ dx, int B, int F) {
  __shared__ float red[2][BN_RL][BN_FX];
  const int fx = threadIdx.x % BN_FX, rl = threadIdx.x / BN_FX;
  const int f = blockIdx.x * BN_FX + fx;
  const bool live = f < F;
  const int fc = live ? f : F - 1;
  float s = 0.0f, q = 0.0f;
  for (int b = rl; b < B; b += BN_RL) {
    const long i = (long)b * F + fc;
    const float d = y[i] > 0.0f ? dy[i] : 0.2f * dy[i];
    s += d; q = fmaf(d, xhat[i], q);
  }
  red[0][rl][fx] = s; red[1][rl][fx] = q;
  __syncthreads();
  s = 0.0f; q = 0.0f;
#pragma unroll
  for (int r = 0; r < BN_RL; ++r) { s += red[0][r][fx]; q += red[1][r][fx]; }
  if (!live) return;
  if (rl == 0) { dbeta[f] = s; dgamma[f] = q; }
  const float n = (float)B;
  const float a = s / n, c = q / n, gi = gamma[f] * invstd[f];
  for (int b = rl; b < B; b += BN_RL) {
    const long i = (long)b * F + f;
    const float d = y[i] > 0.0f ? dy[i] : 0.2f * dy[i];
    dx[i] = gi * (d - a - xhat[i] * c);
  }
}

// ------------------------------------------------------------------ fused Adam
// p -= step_size * m / (sqrt(v) * inv_sqrt_bc2 + eps),  step_size = lr / (1 - b1^t),
// inv_sqrt_bc2 = 1 / sqrt(1 - b2^t)   (torch.optim.Adam, no weight decay / amsgrad).
// gscale multiplies the gradient first (1/world_size after a SUM all-reduce).
extern "C" __global__ __launch_bounds__(256) void s2g_k_adam(float* __restrict__ p, const float* __restrict__ g,
                                                             float* __restrict__ m, float* __restrict__ v, long n,
                                                             float b1, float b2, float eps, float step_size,
                                                             float inv_sqrt_bc2, float gscale) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  // g, m and v are touched once per update and are far larger than the caches for the generator's 46 M
  // parameters: nontemporal accesses keep them from displacing other lines (measured 4.5 -> 4.8 TB/s), and one
  // 16-byte group per thread instead of a capped grid-stride loop gives 4.95 TB/s
#ifndef ADAM_NT
#define ADAM_NT 1
#endif
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
#if ADAM_NT
    typedef float f4v __attribute__((ext_vector_type(4)));
    float4 P = reinterpret_cast<float4*>(p)[i];
    const f4v Gv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(g) + i);
    const f4v Mv = __builtin_nontemporal_load(reinterpret_cast<f4v*>(m) + i);
    const f4v Vv = __builtin_nontemporal_load(reinterpret_cast<f4v*>(v) + i);
    float4 G = {Gv.x, Gv.y, Gv.z, Gv.w}, M = {Mv.x, Mv.y, Mv.z, Mv.w}, V = {Vv.x, Vv.y, Vv.z, Vv.w};
#else
    float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<const float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
#endif
#define ADAM1(c)                                                   \
    { float gg = G.c * gscale;                                     \
      M.c = b1 * M.c + (1.0f - b1) * gg;                           \
      V.c = b2 * V.c + (1.0f - b2) * gg * gg;                      \
      P.c -= step_size * (M.c / (sqrtf(V.c) * inv_sqrt_bc2 + eps)); }
    ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
    reinterpret_cast<float4*>(p)[i] = P;
#if ADAM_NT
    __builtin_nontemporal_store((f4v){M.x, M.y, M.z, M.w}, reinterpret_cast<f4v*>(m) + i);
    __builtin_nontemporal_store((f4v){V.x, V.y, V.z, V.w}, reinterpret_cast<f4v*>(v) + i);
#else
    reinterpret_cast<float4*>(m)[i] = M;
    reinterpret_cast<float4*>(v)[i] = V;
#endif
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gg = g[i] * gscale;
    float mm = b1 * m[i] + (1.0f - b1) * gg;
    float vv = b2 * v[i] + (1.0f - b2) * gg * gg;
    m[i] = mm; v[i] = vv;
    p[i] -= step_size * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
  }
}

// ------------------------------------------------------------------ critic weight packing
// WT0[v][j'] = W_ih0[pt_row(j')][v];  W0p[j'][v] = W_ih0[pt_row(j')][v] (row stride roundup(V,4));
// b0[j'] = b_ih0 + b_hh0, b1[j'] = b_ih1 + b_hh1 (internal gate order j' = 4*unit + gate)
extern "C" __global__ void s2g_k_pack_critic(const float* __restrict__ wih0, const float* __restrict__ bih0,
                                             const float* __restrict__ bhh0, const float* __restrict__ bih1,
                                             const float* __restrict__ bhh1, float* __restrict__ WT0,
                                             float* __restrict__ W0p, float* __restrict__ b0,
                                             float* __restrict__ b1, int V) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < S2G_G * V) {
    int j = i / V, v = i % V;
    float w = wih0[(long)pt_row(j) * V + v];
    W0p[(long)j * ((V + 3) & ~3) + v] = w;      // rows padded to a multiple of 4 floats (pad stays 0)
    WT0[(long)v * S2G_G + j] = w;
  }
  if (i < S2G_G) {
    b0[i] = bih0[pt_row(i)] + bhh0[pt_row(i)];
    b1[i] = bih1[pt_row(i)] + bhh1[pt_row(i)];
  }
}

// x[r][v] = (tokens[r] == v), rows beyond the sequence's length are zero      (wgan_gp.py:277)
// Each thread owns four consecutive floats of the flat [rows * V] output (one 16-byte store; the group may
// straddle two rows, so the row / column pair is stepped per element).
extern "C" __global__ void s2g_k_one_hot(const int* __restrict__ tokens, const int* __restrict__ lengths, int T,
                                         float* __restrict__ x, long rows, int V) {
  // 32-bit index arithmetic (the host rejects rows * V >= 2^31): a 64-bit division costs ~40 instructions
  const unsigned i = 4u * (blockIdx.x * blockDim.x + threadIdx.x);
  const unsigned n = (unsigned)rows * (unsigned)V;
  if (i >= n) return;
  unsigned r = i / (unsigned)V;
  int v = (int)(i - r * (unsigned)V);
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[k] = 0.0f;
    if (i + k < n) {
      const bool live = lengths ? (int)(r % (unsigned)T) < lengths[r / (unsigned)T] : true;
      o[k] = (live && min(max(tokens[r], 0), V - 1) == v) ? 1.0f : 0.0f;      // ids clamped as in the forward sweep
    }
    if (++v == V) { v = 0; ++r; }
  }
  if (i + 3 < n) *reinterpret_cast<float4*>(x + i) = make_float4(o[0], o[1], o[2], o[3]);
  else
    for (int k = 0; i + k < n; ++k) x[i + k] = o[k];
}

// x_hat[b][t][v] = alpha_b * onehot(tok[b][t])[v] + (1 - alpha_b) * fake[b][t][v] (0 for t >= Tf)
// (wgan_gp.py:231-239); rows t >= len_b are written as zero (packing drops them, :241).
// Four consecutive floats per thread; with Tf == T the fake batch has the output's own flat layout and
// is read with one 16-byte load.
extern "C" __global__ void s2g_k_gp_interp(const int* __restrict__ tokens, const int* __restrict__ lengths,
                                           const float* __restrict__ fake, const float* __restrict__ alpha, int B,
                                           int T, int Tf, int V, float* __restrict__ xhat) {
  const unsigned i = 4u * (blockIdx.x * blockDim.x + threadIdx.x);      // 32-bit index arithmetic, as in s2g_k_one_hot
  const unsigned n = (unsigned)B * (unsigned)T * (unsigned)V;
  if (i >= n) return;
  unsigned r = i / (unsigned)V;
  int v = (int)(i - r * (unsigned)V);
  const bool full = i + 3 < n;
  float f[4] = {0.f, 0.f, 0.f, 0.f};
  const bool same = Tf == T;
  if (same && full) {
    const float4 q = *reinterpret_cast<const float4*>(fake + i);
    f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
  }
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[k] = 0.0f;
    if (i + k < n) {
      const int t = (int)(r % (unsigned)T), b = (int)(r / (unsigned)T);
      if (t < lengths[b]) {
        const float a = alpha[b];
        float fk = f[k];
        if (!(same && full)) fk = (t < Tf) ? fake[((long)b * Tf + t) * V + v] : 0.0f;
        o[k] = (min(max(tokens[r], 0), V - 1) == v ? a : 0.0f) + (1.0f - a) * fk;
      }
    }
    if (++v == V) { v = 0; ++r; }
  }
  if (full) *reinterpret_cast<float4*>(xhat + i) = make_float4(o[0], o[1], o[2], o[3]);
  else
    for (int k = 0; i + k < n; ++k) xhat[i + k] = o[k];
}

// Gradient-penalty reduction (wgan_gp.py:255-256) fused with the scaling that seeds the
// second-order sweeps.  One workgroup per sample b: n_b = |g_b|_2 over its T*V entries,
// pen[b] = (n_b - 1)^2, then g_b <- coef_b * g_b in place with
// coef_b = upstream * 2 (n_b - 1) / (n_batch * n_b)   (the direction v_b of sweep C).
extern "C" __global__ __launch_bounds__(1024) void s2g_k_gp_norm_penalty(float* __restrict__ g, long per_sample,
                                                                         float upstream, float n_batch,
                                                                         float* __restrict__ pen,
                                                                         float* __restrict__ norms, int scale_in_place) {
  __shared__ float red[16];
  __shared__ float coef_s;
  const int b = blockIdx.x;
  float* gb = g + (long)b * per_sample;
  float s = 0.0f;
  // 16-byte accesses only when every sample starts 16-byte aligned
  const long n4 = (per_sample & 3) == 0 ? (per_sample >> 2) : 0;
  // four independent 16-byte loads in flight per thread: one workgroup per sample leaves most CUs idle, so
  // each workgroup has to keep its own share of the memory pipeline full
  float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  long i = threadIdx.x;
  for (; i + 3 * (long)blockDim.x < n4; i += 4 * (long)blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(gb)[i], b4 = reinterpret_cast<const float4*>(gb)[i + blockDim.x];
    const float4 c = reinterpret_cast<const float4*>(gb)[i + 2 * blockDim.x];
    const float4 d = reinterpret_cast<const float4*>(gb)[i + 3 * blockDim.x];
    s = fmaf(a.x, a.x, s); s = fmaf(a.y, a.y, s); s = fmaf(a.z, a.z, s); s = fmaf(a.w, a.w, s);
    s1 = fmaf(b4.x, b4.x, s1); s1 = fmaf(b4.y, b4.y, s1); s1 = fmaf(b4.z, b4.z, s1); s1 = fmaf(b4.w, b4.w, s1);
    s2 = fmaf(c.x, c.x, s2); s2 = fmaf(c.y, c.y, s2); s2 = fmaf(c.z, c.z, s2); s2 = fmaf(c.w, c.w, s2);
    s3 = fmaf(d.x, d.x, s3); s3 = fmaf(d.y, d.y, s3); s3 = fmaf(d.z, d.z, s3); s3 = fmaf(d.w, d.w, s3);
  }
  for (; i < n4; i += blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(gb)[i];
    s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
  }
  s = (s + s1) + (s2 + s3);
  for (long i = (n4 << 2) + threadIdx.x; i < per_sample; i += blockDim.x) s = fmaf(gb[i], gb[i], s);
  s = row16_sum(s);
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
    float nb = sqrtf(tot);
    pen[b] = (nb - 1.0f) * (nb - 1.0f);
    if (norms) norms[b] = nb;
    coef_s = nb > 0.0f ? upstream * 2.0f * (nb - 1.0f) / (n_batch * nb) : 0.0f;
  }
  __syncthreads();
  if (scale_in_place) {
    const float c = coef_s;
    for (long i = threadIdx.x; i < n4; i += blockDim.x) {
      float4 v = reinterpret_cast<float4*>(gb)[i];
      v.x *= c; v.y *= c; v.z *= c; v.w *= c;
      reinterpret_cast<float4*>(gb)[i] = v;
    }
    for (long i = (n4 << 2) + threadIdx.x; i < per_sample; i += blockDim.x) gb[i] *= c;
  }
}

// The same reduction without materialising g first.  With g_t = W_ih0^T delta_t (delta = gate
// gradient of layer 0, rows of sample b) and pv_t = (W_ih0 W_ih0^T) delta_t:
//   |g_b|^2 = sum_t delta_t . pv_t        and        W_ih0 (coef_b g_t) = coef_b pv_t,
// so one [rows,200]x[200,200] product gives both the norm and, scaled in place here, the projected
// direction the tangent sweep starts from; the 175-wide g itself is only needed later (for one
// weight-gradient product) and is computed off the critical path.
extern "C" __global__ __launch_bounds__(1024) void s2g_k_gp_norm_penalty_proj(
    const float* __restrict__ delta, float* __restrict__ pv, long per_sample, float upstream, float n_batch,
    float* __restrict__ pen, float* __restrict__ norms) {
  __shared__ float red[16];
  __shared__ float coef_s;
  const int b = blockIdx.x;
  const float4* d4 = reinterpret_cast<const float4*>(delta + (long)b * per_sample);
  float4* p4 = reinterpret_cast<float4*>(pv + (long)b * per_sample);
  const long n4 = per_sample >> 2;
  float s0 = 0.0f, s1 = 0.0f;
  for (long i = threadIdx.x; i < n4; i += blockDim.x) {
    const float4 d = d4[i], p = p4[i];
    s0 = fmaf(d.x, p.x, s0); s1 = fmaf(d.y, p.y, s1); s0 = fmaf(d.z, p.z, s0); s1 = fmaf(d.w, p.w, s1);
  }
  float s = row16_sum(s0 + s1);
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
    const float nb = sqrtf(fmaxf(tot, 0.0f));
    pen[b] = (nb - 1.0f) * (nb - 1.0f);
    if (norms) norms[b] = nb;
    coef_s = nb > 0.0f ? upstream * 2.0f * (nb - 1.0f) / (n_batch * nb) : 0.0f;
  }
  __syncthreads();
  const float c = coef_s;
  for (long i = threadIdx.x; i < n4; i += blockDim.x) {
    float4 p = p4[i];
    p.x *= c; p.y *= c; p.z *= c; p.w *= c;
    p4[i] = p;
  }
}

// The reduction alone, in four blocks per sample (grid (B, 4)) and with pv left unscaled:
// partial[b][q] = sum over the q-th quarter of sample b of delta . pv.  The tangent sweep adds the four
// partials itself (gp_coef_from_partials) and scales its seed; s2g_k_gp_penalty_final turns them into
// pen / norms off the critical path.  Both add in the same fixed order.
extern "C" __global__ __launch_bounds__(1024) void s2g_k_gp_dots(const float* __restrict__ delta,
                                                                 const float* __restrict__ pv, long per_sample,
                                                                 float* __restrict__ partial) {
  __shared__ float red[16];
  const int b = blockIdx.x, q = blockIdx.y;
  const float4* d4 = reinterpret_cast<const float4*>(delta + (long)b * per_sample);
  const float4* p4 = reinterpret_cast<const float4*>(pv + (long)b * per_sample);
  const long n4 = per_sample >> 2;
  const long lo = n4 * q / 4, hi = n4 * (q + 1) / 4;
  float s0 = 0.0f, s1 = 0.0f;
  for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float4 d = d4[i], p = p4[i];
    s0 = fmaf(d.x, p.x, s0); s1 = fmaf(d.y, p.y, s1); s0 = fmaf(d.z, p.z, s0); s1 = fmaf(d.w, p.w, s1);
  }
  float s = row16_sum(s0 + s1);
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
    partial[b * 4 + q] = tot;
  }
}
extern "C" __global__ void s2g_k_gp_penalty_final(const float* __restrict__ partial, int B, float upstream,
                                                  float n_batch, float* __restrict__ pen, float* __restrict__ norms,
                                                  float* __restrict__ coef) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float nb = gp_norm_from_partials(partial + 4 * b);
  pen[b] = (nb - 1.0f) * (nb - 1.0f);
  if (norms) norms[b] = nb;
  if (coef) coef[b] = gp_coef_from_partials(partial + 4 * b, upstream, n_batch);     // the factor sweep C applies
}

// v_b = coef_b * g_b with coef_b = upstream * 2 (n_b - 1) / (n_batch * n_b), out of place
// (used when the upstream gradient only becomes known in a later backward call).
extern "C" __global__ void s2g_k_gp_scale(const float* __restrict__ g, const float* __restrict__ norms,
                                          const float* __restrict__ upstream, float n_batch, long per_sample,
                                          float* __restrict__ v, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float nb = norms[i / per_sample];
  float c = nb > 0.0f ? upstream[0] * 2.0f * (nb - 1.0f) / (n_batch * nb) : 0.0f;
  v[i] = c * g[i];
}

// ------------------------------------------------------------------ critic head and losses
// out[n][j] = fc.weight[j] . hT[n] + fc.bias[j]                              (wgan_gp.py:220-222)
extern "C" __global__ void s2g_k_head_fwd(const float* __restrict__ hT, const float* __restrict__ W,
                                          const float* __restrict__ b, float* __restrict__ out, int N, int J) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * J) return;
  int n = i / J, j = i % J;
  float s = b[j];
  for (int k = 0; k < S2G_H; ++k) s = fmaf(W[j * S2G_H + k], hT[(long)n * S2G_H + k], s);
  out[i] = s;
}
// dhT[n][k] = sum_j dout[n][j] W[j][k]
extern "C" __global__ void s2g_k_head_bwd_dh(const float* __restrict__ dout, const float* __restrict__ W,
                                             float* __restrict__ dhT, int N, int J) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * S2G_H) return;
  int n = i / S2G_H, k = i % S2G_H;
  float s = 0.0f;
  for (int j = 0; j < J; ++j) s = fmaf(dout[n * J + j], W[j * S2G_H + k], s);
  dhT[i] = s;
}
// dW[j][k] (+)= sum_n dout[n][j] hT[n][k];  db[j] (+)= sum_n dout[n][j]     (one thread per (j,k))
extern "C" __global__ void s2g_k_head_bwd_w(const float* __restrict__ dout, const float* __restrict__ hT,
                                            float* __restrict__ dW, float* __restrict__ db, int N, int J,
                                            int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= J * S2G_H) return;
  int j = i / S2G_H, k = i % S2G_H;
  // four independent partial sums: the loads of a group are in flight together instead of one
  // memory round trip per sample
  float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f, q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
  int n = 0;
  for (; n + 4 <= N; n += 4) {
    const float d0 = dout[n * J + j], d1 = dout[(n + 1) * J + j], d2 = dout[(n + 2) * J + j],
                d3 = dout[(n + 3) * J + j];
    p0 = fmaf(d0, hT[(long)n * S2G_H + k], p0);
    p1 = fmaf(d1, hT[(long)(n + 1) * S2G_H + k], p1);
    p2 = fmaf(d2, hT[(long)(n + 2) * S2G_H + k], p2);
    p3 = fmaf(d3, hT[(long)(n + 3) * S2G_H + k], p3);
    q0 += d0; q1 += d1; q2 += d2; q3 += d3;
  }
  for (; n < N; ++n) {
    const float d = dout[n * J + j];
    p0 = fmaf(d, hT[(long)n * S2G_H + k], p0);
    q0 += d;
  }
  const float s = (p0 + p1) + (p2 + p3), sb = (q0 + q1) + (q2 + q3);
  dW[i] = accumulate ? dW[i] + s : s;
  if (k == 0) db[j] = accumulate ? db[j] + sb : sb;
}


// Critic loss terms and d(out) for a batch laid out [real B | fake B | interp B]
// (wgan_gp.py:298-315).  out: [3B,3].  inv_n = 1/B_global, inv_wsum = 1/sum_b w[y_b] (global).
//   dout_real = (-inv_n, 0.5 dCE), dout_fake = (+inv_n, 0.5 dCE), dout_interp = (1, 0, 0)
//   losses[0] = -sum real_validity*inv_n, [1] = sum fake_validity*inv_n, [2] = 0.5*(CE_r+CE_f)
// Single workgroup; B <= 1024.
extern "C" __global__ __launch_bounds__(1024) void s2g_k_critic_loss(const float* __restrict__ out,
                                                                     const long long* __restrict__ labels, int B,
                                                                     float inv_n, float inv_wsum, float w0,
                                                                     float w1, float* __restrict__ dout,
                                                                     float* __restrict__ losses) {
  __shared__ float red[3][16];
  const int b = threadIdx.x;
  float lr = 0.f, lf = 0.f, lce = 0.f;
  if (b < B) {
    int y = (int)labels[b];
    float l, d0, d1;
    const float* o = out + b * 3;
    ce_on_probs(o[1], o[2], y, w0, w1, inv_wsum, l, d0, d1);
    dout[b * 3 + 0] = -inv_n; dout[b * 3 + 1] = 0.5f * d0; dout[b * 3 + 2] = 0.5f * d1;
    lr = -o[0] * inv_n; lce = 0.5f * l;
    o = out + (B + b) * 3;
    ce_on_probs(o[1], o[2], y, w0, w1, inv_wsum, l, d0, d1);
    dout[(B + b) * 3 + 0] = inv_n; dout[(B + b) * 3 + 1] = 0.5f * d0; dout[(B + b) * 3 + 2] = 0.5f * d1;
    lf = o[0] * inv_n; lce += 0.5f * l;
    dout[(2 * B + b) * 3 + 0] = 1.0f; dout[(2 * B + b) * 3 + 1] = 0.0f; dout[(2 * B + b) * 3 + 2] = 0.0f;
  }
  lr = row16_sum(lr); lr += __shfl_xor(lr, 16); lr += __shfl_xor(lr, 32);
  lf = row16_sum(lf); lf += __shfl_xor(lf, 16); lf += __shfl_xor(lf, 32);
  lce = row16_sum(lce); lce += __shfl_xor(lce, 16); lce += __shfl_xor(lce, 32);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lr; red[1][threadIdx.x >> 6] = lf; red[2][threadIdx.x >> 6] = lce; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[threadIdx.x][w];
    losses[threadIdx.x] = s;
  }
}

// The four small kernels between sweep A and sweep B of a critic iteration in ONE single-workgroup
// launch: head forward on [real B | fake B | interp B], the loss terms and d(out) of
// s2g_k_critic_loss, d(hT) for all 3B rows and the fc gradients over the first 2B rows (the interp
// third's d(out) = e0 only seeds the penalty).  Phases talk through LDS (out, d(out) and the fc weight are
// mirrored there; dynamic LDS of (18 B + 3 S2G_H) floats): the kernel sits between two sweeps of every critic
// iteration, and each phase that re-read the previous one's global stores paid a memory round trip.
extern "C" __global__ __launch_bounds__(1024) void s2g_k_critic_head(
    const float* __restrict__ hT, const float* __restrict__ W, const float* __restrict__ bias,
    const long long* __restrict__ labels, int B, float inv_n, float inv_wsum, float w0, float w1,
    float* __restrict__ out, float* __restrict__ dout, float* __restrict__ losses, float* __restrict__ dhT,
    float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float red[3][16];
  extern __shared__ float ch_lds[];
  float* s_out = ch_lds;                 // [3B][3]
  float* s_dout = ch_lds + 9 * B;        // [3B][3]
  float* s_W = ch_lds + 18 * B;          // [3][S2G_H]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int y = tid < B ? (int)labels[tid] : 0;            // fetched beside the head's operands
  if (tid < 3 * S2G_H) s_W[tid] = W[tid];
  for (int i = tid; i < 3 * B * 3; i += nt) {
    const int n = i / 3, j = i - 3 * n;
    float s = bias[j];
    for (int k = 0; k < S2G_H; ++k) s = fmaf(W[j * S2G_H + k], hT[n * S2G_H + k], s);
    out[i] = s;
    s_out[i] = s;
  }
  __syncthreads();
  float lr = 0.f, lf = 0.f, lce = 0.f;
  if (tid < B) {
    const int b = tid;
    float l, d0, d1;
    const float* o = s_out + b * 3;
    ce_on_probs(o[1], o[2], y, w0, w1, inv_wsum, l, d0, d1);
    s_dout[b * 3 + 0] = -inv_n; s_dout[b * 3 + 1] = 0.5f * d0; s_dout[b * 3 + 2] = 0.5f * d1;
    lr = -o[0] * inv_n; lce = 0.5f * l;
    o = s_out + (B + b) * 3;
    ce_on_probs(o[1], o[2], y, w0, w1, inv_wsum, l, d0, d1);
    s_dout[(B + b) * 3 + 0] = inv_n; s_dout[(B + b) * 3 + 1] = 0.5f * d0; s_dout[(B + b) * 3 + 2] = 0.5f * d1;
    lf = o[0] * inv_n; lce += 0.5f * l;
    s_dout[(2 * B + b) * 3 + 0] = 1.0f; s_dout[(2 * B + b) * 3 + 1] = 0.0f; s_dout[(2 * B + b) * 3 + 2] = 0.0f;
  }
  lr = row16_sum(lr); lr += __shfl_xor(lr, 16); lr += __shfl_xor(lr, 32);
  lf = row16_sum(lf); lf += __shfl_xor(lf, 16); lf += __shfl_xor(lf, 32);
  lce = row16_sum(lce); lce += __shfl_xor(lce, 16); lce += __shfl_xor(lce, 32);
  if ((tid & 63) == 0) { red[0][tid >> 6] = lr; red[1][tid >> 6] = lf; red[2][tid >> 6] = lce; }
  __syncthreads();
  if (tid < 3) {
    float s = 0.0f;
    for (int w = 0; w < (nt >> 6); ++w) s += red[tid][w];
    losses[tid] = s;
  }
  for (int i = tid; i < 3 * B * 3; i += nt) dout[i] = s_dout[i];
  for (int i = tid; i < 3 * B * S2G_H; i += nt) {
    const int n = i / S2G_H, k = i - n * S2G_H;
    dhT[i] = fmaf(s_dout[n * 3 + 2], s_W[2 * S2G_H + k], fmaf(s_dout[n * 3 + 1], s_W[S2G_H + k], s_dout[n * 3] * s_W[k]));
  }
  // fc gradients over the 2B real + fake rows: 4 threads per (j, k), quad-summed
  const int q = tid & 3, o4 = tid >> 2;
  if (o4 < 3 * S2G_H) {
    const int j = o4 / S2G_H, k = o4 - j * S2G_H;
    float s = 0.0f, sb = 0.0f;
    // eight rows' loads of hT in flight, then their terms in the same ascending order (one row per iteration
    // paid a cache round trip each: 2B / 4 of them in a row)
    const int nlast = 2 * B - 1;
    for (int nb = q; nb < 2 * B; nb += 32) {
      float hv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) hv[u] = hT[min(nb + 4 * u, nlast) * S2G_H + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int n = nb + 4 * u;
        const float d = n < 2 * B ? s_dout[min(n, nlast) * 3 + j] : 0.0f;
        s = n < 2 * B ? fmaf(d, hv[u], s) : s;
        sb += d;
      }
    }
    s = quad_sum(s);
    sb = quad_sum(sb);
    if (q == 0) {
      dW[o4] = s;
      if (k == 0) db[j] = sb;
    }
  }
}

// Generator loss (wgan_gp.py:333-340): g_loss = -mean(validity) + 0.5 * (0.5 * CE).
// out: [B,3];  dout = (-inv_n, 0.25 dCE);  losses[0] = -sum validity*inv_n, losses[1] = 0.25*CE.
extern "C" __global__ __launch_bounds__(1024) void s2g_k_generator_loss(const float* __restrict__ out,
                                                                        const long long* __restrict__ labels, int B,
                                                                        float inv_n, float inv_wsum, float w0,
                                                                        float w1, float* __restrict__ dout,
                                                                        float* __restrict__ losses) {
  __shared__ float red[2][16];
  const int b = threadIdx.x;
  float lv = 0.f, lce = 0.f;
  if (b < B) {
    float l, d0, d1;
    const float* o = out + b * 3;
    ce_on_probs(o[1], o[2], (int)labels[b], w0, w1, inv_wsum, l, d0, d1);
    dout[b * 3 + 0] = -inv_n; dout[b * 3 + 1] = 0.25f * d0; dout[b * 3 + 2] = 0.25f * d1;
    lv = -o[0] * inv_n; lce = 0.25f * l;
  }
  lv = row16_sum(lv); lv += __shfl_xor(lv, 16); lv += __shfl_xor(lv, 32);
  lce = row16_sum(lce); lce += __shfl_xor(lce, 16); lce += __shfl_xor(lce, 32);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lv; red[1][threadIdx.x >> 6] = lce; }
  __syncthreads();
  if (threadIdx.x < 2) {
    float s = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[threadIdx.x][w];
    losses[threadIdx.x] = s;
  }
}

// Sequence descriptors for the fused critic batch [real B | fake B | interp B]:
//   real   : tokens, len = lengths[b]
//   fake   : projection rows b*Tf.., len = Tf                       (dense call, wgan_gp.py:302)
//   interp : alpha*tokens + (1-alpha)*projection rows (t < Tf), len = lengths[b]   (:239-241)
// followed by a 4th group repeating interp with group-relative rows.  seqs holds 4B entries.
// use_mask: dropout-mask rows coincide with workspace rows.
extern "C" __global__ void s2g_k_critic_batch_desc(const int* __restrict__ lengths, const float* __restrict__ alpha,
                                                   int B, int T, int Tf, int use_mask, S2GSeq* __restrict__ seqs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * B) return;
  int kind = i / B, b = i % B;
  const int Tw = T > Tf ? T : Tf;     // workspace rows per sequence
  const int len_b = min(max(lengths[b], 0), T);      // never beyond the rows a sequence owns
  S2GSeq s;
  // kind 3 repeats the interp sequences with rows counted from the first interp row, for the
  // second-order sweeps whose buffers hold the interp sequences only
  const int r0 = (kind == 3 ? b : i) * Tw;
  s.ws_row = r0;
  s.mask_off = use_mask ? r0 : -1;
  if (kind == 0) {
    s.len = len_b; s.tok_off = b * T; s.p_off = -1; s.p_len = 0; s.ca = 1.0f; s.cp = 0.0f;
  } else if (kind == 1) {
    s.len = Tf; s.tok_off = -1; s.p_off = b * Tf; s.p_len = Tf; s.ca = 0.0f; s.cp = 1.0f;
  } else {
    float a = alpha[b];
    s.len = len_b; s.tok_off = b * T; s.p_off = b * Tf; s.p_len = Tf; s.ca = a; s.cp = 1.0f - a;
  }
  seqs[i] = s;
}

// Dense form of the interpolation for callers that hold the real batch as a dense [B,T,V] tensor
// (compute_gradient_penalty's signature, wgan_gp.py:225-239).
extern "C" __global__ void s2g_k_gp_interp_dense(const float* __restrict__ real, const int* __restrict__ lengths,
                                                 const float* __restrict__ fake, const float* __restrict__ alpha,
                                                 int B, int T, int Tf, int V, float* __restrict__ xhat) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * T * V) return;
  int v = (int)(i % V);
  long r = i / V;
  int t = (int)(r % T), b = (int)(r / T);
  float out = 0.0f;
  if (t < lengths[b]) {
    float a = alpha[b];
    float f = (t < Tf) ? fake[((long)b * Tf + t) * V + v] : 0.0f;
    out = a * real[i] + (1.0f - a) * f;
  }
  xhat[i] = out;
}
