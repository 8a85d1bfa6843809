// Host side of libs2g: argument checks + kernel launches behind the C ABI of include/s2g.h.
// Single translation unit: the kernel files are included so every launch sees its kernel.
#include "s2g_common.h"
#include "lstm_fwd.hip"
#include "lstm_bwd.hip"
#include "gemm.hip"
#include "gemm_bf16.hip"
#include "sampler.hip"
#include "elementwise.hip"
#include "attention.hip"
#include "lstm64.hip"
#include "graph.hip"
#include "linear_adam.hip"
#include <math.h>
#include <mutex>

#define ST(s) ((hipStream_t)(s))
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -2)
static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

extern "C" {

int s2g_version(void) { return 100; }

int s2g_gumbel_softmax_fwd(const float* logits, const float* U, const float* bias, int bias_rows, float* y_out,
                           float* y_soft, long rows, int V, float tau, int mode, int hard, void* stream) {
  if (!logits || !y_out || rows <= 0 || V <= 0 || V > 64 * MAXPER || mode < 0 || mode > 2) return -1;
  if (mode == 0 && !U) return -1;
  if (mode != 2 && !(tau > 0.0f)) return -1;
  if (hard && !y_soft) return -1;
  if (bias && bias_rows <= 0) return -1;
  if (V <= 256)
    hipLaunchKernelGGL(s2g_k_gumbel_softmax_fwd_v4<false>, dim3(cdiv(rows, 4 * SAMPLER_ROWS)), dim3(256), 0, ST(stream), logits, U,
                       bias, bias_rows, y_out, y_soft, rows, V, 1.0f / tau, mode, hard, 0ULL, 0ULL);
  else
    hipLaunchKernelGGL(s2g_k_gumbel_softmax_fwd, dim3(cdiv(rows, 4)), dim3(256), 0, ST(stream), logits, U, bias,
                       bias_rows, y_out, y_soft, rows, V, 1.0f / tau, mode, hard);
  return LAUNCH_OK();
}

int s2g_gumbel_softmax_fwd_rng(const float* logits, const float* bias, int bias_rows, float* y_out, float* y_soft,
                               long rows, int V, float tau, int hard, unsigned long long seed,
                               unsigned long long offset, void* stream) {
  if (!logits || !y_out || rows <= 0 || V <= 0 || V > 256 || !(tau > 0.0f)) return -1;
  if (hard && !y_soft) return -1;
  if (bias && bias_rows <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_gumbel_softmax_fwd_v4<true>, dim3(cdiv(rows, 4 * SAMPLER_ROWS)), dim3(256), 0, ST(stream), logits,
                     (const float*)nullptr, bias, bias_rows, y_out, y_soft, rows, V, 1.0f / tau, 0, hard, seed, offset);
  return LAUNCH_OK();
}

int s2g_rng_fill(float* out, long n, int kind, unsigned long long seed, unsigned long long offset, void* stream) {
  if (!out || n <= 0 || (kind != 0 && kind != 1)) return -1;
  hipLaunchKernelGGL(s2g_k_rng_fill, dim3(cdiv((n + 3) / 4, 256)), dim3(256), 0, ST(stream), out, n, kind, seed, offset);
  return LAUNCH_OK();
}

int s2g_dropout_mask_rng(float* mask, long rows, int width, int live_cols, float p, unsigned long long seed,
                         unsigned long long offset, void* stream) {
  if (!mask || rows <= 0 || !(p >= 0.0f) || !(p < 1.0f) || ((uintptr_t)mask & 15)) return -1;
  if (width <= 0 || (width & 3) || live_cols <= 0 || live_cols > width) return -1;
  const long n4 = rows * (width / 4);
  hipLaunchKernelGGL(s2g_k_dropout_mask_rng, dim3(cdiv(n4, 256)), dim3(256), 0, ST(stream), mask, n4, p,
                     1.0f / (1.0f - p), seed, offset, width / 4, live_cols);
  return LAUNCH_OK();
}

int s2g_gumbel_softmax_bwd(const float* y_soft, const float* dy, float* dlogits, long rows, int V, float scale,
                           void* stream) {
  if (!y_soft || !dy || !dlogits || rows <= 0 || V <= 0 || V > 64 * MAXPER) return -1;
  if (V <= 256)
    hipLaunchKernelGGL(s2g_k_gumbel_softmax_bwd_v4, dim3(cdiv(rows, 4 * SAMPLER_ROWS)), dim3(256), 0, ST(stream), y_soft, dy, dlogits,
                       rows, V, scale);
  else
    hipLaunchKernelGGL(s2g_k_gumbel_softmax_bwd, dim3(cdiv(rows, 4)), dim3(256), 0, ST(stream), y_soft, dy,
                       dlogits, rows, V, scale);
  return LAUNCH_OK();
}

int s2g_argmax_rows(const float* y, long long* tokens, long rows, int V, void* stream) {
  if (!y || !tokens || rows <= 0 || V <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_argmax_rows, dim3(cdiv(rows, 4)), dim3(256), 0, ST(stream), y, tokens, rows, V);
  return LAUNCH_OK();
}

int s2g_gemm_f32(const GemmArgs* args, int nsplit, void* stream) {
  if (!args || !args->A || !args->B || !args->C) return -1;
  if (args->segT < 0 || (args->shift && args->segT <= 0)) return -1;
  return s2g_launch_gemm(args, nsplit, ST(stream));
}

int s2g_gemm_tall(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
                  void* stream) {
  if (!A || !B || !C || M <= 0 || K <= 0 || N <= 160 || N > 224) return -1;
  if ((ldb & 3) || ldb < ((N + 3) & ~3) || (((uintptr_t)B) & 15) || lda < K || ldc < N) return -1;
  return s2g_launch_gemm_tall(A, lda, B, ldb, C, ldc, M, N, K, ST(stream));
}

int s2g_gemm_tall_bf16(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
                       void* stream) {
  if (!A || !B || !C || M <= 0 || K <= 0 || K > TB_KMAX || N <= 0 || N > 224) return -1;
  if (lda < K || ldb < N || ldc < N) return -1;
  return s2g_launch_gemm_tall_bf16(A, lda, B, ldb, C, ldc, M, N, K, ST(stream));
}

int s2g_linear_small_batch(const float* W, const float* x, const float* bias, float* y, int n_out, int batch, int K,
                           int max_workgroups, void* stream) {
  if (!W || !x || !y || n_out <= 0 || batch <= 0 || batch > 64 || K <= 0) return -1;
  if ((n_out & 15) || (K & 127) || K > 512) return -1;                      // x^T must fit in LDS: K * 272 bytes
  if ((((uintptr_t)W) | ((uintptr_t)y) | ((uintptr_t)bias)) & 15) return -1;
  if (max_workgroups < 0) return -1;
  return s2g_launch_linear_wstream(W, x, bias, y, n_out, batch, K, max_workgroups, ST(stream));
}

int s2g_linear_wgrad_adam(const float* dy, const float* a, float* W, float* m, float* v, float* grad_w, int n_out,
                          int batch, int K, float lr, float b1, float b2, float eps, int step, int max_workgroups,
                          void* stream) {
  if (!dy || !a || !W || !m || !v || n_out <= 0 || batch <= 0 || batch > 64 || K <= 0 || step < 1) return -1;
  if ((K & 255) || K > 512 || max_workgroups < 0) return -1;                // a workgroup owns 256 columns
  if ((((uintptr_t)W) | ((uintptr_t)m) | ((uintptr_t)v) | ((uintptr_t)grad_w)) & 15) return -1;
  const double bc1 = 1.0 - pow((double)b1, (double)step);                   // as s2g_adam
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  return s2g_launch_linear_wgrad_adam(dy, a, W, m, v, grad_w, n_out, batch, K, b1, b2, eps, (float)((double)lr / bc1),
                                      (float)(1.0 / sqrt(bc2)), max_workgroups, ST(stream));
}

int s2g_small_linear_grads(int n_jobs, const float* const* dy, const float* const* x, float* const* dW,
                           float* const* db, const int* n_out, const int* n_in, int batch, void* stream) {
  if (n_jobs < 1 || n_jobs > 4 || !dy || !x || !dW || !db || !n_out || !n_in || batch < 1 || batch > 4096) return -1;
  S2GSmallDw a;
  a.n_jobs = n_jobs;
  a.batch = batch;
  for (int j = 0; j < n_jobs; ++j) {
    if (!dy[j] || !x[j] || !dW[j] || n_out[j] < 1 || n_in[j] < 4 || (n_in[j] & 3)) return -1;
    if ((long)n_out[j] * n_in[j] > (1L << 26)) return -1;                   // small layers only
    if ((((uintptr_t)x[j]) | ((uintptr_t)dW[j])) & 15) return -1;
    a.dy[j] = dy[j]; a.x[j] = x[j]; a.dW[j] = dW[j]; a.db[j] = db[j];
    a.n_out[j] = n_out[j]; a.n_in[j] = n_in[j];
  }
  return s2g_launch_small_dw(&a, ST(stream));
}

int s2g_generate_tokens(const float* W, const float* x, const float* bias, const float* U, long long* tokens, int L,
                        int V, int batch, int K, unsigned long long seed, unsigned long long offset, void* stream) {
  if (!W || !x || !tokens || L <= 0 || V <= 0 || batch <= 0 || batch > 64 || K <= 0) return -1;
  if ((K & 127) || K > 512 || ((V + 15) >> 4) > GT_MAXTILES / 2) return -1;
  if ((((uintptr_t)W) | ((uintptr_t)x)) & 15) return -1;
  return s2g_launch_generate_tokens(W, x, bias, U, tokens, L, V, batch, K, seed, offset, ST(stream));
}

int s2g_gemm_grouped(const GemmArgs* table_dev, const int* prefix_dev, int nprob, int total_blocks, int nz,
                     void* stream) {
  if (!table_dev || !prefix_dev || nprob <= 0 || nprob > 64 || total_blocks <= 0 || nz < 0) return -1;
  if (nz > 0 && total_blocks % nz) return -1;
  return s2g_launch_gemm_grouped(table_dev, prefix_dev, nprob, total_blocks, nz, ST(stream));
}
int s2g_reduce_slabs_weighted(const float* slabs, long slab, int nslab, const float* weights, float* out, long n,
                              void* stream) {
  if (!slabs || !weights || !out || nslab <= 0 || n <= 0 || slab < n) return -1;
  return s2g_launch_reduce_slabs_weighted(slabs, slab, nslab, weights, out, n, ST(stream));
}
int s2g_gemm_grouped_bf16(const GemmArgs* table_dev, const int* prefix_dev, int nprob, int total_blocks, int nz,
                          void* stream) {
  if (!table_dev || !prefix_dev || nprob <= 0 || nprob > 64 || total_blocks <= 0 || nz < 0) return -1;
  if (nz > 0 && total_blocks % nz) return -1;
  return s2g_launch_gemm_grouped_bf16(table_dev, prefix_dev, nprob, total_blocks, nz, ST(stream));
}

int s2g_reduce_slabs(const float* slabs, long slab, int nslab, int rows, int cols, float* out, int permute_rows,
                     int accumulate, float scale, void* stream) {
  if (!slabs || !out || nslab <= 0 || rows <= 0 || cols <= 0) return -1;
  if (permute_rows && rows != S2G_G) return -1;
  hipLaunchKernelGGL(s2g_k_reduce_slabs, dim3(cdiv((long)rows * cols, 64)), dim3(256), 0, ST(stream), slabs,
                     slab, nslab, rows, cols, out, permute_rows, accumulate, scale);
  return LAUNCH_OK();
}

struct S2GReduce {      // mirrors include/s2g.h
  const float* slabs; long slab; float* out;
  int nslab, rows, cols;
};
int s2g_reduce_slabs_multi(const S2GReduce* jobs, int njobs, int permute_rows, int accumulate, float scale,
                           void* stream) {
  if (!jobs || njobs <= 0 || njobs > S2G_MAX_REDUCE_JOBS) return -1;
  ReduceJobs a;
  int blocks = 0;
  for (int k = 0; k < njobs; ++k) {
    const S2GReduce& q = jobs[k];
    if (!q.slabs || !q.out || q.nslab <= 0 || q.rows <= 0 || q.cols <= 0) return -1;
    if (permute_rows && q.rows != S2G_G) return -1;
    a.j[k] = ReduceJob{q.slabs, q.slab, q.out, q.nslab, q.rows, q.cols, blocks};
    blocks += (int)cdiv((long)q.rows * q.cols, 64);
  }
  a.njobs = njobs;
  hipLaunchKernelGGL(s2g_k_reduce_slabs_multi, dim3(blocks), dim3(256), 0, ST(stream), a, permute_rows, accumulate,
                     scale);
  return LAUNCH_OK();
}
struct S2GCriticUpdate {
  S2GReduce job[4];
  long poff[4];
  float *p, *g, *m, *v;
  long o_bih0, o_bhh0, o_bih1, o_bhh1, o_fcw, o_fcb;
  int n_fcw, n_fcb, V, Vp;
  float *WT0, *W0p, *b0, *b1;
  float lr, b1c, b2c, eps;
  int step;
};
int s2g_critic_update(const S2GCriticUpdate* u, void* stream) {
  if (!u || !u->p || !u->g || !u->m || !u->v || !u->WT0 || !u->W0p || !u->b0 || !u->b1 || u->step < 1) return -1;
  if (u->V <= 0 || u->Vp < u->V || u->n_fcw < 0 || u->n_fcb < 0) return -1;
  CriticUpdate a;
  int blocks = 0;
  for (int k = 0; k < 4; ++k) {
    const S2GReduce& q = u->job[k];
    if (!q.slabs || !q.out || q.nslab <= 0 || q.rows != S2G_G || q.cols <= 0 || u->poff[k] < 0) return -1;
    a.j[k] = ReduceJob{q.slabs, q.slab, q.out, q.nslab, q.rows, q.cols, blocks};
    a.poff[k] = u->poff[k];
    blocks += (int)cdiv((long)q.rows * q.cols, 64);
  }
  if (u->job[0].cols != u->V) return -1;
  a.p = u->p; a.g = u->g; a.m = u->m; a.v = u->v;
  a.o_bih0 = u->o_bih0; a.o_bhh0 = u->o_bhh0; a.o_bih1 = u->o_bih1; a.o_bhh1 = u->o_bhh1;
  a.o_fcw = u->o_fcw; a.o_fcb = u->o_fcb; a.n_fcw = u->n_fcw; a.n_fcb = u->n_fcb;
  a.tail_block = blocks; a.V = u->V; a.Vp = u->Vp;
  a.WT0 = u->WT0; a.W0p = u->W0p; a.b0 = u->b0; a.b1 = u->b1;
  const double bc1 = 1.0 - pow((double)u->b1c, (double)u->step);
  const double bc2 = 1.0 - pow((double)u->b2c, (double)u->step);
  a.b1c = u->b1c; a.b2c = u->b2c; a.eps = u->eps;
  a.step_size = (float)((double)u->lr / bc1);
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  blocks += (int)cdiv(2 * S2G_G + u->n_fcw + u->n_fcb, 256);
  return s2g_launch_critic_update(&a, blocks, ST(stream));
}
int s2g_colsum(const float* X, long ld, long rows, int cols, float* part, long rows_per_chunk, void* stream) {
  if (!X || !part || rows <= 0 || cols <= 0 || rows_per_chunk <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_colsum, dim3(cdiv(cols, 64), cdiv(rows, rows_per_chunk)), dim3(256), 0, ST(stream), X,
                     ld, rows, cols, part, rows_per_chunk);
  return LAUNCH_OK();
}
int s2g_bias_sums(const float* X0, const float* X1, long rows, float* part, const float* hdotT, int B, float* bih0,
                  float* bhh0, float* bih1, float* bhh1, float* fcw0, int accumulate, void* stream) {
  if (!X0 || !X1 || !part || rows <= 0 || !bih0 || !bhh0 || !bih1 || !bhh1) return -1;
  if (hdotT && (!fcw0 || B <= 0)) return -1;
  const int nchunk = (int)cdiv(rows, (long)BIAS_CHUNK);
  hipLaunchKernelGGL(s2g_k_bias_sums_part, dim3(cdiv(S2G_G, 64), nchunk, 2), dim3(256), 0, ST(stream), X0, X1, rows,
                     part);
  hipLaunchKernelGGL(s2g_k_bias_sums_final, dim3(8), dim3(256), 0, ST(stream), part, nchunk, (long)nchunk * S2G_G,
                     hdotT, B, bih0, bhh0, bih1, bhh1, fcw0, accumulate);
  return LAUNCH_OK();
}
// s2g_bias_sums for callers that cannot allocate (HIP graph capture, serving loops): the partial sums
// live in a file-scope device array instead of a caller-provided workspace.  Every GPU holds its own
// copy of the array at its own address, looked up once per device (the array's name is only a
// host-side handle for that lookup, never a device address).
#define S2G_NOALLOC_MAX_ROWS 262144L
static __device__ __align__(256) float s2g_noalloc_scratch[2 * (S2G_NOALLOC_MAX_ROWS / BIAS_CHUNK) * S2G_G];
#define S2G_MAX_DEVICES 64
static float* s2g_noalloc_ptr[S2G_MAX_DEVICES];
static std::mutex s2g_noalloc_lock;

// device address of the scratch array on the current device (nullptr + error code on failure)
static int s2g_noalloc_scratch_ptr(float** out) {
  int dev = -1;
  S2G_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= S2G_MAX_DEVICES) return -1;
  std::lock_guard<std::mutex> guard(s2g_noalloc_lock);
  if (!s2g_noalloc_ptr[dev]) {
    float* p = nullptr;
    S2G_CHECK(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(s2g_noalloc_scratch)));
    if (!p) return -2;
    s2g_noalloc_ptr[dev] = p;
  }
  *out = s2g_noalloc_ptr[dev];
  return 0;
}

int s2g_bias_sums_noalloc(const float* X0, const float* X1, long rows, const float* hdotT, int B, float* bih0,
                          float* bhh0, float* bih1, float* bhh1, float* fcw0, int accumulate, void* stream) {
  if (!X0 || !X1 || rows <= 0 || !bih0 || !bhh0 || !bih1 || !bhh1) return -1;
  if (hdotT && (!fcw0 || B <= 0)) return -1;
  if (rows > S2G_NOALLOC_MAX_ROWS) return -3;          // beyond the static scratch: use s2g_bias_sums
  float* part = nullptr;
  const int rc = s2g_noalloc_scratch_ptr(&part);
  if (rc != 0) return rc;
  // pass 1 writes every partial row pass 2 reads (nchunk rows per layer), so nothing of the array has to
  // start at zero; what the last call left beyond them is never read
  const int nchunk = (int)cdiv(rows, (long)BIAS_CHUNK);
  hipLaunchKernelGGL(s2g_k_bias_sums_part, dim3(cdiv(S2G_G, 64), nchunk, 2), dim3(256), 0, ST(stream), X0, X1, rows,
                     part);
  hipLaunchKernelGGL(s2g_k_bias_sums_final, dim3(8), dim3(256), 0, ST(stream), part, nchunk, (long)nchunk * S2G_G,
                     hdotT, B, bih0, bhh0, bih1, bhh1, fcw0, accumulate);
  return LAUNCH_OK();
}

int s2g_bias_sums_final(const float* sums, int nseq, long layer_stride, const float* hdotT, int B, float* bih0,
                        float* bhh0, float* bih1, float* bhh1, float* fcw0, int accumulate, void* stream) {
  if (!sums || nseq <= 0 || layer_stride < (long)nseq * S2G_G || !bih0 || !bhh0 || !bih1 || !bhh1) return -1;
  if (hdotT && (!fcw0 || B <= 0)) return -1;
  hipLaunchKernelGGL(s2g_k_bias_sums_final, dim3(8), dim3(256), 0, ST(stream), sums, nseq, layer_stride, hdotT, B,
                     bih0, bhh0, bih1, bhh1, fcw0, accumulate);
  return LAUNCH_OK();
}

int s2g_lstm2_fwd(const S2GSeq* seqs, int nseq, int max_len, int vocab, const int* tokens, const float* WT0, const float* P,
                  const float* mask, const float* b0, const float* b1, const float* whh0, const float* wih1,
                  const float* whh1, float* gates0, float* cell0, float* gates1, float* cell1, float* hT,
                  void* stream) {
  if (!seqs || nseq <= 0 || max_len < 0 || max_len > 16384) return -1;
  if (tokens && (vocab <= 0 || !WT0)) return -1;
  if (!b0 || !b1 || !whh0 || !wih1 || !whh1 || !gates0 || !cell0 || !gates1 || !cell1 || !hT) return -1;
  size_t lds = FWD_LDS_FLOATS * sizeof(float) + (size_t)max_len * sizeof(int);
  hipLaunchKernelGGL(s2g_k_lstm2_fwd, dim3(nseq), dim3(FWD_THREADS), lds, ST(stream), seqs, tokens, WT0, P, mask, b0, b1,
                     whh0, wih1, whh1, gates0, cell0, gates1, cell1, hT, vocab > 0 ? vocab : 1, max_len);
  return LAUNCH_OK();
}

int s2g_lstm2_bwd(const S2GSeq* seqs, int nseq, int rows_per_seq, const float* mask, const float* whh0,
                  const float* wih1, const float* whh1, const float* gates0, const float* cell0,
                  const float* gates1, const float* cell1, const float* dhT, float* delta0, float* delta1,
                  float* adj0, float* adj1, int adj_seq0, int adj_row0, const S2GHead* head, float* bsum,
                  void* stream) {
  if (!seqs || nseq <= 0 || rows_per_seq <= 0) return -1;
  if (!whh0 || !wih1 || !whh1 || !gates0 || !cell0 || !gates1 || !cell1 || !delta0 || !delta1) return -1;
  if ((adj0 == nullptr) != (adj1 == nullptr)) return -1;
  if (adj0 && (adj_seq0 < 0 || adj_row0 < 0)) return -1;
  S2GHead h = {};
  if (head) {
    h = *head;
    if (!h.hT || !h.W || !h.bias || !h.labels || h.B <= 0 || (h.mode != 0 && h.mode != 1)) return -1;
    if (h.mode == 0 ? nseq != 3 * h.B : nseq != h.B) return -1;
  } else if (!dhT) {
    return -1;
  }
  hipLaunchKernelGGL(s2g_k_lstm2_bwd, dim3(nseq), dim3(BWD_THREADS), 0, ST(stream), seqs, mask, whh0, wih1, whh1, gates0,
                     cell0, gates1, cell1, dhT, delta0, delta1, adj0, adj1, rows_per_seq, adj_seq0, adj_row0, h, bsum);
  return LAUNCH_OK();
}

int s2g_lstm2_tangent(const S2GSeq* seqs, int nseq, const float* Xdot, const float* mask, const float* whh0,
                      const float* wih1, const float* whh1, const float* gates0, const float* cell0,
                      const float* gates1, const float* cell1, float* adot0, float* celld0, float* adot1,
                      float* celld1, float* hdotT, const float* gp_partial, float gp_upstream, float gp_n_batch,
                      void* stream) {
  if (!seqs || nseq <= 0 || !Xdot) return -1;
  if (!whh0 || !wih1 || !whh1 || !gates0 || !cell0 || !gates1 || !cell1) return -1;
  if (!adot0 || !celld0 || !adot1 || !celld1 || !hdotT) return -1;
  if (gp_partial && !(gp_n_batch > 0.0f)) return -1;
  hipLaunchKernelGGL(s2g_k_lstm2_tangent, dim3(nseq), dim3(FWD_THREADS), 0, ST(stream), seqs, Xdot, mask, whh0, wih1,
                     whh1, gates0, cell0, gates1, cell1, adot0, celld0, adot1, celld1, hdotT, gp_partial, gp_upstream,
                     gp_n_batch);
  return LAUNCH_OK();
}
int s2g_gp_dots(const float* delta, const float* pv, int B, long per_sample, float* partial, void* stream) {
  if (!delta || !pv || !partial || B <= 0 || per_sample <= 0 || (per_sample & 3)) return -1;
  if (((uintptr_t)delta | (uintptr_t)pv) & 15) return -1;
  hipLaunchKernelGGL(s2g_k_gp_dots, dim3(B, 4), dim3(1024), 0, ST(stream), delta, pv, per_sample, partial);
  return LAUNCH_OK();
}
int s2g_gp_penalty_final(const float* partial, int B, float upstream, float n_batch, float* pen, float* norms,
                         float* coef, void* stream) {
  if (!partial || !pen || B <= 0 || (coef && !(n_batch > 0.0f))) return -1;
  hipLaunchKernelGGL(s2g_k_gp_penalty_final, dim3(cdiv(B, 64)), dim3(64), 0, ST(stream), partial, B, upstream, n_batch,
                     pen, norms, coef);
  return LAUNCH_OK();
}

int s2g_lstm2_bwd2(const S2GSeq* seqs, int nseq, int rows_per_seq, const float* mask, const float* whh0,
                   const float* wih1, const float* whh1, const float* gates0, const float* cell0,
                   const float* gates1, const float* cell1, const float* adot0, const float* celld0,
                   const float* adot1, const float* celld1, const float* adj0, const float* adj1, float* abar0,
                   float* abar1, float* bsum, void* stream) {
  if (!seqs || nseq <= 0 || rows_per_seq <= 0) return -1;
  if (!whh0 || !wih1 || !whh1 || !gates0 || !cell0 || !gates1 || !cell1) return -1;
  if (!adot0 || !celld0 || !adot1 || !celld1 || !adj0 || !adj1 || !abar0 || !abar1) return -1;
  hipLaunchKernelGGL(s2g_k_lstm2_bwd2, dim3(nseq), dim3(BWD_THREADS), 0, ST(stream), seqs, mask, whh0, wih1, whh1, gates0,
                     cell0, gates1, cell1, adot0, celld0, adot1, celld1, adj0, adj1, abar0, abar1, rows_per_seq, bsum);
  return LAUNCH_OK();
}

int s2g_pack_critic(const float* wih0, const float* bih0, const float* bhh0, const float* bih1, const float* bhh1,
                    float* WT0, float* W0p, float* b0, float* b1, int V, void* stream) {
  if (!wih0 || !bih0 || !bhh0 || !bih1 || !bhh1 || !WT0 || !W0p || !b0 || !b1 || V <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_pack_critic, dim3(cdiv((long)S2G_G * V, 256)), dim3(256), 0, ST(stream), wih0, bih0,
                     bhh0, bih1, bhh1, WT0, W0p, b0, b1, V);
  return LAUNCH_OK();
}

int s2g_critic_batch_desc(const int* lengths, const float* alpha, int B, int T, int Tf, int use_mask, S2GSeq* seqs,
                          void* stream) {
  if (!lengths || !alpha || !seqs || B <= 0 || T <= 0 || Tf <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_critic_batch_desc, dim3(cdiv(4 * B, 256)), dim3(256), 0, ST(stream), lengths, alpha, B,
                     T, Tf, use_mask, seqs);
  return LAUNCH_OK();
}

int s2g_head_fwd(const float* hT, const float* W, const float* b, float* out, int N, int J, void* stream) {
  if (!hT || !W || !b || !out || N <= 0 || J <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_head_fwd, dim3(cdiv((long)N * J, 256)), dim3(256), 0, ST(stream), hT, W, b, out, N, J);
  return LAUNCH_OK();
}
int s2g_head_bwd_dh(const float* dout, const float* W, float* dhT, int N, int J, void* stream) {
  if (!dout || !W || !dhT || N <= 0 || J <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_head_bwd_dh, dim3(cdiv((long)N * S2G_H, 256)), dim3(256), 0, ST(stream), dout, W, dhT,
                     N, J);
  return LAUNCH_OK();
}
int s2g_head_bwd_w(const float* dout, const float* hT, float* dW, float* db, int N, int J, int accumulate,
                   void* stream) {
  if (!dout || !hT || !dW || !db || N <= 0 || J <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_head_bwd_w, dim3(cdiv((long)J * S2G_H, 64)), dim3(64), 0, ST(stream), dout, hT, dW, db,
                     N, J, accumulate);
  return LAUNCH_OK();
}
int s2g_critic_loss(const float* out, const long long* labels, int B, float inv_n, float inv_wsum, float w0,
                    float w1, float* dout, float* losses, void* stream) {
  if (!out || !labels || !dout || !losses || B <= 0 || B > 1024) return -1;
  hipLaunchKernelGGL(s2g_k_critic_loss, dim3(1), dim3(cdiv(B, 64) * 64), 0, ST(stream), out, labels, B, inv_n,
                     inv_wsum, w0, w1, dout, losses);
  return LAUNCH_OK();
}
int s2g_critic_head(const float* hT, const float* W, const float* bias, const long long* labels, int B, float inv_n,
                    float inv_wsum, float w0, float w1, float* out, float* dout, float* losses, float* dhT,
                    float* dW, float* db, void* stream) {
  if (!hT || !W || !bias || !labels || !out || !dout || !losses || !dhT || !dW || !db || B <= 0 || B > 1024)
    return -1;
  const size_t lds = (size_t)(18 * B + 3 * S2G_H) * sizeof(float);     // out, d(out) and the fc weight mirrored in LDS
  static S2GLdsCap lds_cap;                                                // B = 1024 needs 74 KB: above the default cap
  if (s2g_lds_cap(lds_cap, (const void*)s2g_k_critic_head, 80 * 1024) != 0) return -2;
  hipLaunchKernelGGL(s2g_k_critic_head, dim3(1), dim3(1024), lds, ST(stream), hT, W, bias, labels, B, inv_n, inv_wsum,
                     w0, w1, out, dout, losses, dhT, dW, db);
  return LAUNCH_OK();
}
int s2g_generator_loss(const float* out, const long long* labels, int B, float inv_n, float inv_wsum, float w0,
                       float w1, float* dout, float* losses, void* stream) {
  if (!out || !labels || !dout || !losses || B <= 0 || B > 1024) return -1;
  hipLaunchKernelGGL(s2g_k_generator_loss, dim3(1), dim3(cdiv(B, 64) * 64), 0, ST(stream), out, labels, B, inv_n,
                     inv_wsum, w0, w1, dout, losses);
  return LAUNCH_OK();
}

int s2g_one_hot(const int* tokens, const int* lengths, int T, float* x, long rows, int V, void* stream) {
  if (!tokens || !x || rows <= 0 || V <= 0 || (lengths && T <= 0) || rows * V >= (1L << 31)) return -1;
  if (((uintptr_t)x) & 15) return -1;
  hipLaunchKernelGGL(s2g_k_one_hot, dim3(cdiv(cdiv(rows * V, 4), 256)), dim3(256), 0, ST(stream), tokens, lengths, T, x,
                     rows, V);
  return LAUNCH_OK();
}
int s2g_gp_interp(const int* tokens, const int* lengths, const float* fake, const float* alpha, int B, int T,
                  int Tf, int V, float* xhat, void* stream) {
  if (!tokens || !lengths || !fake || !alpha || !xhat || B <= 0 || T <= 0 || Tf <= 0 || V <= 0) return -1;
  if ((long)B * T * V >= (1L << 31)) return -1;
  if ((((uintptr_t)xhat) | ((uintptr_t)fake)) & 15) return -1;
  hipLaunchKernelGGL(s2g_k_gp_interp, dim3(cdiv(cdiv((long)B * T * V, 4), 256)), dim3(256), 0, ST(stream), tokens,
                     lengths, fake, alpha, B, T, Tf, V, xhat);
  return LAUNCH_OK();
}
int s2g_gp_norm_penalty(float* g, int B, long per_sample, float upstream, float n_batch, float* pen, float* norms,
                        int scale_in_place, void* stream) {
  if (!g || !pen || B <= 0 || per_sample <= 0 || !(n_batch > 0.0f)) return -1;
  if (((uintptr_t)g) & 15) return -1;
  hipLaunchKernelGGL(s2g_k_gp_norm_penalty, dim3(B), dim3(1024), 0, ST(stream), g, per_sample, upstream, n_batch,
                     pen, norms, scale_in_place);
  return LAUNCH_OK();
}
int s2g_gp_norm_penalty_proj(const float* delta, float* pv, int B, long per_sample, float upstream, float n_batch,
                             float* pen, float* norms, void* stream) {
  if (!delta || !pv || !pen || B <= 0 || per_sample <= 0 || (per_sample & 3) || !(n_batch > 0.0f)) return -1;
  if ((((uintptr_t)delta) | ((uintptr_t)pv)) & 15) return -1;
  hipLaunchKernelGGL(s2g_k_gp_norm_penalty_proj, dim3(B), dim3(1024), 0, ST(stream), delta, pv, per_sample, upstream,
                     n_batch, pen, norms);
  return LAUNCH_OK();
}
int s2g_gp_scale(const float* g, const float* norms, const float* upstream, float n_batch, long per_sample,
                 float* v, int B, void* stream) {
  if (!g || !norms || !upstream || !v || B <= 0 || per_sample <= 0) return -1;
  long total = (long)B * per_sample;
  hipLaunchKernelGGL(s2g_k_gp_scale, dim3(cdiv(total, 256)), dim3(256), 0, ST(stream), g, norms, upstream, n_batch,
                     per_sample, v, total);
  return LAUNCH_OK();
}

int s2g_embed_mul(const float* z, const long long* labels, const float* emb, float* out, int B, int D,
                  int n_classes, void* stream) {
  if (!z || !labels || !emb || !out || B <= 0 || D <= 0 || n_classes <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_embed_mul, dim3(cdiv((long)B * D, 256)), dim3(256), 0, ST(stream), z, labels, emb, out,
                     B, D, n_classes);
  return LAUNCH_OK();
}
int s2g_embed_mul_bwd(const float* z, const long long* labels, const float* dout, float* demb, int B, int D,
                      int n_classes, void* stream) {
  if (!z || !labels || !dout || !demb || B <= 0 || D <= 0 || n_classes <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_embed_mul_bwd, dim3(cdiv((long)n_classes * D * 16, 256)), dim3(256), 0, ST(stream), z,
                     labels, dout, demb, B, D, n_classes);
  return LAUNCH_OK();
}
int s2g_lrelu_bwd(const float* y, float* dy, long n, void* stream) {
  if (!y || !dy || n <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_lrelu_bwd, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), y, dy, n);
  return LAUNCH_OK();
}
int s2g_relu_bwd(const float* y, float* dy, long n, void* stream) {
  if (!y || !dy || n <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_relu_bwd, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), y, dy, n);
  return LAUNCH_OK();
}
int s2g_add_layernorm_fwd(const float* x, const float* r, const float* rmask, const float* gamma, const float* beta,
                          float eps, float* y, float* xhat, float* rstd, long rows, int E, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || E <= 0 || E > 64 * LN_MAXPER || !(eps >= 0.0f)) return -1;
  if (rmask && !r) return -1;
  hipLaunchKernelGGL(s2g_k_add_layernorm_fwd, dim3(cdiv(rows, 4)), dim3(256), 0, ST(stream), x, r, rmask, gamma, beta,
                     eps, y, xhat, rstd, rows, E);
  return LAUNCH_OK();
}
int s2g_add_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dz,
                          float* dyxh, long rows, int E, void* stream) {
  if (!dy || !xhat || !rstd || !gamma || !dz || !dyxh || rows <= 0 || E <= 0 || E > 64 * LN_MAXPER) return -1;
  hipLaunchKernelGGL(s2g_k_add_layernorm_bwd, dim3(cdiv(rows, 4)), dim3(256), 0, ST(stream), dy, xhat, rstd, gamma, dz,
                     dyxh, rows, E);
  return LAUNCH_OK();
}
int s2g_dropout_mask(float* u, long rows, float p, void* stream) {
  if (!u || rows <= 0 || !(p >= 0.0f) || !(p < 1.0f) || ((uintptr_t)u & 15)) return -1;
  const long n4 = rows * (S2G_MASKW / 4);
  hipLaunchKernelGGL(s2g_k_dropout_mask, dim3(cdiv(n4, 256)), dim3(256), 0, ST(stream), u, n4, p, 1.0f / (1.0f - p));
  return LAUNCH_OK();
}
int s2g_bn_stats(const float* x, float* stats, int B, int F, void* stream) {
  if (!x || !stats || B <= 0 || F <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_bn_stats, dim3(cdiv(F, 64)), dim3(64), 0, ST(stream), x, stats, B, F);
  return LAUNCH_OK();
}
int s2g_bn_lrelu_apply(const float* x, const float* stats, float n_global, const float* gamma, const float* beta,
                       float* run_mean, float* run_var, float eps, float momentum, int update_running, float* y,
                       float* xhat, float* invstd, int B, int F, void* stream) {
  if (!x || !gamma || !beta || !run_mean || !run_var || !y || B <= 0 || F <= 0) return -1;
  if (stats && !(n_global > 0.0f)) return -1;
  hipLaunchKernelGGL(s2g_k_bn_lrelu_apply, dim3(cdiv(F, 64)), dim3(64), 0, ST(stream), x, stats, n_global, gamma,
                     beta, run_mean, run_var, eps, momentum, update_running, y, xhat, invstd, B, F);
  return LAUNCH_OK();
}
int s2g_bn_lrelu_bwd_stats(const float* y, float* dy, const float* xhat, float* sums, int B, int F, void* stream) {
  if (!y || !dy || !xhat || !sums || B <= 0 || F <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_bn_lrelu_bwd_stats, dim3(cdiv(F, 64)), dim3(64), 0, ST(stream), y, dy, xhat, sums, B, F);
  return LAUNCH_OK();
}
int s2g_bn_bwd_apply(const float* dyh, const float* xhat, const float* sums, float n_global, const float* gamma,
                     const float* invstd, float* dx, int B, int F, void* stream) {
  if (!dyh || !xhat || !sums || !gamma || !invstd || !dx || B <= 0 || F <= 0 || !(n_global > 0.0f)) return -1;
  hipLaunchKernelGGL(s2g_k_bn_bwd_apply, dim3(cdiv(F, 64)), dim3(64), 0, ST(stream), dyh, xhat, sums, n_global,
                     gamma, invstd, dx, B, F);
  return LAUNCH_OK();
}
int s2g_bn_lrelu_fwd_fused(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var,
                           float eps, float momentum, int update_running, float* y, float* xhat, float* invstd,
                           int B, int F, void* stream) {
  if (!x || !gamma || !beta || !run_mean || !run_var || !y || !xhat || !invstd || B <= 0 || F <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_bn_lrelu_fwd_fused, dim3(cdiv(F, BN_FX)), dim3(BN_FX * BN_RL), 0, ST(stream), x, gamma,
                     beta, run_mean, run_var, eps, momentum, update_running, y, xhat, invstd, B, F);
  return LAUNCH_OK();
}
int s2g_bn_lrelu_bwd_fused(const float* y, const float* dy, const float* xhat, const float* gamma,
                           const float* invstd, float* dgamma, float* dbeta, float* dx, int B, int F,
                           void* stream) {
  if (!y || !dy || !xhat || !gamma || !invstd || !dgamma || !dbeta || !dx || B <= 0 || F <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_bn_lrelu_bwd_fused, dim3(cdiv(F, BN_FX)), dim3(BN_FX * BN_RL), 0, ST(stream), y, dy, xhat,
                     gamma, invstd, dgamma, dbeta, dx, B, F);
  return LAUNCH_OK();
}

int s2g_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
             int step, float gscale, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) return -1;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return -1;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  if (n >= (1L << 40)) return -1;
  const unsigned blocks = cdiv(n, 1024);        // one 16-byte group per thread (the loop in the kernel runs once)
  hipLaunchKernelGGL(s2g_k_adam, dim3(blocks), dim3(256), 0, ST(stream), p, g, m, v, n, b1, b2, eps, step_size,
                     inv_sqrt_bc2, gscale);
  return LAUNCH_OK();
}

// ---- weight-clipping WGAN variant (reference wgan.py): hidden size 64
int s2g_lstm64_fwd(const float* pin, const float* whh, const int* lengths, int nseq, int T, float* gates, float* cst,
                   float* hout, const float* out_mask, float* hdrop, void* stream) {
  if (!pin || !whh || !gates || !cst || !hout || nseq <= 0 || T <= 0) return -1;
  if (out_mask && !hdrop) return -1;
  hipLaunchKernelGGL(s2g_k_lstm64_fwd, dim3(nseq), dim3(X_G), 0, ST(stream), pin, whh, lengths, T, gates, cst, hout,
                     out_mask, hdrop);
  return LAUNCH_OK();
}
int s2g_lstm64_stack_fwd(const S2GLstm64Stack* st, int layers, const int* lengths, int nseq, int T, void* stream) {
  if (!st || (layers != 2 && layers != 3) || nseq <= 0 || T <= 0) return -1;
  const bool tok = st->tokens != nullptr;
  if (tok ? (!st->tab || !st->labels || !st->tabl || st->V <= 0 || st->n_classes <= 0) : !st->pin0) return -1;
  for (int l = 0; l < layers; ++l) {
    if (!st->whh[l]) return -1;
    if (l > 0 && (!st->wih[l] || !st->bsum[l])) return -1;
    if (st->mask[l] && (l + 1 >= layers || !st->hdrop[l])) return -1;
  }
  if (!st->hout[layers - 1]) return -1;
  if (layers == 2 && tok)
    hipLaunchKernelGGL((s2g_k_lstm64_stack_fwd<2, true, true>), dim3(nseq), dim3(X_G * 3), 0, ST(stream), *st, lengths, T);
  else if (layers == 2)
    hipLaunchKernelGGL((s2g_k_lstm64_stack_fwd<2, false, true>), dim3(nseq), dim3(X_G * 3), 0, ST(stream), *st, lengths, T);
  else if (tok)
    hipLaunchKernelGGL((s2g_k_lstm64_stack_fwd<3, true, false>), dim3(nseq), dim3(X_G * 3), 0, ST(stream), *st, lengths, T);
  else
    hipLaunchKernelGGL((s2g_k_lstm64_stack_fwd<3, false, false>), dim3(nseq), dim3(X_G * 3), 0, ST(stream), *st, lengths, T);
  return LAUNCH_OK();
}
int s2g_lstm64_bwd(const float* dh_in, const float* whh, const int* lengths, int nseq, int T, const float* gates,
                   const float* cst, float* delta, void* stream) {
  if (!dh_in || !whh || !gates || !cst || !delta || nseq <= 0 || T <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_lstm64_bwd, dim3(nseq), dim3(X_G), 0, ST(stream), dh_in, whh, lengths, T, gates, cst, delta);
  return LAUNCH_OK();
}
int s2g_lstm64_stack_bwd(const S2GLstm64StackBwd* st, const int* lengths, int nseq, int T, void* stream) {
  if (!st || nseq <= 0 || T <= 0 || !st->dh_top || !st->wih1) return -1;
  for (int l = 0; l < 2; ++l)
    if (!st->whh[l] || !st->gates[l] || !st->cst[l] || !st->delta[l]) return -1;
  hipLaunchKernelGGL(s2g_k_lstm64_stack_bwd, dim3(nseq), dim3(3 * X_G), 0, ST(stream), *st, lengths, T);
  return LAUNCH_OK();
}
int s2g_embed_concat(const int* tokens, const float* emb, int V, const long long* labels, const float* lab_emb,
                     int n_classes, int T, long rows, int soft, float* x, void* stream) {
  if (!emb || !labels || !lab_emb || !x || V <= 0 || n_classes <= 0 || T <= 0 || rows <= 0) return -1;
  if (!soft && !tokens) return -1;
  hipLaunchKernelGGL(s2g_k_embed_concat, dim3(cdiv(rows * 64, 256)), dim3(256), 0, ST(stream), tokens, emb, V, labels,
                     lab_emb, n_classes, T, rows, soft, x);
  return LAUNCH_OK();
}
long s2g_embed_grad_tokens_scratch(long rows, int V) {
  if (rows <= 0 || V <= 0) return -1;
  return ((rows + EG_CHUNK - 1) / EG_CHUNK) * (long)V * 32;
}
static int bucket_rows_by_token(const int* tokens, const int* lengths, int T, long rows, int V, const float* x, int ld,
                                int width, float* out, int ldo, int accumulate, float* part, void* stream,
                                int x_seg = 0, int transposed = 0);
int s2g_embed_grad_tokens(const int* tokens, const int* lengths, int T, long rows, int V, const float* dx, float* demb,
                          int accumulate, float* part, void* stream) {
  return bucket_rows_by_token(tokens, lengths, T, rows, V, dx, 64, 32, demb, 32, accumulate, part, stream);
}
int s2g_bucket_rows_by_token(const int* tokens, const int* lengths, int T, long rows, int V, const float* x, int ld,
                             int width, float* out, float* part, void* stream) {
  return bucket_rows_by_token(tokens, lengths, T, rows, V, x, ld, width, out, width, 0, part, stream);
}
int s2g_bucket_rows_by_token_t(const int* tokens, const int* lengths, int T, long rows, int V, const float* x, int ld,
                               int x_seg, int width, float* out, int ldo, float* part, void* stream) {
  if (x_seg < T || ldo < V) return -1;
  return bucket_rows_by_token(tokens, lengths, T, rows, V, x, ld, width, out, ldo, 0, part, stream, x_seg, 1);
}
int s2g_bucket_rows_by_label(const long long* labels, int B, int n_classes, const float* x, int width, float* out,
                             void* stream) {
  if (!labels || !x || !out || B <= 0 || n_classes <= 0 || width <= 0 || width > 1024) return -1;
  hipLaunchKernelGGL(s2g_k_bucket_rows_by_label, dim3(n_classes), dim3(width), 0, ST(stream), labels, B, x, width, out);
  return LAUNCH_OK();
}
#ifndef S2G_BUCKET_BLOCKS
#define S2G_BUCKET_BLOCKS 256
#endif
static int bucket_rows_by_token(const int* tokens, const int* lengths, int T, long rows, int V, const float* dx, int ld,
                                int width, float* demb, int ldo, int accumulate, float* part, void* stream, int x_seg,
                                int transposed) {
  if (!tokens || !lengths || !dx || !demb || !part || T <= 0 || rows <= 0 || V <= 0) return -1;
  if (rows % T) return -1;                                         // rows = sequences x T: lengths[] is read per sequence
  if (width <= 0 || ld < width || (!transposed && (width % 32 || ldo < width))) return -1;
  if (x_seg <= 0) x_seg = T;
  const size_t lds = (size_t)EG_GROUPS * V * 32 * sizeof(float);   // four tables of V x 32 floats in LDS
  if (lds > 160 * 1024) return -1;                                 // V <= 320
  static S2GLdsCap lds_cap;
  if (s2g_lds_cap(lds_cap, (const void*)s2g_k_embed_grad_tokens_part, 160 * 1024) != 0) return -2;
  const int nsl = (width + 31) / 32;
  // rows per workgroup: about one workgroup per CU over all column slices (a workgroup zeroes and flushes whole
  // tables, so short chunks of wide matrices are all overhead); never below EG_CHUNK, the unit the scratch is sized by
  long chunk = ((rows * nsl / S2G_BUCKET_BLOCKS + EG_CHUNK - 1) / EG_CHUNK) * EG_CHUNK;
  chunk = chunk < EG_CHUNK ? EG_CHUNK : (chunk > 16 * EG_CHUNK ? 16 * EG_CHUNK : chunk);
  const int nchunks = (int)((rows + chunk - 1) / chunk);
  hipLaunchKernelGGL(s2g_k_embed_grad_tokens_part, dim3(nchunks, nsl), dim3(32 * EG_GROUPS), lds, ST(stream), tokens,
                     lengths, T, rows, V, dx, ld, (int)chunk, part, width, x_seg);
  if (transposed)
    hipLaunchKernelGGL(s2g_k_embed_grad_final_t, dim3(cdiv((long)V * 32, 256), nsl), dim3(256), 0, ST(stream), part,
                       nchunks, V * 32, width, demb, ldo);
  else
    hipLaunchKernelGGL(s2g_k_embed_grad_final, dim3(cdiv((long)V * 32, 256), nsl), dim3(256), 0, ST(stream), part,
                       nchunks, V * 32, demb, ldo, accumulate);
  return LAUNCH_OK();
}
int s2g_embed_grad_labels(const long long* labels, const int* lengths, int T, int B, int n_classes, const float* dx,
                          float* dlab, int accumulate, float* part, void* stream) {
  if (!labels || !dx || !dlab || !part || T <= 0 || B <= 0 || n_classes <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_embed_grad_labels_part, dim3(B), dim3(256), 0, ST(stream), lengths, T, dx, part);
  hipLaunchKernelGGL(s2g_k_embed_grad_labels, dim3(n_classes), dim3(32), 0, ST(stream), labels, B, part, dlab,
                     accumulate);
  return LAUNCH_OK();
}
int s2g_attnpool_fwd(float* pre, const float* out, const float* wc, const float* wl, const float* bl, int nseq, int T,
                     int Tp, float* weights, float* pooled, float* dval, void* stream) {
  if (!pre || !out || !wc || !wl || !bl || !weights || !pooled || !dval || nseq <= 0 || T <= 0 || Tp <= 0 || Tp > T ||
      Tp > 12288)
    return -1;
  hipLaunchKernelGGL(s2g_k_attnpool_fwd, dim3(nseq), dim3(64 * AP_NW), (size_t)Tp * sizeof(float), ST(stream), pre, out, wc, wl,
                     bl, T, Tp, weights, pooled, dval);
  return LAUNCH_OK();
}
int s2g_attnpool_bwd(float* a_dpre, const float* out, const float* wc, const float* wl, const float* weights,
                     const float* dD, int nseq, int T, int Tp, float* dout, float* dwc_part, const float* dpooled,
                     const float* dweights, void* stream) {
  if (!a_dpre || !out || !wc || !weights || !dout || !dwc_part || nseq <= 0 || T <= 0 || Tp <= 0 || Tp > T || Tp > 12288)
    return -1;
  if (!dpooled && (!dD || !wl)) return -1;               // the upstream gradient: of the pooled vector, or of D through wl
  hipLaunchKernelGGL(s2g_k_attnpool_bwd, dim3(nseq), dim3(64 * AP_NW), (size_t)Tp * sizeof(float), ST(stream), a_dpre, out, wc,
                     wl, weights, dD, T, Tp, dout, dwc_part, dpooled, dweights);
  return LAUNCH_OK();
}
int s2g_wasserstein_loss(const float* dval, int B, float* dD, float* loss, void* stream) {
  if (!dval || !dD || !loss || B <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_wasserstein_loss, dim3(1), dim3(256), 0, ST(stream), dval, B, dD, loss);
  return LAUNCH_OK();
}
int s2g_clamp(float* p, long n, float clip, void* stream) {
  if (!p || n <= 0 || !(clip >= 0.0f)) return -1;
  hipLaunchKernelGGL(s2g_k_clamp, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), p, n, clip);
  return LAUNCH_OK();
}
int s2g_mul_inplace(float* a, const float* b, long n, void* stream) {
  if (!a || !b || n <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_mul_inplace, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), a, b, n);
  return LAUNCH_OK();
}
int s2g_add_inplace(float* a, const float* b, long n, void* stream) {
  if (!a || !b || n <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_add_inplace, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), a, b, n);
  return LAUNCH_OK();
}

struct S2GAttn {        // mirrors include/s2g.h
  int N, H, T, D;
  long ld, ldo;
  float scale;
};
static int attn_dims(const S2GAttn* a, AttnDims* d) {
  if (!a || a->N <= 0 || a->H <= 0 || a->T <= 0 || a->T > 2048 || a->D <= 0 || a->D > ATT_MAXD || (a->D & 15)) return -1;
  if (a->ld < (long)a->H * a->D || a->ldo < (long)a->H * a->D || !(a->scale > 0.0f)) return -1;
  *d = AttnDims{a->N, a->H, a->T, a->D, a->ld, a->ldo, a->scale};
  return 0;
}
int s2g_attention_fwd(const float* Q, const float* K, const float* V, const float* attn_mask, const float* pad_mask,
                      float* out, float* P, const S2GAttn* dims, void* stream) {
  AttnDims d;
  if (!Q || !K || !V || !out || attn_dims(dims, &d) != 0) return -1;
  return s2g_launch_attn_fwd(Q, K, V, attn_mask, pad_mask, out, P, d, ST(stream));
}
int s2g_attention_bwd(const float* Q, const float* K, const float* V, const float* attn_mask, const float* pad_mask,
                      const float* P, const float* dO, float* dS, float* dQ, float* dK, float* dV, const S2GAttn* dims,
                      void* stream) {
  AttnDims d;
  if (!Q || !K || !V || !P || !dO || !dS || !dQ || !dK || !dV || attn_dims(dims, &d) != 0) return -1;
  return s2g_launch_attn_bwd(Q, K, V, attn_mask, pad_mask, P, dO, dS, dQ, dK, dV, d, ST(stream));
}

// ---- sequence -> graph node features (reference graph_encoder.py:57-133)
long s2g_graph_features_workspace(int n_seq, int V) {
  if (n_seq <= 0 || V <= 0 || V > GR_NMAX) return -1;
  return (long)n_seq * V * V * (long)sizeof(double);
}
int s2g_graph_features(const int* tokens, const int* lengths, int n_seq, int L, int V, double* feat, int* info,
                       double* work, long work_bytes, void* stream) {
  if (!tokens || !feat || !info || !work || n_seq <= 0 || L <= 0 || V <= 0) return -1;
  if (V > GR_NMAX || L > 65535) return -3;                    // node bitmasks are 256 wide; counts are 16-bit
  if (work_bytes < s2g_graph_features_workspace(n_seq, V)) return -3;
  GraphArgs a;
  a.tokens = tokens; a.lengths = lengths; a.L = L; a.V = V; a.feat = feat; a.info = info; a.work = work;
  const long emax = (long)V * V < (long)L - 1 ? (long)V * V : (long)L - 1;
  a.ecap = emax < 1 ? 1 : (int)emax;
  a.state_off = graph_state_off(V, a.ecap);
  if (a.state_off < 0) return -3;                             // the sequence's edge lists do not fit in LDS
  const int lds = graph_lds_layout(V, a.ecap, a.state_off).total;
  static S2GLdsCap lds_cap;
  if (s2g_lds_cap(lds_cap, (const void*)s2g_k_graph_features, (size_t)lds) != 0) return -2;
  hipLaunchKernelGGL(s2g_k_graph_features, dim3(n_seq), dim3(GR_THREADS), lds, ST(stream), a);
  return LAUNCH_OK();
}

int s2g_stream_create_cu_mask(const unsigned* mask, int nwords, void** stream_out) {
  if (!mask || nwords <= 0 || !stream_out) return -1;
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask);
  if (e != hipSuccess) {
    (void)hipGetLastError();      // the runtime refuses some mask shapes; do not leave the error for the next launch check
    return -(int)e - 1000;
  }
  *stream_out = (void*)s;
  return 0;
}

int s2g_stream_destroy(void* stream) {
  if (!stream) return -1;
  S2G_CHECK(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

}  // extern "C"

extern "C" int s2g_gp_interp_dense(const float* real, const int* lengths, const float* fake, const float* alpha,
                                   int B, int T, int Tf, int V, float* xhat, void* stream) {
  if (!real || !lengths || !fake || !alpha || !xhat || B <= 0 || T <= 0 || Tf <= 0 || V <= 0) return -1;
  hipLaunchKernelGGL(s2g_k_gp_interp_dense, dim3(cdiv((long)B * T * V, 256)), dim3(256), 0, ST(stream), real,
                     lengths, fake, alpha, B, T, Tf, V, xhat);
  return LAUNCH_OK();
}
