/* libs2g -- C ABI of the MI355X (gfx950) kernels behind the WGAN-GP training step of
 * sequence-to-graph-gan.
 *
 * The reference is pure Python/PyTorch and has no FFI; the boundary a maintainer sees is the
 * Python module surface of wgan_gp.py (Generator, Discriminator, CustomSoftmax,
 * compute_gradient_penalty, train_wgan).  Each entry point below replaces the PyTorch
 * operation(s) cited next to it and is what a ctypes binding in wgan_gp.py calls.
 *
 * Conventions: every function enqueues work on `stream` (a hipStream_t passed as void*) and
 * returns immediately; it never allocates, frees or synchronises.  Return value 0 = enqueued,
 * negative = rejected (bad argument: -1, launch failure: -2; nothing was enqueued for -1).
 * All pointers are device pointers to fp32 data unless stated; matrices are row-major.
 *
 * Internal gate order.  torch.nn.LSTM stores the 4H gate rows as blocks i|f|g|o (row q*H+u).
 * Every 200-wide buffer of this library (projections, saved gates, gate gradients) uses
 * row j' = 4*u + q instead, so the four gates of one hidden unit sit in one quad of lanes.
 * s2g_pack_critic builds the permuted copies of W_ih0; weight gradients are returned in
 * torch order by s2g_reduce_slabs(permute_rows = 1).
 */
#ifndef S2G_H
#define S2G_H
#ifdef __cplusplus
extern "C" {
#endif

#define S2G_HIDDEN 50
#define S2G_GATES 200
#define S2G_MASK_WIDTH 52

int s2g_version(void);

/* One sequence of a critic batch; xproj_t = ca * WT0[token_t] + cp * P[p_off + t] (t < p_len) + b0.
 * All sequences of a launch own rows [ws_row, ws_row + len) of every workspace buffer. */
typedef struct S2GSeq {
  int len, tok_off, p_off, p_len;
  float ca, cp;
  int mask_off, ws_row;
} S2GSeq;

/* ---- sampler: CustomSoftmax.forward / gumbel_softmax / relaxed_bernoulli (wgan_gp.py:36-104).
 * mode 0 gumbel (needs U, the torch.rand draw of :58), 1 softmax(logits/tau), 2 softmax(logits).
 * hard != 0: y_out = one-hot(argmax), y_soft (may be NULL when hard == 0) keeps the soft sample
 * for the straight-through backward (:80-84).  bias (optional) is added to the logits first,
 * indexed [(row % bias_rows) * V + j] (the last generator Linear's bias, wgan_gp.py:159). */
int s2g_gumbel_softmax_fwd(const float* logits, const float* U, const float* bias, int bias_rows,
                           float* y_out, float* y_soft, long rows, int V, float tau, int mode, int hard,
                           void* stream);
/* mode 0 with the uniforms drawn inside the kernel (Philox4x32-10 keyed by `seed`; element (row, 4l..4l+3)
 * takes the four outputs of counter (row * 64 + l, offset)): replaces torch.rand(logits.size()).to(device)
 * of wgan_gp.py:58 without a [rows, V] tensor of draws.  A draw is a pure function of (seed, offset,
 * position); the caller advances offset by one per launch.  V <= 256. */
int s2g_gumbel_softmax_fwd_rng(const float* logits, const float* bias, int bias_rows, float* y_out, float* y_soft,
                               long rows, int V, float tau, int hard, unsigned long long seed,
                               unsigned long long offset, void* stream);
/* out[0..n) = uniform [0,1) draws (kind 0: torch.rand, wgan_gp.py:228) or standard normal draws (kind 1:
 * torch.randn, :291) from the same counter-based generator. */
int s2g_rng_fill(float* out, long n, int kind, unsigned long long seed, unsigned long long offset, void* stream);
/* autograd of the above: dlogits = scale * y * (dy - <dy, y>), scale = 1/tau (modes 0,1) or 1. */
int s2g_gumbel_softmax_bwd(const float* y_soft, const float* dy, float* dlogits, long rows, int V,
                           float scale, void* stream);
/* tokens = y.argmax(-1) as int64 (generate_fake_data.py:66). */
int s2g_argmax_rows(const float* y, long long* tokens, long rows, int V, void* stream);

/* ---- exact-fp32 MFMA GEMM: nn.Linear / the W_ih product of nn.LSTM / every weight gradient.
 * C[m][n] = act(alpha * sum_k A[m*sam + ka*sak] * B[kb*sbk + n*sbn] + bias[n]) (+ C if accumulate);
 * segT > 0 maps k = b*segT + t to ka = b*a_seg + t, kb = b*b_seg + t (- 1 and zero at t == 0 when
 * shift); nsplit > 1 writes K-slices to C + z*slab.  act 1 = LeakyReLU(0.2) (wgan_gp.py:151). */
typedef struct S2GGemm {
  const float* A; long sam, sak;
  const float* B; long sbk, sbn;
  float* C; long ldc; long slab;
  const float* bias;
  int M, N, K, kchunk, segT, a_seg, b_seg, shift, accumulate, act;
  float alpha;
  const float* bscale;   /* grouped launches only, segT > 0: row k of B is multiplied by bscale[k / segT] (one factor
                          * per sequence; NULL: none).  s2g_gemm_f32 ignores it. */
} S2GGemm;
int s2g_gemm_f32(const S2GGemm* args, int nsplit, void* stream);
/* the tall products of the critic's input layer (x W_ih0^T, delta W_ih0, delta (W_ih0 W_ih0^T);
 * the W_ih part of nn.LSTM and its autograd, wgan_gp.py:189,217,247-254): C[M,N] = A[M,K] B[K,N],
 * row-major, 160 < N <= 224, any K; ldb % 4 == 0, B 16-byte aligned and readable (zeros) up to
 * column roundup(N, 4).  A and B are each read once per 64-row block. */
int s2g_gemm_tall(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
                  void* stream);
/* the same products with both operands rounded to bf16 and fp32 accumulation (v_mfma_f32_16x16x32_bf16):
 * the optional reduced-precision path of the step (results within 2e-2 relative of the fp32 reference);
 * N <= 224, K <= 224, any lda / ldb / alignment. */
int s2g_gemm_tall_bf16(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
                       void* stream);
/* nn.Linear forward for a small batch and many output features (the generator's 512 -> L*V output layer,
 * wgan_gp.py:159): y[batch, n_out] = x[batch, K] W[n_out, K]^T + bias; batch <= 64, n_out % 16 == 0,
 * K % 128 == 0, K <= 512, W / y / bias 16-byte aligned.  W is streamed from HBM exactly once.
 * max_workgroups: the number of CUs `stream` can reach when it is CU-masked (a workgroup takes a whole
 * CU; more workgroups than that would run in two rounds), 0 = all CUs of the device. */
int s2g_linear_small_batch(const float* W, const float* x, const float* bias, float* y, int n_out, int batch, int K,
                           int max_workgroups, void* stream);
/* Weight gradient of the same layer fused with its Adam step: what loss.backward() leaves in
 * model[8].weight.grad (wgan_gp.py:341, dW[n_out, K] = dy[batch, n_out]^T a[batch, K]) and
 * optimizer_G.step() (wgan_gp.py:342, torch.optim.Adam without weight decay / amsgrad) then does to W and
 * its moments m, v, in ONE pass over W, m and v; the arithmetic per element is s2g_adam's.  grad_w (optional,
 * may be NULL) also receives dW.  batch <= 64, K = 256 or 512, W / m / v / grad_w 16-byte aligned;
 * step = 1-based count of this update; max_workgroups as for s2g_linear_small_batch.  Single process only:
 * with several ranks the gradient has to be all-reduced before Adam sees it. */
int s2g_linear_wgrad_adam(const float* dy, const float* a, float* W, float* m, float* v, float* grad_w, int n_out,
                          int batch, int K, float lr, float b1, float b2, float eps, int step, int max_workgroups,
                          void* stream);
/* Weight and bias gradients of up to four small nn.Linear layers in one launch (the generator's
 * 100 -> 128 -> 256 -> 512 stack, wgan_gp.py:147-156; what loss.backward(), wgan_gp.py:341, leaves in their
 * .grad): for job j, dW[j][n_out, n_in] = dy[j][batch, n_out]^T x[j][batch, n_in] and (db[j] not NULL)
 * db[j][n_out] = column sums of dy[j].  The arrays are HOST arrays of n_jobs (1..4) entries; n_in % 4 == 0,
 * x[j] / dW[j] 16-byte aligned, batch <= 4096. */
int s2g_small_linear_grads(int n_jobs, const float* const* dy, const float* const* x, float* const* dW,
                           float* const* db, const int* n_out, const int* n_in, int batch, void* stream);
/* Generator output layer + Gumbel-softmax + argmax in one kernel: what generate_fake_data.py:64-66 computes
 * as generator(noise, labels, tau, hard).argmax(-1) from the last hidden activations x [batch, K]
 * (wgan_gp.py:159, 169-171): tokens[b][t] = argmax_v (x[b] . W[t V + v] + bias[t V + v] + gumbel(u[b][t][v])),
 * int64 [batch, L]; the [batch, L, V] logits and samples are never written.  The argmax of the sample equals
 * the argmax of the perturbed logits for every tau > 0 and for hard and soft sampling alike (first index
 * on ties, as torch.argmax).  U [batch, L, V]: the uniforms of wgan_gp.py:58, or NULL: drawn in the kernel
 * with the numbering of s2g_gumbel_softmax_fwd_rng (same seed / offset -> same tokens as that sampler's
 * argmax).  batch <= 64, K % 128 == 0, K <= 512, V <= 384, W and x 16-byte aligned. */
int s2g_generate_tokens(const float* W, const float* x, const float* bias, const float* U, long long* tokens, int L,
                        int V, int batch, int K, unsigned long long seed, unsigned long long offset, void* stream);
/* nprob problems in ONE launch.  table_dev / prefix_dev are DEVICE arrays: S2GGemm[nprob] (every
 * problem of the weight-gradient shape: A = [rows, M] with sam == 1, M % 4 == 0, M <= 208,
 * sak % 4 == 0 and a 16-byte aligned base, kchunk > 0 set by the caller, no bias / act /
 * accumulate) and int[nprob + 1] block prefix sums; problem p owns ceil(N/64) * nsplit_p blocks,
 * each computing all M rows of 64 columns over one K-chunk.
 * nz > 0: every problem has exactly nz K-chunks and prefix_dev holds prefix sums of ceil(N/64), the
 * output tiles per problem; blocks are then ordered K-chunk first (total_blocks = nz * prefix[nprob]), so
 * problems that read the same rows of an operand find them in L2. */
int s2g_gemm_grouped(const S2GGemm* table_dev, const int* prefix_dev, int nprob, int total_blocks, int nz,
                     void* stream);
/* The same launch with both operands rounded to bf16 and fp32 accumulation (v_mfma_f32_16x16x32_bf16): the
 * reduced-precision form of the weight gradients (autograd of the W_ih / W_hh products of nn.LSTM, wgan_gp.py:189,
 * 317) for the 2e-2 path; same tables, block order and slabs; every kchunk a multiple of 32. */
int s2g_gemm_grouped_bf16(const S2GGemm* table_dev, const int* prefix_dev, int nprob, int total_blocks, int nz,
                          void* stream);
int s2g_reduce_slabs(const float* slabs, long slab, int nslab, int rows, int cols, float* out,
                     int permute_rows, int accumulate, float scale, void* stream);
/* out[i] = sum_z weights[z] * slabs[z * slab + i] (ascending z, i < n <= slab): with one split-K slab per sequence, a
 * per-sequence weighted sum of outer products (the first-order factor of the penalty's weight gradient, autograd of
 * wgan_gp.py:247-256) gets its weights after the product has run. */
int s2g_reduce_slabs_weighted(const float* slabs, long slab, int nslab, const float* weights, float* out, long n,
                              void* stream);
/* up to 8 such reductions in one launch; `jobs` is a HOST array (copied into the kernel arguments) */
typedef struct S2GReduce {
  const float* slabs; long slab; float* out;
  int nslab, rows, cols;
} S2GReduce;
int s2g_reduce_slabs_multi(const S2GReduce* jobs, int njobs, int permute_rows, int accumulate, float scale,
                           void* stream);
/* one process only (no all-reduce between gradient and update): the final slab reduction of the four
 * LSTM weight gradients (job[0..3] = W_ih0, W_hh0, W_ih1, W_hh1, as in s2g_reduce_slabs_multi with
 * permute_rows), optimizer_D.step() (Adam, wgan_gp.py:318) on every critic parameter, and the packed
 * copies of s2g_pack_critic, in ONE launch.  p/g/m/v: the flat parameter, gradient and moment buffers;
 * poff / o_*: element offsets of the parameters in them; the gradients of the biases and the fc head
 * are read from g, those of the four matrices are also written to job[k].out. */
typedef struct S2GCriticUpdate {
  S2GReduce job[4];
  long poff[4];
  float *p, *g, *m, *v;
  long o_bih0, o_bhh0, o_bih1, o_bhh1, o_fcw, o_fcb;
  int n_fcw, n_fcb, V, Vp;
  float *WT0, *W0p, *b0, *b1;
  float lr, b1, b2, eps;
  int step;
} S2GCriticUpdate;
int s2g_critic_update(const S2GCriticUpdate* u, void* stream);
/* part[chunk][c] = sum over a chunk of rows of X[r*ld + c]; nchunk = ceil(rows / rows_per_chunk). */
int s2g_colsum(const float* X, long ld, long rows, int cols, float* part, long rows_per_chunk, void* stream);
/* bias gradients of the two LSTM layers (what backward() accumulates into lstm.bias_ih/_hh_l{0,1},
 * wgan_gp.py:317): column sums of the gate gradients X0 / X1 [rows, 200] (internal gate order), written
 * (accumulate == 0) or added (1) in torch row order to bias_ih AND bias_hh of each layer (they share one
 * gradient).  hdotT [B, 50] (optional): its column sum is added to fcw0[50], the penalty's part of
 * fc.weight[0].  part: scratch of 2 * ceil(rows / 512) * 200 floats.  Two launches. */
int s2g_bias_sums(const float* X0, const float* X1, long rows, float* part, const float* hdotT, int B,
                  float* bih0, float* bhh0, float* bih1, float* bhh1, float* fcw0, int accumulate, void* stream);

/* ---- critic LSTM sweeps: nn.LSTM(175, 50, 2, batch_first, dropout) inside
 * Discriminator.forward (wgan_gp.py:189, 217-220) and its autograd.
 * Buffers indexed by workspace row: gates*, delta*, adot*, abar*, cell*, celld* are [rows, 200];
 * adj* is [rows, 100]; mask is [rows, 52] (dropout multiplier of layer-0 outputs, NULL = none);
 * hT/dhT/hdotT are [nseq, 50].  whh0/wih1/whh1 are the torch parameters [200, 50]. */
/* tokens (optional) must lie in [0, vocab), the rows of WT0 (F.one_hot raises otherwise, wgan_gp.py:277;
 * here an out-of-range id is clamped to the table); sequences longer than max_len are cut to max_len. */
int s2g_lstm2_fwd(const S2GSeq* seqs, int nseq, int max_len, int vocab, const int* tokens, const float* WT0,
                  const float* P, const float* mask, const float* b0, const float* b1, const float* whh0,
                  const float* wih1, const float* whh1, float* gates0, float* cell0, float* gates1,
                  float* cell1, float* hT, void* stream);
/* reverse sweep for d(loss)/d(h1_final) = dhT: what d_loss.backward() (wgan_gp.py:317) and
 * autograd.grad(..., inputs=interpolates) (:247-254) run through the LSTM.  adj0/adj1 may be NULL;
 * otherwise sequences n >= adj_seq0 store (dh_t, dc_t) at adj row (ws_row + t - adj_row0).
 * rows_per_seq: workspace rows reserved per sequence (rows >= len are zero-filled in delta*). */
/* head (optional, a HOST struct): sweep B evaluates the fc head (wgan_gp.py:221) and the derivative of
 * the loss terms (wgan_gp.py:299-313 / 333-340) of each sequence itself and ignores dhT, so that no
 * launch sits between the forward and the reverse sweep.  mode 0: nseq = 3B laid out
 * [real B | fake B | interp B] with d(out) = (-inv_n, .5 dCE), (+inv_n, .5 dCE), (1, 0, 0);
 * mode 1: nseq = B fakes of the generator pass, d(out) = (-inv_n, .25 dCE). */
typedef struct S2GHead {
  const float* hT;          /* [nseq, 50] final hidden states written by s2g_lstm2_fwd */
  const float* W;           /* fc.weight [3, 50] */
  const float* bias;        /* fc.bias [3] */
  const long long* labels;  /* [B] */
  int B, mode;
  float inv_n, inv_wsum, w0, w1;
} S2GHead;
int s2g_lstm2_bwd(const S2GSeq* seqs, int nseq, int rows_per_seq, const float* mask, const float* whh0,
                  const float* wih1, const float* whh1, const float* gates0, const float* cell0,
                  const float* gates1, const float* cell1, const float* dhT, float* delta0, float* delta1,
                  float* adj0, float* adj1, int adj_seq0, int adj_row0, const struct S2GHead* head,
                  float* bsum, void* stream);
/* tangent (forward-mode) sweep along an input direction whose projection is Xdot [rows, 200];
 * with s2g_lstm2_bwd2 it replaces the double backward that create_graph=True makes autograd
 * run for the gradient penalty (wgan_gp.py:251, 256, 317).  With gp_partial (the [nseq,4] output
 * of s2g_gp_dots) sequence b's direction is Xdot_b times the penalty coefficient
 * gp_upstream * 2 (|g_b| - 1) / (gp_n_batch |g_b|), evaluated by the sweep itself; NULL: Xdot as is. */
int s2g_lstm2_tangent(const S2GSeq* seqs, int nseq, const float* Xdot, const float* mask, const float* whh0,
                      const float* wih1, const float* whh1, const float* gates0, const float* cell0,
                      const float* gates1, const float* cell1, float* adot0, float* celld0, float* adot1,
                      float* celld1, float* hdotT, const float* gp_partial, float gp_upstream, float gp_n_batch,
                      void* stream);
int s2g_lstm2_bwd2(const S2GSeq* seqs, int nseq, int rows_per_seq, const float* mask, const float* whh0,
                   const float* wih1, const float* whh1, const float* gates0, const float* cell0,
                   const float* gates1, const float* cell1, const float* adot0, const float* celld0,
                   const float* adot1, const float* celld1, const float* adj0, const float* adj1,
                   float* abar0, float* abar1, float* bsum, void* stream);
/* s2g_bias_sums without the `part` workspace, for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  The partial sums go to a file-scope device array of the library (800 KB, held by every GPU
 * the library runs on until the process ends), so rows <= 262144; more rows return -3 and nothing is
 * enqueued.  Calls no hipMalloc / hipFree.  All calls on one GPU share that array: two calls whose
 * kernels may overlap (different streams without an ordering event) race; same-stream calls are safe.
 * Results are bit-identical to s2g_bias_sums. */
int s2g_bias_sums_noalloc(const float* X0, const float* X1, long rows, const float* hdotT, int B, float* bih0,
                          float* bhh0, float* bih1, float* bhh1, float* fcw0, int accumulate, void* stream);
/* bsum (optional, both reverse sweeps): [2][nseq][200]; sequence n's sums over time of its layer-0
 * (bsum[0][n]) and layer-1 (bsum[1][n]) gate gradients, i.e. its share of the bias gradients, left
 * behind by the sweep at no extra pass over memory.  s2g_bias_sums_final adds up the first nseq rows
 * of each layer (layers layer_stride floats apart) into bias_ih / bias_hh as s2g_bias_sums does. */
int s2g_bias_sums_final(const float* sums, int nseq, long layer_stride, const float* hdotT, int B, float* bih0,
                        float* bhh0, float* bih1, float* bhh1, float* fcw0, int accumulate, void* stream);

/* permuted copies of lstm.weight_ih_l0 and the summed biases (Discriminator parameters,
 * wgan_gp.py:189): WT0 [V,200], W0p [200, roundup(V,4)] (pad columns must
 * be zero on entry and are left untouched), b0/b1 [200]. */
int s2g_pack_critic(const float* wih0, const float* bih0, const float* bhh0, const float* bih1,
                    const float* bhh1, float* WT0, float* W0p, float* b0, float* b1, int V, void* stream);
/* sequence descriptors of the fused critic batch [real B | fake B | interp B | interp B with
 * group-relative rows] (wgan_gp.py:297-307); seqs has room for 4B entries */
int s2g_critic_batch_desc(const int* lengths, const float* alpha, int B, int T, int Tf, int use_mask,
                          S2GSeq* seqs, void* stream);

/* ---- critic head and losses: self.fc (wgan_gp.py:222) and lines 298-315 / 333-340 */
int s2g_head_fwd(const float* hT, const float* W, const float* b, float* out, int N, int J, void* stream);
int s2g_head_bwd_dh(const float* dout, const float* W, float* dhT, int N, int J, void* stream);
int s2g_head_bwd_w(const float* dout, const float* hT, float* dW, float* db, int N, int J, int accumulate,
                   void* stream);
int s2g_critic_loss(const float* out, const long long* labels, int B, float inv_n, float inv_wsum, float w0,
                    float w1, float* dout, float* losses, void* stream);
int s2g_generator_loss(const float* out, const long long* labels, int B, float inv_n, float inv_wsum,
                       float w0, float w1, float* dout, float* losses, void* stream);
/* s2g_head_fwd (3B rows) + s2g_critic_loss + s2g_head_bwd_dh (3B rows) + s2g_head_bwd_w (first 2B
 * rows, no accumulate) of one critic iteration in a single launch; hT/out/dout/dhT as in those. */
int s2g_critic_head(const float* hT, const float* W, const float* bias, const long long* labels, int B,
                    float inv_n, float inv_wsum, float w0, float w1, float* out, float* dout, float* losses,
                    float* dhT, float* dW, float* db, void* stream);

/* ---- gradient penalty pieces: compute_gradient_penalty (wgan_gp.py:225-257) */
int s2g_one_hot(const int* tokens, const int* lengths, int T, float* x, long rows, int V, void* stream);
int s2g_gp_interp(const int* tokens, const int* lengths, const float* fake, const float* alpha, int B, int T,
                  int Tf, int V, float* xhat, void* stream);
/* same with the real batch given dense [B,T,V] (the argument compute_gradient_penalty receives) */
int s2g_gp_interp_dense(const float* real, const int* lengths, const float* fake, const float* alpha, int B,
                        int T, int Tf, int V, float* xhat, void* stream);
/* per sample b: norms[b] = |g_b|, pen[b] = (|g_b| - 1)^2 and, when scale_in_place,
 * g_b <- upstream * 2 (|g_b| - 1) / (n_batch |g_b|) * g_b */
int s2g_gp_norm_penalty(float* g, int B, long per_sample, float upstream, float n_batch, float* pen,
                        float* norms, int scale_in_place, void* stream);
/* the same penalty from the layer-0 gate gradients delta [rows,200] and pv = delta (W_ih0 W_ih0^T):
 * |g_b|^2 = sum_t delta_t . pv_t, so the 175-wide gradients.view(B,-1) of wgan_gp.py:255 need not
 * exist yet; pv_b <- coef_b * pv_b in place (the projected direction the tangent sweep consumes) */
int s2g_gp_norm_penalty_proj(const float* delta, float* pv, int B, long per_sample, float upstream,
                             float n_batch, float* pen, float* norms, void* stream);
/* the reduction of s2g_gp_norm_penalty_proj alone, in four blocks per sample and with pv left
 * unscaled: partial[b][q] = sum over the q-th quarter of sample b of delta . pv; the tangent sweep
 * takes the coefficient from the partials, s2g_gp_penalty_final gives pen / norms (wgan_gp.py:255-256) */
int s2g_gp_dots(const float* delta, const float* pv, int B, long per_sample, float* partial, void* stream);
int s2g_gp_penalty_final(const float* partial, int B, float upstream, float n_batch, float* pen, float* norms,
                         float* coef, void* stream);
int s2g_gp_scale(const float* g, const float* norms, const float* upstream, float n_batch, long per_sample,
                 float* v, int B, void* stream);

/* ---- generator glue: Generator.forward (wgan_gp.py:165-171), BatchNorm1d(n, eps=0.8) + LeakyReLU */
int s2g_embed_mul(const float* z, const long long* labels, const float* emb, float* out, int B, int D,
                  int n_classes, void* stream);
int s2g_embed_mul_bwd(const float* z, const long long* labels, const float* dout, float* demb, int B, int D,
                      int n_classes, void* stream);
int s2g_lrelu_bwd(const float* y, float* dy, long n, void* stream);
/* ---- transformer encoder layer pieces (reference transformer.py:285-317, CustomTransformerEncoderLayer.forward).
 * F.relu of the feed-forward block (:309) is the GEMM epilogue act = 2; its backward: dy <- y > 0 ? dy : 0. */
int s2g_relu_bwd(const float* y, float* dy, long n, void* stream);
/* src = norm(src + dropout(src2)) (:305-306, :310-311): y = LayerNorm(x + r * rmask) over the E <= 1024 features of
 * each of `rows` rows (r, rmask NULL: no residual / no dropout multiplier); xhat [rows, E] and rstd [rows] (may be
 * NULL) are saved for the backward.  nn.LayerNorm arithmetic: biased variance, eps inside the root. */
int s2g_add_layernorm_fwd(const float* x, const float* r, const float* rmask, const float* gamma, const float* beta,
                          float eps, float* y, float* xhat, float* rstd, long rows, int E, void* stream);
/* dz = d loss / d (x + r * rmask) from dy; dyxh = dy * xhat (column sums: d gamma; those of dy: d beta). */
int s2g_add_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dz,
                          float* dyxh, long rows, int E, void* stream);
/* inter-layer dropout of nn.LSTM(175, 50, 2, dropout=p) in train mode (wgan_gp.py:189): uniform draws
 * u[rows][52] -> multiplier (u >= p) / (1 - p) in place, columns 50 and 51 (row padding) -> 0 */
int s2g_dropout_mask(float* u, long rows, float p, void* stream);
/* such a mask [rows][width] with the uniforms drawn inside the kernel (one write pass, no tensor of draws):
 * columns < live_cols hold (u >= p) / (1 - p), the padding columns 0; width % 4 == 0 (52 / 50 for the H = 50
 * critic, 64 / 64 for the H = 64 critic of wgan.py) */
int s2g_dropout_mask_rng(float* mask, long rows, int width, int live_cols, float p, unsigned long long seed,
                         unsigned long long offset, void* stream);
int s2g_bn_stats(const float* x, float* stats, int B, int F, void* stream);
int s2g_bn_lrelu_apply(const float* x, const float* stats, float n_global, const float* gamma,
                       const float* beta, float* run_mean, float* run_var, float eps, float momentum,
                       int update_running, float* y, float* xhat, float* invstd, int B, int F, void* stream);
int s2g_bn_lrelu_bwd_stats(const float* y, float* dy, const float* xhat, float* sums, int B, int F,
                           void* stream);
int s2g_bn_bwd_apply(const float* dyh, const float* xhat, const float* sums, float n_global,
                     const float* gamma, const float* invstd, float* dx, int B, int F, void* stream);
/* single-process forms (batch statistics of nn.BatchNorm1d in train mode need no all-reduce then,
 * wgan_gp.py:150,156-157): s2g_bn_stats + s2g_bn_lrelu_apply, and s2g_bn_lrelu_bwd_stats +
 * s2g_bn_bwd_apply with dgamma / dbeta written directly, each in one launch */
int s2g_bn_lrelu_fwd_fused(const float* x, const float* gamma, const float* beta, float* run_mean,
                           float* run_var, float eps, float momentum, int update_running, float* y,
                           float* xhat, float* invstd, int B, int F, void* stream);
int s2g_bn_lrelu_bwd_fused(const float* y, const float* dy, const float* xhat, const float* gamma,
                           const float* invstd, float* dgamma, float* dbeta, float* dx, int B, int F,
                           void* stream);

/* ---- optimiser: torch.optim.Adam(lr, betas) as built at wgan_gp.py:405-406; `step` counts from 1;
 * the gradient is multiplied by gscale first (1/world_size after a SUM all-reduce). */
int s2g_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
             int step, float gscale, void* stream);

/* ---- attention core of MaskedMultiheadAttention.forward (transformer.py:243-274) and its autograd.
 * Q, K, V: [N, T, H, D] views with row stride ld (the three chunks of the fused in-projection, :235-241),
 * out / dO: [N, T, H D] with row stride ldo, P and dS: [N, H, T, T].  The reference's conventions are kept:
 * energy <- -1e3 (finite) where attn_mask[q][k] == 0 ([T, T] floats, optional) and where pad_mask[n][k] == 0
 * ([N, T] floats, optional; the reference's helper puts 1 at PAD positions, so the NON-pad keys are filled),
 * softmax(energy * scale) with scale = 1 / sqrt(embed_size).  D % 16 == 0, D <= 64, T <= 2048. */
typedef struct S2GAttn {
  int N, H, T, D;
  long ld, ldo;
  float scale;
} S2GAttn;
int s2g_attention_fwd(const float* Q, const float* K, const float* V, const float* attn_mask, const float* pad_mask,
                      float* out, float* P, const S2GAttn* dims, void* stream);
/* gradients of the above given dO: dQ, dK, dV (layout and stride of Q); dS is scratch [N, H, T, T] */
int s2g_attention_bwd(const float* Q, const float* K, const float* V, const float* attn_mask, const float* pad_mask,
                      const float* P, const float* dO, float* dS, float* dQ, float* dK, float* dV, const S2GAttn* dims,
                      void* stream);

/* ---- weight-clipping WGAN variant (reference wgan.py).  Hidden size 64 throughout (wgan.py:50, 96); buffers
 * are [rows, .] with sequence n owning rows [n T, n T + len_n).
 * One LSTM layer over all its steps (nn.LSTM(64, 64, ...) of wgan.py:58, 100): pin [rows, 256] = x W_ih^T + b_ih + b_hh
 * (a batched product done beforehand), whh the torch parameter [256, 64]; saves the activated gates [rows, 256]
 * (torch order i|f|g|o), the cell states [rows, 64] and writes the hidden states hout [rows, 64] (zeros beyond a
 * sequence's length, as pad_packed_sequence returns them, wgan.py:114).  lengths NULL: every sequence has T steps.
 * out_mask / hdrop (optional): hdrop = hout * out_mask, the next layer's input under nn.LSTM's dropout. */
int s2g_lstm64_fwd(const float* pin, const float* whh, const int* lengths, int nseq, int T, float* gates, float* cst,
                   float* hout, const float* out_mask, float* hdrop, void* stream);
/* All layers of that stack in one sweep (layers = 2 or 3; nn.LSTM(64, 64, num_layers) of wgan.py:58, 100): layer
 * l + 1 runs one step behind layer l and takes its input mask[l] * h_l (nn.LSTM's inter-layer dropout, mask NULL =
 * none) through on-chip memory, so only layer 0 needs the batched input product pin0.  Per layer: whh [256, 64];
 * from layer 1 on wih [256, 64] and bsum [256] = b_ih + b_hh; outputs as s2g_lstm64_fwd writes them, each
 * optional (NULL = not stored) except the top layer's hout. */
typedef struct S2GLstm64Stack {
  const float* pin0;
  const float* whh[3];
  const float* wih[3];
  const float* bsum[3];
  float* gates[3];
  float* cst[3];
  float* hout[3];
  const float* mask[3];
  float* hdrop[3];
  /* token mode (tokens != NULL; pin0 unused): layer 0's input part of step t is tab[token_t] + tabl[label of the
   * sequence], tab [V, 256] = emb W_ih0[:, :32]^T + b_ih0 + b_hh0, tabl [n_classes, 256] = label_emb W_ih0[:, 32:]^T:
   * what the lookup branch of RelaxedEmbedding followed by the concat and the LSTM's input product compute
   * (wgan.py:39-42, 107-111), without the [rows, 64] and [rows, 256] intermediates */
  const int* tokens;
  const float* tab;
  const long long* labels;
  const float* tabl;
  int V, n_classes;
} S2GLstm64Stack;
int s2g_lstm64_stack_fwd(const S2GLstm64Stack* st, int layers, const int* lengths, int nseq, int T, void* stream);
/* its autograd: dh_in [rows, 64] = gradient reaching every h_t from above -> delta [rows, 256], the gradient of
 * the gate pre-activations (zeros beyond the length); weight / input gradients are batched products of delta. */
int s2g_lstm64_bwd(const float* dh_in, const float* whh, const int* lengths, int nseq, int T, const float* gates,
                   const float* cst, float* delta, void* stream);
/* the autograd of a TWO-layer stack in one reverse sweep (Discriminator's nn.LSTM(64, 64, 2, dropout), wgan.py:100):
 * dh_top [rows, 64] reaches the top layer's h; the gradient reaching layer 0's h, mask0 * (delta1 W_ih1), stays on
 * chip; delta[l] [rows, 256] as s2g_lstm64_bwd writes them.  mask0 NULL = no dropout between the layers. */
typedef struct S2GLstm64StackBwd {
  const float* dh_top;
  const float* whh[2];
  const float* wih1;
  const float* gates[2];
  const float* cst[2];
  const float* mask0;
  float* delta[2];
  float* bias_part[2];   /* optional [nseq, 256]: per-sequence sums of delta over time (sum them for db) */
} S2GLstm64StackBwd;
int s2g_lstm64_stack_bwd(const S2GLstm64StackBwd* st, const int* lengths, int nseq, int T, void* stream);
/* x [rows, 64] = (emb[token] | lab_emb[label of the sequence]): label_embedding / RelaxedEmbedding lookup + concat
 * (wgan.py:39-42, 107-111); soft != 0: columns 0..31 already hold y @ emb (the matmul branch, :45). */
int s2g_embed_concat(const int* tokens, const float* emb, int V, const long long* labels, const float* lab_emb,
                     int n_classes, int T, long rows, int soft, float* x, void* stream);
/* gradients of the two lookups from dx [rows, 64] (live rows only; what backward() leaves in
 * sequence_embedding.embed_layer.weight.grad / label_embedding.weight.grad, wgan.py:39-42, 107-111), each as a
 * two-stage sum with a fixed order of additions.  part: scratch of s2g_embed_grad_tokens_scratch(rows, V) floats
 * (tokens; V <= 320) / B * 32 floats (labels). */
long s2g_embed_grad_tokens_scratch(long rows, int V);
int s2g_embed_grad_tokens(const int* tokens, const int* lengths, int T, long rows, int V, const float* dx, float* demb,
                          int accumulate, float* part, void* stream);
/* the same two-stage ordered sum for any [rows, width] matrix x (row stride ld, width a multiple of 32):
 * out [V, width] = sum over live rows with token v of x[row]; part: s2g_embed_grad_tokens_scratch(rows, V) * width / 32
 * floats.  With x = the gate gradients of LSTM layer 0 this is all the token path of wgan.py:107-111 needs for the
 * gradients of W_ih0 and of the embedding (both are products of these [V, 256] sums with small matrices). */
int s2g_bucket_rows_by_token(const int* tokens, const int* lengths, int T, long rows, int V, const float* x, int ld,
                             int width, float* out, float* part, void* stream);
/* The same bucketed sums for any width (the last 32-column slice may be partial) over rows laid out x_seg >= T rows
 * per sequence (row of step t of sequence n = n * x_seg + t; tokens / lengths still count T per sequence), written
 * TRANSPOSED: out[c * ldo + v] = sum over live rows with token v of x[row][c].  With x = the layer-0 gate gradients of
 * sequences whose input is one-hot this is that weight gradient itself, dW_ih0[gate row][token] (the autograd of
 * F.one_hot(...).float() through nn.LSTM, wgan_gp.py:277, 317), without a one-hot operand or a product with zeros.
 * part: s2g_embed_grad_tokens_scratch(rows, V) * ceil(width / 32) floats. */
int s2g_bucket_rows_by_token_t(const int* tokens, const int* lengths, int T, long rows, int V, const float* x, int ld,
                               int x_seg, int width, float* out, int ldo, float* part, void* stream);
/* out [n_classes, width] = sum over sequences with label l (ascending) of x[n] (x [B, width], width <= 1024) */
int s2g_bucket_rows_by_label(const long long* labels, int B, int n_classes, const float* x, int width, float* out,
                             void* stream);
int s2g_embed_grad_labels(const long long* labels, const int* lengths, int T, int B, int n_classes, const float* dx,
                          float* dlab, int accumulate, float* part, void* stream);
/* Attention pooling + output Linear of the critic (wgan.py:82-89, 116-118) over the first Tp = max(lengths) steps:
 * pre [rows, 64] = out Wa^T + ba on entry, tanh(pre) on return; weights [nseq, T], pooled [nseq, 64], dval [nseq]. */
int s2g_attnpool_fwd(float* pre, const float* out, const float* wc, const float* wl, const float* bl, int nseq, int T,
                     int Tp, float* weights, float* pooled, float* dval, void* stream);
/* its autograd for dD [nseq]: a_dpre (tanh(pre) on entry) <- d(pre); dout [rows, 64] = the direct part of d(out);
 * dwc_part [nseq, 64] = every sequence's share of d(context_vector.weight).
 * dpooled [nseq, 64] (or NULL): the gradient of the pooled vector itself, used instead of dD[n] * wl (Attention.forward
 * returns the pooled vector, wgan.py:88-89; dD and wl may then be NULL); dweights [nseq, T] (or NULL): a gradient
 * on the attention weights, its second return value. */
int s2g_attnpool_bwd(float* a_dpre, const float* out, const float* wc, const float* wl, const float* weights,
                     const float* dD, int nseq, int T, int Tp, float* dout, float* dwc_part, const float* dpooled,
                     const float* dweights, void* stream);
/* d_loss = -(mean(D_real) - mean(D_fake)) (wgan.py:145-147) for dval = [real B | fake B], and its derivative dD */
int s2g_wasserstein_loss(const float* dval, int B, float* dD, float* loss, void* stream);
/* p <- clamp(p, -clip, clip) (wgan.py:155-158) */
int s2g_clamp(float* p, long n, float clip, void* stream);
int s2g_mul_inplace(float* a, const float* b, long n, void* stream);
/* a += b: the gradients of the fake pass (computed beside the real pass into a buffer of their own) join those of
 * the real pass, as backward() accumulates them in one .grad (wgan.py:150) */
int s2g_add_inplace(float* a, const float* b, long n, void* stream);

/* ---- sequence -> graph.  Node features of the transition graph of each token sequence, what
 * GraphEncoder.encode (graph_encoder.py:57-133) computes per sequence with networkx (graph_encoder.py:86-91):
 * one node per distinct symbol, an edge u -> v weighted by the number of times v follows u;
 * feat [n_seq, V, 4] float64 = (katz_centrality_numpy(weight), betweenness_centrality, closeness_centrality,
 * pagerank(weight)) at the symbol's id, zeros for absent symbols, for single-node graphs and (Katz only) for
 * two-node graphs, as the reference assigns them.  tokens [n_seq, L] int32 symbol ids in [0, V), V <= 256,
 * L <= 65535; lengths [n_seq] or NULL.  info [n_seq, 4] int32 = nodes, status, pagerank iterations, distinct
 * edges; status bit 0: pagerank did not converge in 100 iterations (networkx raises
 * PowerIterationFailedConvergence), bit 1: the Katz matrix I - 0.1 W^T is singular (numpy raises LinAlgError);
 * the host mirror raises the same.  work: s2g_graph_features_workspace(n_seq, V) bytes of device memory for
 * the Katz matrices that do not fit in LDS (n > ~118 at V = 175).  Returns -3 for sizes beyond these limits. */
int s2g_graph_features(const int* tokens, const int* lengths, int n_seq, int L, int V, double* feat, int* info,
                       double* work, long work_bytes, void* stream);
long s2g_graph_features_workspace(int n_seq, int V);

/* ---- streams.  No counterpart in the reference (its step is one serial autograd graph,
 * wgan_gp.py:297-317): the second-order sweeps of the penalty run one workgroup per sequence and
 * occupy B of the 256 CUs, so the host runs the work that does not depend on them on a second HIP
 * stream confined to the other CUs.  mask: bit i of word i/32 enables CU i for kernels launched on
 * the stream (hipExtStreamCreateWithCUMask). */
int s2g_stream_create_cu_mask(const unsigned* mask, int nwords, void** stream_out);
int s2g_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif
