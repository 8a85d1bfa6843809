"""Unit checks of single libs2g entry points against plain PyTorch fp32/fp64 references of the same
operation (the composite paths are covered by test_gpu_step.py / test_gpu_sweeps.py)."""
import ctypes

import pytest
import torch

import common

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3          # relative fp32 bound of the north star; the kernels land ~1e-6


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,K,N,ldb", [(32768, 175, 200, 200), (1000, 200, 200, 200), (517, 200, 175, 176),
                                         (64, 33, 161, 164), (129, 7, 224, 224)])
def test_gemm_tall_matches_torch(M, K, N, ldb):
    from s2g import ops
    from s2g.ops import ptr
    g = common.gen(11)
    A = torch.randn(M, K, generator=g).to(DEV)
    B = torch.zeros(K, ldb)
    B[:, :N] = torch.randn(K, N, generator=g)
    B = B.to(DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm_tall(ptr(A), K, ptr(B), ldb, ptr(C), N, M, N, K)
    torch.cuda.synchronize()
    assert rel(C, A.double() @ B[:, :N].double()) < 1e-5


def test_gemm_tall_rejects_bad_shapes():
    from s2g import lib
    L = lib.get()
    x = torch.zeros(64, 256, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    assert L.s2g_gemm_tall(x.data_ptr(), 200, x.data_ptr(), 200, x.data_ptr(), 200, 64, 100, 200, s) == -1   # N <= 160
    assert L.s2g_gemm_tall(x.data_ptr(), 200, x.data_ptr(), 175, x.data_ptr(), 175, 64, 175, 200, s) == -1   # ldb % 4
    assert L.s2g_gemm_tall(x.data_ptr(), 200, x.data_ptr(), 256, x.data_ptr(), 256, 64, 230, 200, s) == -1   # N > 224


@pytest.mark.parametrize("fn,tol", [("s2g_gemm_grouped", 1e-5), ("s2g_gemm_grouped_bf16", 1e-5)])
def test_gemm_grouped_matches_torch(fn, tol):
    """Three weight-gradient products in one launch: a wide one on contiguous rows with a ragged K, a
    narrow one whose B operand is strided (every 4th float) and reads the previous step inside each
    sequence (zero at step 0), and a wide one whose operands keep a different number of rows per sequence.
    The bf16 form is held to the same bound against a float64 product of the bf16-ROUNDED operands (its
    only other difference from the fp32 form is the order of the fp32 additions)."""
    from s2g.lib import S2GGemm, call
    from s2g.ops import ptr, stream_ptr
    g = common.gen(21)
    nb, T, Tw = 5, 37, 41
    K1 = 1003
    rnd = (lambda t: t.bfloat16().float()) if fn.endswith("bf16") else (lambda t: t)
    A1 = rnd(torch.randn(K1, 200, generator=g)).to(DEV)
    B1 = rnd(torch.randn(K1, 175, generator=g)).to(DEV)
    A2 = rnd(torch.randn(nb * Tw, 200, generator=g)).to(DEV)
    cell = rnd(torch.randn(nb * Tw, 50, 4, generator=g)).to(DEV)
    A3 = rnd(torch.randn(nb * Tw, 200, generator=g)).to(DEV)
    B3 = rnd(torch.randn(nb * T, 175, generator=g)).to(DEV)
    want1 = A1.double().T @ B1.double()
    h = cell[:, :, 2].double().view(nb, Tw, 50)
    hprev = torch.zeros_like(h)
    hprev[:, 1:] = h[:, :-1]
    want2 = A2.double().T @ hprev.reshape(nb * Tw, 50)
    want3 = A3.double().view(nb, Tw, 200)[:, :T].reshape(nb * T, 200).T @ B3.double()
    probs = [(A1, K1, ptr(B1), 175, 1, 0, 0, 0, 0, 175, 256), (A2, nb * Tw, ptr(cell, 2), 200, 4, Tw, Tw, Tw, 1, 50, 96),
             (A3, nb * T, ptr(B3), 175, 1, T, Tw, T, 0, 175, 64)]
    arr = (S2GGemm * len(probs))()
    outs, prefix, blocks = [], [], 0
    for i, (A, K, Bp, sbk, sbn, segT, a_seg, b_seg, shift, N, kchunk) in enumerate(probs):
        ns = (K + kchunk - 1) // kchunk
        out = torch.full((ns, 200, N), float("nan"), device=DEV)
        outs.append(out)
        arr[i] = S2GGemm(ptr(A), 1, 200, Bp, sbk, sbn, ptr(out), N, 200 * N, None, 200, N, K, kchunk, segT, a_seg,
                         b_seg, shift, 0, 0, 1.0)
        prefix.append(blocks)
        blocks += ns * ((N + 63) // 64)          # a workgroup owns all 200 rows of 64 columns over one K-chunk
    prefix.append(blocks)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    pre = torch.tensor(prefix, dtype=torch.int32, device=DEV)
    call(fn, ptr(raw), ptr(pre), len(probs), blocks, 0, stream_ptr())
    torch.cuda.synchronize()
    for out, want in zip(outs, (want1, want2, want3)):
        assert rel(out.double().sum(0), want) < tol

    # K-chunk-first block order: every problem cut into the same number of chunks, prefix counts tiles
    K = nb * Tw
    probs = [(A2, K, ptr(cell, 2), 200, 4, Tw, Tw, Tw, 1, 50), (A2, K, ptr(B1), 175, 1, 0, 0, 0, 0, 175),
             (A3, K, ptr(cell, 1), 200, 4, 0, 0, 0, 0, 50)]
    kchunk = 64
    nz = (K + kchunk - 1) // kchunk
    arr = (S2GGemm * len(probs))()
    outs, prefix, tiles = [], [], 0
    for i, (A, K_, Bp, sbk, sbn, segT, a_seg, b_seg, shift, N) in enumerate(probs):
        out = torch.full((nz, 200, N), float("nan"), device=DEV)
        outs.append(out)
        arr[i] = S2GGemm(ptr(A), 1, 200, Bp, sbk, sbn, ptr(out), N, 200 * N, None, 200, N, K_, kchunk, segT, a_seg,
                         b_seg, shift, 0, 0, 1.0)
        prefix.append(tiles)
        tiles += (N + 63) // 64
    prefix.append(tiles)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    pre = torch.tensor(prefix, dtype=torch.int32, device=DEV)
    call(fn, ptr(raw), ptr(pre), len(probs), tiles * nz, nz, stream_ptr())
    torch.cuda.synchronize()
    wants = (want2, A2.double().T @ B1.double()[:K], A3.double().T @ cell[:, :, 1].double())
    for out, want in zip(outs, wants):
        assert rel(out.double().sum(0), want) < tol


def test_dropout_mask_matches_torch():
    from s2g.lib import call
    from s2g.ops import ptr, stream_ptr
    rows, p = 4099, 0.3
    u = torch.rand(rows, 52, generator=common.gen(12)).to(DEV)
    want = (u >= p).float() / (1.0 - p)
    want[:, 50:] = 0.0
    call("s2g_dropout_mask", ptr(u), rows, p, stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(u, want)


def test_gp_norm_penalty_proj_matches_formula():
    """pen_b = (|g_b| - 1)^2 with |g_b|^2 = sum_t delta_t . pv_t, and pv_b scaled by
    lambda * 2 (n_b - 1) / (n_batch n_b) in place (wgan_gp.py:255-256 and the seed of the tangent sweep)."""
    from s2g.lib import call
    from s2g.ops import ptr, stream_ptr
    B, T, lam, nb = 5, 37, 10.0, 7.0
    g = common.gen(13)
    W = torch.randn(200, 175, generator=g) * 0.2
    delta = (torch.randn(B * T, 200, generator=g) * 0.05).to(DEV)
    M2 = (W @ W.T).to(DEV)
    pv = (delta @ M2).contiguous()
    pv0 = pv.clone()
    pen = torch.empty(B, device=DEV)
    norms = torch.empty(B, device=DEV)
    call("s2g_gp_norm_penalty_proj", ptr(delta), ptr(pv), B, T * 200, lam, nb, ptr(pen), ptr(norms), stream_ptr())
    torch.cuda.synchronize()
    gfull = (delta.double().cpu() @ W.double()).view(B, -1)
    n = gfull.norm(dim=1)
    assert rel(norms.cpu(), n) < TOL
    assert rel(pen.cpu(), (n - 1) ** 2) < TOL
    coef = lam * 2 * (n - 1) / (nb * n)
    assert rel(pv.cpu().view(B, -1), pv0.double().cpu().view(B, -1) * coef[:, None]) < TOL
    # the reduction alone (four partial sums per sample, pv untouched) and the values from the partials
    part = torch.empty(B, 4, device=DEV)
    pen2, norms2 = torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    call("s2g_gp_dots", ptr(delta), ptr(pv0), B, T * 200, ptr(part), stream_ptr())
    call("s2g_gp_penalty_final", ptr(part), B, 0.0, 1.0, ptr(pen2), ptr(norms2), None, stream_ptr())
    torch.cuda.synchronize()
    assert rel(part.double().sum(1).cpu(), n ** 2) < TOL
    assert rel(norms2.cpu(), n) < TOL
    assert rel(pen2.cpu(), (n - 1) ** 2) < TOL
    assert rel(norms2.cpu(), ((part[:, 0] + part[:, 1]) + (part[:, 2] + part[:, 3])).clamp_min(0).sqrt().cpu()) < 1e-6


def test_reduce_slabs_multi_matches_single():
    from s2g.lib import S2GReduce, call
    from s2g.ops import ptr, stream_ptr
    g = common.gen(14)
    jobs = []
    for cols, nslab in ((175, 37), (50, 128), (50, 5)):
        slabs = torch.randn(nslab, 200 * cols, generator=g).to(DEV)
        jobs.append((slabs, cols, nslab, torch.empty(200, cols, device=DEV), torch.empty(200, cols, device=DEV)))
    arr = (S2GReduce * len(jobs))()
    for k, (slabs, cols, nslab, out_m, out_s) in enumerate(jobs):
        arr[k] = S2GReduce(ptr(slabs), 200 * cols, ptr(out_m), nslab, 200, cols)
        call("s2g_reduce_slabs", ptr(slabs), 200 * cols, nslab, 200, cols, ptr(out_s), 1, 0, 1.0, stream_ptr())
    call("s2g_reduce_slabs_multi", arr, len(jobs), 1, 0, 1.0, stream_ptr())
    torch.cuda.synchronize()
    perm = torch.tensor([(j & 3) * 50 + (j >> 2) for j in range(200)])
    for slabs, cols, nslab, out_m, out_s in jobs:
        assert torch.equal(out_m, out_s)
        want = torch.empty(200, cols, dtype=torch.float64)
        want[perm] = slabs.double().cpu().sum(0).view(200, cols)
        assert rel(out_m.cpu(), want) < 1e-5


@pytest.mark.parametrize("B", [1, 5, 64, 257])
def test_critic_head_matches_separate_kernels(B):
    """The one-launch head agrees with head forward + critic loss + head backward launched one by one
    (fc(h_T), wgan_gp.py:221; loss terms, wgan_gp.py:299-313)."""
    from s2g.lib import call
    from s2g.ops import ptr, stream_ptr
    g = common.gen(15 + B)
    hT = torch.randn(3 * B, 50, generator=g).to(DEV)
    W = (torch.randn(3, 50, generator=g) * 0.3).to(DEV)
    bias = torch.randn(3, generator=g).to(DEV)
    labels = torch.randint(0, 2, (B,), generator=g).to(DEV)
    inv_n, inv_wsum, w0, w1 = 1.0 / B, 1.0 / float((1.0 + labels.float()).sum()), 1.0, 2.0
    st = stream_ptr()

    def bufs():
        return [torch.full(s, float("nan"), device=DEV) for s in ((3 * B, 3), (3 * B, 3), (3,), (3 * B, 50), (3, 50), (3,))]

    out, dout, losses, dhT, dW, db = bufs()
    call("s2g_head_fwd", ptr(hT), ptr(W), ptr(bias), ptr(out), 3 * B, 3, st)
    call("s2g_critic_loss", ptr(out), ptr(labels), B, inv_n, inv_wsum, w0, w1, ptr(dout), ptr(losses), st)
    call("s2g_head_bwd_dh", ptr(dout), ptr(W), ptr(dhT), 3 * B, 3, st)
    call("s2g_head_bwd_w", ptr(dout), ptr(hT), ptr(dW), ptr(db), 2 * B, 3, 0, st)
    out2, dout2, losses2, dhT2, dW2, db2 = bufs()
    call("s2g_critic_head", ptr(hT), ptr(W), ptr(bias), ptr(labels), B, inv_n, inv_wsum, w0, w1, ptr(out2),
         ptr(dout2), ptr(losses2), ptr(dhT2), ptr(dW2), ptr(db2), st)
    torch.cuda.synchronize()
    assert rel(out2, hT.double() @ W.double().T + bias.double()) < 1e-5
    for a, b in ((out2, out), (dout2, dout), (losses2, losses), (dhT2, dhT), (dW2, dW), (db2, db)):
        assert torch.isfinite(a).all()
        assert rel(a, b) < 1e-5


@pytest.mark.parametrize("B,F", [(64, 256), (64, 512), (7, 50), (1, 33)])
def test_bn_fused_matches_two_kernel_form(B, F):
    """One-launch BatchNorm(eps=0.8)+LeakyReLU forward/backward (wgan_gp.py:150,156-157) against the
    statistics + apply pair that data-parallel runs use, and against torch."""
    from s2g.lib import call
    from s2g.ops import ptr, stream_ptr
    g = common.gen(30 + B)
    x = torch.randn(B, F, generator=g).to(DEV) * 2 + 0.5
    gamma = (torch.rand(F, generator=g) + 0.5).to(DEV)
    beta = torch.randn(F, generator=g).to(DEV)
    dy = torch.randn(B, F, generator=g).to(DEV)
    st = stream_ptr()
    outs = []
    for fused in (False, True):
        rm, rv = torch.zeros(F, device=DEV), torch.ones(F, device=DEV)
        y, xhat, inv = torch.empty_like(x), torch.empty_like(x), torch.empty(F, device=DEV)
        dgamma, dbeta, dx = torch.empty(F, device=DEV), torch.empty(F, device=DEV), torch.empty_like(x)
        if fused:
            call("s2g_bn_lrelu_fwd_fused", ptr(x), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.8, 0.1, 1, ptr(y),
                 ptr(xhat), ptr(inv), B, F, st)
            call("s2g_bn_lrelu_bwd_fused", ptr(y), ptr(dy), ptr(xhat), ptr(gamma), ptr(inv), ptr(dgamma), ptr(dbeta),
                 ptr(dx), B, F, st)
        else:
            stats, sums, d = torch.empty(2 * F, device=DEV), torch.empty(2 * F, device=DEV), dy.clone()
            call("s2g_bn_stats", ptr(x), ptr(stats), B, F, st)
            call("s2g_bn_lrelu_apply", ptr(x), ptr(stats), float(B), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.8, 0.1,
                 1, ptr(y), ptr(xhat), ptr(inv), B, F, st)
            call("s2g_bn_lrelu_bwd_stats", ptr(y), ptr(d), ptr(xhat), ptr(sums), B, F, st)
            call("s2g_bn_bwd_apply", ptr(d), ptr(xhat), ptr(sums), float(B), ptr(gamma), ptr(inv), ptr(dx), B, F, st)
            dbeta, dgamma = sums[:F].clone(), sums[F:].clone()
        torch.cuda.synchronize()
        outs.append((y, xhat, inv, rm, rv, dgamma, dbeta, dx))
    for a, b in zip(*outs):
        assert torch.isfinite(b).all()
        assert rel(b, a) < 1e-5
    # torch on the same numbers (float64)
    xd = x.double().cpu().requires_grad_(True)
    bn = torch.nn.BatchNorm1d(F, 0.8).double()
    with torch.no_grad():
        bn.weight.copy_(gamma.double().cpu()); bn.bias.copy_(beta.double().cpu())
    if B > 1:
        yt = torch.nn.functional.leaky_relu(bn(xd), 0.2)
        yt.backward(dy.double().cpu())
        y, _, _, rm, rv, dgamma, dbeta, dx = outs[1]
        for got, want in ((y, yt), (rm, bn.running_mean), (rv, bn.running_var), (dgamma, bn.weight.grad),
                          (dbeta, bn.bias.grad), (dx, xd.grad)):
            assert rel(got.cpu(), want.detach()) < TOL


def test_embed_mul_bwd_matches_torch():
    from s2g.lib import call
    from s2g.ops import ptr, stream_ptr
    g = common.gen(41)
    for B in (64, 5, 1):
        z = torch.randn(B, 100, generator=g).to(DEV)
        dout = torch.randn(B, 100, generator=g).to(DEV)
        labels = torch.randint(0, 2, (B,), generator=g).to(DEV)
        demb = torch.full((2, 100), float("nan"), device=DEV)
        call("s2g_embed_mul_bwd", ptr(z), ptr(labels), ptr(dout), ptr(demb), B, 100, 2, stream_ptr())
        torch.cuda.synchronize()
        want = torch.zeros(2, 100, dtype=torch.float64)
        want.index_add_(0, labels.cpu(), (z.double() * dout.double()).cpu())
        assert (demb.double().cpu() - want).abs().max() < 1e-4


@pytest.mark.parametrize("rows,B", [(32768, 64), (1000, 5), (3, 1)])
def test_bias_sums_matches_torch(rows, B):
    """bias_ih = bias_hh gradient = column sums of the gate gradients in torch row order (autograd of the
    b_ih + b_hh terms of nn.LSTM, wgan_gp.py:189,317), plus the fc.weight[0] part of the penalty."""
    from s2g.lib import call
    from s2g.ops import ptr, stream_ptr
    g = common.gen(50 + B)
    x0 = torch.randn(rows, 200, generator=g).to(DEV)
    x1 = torch.randn(rows, 200, generator=g).to(DEV)
    hd = torch.randn(B, 50, generator=g).to(DEV)
    part = torch.empty(2 * ((rows + 511) // 512) * 200, device=DEV)
    perm = torch.tensor([(j & 3) * 50 + (j >> 2) for j in range(200)])
    base = torch.randn(4, 200, generator=g)
    fcw = torch.randn(3, 50, generator=g)
    for accumulate in (0, 1):
        b = [base[k].clone().to(DEV) for k in range(4)]
        f = fcw.clone().to(DEV)
        call("s2g_bias_sums", ptr(x0), ptr(x1), rows, ptr(part), ptr(hd) if accumulate else None, B, ptr(b[0]),
             ptr(b[1]), ptr(b[2]), ptr(b[3]), ptr(f), accumulate, stream_ptr())
        torch.cuda.synchronize()
        for layer, x in enumerate((x0, x1)):
            want = torch.zeros(200, dtype=torch.float64)
            want[perm] = x.double().cpu().sum(0)
            if accumulate:
                want = want + base[2 * layer].double()
            assert torch.equal(b[2 * layer], b[2 * layer + 1])
            assert (b[2 * layer].double().cpu() - want).abs().max() < 1e-3 * max(1.0, want.abs().max().item())
        want_f = fcw.double().clone()
        if accumulate:
            want_f[0] += hd.double().cpu().sum(0)
        assert (f.double().cpu() - want_f).abs().max() < 1e-4


def test_cu_masked_stream_runs_kernels():
    from s2g import lib
    from s2g.lib import call
    from s2g.ops import ptr
    L = lib.get()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(n_cu * 3 // 4):
        mask[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    rc = L.s2g_stream_create_cu_mask(mask, words, ctypes.byref(h))
    if rc != 0:
        pytest.skip(f"runtime refused the CU mask (rc {rc})")
    u = torch.full((8, 52), 0.9, device=DEV)
    torch.cuda.synchronize()
    call("s2g_dropout_mask", ptr(u), 8, 0.5, h.value)
    st = torch.cuda.ExternalStream(h.value, device=torch.device(DEV))
    st.synchronize()
    assert float(u[0, 0]) == 2.0 and float(u[0, 51]) == 0.0
    assert L.s2g_stream_destroy(h) == 0
    assert L.s2g_stream_create_cu_mask(None, words, ctypes.byref(h)) == -1


@pytest.mark.parametrize("B,N,K", [(64, 89600, 512), (8, 4208, 512), (13, 4112, 128), (12, 4099, 77)])
def test_linear_small_batch_matches_torch(B, N, K):
    """y = x W^T + b through the weight-streaming kernel (the last shape falls back to the generic GEMM)."""
    from s2g import ops
    g = common.gen(31)
    x = torch.randn(B, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    y = torch.full((B, N), float("nan"), device=DEV)
    ops.linear_fwd_small_batch(x, W, b, y)
    torch.cuda.synchronize()
    assert rel(y, x.double() @ W.double().T + b.double()) < 1e-5


def test_adam_matches_reference_golden(golden):
    """s2g_adam against torch.optim.Adam(lr=2e-4, betas=(0.5, 0.999)) as recorded from the reference's
    optimizer (g6): three consecutive steps, gradients spanning three decades; the tensor whose gradient
    is None at step 1 is skipped by torch (no moment update, no step-count increment), which the host
    mirrors by not launching for it and keeping a step count per tensor."""
    from s2g import ops
    G = golden("g6_adam.pt")
    state = {}
    for name, w0 in (("w", G["w0"]), ("w2", G["w20"])):
        n = (w0.numel() + 3) // 4 * 4                  # 16-byte granularity of the kernel's accesses
        buf = torch.zeros(4, n, device=DEV)             # p, g, m, v (16-byte aligned rows)
        buf[0, :w0.numel()] = w0.reshape(-1).to(DEV)
        state[name] = [buf, 0, w0.numel()]
    for s in range(3):
        for name, key in (("w", f"gw{s}"), ("w2", f"gw2{s}")):
            buf, t, n = state[name]
            if key not in G:
                continue                                 # grad is None: torch skips the tensor
            buf[1, :n] = G[key].reshape(-1).to(DEV)
            state[name][1] = t = t + 1
            ops.adam(buf[0], buf[1], buf[2], buf[3], 2e-4, 0.5, 0.999, 1e-8, t)
        torch.cuda.synchronize()
        for name, key in (("w", f"w{s + 1}"), ("w2", f"w2{s + 1}")):
            buf, _, n = state[name]
            want = G[key].reshape(-1)
            got = buf[0, :n].cpu()
            assert rel(got, want) < 1e-6, (name, s)
            # the UPDATE itself (parameters hide a wrong step behind their norm): p_new - p_old
            prev = G[key[:-1] + str(s)].reshape(-1)
            if float((want - prev).abs().max()) > 0:
                assert rel(got - prev, want - prev) < 1e-3, (name, s)


def test_loss_kernels_match_reference_golden(golden):
    """s2g_critic_loss / s2g_generator_loss / s2g_critic_head against CrossEntropyLoss(weight=[1,2]) applied
    to softmax probabilities (the reference's double softmax, wgan_gp.py:268,299,310,338) as recorded in g5:
    loss value and d loss / d logits."""
    from s2g.lib import call
    from s2g.ops import ptr, stream_ptr
    G = golden("g5_ce.pt")
    logits, labels = G["logits"], G["labels"]
    B = labels.numel()
    wsum = float(torch.tensor([1.0, 2.0])[labels].sum())
    g = common.gen(71)
    val = torch.randn(3 * B, generator=g)
    out = torch.zeros(3 * B, 3)
    out[:, 0] = val
    out[:B, 1:] = logits
    out[B:2 * B, 1:] = logits.flip(0)                  # a different set of logits for the fake third
    lab = labels.to(DEV)
    o = out.to(DEV)
    dout = torch.full((3 * B, 3), float("nan"), device=DEV)
    losses = torch.full((4,), float("nan"), device=DEV)
    call("s2g_critic_loss", ptr(o), ptr(lab), B, 1.0 / B, 1.0 / wsum, 1.0, 2.0, ptr(dout), ptr(losses), stream_ptr())
    torch.cuda.synchronize()
    crit = torch.nn.CrossEntropyLoss(torch.tensor([1.0, 2.0]))
    lg2 = logits.flip(0).clone().requires_grad_(True)
    ce2 = crit(torch.softmax(lg2, -1), labels)
    (dl2,) = torch.autograd.grad(ce2, lg2)
    d = dout.cpu()
    assert abs(float(losses[0]) + float(val[:B].mean())) < 1e-6
    assert abs(float(losses[1]) - float(val[B:2 * B].mean())) < 1e-6
    assert abs(float(losses[2]) - 0.5 * (float(G["loss"]) + float(ce2))) < 1e-6
    assert rel(d[:B, 1:], 0.5 * G["dlogits"]) < 1e-5
    assert rel(d[B:2 * B, 1:], 0.5 * dl2) < 1e-5
    assert torch.all(d[:B, 0] == -1.0 / B) and torch.all(d[B:2 * B, 0] == 1.0 / B)
    assert torch.equal(d[2 * B:], torch.tensor([[1.0, 0.0, 0.0]]).repeat(B, 1))
    # generator pass: -mean(validity) + 0.5 * (0.5 * CE)   (wgan_gp.py:338-340)
    og = out[:B].contiguous().to(DEV)
    dg = torch.full((B, 3), float("nan"), device=DEV)
    lg = torch.full((4,), float("nan"), device=DEV)
    call("s2g_generator_loss", ptr(og), ptr(lab), B, 1.0 / B, 1.0 / wsum, 1.0, 2.0, ptr(dg), ptr(lg), stream_ptr())
    torch.cuda.synchronize()
    assert abs(float(lg[0]) + float(val[:B].mean())) < 1e-6
    assert abs(float(lg[1]) - 0.25 * float(G["loss"])) < 1e-6
    assert rel(dg.cpu()[:, 1:], 0.25 * G["dlogits"]) < 1e-5


@pytest.mark.parametrize("M,N,K", [(32768, 200, 175), (32768, 200, 200), (4096, 175, 200), (130, 175, 200), (5, 33, 7)])
def test_gemm_tall_bf16_matches_bf16_rounded_reference(M, N, K):
    """s2g_gemm_tall_bf16 = (A rounded to bf16) @ (B rounded to bf16) accumulated in fp32: against a float64
    product of the ROUNDED operands it must be fp32-accurate (the rounding is the only approximation), and
    against the unrounded product within the 2e-2 the reduced-precision path is entitled to."""
    from s2g import ops
    from s2g.ops import ptr
    g = common.gen(91)
    A = torch.randn(M, K, generator=g).to(DEV)
    B = (torch.randn(K, N, generator=g) * 0.1).to(DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm_tall(ptr(A), K, ptr(B), N, ptr(C), N, M, N, K, bf16=True)
    torch.cuda.synchronize()
    Ar, Br = A.bfloat16().double(), B.bfloat16().double()
    assert rel(C, Ar @ Br) < 1e-5
    assert rel(C, A.double() @ B.double()) < 2e-2


def test_gemm_tall_bf16_strided_operands():
    """padded leading dimensions, as the critic uses them (W0p rows are padded to 176 floats)"""
    from s2g import ops
    from s2g.ops import ptr
    g = common.gen(92)
    M, N, K = 1000, 175, 200
    A = torch.randn(M, 208, generator=g).to(DEV)
    B = torch.zeros(K, 176)
    B[:, :N] = torch.randn(K, N, generator=g) * 0.1
    B = B.to(DEV)
    C = torch.full((M, 180), float("nan"), device=DEV)
    ops.gemm_tall(ptr(A), 208, ptr(B), 176, ptr(C), 180, M, N, K, bf16=True)
    torch.cuda.synchronize()
    want = A[:, :K].bfloat16().double() @ B[:, :N].bfloat16().double()
    assert rel(C[:, :N], want) < 1e-5
    assert torch.isnan(C[:, N:]).all()          # nothing written outside the N columns


@pytest.mark.parametrize("B,N,K", [(64, 89600, 512), (5, 1000, 256), (16, 4112, 512)])
def test_linear_wgrad_adam_matches_unfused(B, N, K):
    """The fused weight-gradient + Adam step of the generator's output layer against torch.optim.Adam on the
    float64 gradient dy^T a, two updates in a row (the second with moments that are no longer zero); the
    optional gradient output against the same product.  N = 1000 and 4112 leave a partial / odd last block."""
    from s2g import lib, ops
    g = common.gen(77)
    W0 = (torch.randn(N, K, generator=g) * 0.05)
    Wr = W0.double().clone().requires_grad_(True)
    opt = torch.optim.Adam([Wr], lr=2e-4, betas=(0.5, 0.999), eps=1e-8)
    W = W0.to(DEV)
    m, v = torch.zeros_like(W), torch.zeros_like(W)
    grad = torch.full_like(W, float("nan"))
    for step in (1, 2):
        dy = (torch.randn(B, N, generator=g) * 1e-3)
        a = torch.randn(B, K, generator=g)
        gref = dy.double().t() @ a.double()
        Wr.grad = gref.clone()
        W_before = Wr.detach().clone()
        opt.step()
        Wd_before = W.clone()
        ops.linear_wgrad_adam(dy.to(DEV), a.to(DEV), W, m, v, 2e-4, 0.5, 0.999, 1e-8, step,
                              grad=grad if step == 1 else None, workgroups=0 if step == 1 else 192)
        torch.cuda.synchronize()
        if step == 1:
            assert rel(grad.cpu(), gref) < 1e-5
        st = opt.state[Wr]
        assert rel(m.cpu(), st["exp_avg"]) < 1e-5
        assert rel(v.cpu(), st["exp_avg_sq"]) < 1e-4      # 1 - 0.999 in fp32 is 1.00005e-3
        # the UPDATE, not the parameter (an Adam step moves an element by ~lr); where |g| is within a few orders
        # of eps * sqrt(v)'s rounding the step is ill-conditioned, so compare where the gradient is not tiny
        firm = gref.abs() >= 1e-6
        du, du_ref = (W.cpu().double() - Wd_before.cpu().double())[firm], (Wr.detach() - W_before)[firm]
        assert float((du - du_ref).norm() / du_ref.norm()) < 1e-3
    L = lib.get()
    s = torch.cuda.current_stream().cuda_stream
    p = W.data_ptr()
    assert L.s2g_linear_wgrad_adam(p, p, p, p, p, None, N, 65, K, 2e-4, 0.5, 0.999, 1e-8, 1, 0, s) == -1     # batch > 64
    assert L.s2g_linear_wgrad_adam(p, p, p, p, p, None, N, 8, 128, 2e-4, 0.5, 0.999, 1e-8, 1, 0, s) == -1    # K = 256 or 512
    assert L.s2g_linear_wgrad_adam(p, p, p + 4, p, p, None, N, 8, K, 2e-4, 0.5, 0.999, 1e-8, 1, 0, s) == -1  # alignment
    assert L.s2g_linear_wgrad_adam(p, p, p, p, p, None, N, 8, K, 2e-4, 0.5, 0.999, 1e-8, 0, 0, s) == -1      # step >= 1


@pytest.mark.parametrize("M", [64, 5])
def test_small_linear_grads_match_torch(M):
    """One launch for the weight and bias gradients of the generator's three small Linear layers against the
    float64 products dy^T x and column sums (shapes of wgan_gp.py:147-156)."""
    from s2g import lib, ops
    g = common.gen(91)
    jobs, refs = [], []
    for n_out, n_in, with_bias in ((512, 256, True), (256, 128, True), (128, 100, False)):
        dy, x = torch.randn(M, n_out, generator=g), torch.randn(M, n_in, generator=g)
        dW = torch.full((n_out, n_in), float("nan"), device=DEV)
        db = torch.full((n_out,), float("nan"), device=DEV) if with_bias else None
        jobs.append((dy.to(DEV), x.to(DEV), dW, db))
        refs.append((dy.double().t() @ x.double(), dy.double().sum(0)))
    ops.small_linear_grads(jobs)
    torch.cuda.synchronize()
    for (dy, x, dW, db), (rW, rb) in zip(jobs, refs):
        assert rel(dW.cpu(), rW) < 1e-5
        if db is not None:
            assert rel(db.cpu(), rb) < 1e-5
    with pytest.raises(lib.S2GError):
        ops.small_linear_grads([(jobs[0][0], jobs[1][1], jobs[0][2], None)])      # dW shape does not match x


@pytest.mark.parametrize("B,T,Tw,V,W", [(5, 37, 41, 175, 200), (64, 512, 512, 175, 200), (3, 16, 16, 23, 40)])
def test_bucketed_row_sums_transposed_match_index_add(B, T, Tw, V, W):
    """s2g_bucket_rows_by_token_t: out[c][v] = sum over live rows with token v of x[row][c] for a width that is no
    multiple of 32, rows laid out Tw >= T per sequence, ragged lengths -- the real third of dW_ih0 -- against
    index_add in float64."""
    from s2g.lib import call, get
    from s2g.ops import ptr, stream_ptr
    g = common.gen(77 + B)
    tokens = torch.randint(0, V, (B, T), generator=g, dtype=torch.int32)
    lengths = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
    lengths[0] = T
    x = torch.randn(B * Tw + 3, W, generator=g)              # a few rows beyond the last sequence: never read as live
    want = torch.zeros(V, W, dtype=torch.float64)
    for b in range(B):
        n = int(lengths[b])
        want.index_add_(0, tokens[b, :n].long(), x[b * Tw:b * Tw + n].double())
    out = torch.full((W, V), float("nan"), device=DEV)
    part = torch.empty(int(get().s2g_embed_grad_tokens_scratch(B * T, V)) * ((W + 31) // 32), device=DEV)
    tok_d, len_d, x_d = tokens.to(DEV), lengths.to(DEV), x.to(DEV)
    call("s2g_bucket_rows_by_token_t", ptr(tok_d), ptr(len_d), T, B * T, V, ptr(x_d), W, Tw, W, ptr(out), V, ptr(part),
         stream_ptr())
    torch.cuda.synchronize()
    assert rel(out.double().cpu(), want.T) < 1e-5


@pytest.mark.parametrize("fn", ["s2g_gemm_grouped", "s2g_gemm_grouped_bf16"])
def test_gemm_grouped_row_scale(fn):
    """S2GGemm.bscale: one factor per sequence on the rows of B -- the weighted Gram product
    sum_rows coef[seq(row)] A[row]^T A[row] of the penalty's first-order factor."""
    from s2g.lib import S2GGemm, call
    from s2g.ops import ptr, stream_ptr
    g = common.gen(91)
    nb, T = 7, 53
    rnd = (lambda t: t.bfloat16().float()) if fn.endswith("bf16") else (lambda t: t)
    A = rnd(torch.randn(nb * T, 200, generator=g)).to(DEV)
    # factors that are exact in bf16 too (the kernel multiplies the fp32 row before rounding it)
    coef = torch.tensor([0.5, -2.0, 1.0, 4.0, 0.25, -1.0, 8.0]).to(DEV)
    want = (A.double() * coef.double().repeat_interleave(T).unsqueeze(1)).T @ A.double()
    kchunk = 64
    K = nb * T
    nz = (K + kchunk - 1) // kchunk
    out = torch.full((nz, 200, 200), float("nan"), device=DEV)
    arr = (S2GGemm * 1)()
    arr[0] = S2GGemm(ptr(A), 1, 200, ptr(A), 200, 1, ptr(out), 200, 200 * 200, None, 200, 200, K, kchunk, T, T, T, 0, 0, 0,
                     1.0, ptr(coef))
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    pre = torch.tensor([0, 4], dtype=torch.int32, device=DEV)
    call(fn, ptr(raw), ptr(pre), 1, 4 * nz, nz, stream_ptr())
    torch.cuda.synchronize()
    assert rel(out.double().sum(0), want.T) < 1e-5
