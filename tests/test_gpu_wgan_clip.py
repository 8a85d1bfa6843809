"""The weight-clipping WGAN variant (reference wgan.py) on the HIP kernels: three critic updates
(CWGAN.train_step lines 174-198: D(real tokens), G(noise).detach(), D(fake), Wasserstein loss, Adam, clamp)
against what the reference's own classes produced (g12), and a larger ragged batch with inter-layer dropout
masks against the CPU restatement.  1e-3 relative, the fp32 bound."""
import pytest
import torch

import common
from conftest import assert_close, assert_update_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


def engines(seed, V=175, dropout=0.0):
    from s2g.wgan_clip import ClipCriticEngine, ClipGeneratorEngine
    pd, pg = common.wgan_clip_params(seed, V)
    D = ClipCriticEngine(V, 2, torch.device(DEV), dropout=dropout)
    G = ClipGeneratorEngine(V, 3, torch.device(DEV))
    D.load(pd); G.load(pg)
    return D, G, pd, pg


def test_three_critic_updates_match_reference_golden(golden):
    Gd = golden("g12_wgan_clip.pt")
    V, B, T = 175, 6, 40
    D, G, pd, _ = engines(1401, V)
    for it in range(3):
        tokens, labels, lengths = common.batch(1410 + it, B, T, vocab=V, min_len=4)
        g = common.gen(1420 + it)
        noise = torch.randint(0, V, (B, 100), generator=g)
        U = torch.rand((B, 100, V), generator=g)
        fake = G.forward(noise.to(torch.int32).to(DEV), labels.to(DEV), 0.5, U=U.to(DEV))
        loss = D.critic_step(tokens.to(torch.int32).to(DEV).contiguous(), labels.to(DEV), lengths.to(torch.int32).to(DEV),
                             int(lengths.max()), fake, 2e-4, 0.5, 0.999, 0.1, masks_real=None, masks_fake=None)
        torch.cuda.synchronize()
        assert abs(float(loss) - float(Gd[f"d_loss{it}"])) < TOL * max(1.0, abs(float(Gd[f"d_loss{it}"])))
        for n, gv in D.grads.views.items():
            assert_close(gv, Gd[f"grad{it}.{n}"], TOL, f"grad{it} {n}", atol=1e-9)
    for n, v in D.params.views.items():
        assert_close(v, Gd[f"D.{n}"], TOL, n)
        if n != "linear.bias":          # d(loss)/d(linear.bias) = -1 + 1 = 0: the reference never moves it
            assert_update_close(v, pd[n], Gd[f"D.{n}"], TOL, n)
        assert float(v.abs().max()) <= 0.1 + 1e-7


def test_ragged_batch_with_dropout_masks_matches_oracle():
    from oracle import wgan_clip_cpu as O
    V, B, T = 175, 16, 96
    D, G, pd, pg = engines(1501, V, dropout=0.3)
    tokens, labels, lengths = common.batch(1510, B, T, vocab=V, min_len=2)
    lengths[0] = 70                                            # the longest sequence is shorter than the padded width
    lengths = torch.sort(lengths.clamp(max=70), descending=True).values
    g = common.gen(1520)
    noise = torch.randint(0, V, (B, 100), generator=g)
    U = torch.rand((B, 100, V), generator=g)
    m_real = (torch.rand((B, T, 64), generator=g) > 0.3).float() / 0.7
    m_fake = (torch.rand((B, 100, 64), generator=g) > 0.3).float() / 0.7
    pd_o = {k: v.clone() for k, v in pd.items()}
    opt = O.Adam(pd_o, 2e-4, 0.5, 0.999)
    d_ref, g_ref = O.critic_step(pd_o, opt, pg, tokens, labels, lengths, noise, U, 0.5, 0.1,
                                 masks_real=[m_real, None], masks_fake=[m_fake, None])
    fake = G.forward(noise.to(torch.int32).to(DEV), labels.to(DEV), 0.5, U=U.to(DEV))
    fake_ref = O.generator_forward(pg, noise, labels, U, 0.5)
    assert_close(fake, fake_ref, TOL, "generator sample")
    loss = D.critic_step(tokens.to(torch.int32).to(DEV).contiguous(), labels.to(DEV), lengths.to(torch.int32).to(DEV),
                         int(lengths.max()), fake, 2e-4, 0.5, 0.999, 0.1,
                         masks_real=[m_real.reshape(B * T, 64).to(DEV), None],
                         masks_fake=[m_fake.reshape(B * 100, 64).to(DEV), None])
    torch.cuda.synchronize()
    assert abs(float(loss) - float(d_ref)) < TOL * max(1.0, abs(float(d_ref)))
    for n, gv in D.grads.views.items():
        assert_close(gv, g_ref[n], TOL, f"grad {n}", atol=1e-9)
    for n, v in D.params.views.items():
        assert_close(v, pd_o[n], TOL, n)


def test_device_dropout_masks_have_the_right_rate():
    D, _, _, _ = engines(1601, dropout=0.3)
    m = D.dropout_masks(4096)
    torch.cuda.synchronize()
    assert m[-1] is None and tuple(m[0].shape) == (4096, 64)
    keep = (m[0] != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01 and abs(float(m[0].max()) - 1.0 / 0.7) < 1e-6


def test_module_surface_and_train_step_follow_the_reference(golden):
    """wgan.py's classes: state_dict keys of the reference, CWGAN.train_step = the reference's critic update on a
    one-batch loader (the first d_loss of g12), and the reference's AttributeError on a second batch."""
    import argparse
    import wgan as W
    Gd = golden("g12_wgan_clip.pt")
    V, B, T = 175, 6, 40
    pd, pg = common.wgan_clip_params(1401, V)
    G = W.Generator(V, 2, 3, softmax="gumbel_softmax")
    D = W.Discriminator(V, 2, 2, dropout=0.0)
    assert set(D.state_dict()) == set(pd) and set(G.state_dict()) == set(pg)
    G.load_state_dict(pg); D.load_state_dict(pd)
    opt = argparse.Namespace(lr=2e-4, beta1=0.5, beta2=0.999, clip_value=0.1, n_critic=5)
    model = W.CWGAN(G, D, opt)
    tokens, labels, lengths = common.batch(1410, B, T, vocab=V, min_len=4)
    g = common.gen(1420)
    noise = torch.randint(0, V, (B, 100), generator=g)
    U = torch.rand((B, 100, V), generator=g)
    real_randint, real_rand = torch.randint, torch.rand
    torch.randint = lambda *a, **k: noise.clone()
    torch.rand = lambda *a, **k: U.clone()
    try:
        d_loss, g_loss = model.train_step([(tokens, labels, lengths)], 0.5, False)
        assert abs(d_loss - float(Gd["d_loss0"])) < TOL and g_loss == 0.0
        out = D(tokens.to(DEV), labels.to(DEV), lengths)
        assert tuple(out.shape) == (B, 1) and torch.isfinite(out).all()
        with pytest.raises(AttributeError, match="loss"):
            model.train_step([(tokens, labels, lengths)] * 2, 0.5, False)
    finally:
        torch.randint, torch.rand = real_randint, real_rand
    for n, v in D.state_dict().items():            # the module's parameters ARE the engine's (two updates by now)
        assert float(v.abs().max()) <= 0.1 + 1e-7


@pytest.mark.parametrize("B,T,V", [(5, 37, 175), (128, 1024, 175)])
def test_embedding_gradients_match_index_add(B, T, V):
    """The two-stage embedding gradients (tokens and labels) against index_add in float64: live rows only,
    written and accumulated, at a small ragged shape and at the benchmark's shape (rows span many chunks)."""
    from s2g.lib import call, get
    from s2g.ops import ptr, stream_ptr
    g = common.gen(5)
    tokens = torch.randint(0, V, (B, T), generator=g, dtype=torch.int32)
    lengths = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
    lengths[0] = T
    labels = torch.randint(0, 2, (B,), generator=g)
    dx = torch.randn(B * T, 64, generator=g)
    live = (torch.arange(T)[None, :] < lengths[:, None]).reshape(-1)
    ref_e = torch.zeros(V, 32, dtype=torch.float64).index_add_(0, tokens.reshape(-1)[live].long(), dx[live, :32].double())
    per_seq = (dx[:, 32:].double() * live[:, None]).view(B, T, 32).sum(1)
    ref_l = torch.zeros(2, 32, dtype=torch.float64).index_add_(0, labels, per_seq)
    L = get()
    n_part = L.s2g_embed_grad_tokens_scratch(B * T, V)
    assert n_part >= V * 32 and n_part % (V * 32) == 0          # one [V, 32] table per chunk of rows
    part = torch.empty(max(n_part, B * 32), device=DEV)
    de = torch.full((V, 32), 0.5, device=DEV)
    dl = torch.full((2, 32), 0.25, device=DEV)
    args_t = (ptr(tokens.to(DEV)), ptr(lengths.to(DEV)), T, B * T, V, ptr(dx.to(DEV)))
    tk, ln, lb, dxd = tokens.to(DEV), lengths.to(DEV), labels.to(DEV), dx.to(DEV)
    for accumulate, base_e, base_l in ((0, 0.0, 0.0), (1, 1.0, 1.0)):
        call("s2g_embed_grad_tokens", ptr(tk), ptr(ln), T, B * T, V, ptr(dxd), ptr(de), accumulate, ptr(part), stream_ptr())
        call("s2g_embed_grad_labels", ptr(lb), ptr(ln), T, B, 2, ptr(dxd), ptr(dl), accumulate, ptr(part), stream_ptr())
        torch.cuda.synchronize()
        k = 1 + accumulate
        assert (de.cpu().double() - k * ref_e).abs().max() <= 1e-5 * ref_e.abs().max()
        assert (dl.cpu().double() - k * ref_l).abs().max() <= 1e-5 * ref_l.abs().max()


@pytest.mark.parametrize("B,T,V,W", [(5, 37, 175, 256), (128, 1024, 175, 256), (7, 50, 33, 64)])
def test_bucketed_row_sums_match_index_add(B, T, V, W):
    """s2g_bucket_rows_by_token / _by_label against index_add in float64 (live rows only; the benchmark's shape
    spans many row chunks and all eight column slices)."""
    from s2g.lib import call, get
    from s2g.ops import ptr, stream_ptr
    g = common.gen(6)
    tokens = torch.randint(0, V, (B, T), generator=g, dtype=torch.int32)
    lengths = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
    lengths[0] = T
    labels = torch.randint(0, 2, (B,), generator=g)
    x = torch.randn(B * T, W, generator=g)
    live = (torch.arange(T)[None, :] < lengths[:, None]).reshape(-1)
    ref = torch.zeros(V, W, dtype=torch.float64).index_add_(0, tokens.reshape(-1)[live].long(), x[live].double())
    xb = torch.randn(B, W, generator=g)
    ref_l = torch.zeros(2, W, dtype=torch.float64).index_add_(0, labels, xb.double())
    part = torch.empty(get().s2g_embed_grad_tokens_scratch(B * T, V) * (W // 32), device=DEV)
    out, out_l = torch.full((V, W), 7.0, device=DEV), torch.full((2, W), 7.0, device=DEV)
    tk, ln, xd, xbd, lb = tokens.to(DEV), lengths.to(DEV), x.to(DEV), xb.to(DEV), labels.to(DEV)
    call("s2g_bucket_rows_by_token", ptr(tk), ptr(ln), T, B * T, V, ptr(xd), W, W, ptr(out), ptr(part), stream_ptr())
    call("s2g_bucket_rows_by_label", ptr(lb), B, 2, ptr(xbd), W, ptr(out_l), stream_ptr())
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    assert (out_l.cpu().double() - ref_l).abs().max() <= 1e-5 * ref_l.abs().max()


@pytest.mark.parametrize("layers,use_mask", [(2, True), (2, False), (3, False)])
def test_stack_sweeps_match_the_per_layer_sweeps(layers, use_mask):
    """s2g_lstm64_stack_fwd (and, for two layers, s2g_lstm64_stack_bwd) against one s2g_lstm64_fwd / _bwd launch
    per layer with the batched products in between, on a ragged batch: same saved planes, outputs and gate
    gradients to 1e-5 (the products are summed in another order)."""
    import ctypes
    from s2g.lib import S2GLstm64Stack, S2GLstm64StackBwd, call
    from s2g import ops
    from s2g.ops import ptr, stream_ptr
    B, T, H, G4 = 9, 45, 64, 256
    g = common.gen(77 + layers)
    lengths = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
    lengths[0], lengths[1] = T, 1
    rows = B * T
    dev = torch.device(DEV)
    W = [dict(ih=(0.2 * torch.randn(G4, H, generator=g)).to(dev), hh=(0.2 * torch.randn(G4, H, generator=g)).to(dev),
              b=(0.1 * torch.randn(G4, generator=g)).to(dev)) for _ in range(layers)]
    x = torch.randn(rows, H, generator=g).to(dev)
    masks = [((torch.rand(rows, H, generator=g) > 0.3).float() / 0.7).to(dev) if use_mask and l + 1 < layers else None
             for l in range(layers)]
    ln = lengths.to(dev)
    # per-layer path
    ref, inp = [], x
    for l in range(layers):
        pin = torch.empty(rows, G4, device=dev)
        ops.linear_fwd(inp, W[l]["ih"], W[l]["b"], pin)
        gates, cst, hout = torch.empty(rows, G4, device=dev), torch.empty(rows, H, device=dev), torch.empty(rows, H, device=dev)
        hdrop = torch.empty(rows, H, device=dev) if masks[l] is not None else None
        call("s2g_lstm64_fwd", ptr(pin), ptr(W[l]["hh"]), ptr(ln), B, T, ptr(gates), ptr(cst), ptr(hout), ptr(masks[l]),
             ptr(hdrop), stream_ptr())
        ref.append(dict(gates=gates, cst=cst, hout=hout, hdrop=hdrop))
        inp = hdrop if hdrop is not None else hout
    # one sweep
    pin0 = torch.empty(rows, G4, device=dev)
    ops.linear_fwd(x, W[0]["ih"], W[0]["b"], pin0)
    st, got = S2GLstm64Stack(), []
    st.pin0 = ptr(pin0)
    for l in range(layers):
        o = dict(gates=torch.empty(rows, G4, device=dev), cst=torch.empty(rows, H, device=dev),
                 hout=torch.full((rows, H), 9.0, device=dev),
                 hdrop=torch.full((rows, H), 9.0, device=dev) if masks[l] is not None else None)
        st.whh[l], st.gates[l], st.cst[l], st.hout[l] = ptr(W[l]["hh"]), ptr(o["gates"]), ptr(o["cst"]), ptr(o["hout"])
        st.mask[l], st.hdrop[l] = ptr(masks[l]), ptr(o["hdrop"])
        if l:
            st.wih[l], st.bsum[l] = ptr(W[l]["ih"]), ptr(W[l]["b"])
        got.append(o)
    call("s2g_lstm64_stack_fwd", ctypes.byref(st), layers, ptr(ln), B, T, stream_ptr())
    torch.cuda.synchronize()
    live = (torch.arange(T)[None, :] < lengths[:, None]).reshape(-1).to(dev)
    for l in range(layers):
        for k in ("gates", "cst"):                      # saved planes: rows of live steps only
            assert_close(got[l][k][live], ref[l][k][live], 1e-5, f"layer {l} {k}", atol=1e-6)
        assert_close(got[l]["hout"], ref[l]["hout"], 1e-5, f"layer {l} hout", atol=1e-6)      # zeros beyond the length
        if masks[l] is not None:
            assert_close(got[l]["hdrop"], ref[l]["hdrop"], 1e-5, f"layer {l} hdrop", atol=1e-6)
    if layers != 2:
        return
    dh = torch.randn(rows, H, generator=g).to(dev)
    d1 = torch.empty(rows, G4, device=dev)
    call("s2g_lstm64_bwd", ptr(dh), ptr(W[1]["hh"]), ptr(ln), B, T, ptr(ref[1]["gates"]), ptr(ref[1]["cst"]), ptr(d1),
         stream_ptr())
    dinp = d1 @ W[1]["ih"]
    if masks[0] is not None:
        dinp = dinp * masks[0]
    d0 = torch.empty(rows, G4, device=dev)
    call("s2g_lstm64_bwd", ptr(dinp.contiguous()), ptr(W[0]["hh"]), ptr(ln), B, T, ptr(ref[0]["gates"]), ptr(ref[0]["cst"]),
         ptr(d0), stream_ptr())
    sb = S2GLstm64StackBwd()
    f = [torch.full((rows, G4), 9.0, device=dev) for _ in range(2)]
    bp = [torch.empty(B, G4, device=dev) for _ in range(2)]
    sb.dh_top, sb.wih1, sb.mask0 = ptr(dh), ptr(W[1]["ih"]), ptr(masks[0])
    for l in range(2):
        sb.whh[l], sb.gates[l], sb.cst[l], sb.delta[l], sb.bias_part[l] = (ptr(W[l]["hh"]), ptr(ref[l]["gates"]),
                                                                           ptr(ref[l]["cst"]), ptr(f[l]), ptr(bp[l]))
    call("s2g_lstm64_stack_bwd", ctypes.byref(sb), ptr(ln), B, T, stream_ptr())
    torch.cuda.synchronize()
    for l, d in ((1, d1), (0, d0)):
        assert_close(f[l], d, 1e-4, f"delta{l}", atol=1e-6)
        assert_close(bp[l], d.view(B, T, G4).sum(1), 1e-4, f"bias sums {l}", atol=1e-5)


@pytest.mark.parametrize("B,T", [(12, 64), (48, 384)])
def test_one_stream_and_overlapped_critic_steps_agree(B, T):
    """critic_step with everything on the caller's stream (overlap = False: the fake pass accumulates into the real
    pass's gradients, no side streams) against the default schedule (fake pass and weight-gradient products on side
    streams, two gradient buffers added at the end): same loss, gradients and updated weights to 1e-5, on a ragged
    batch with dropout masks.  Each schedule runs the FIRST step of a fresh engine (cold split-K scratch: the step
    allocates while the side streams still read the reverse sweep's outputs; the larger shape makes those
    buffers big enough that a block freed too early would be handed out again)."""
    V = 175
    tokens, labels, lengths = common.batch(1710, B, T, vocab=V, min_len=1)
    g = common.gen(1720)
    noise = torch.randint(0, V, (B, 100), generator=g)
    U = torch.rand((B, 100, V), generator=g)
    m_real = ((torch.rand((B * T, 64), generator=g) > 0.3).float() / 0.7).to(DEV)
    m_fake = ((torch.rand((B * 100, 64), generator=g) > 0.3).float() / 0.7).to(DEV)
    res = []
    for overlap in (False, True):
        D, G, _, _ = engines(1701, V, dropout=0.3)
        D.overlap = overlap
        fake = G.forward(noise.to(torch.int32).to(DEV), labels.to(DEV), 0.5, U=U.to(DEV))
        loss = D.critic_step(tokens.to(torch.int32).to(DEV).contiguous(), labels.to(DEV), lengths.to(torch.int32).to(DEV),
                             int(lengths.max()), fake, 2e-4, 0.5, 0.999, 0.1, masks_real=[m_real, None],
                             masks_fake=[m_fake, None])
        torch.cuda.synchronize()
        res.append((float(loss), {n: v.clone() for n, v in D.grads.views.items()},
                    {n: v.clone() for n, v in D.params.views.items()}))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * max(1.0, abs(res[0][0]))
    for n in res[0][1]:
        assert_close(res[1][1][n], res[0][1][n], 1e-5, f"grad {n}", atol=1e-9)
        assert_close(res[1][2][n], res[0][2][n], 1e-5, n)


def test_critic_update_at_full_size_matches_reference_golden(golden):
    """BASELINE's shape for this variant, B = 128, L = 1024, V = 175 (ragged lengths, no dropout): one critic
    update of the reference's own CWGAN.train_step (g14) against critic_step through the two-layer stack sweeps
    (s2g_k_lstm64_stack_fwd / _bwd over 1 024 steps), the token tables and the bucketed sums: loss, every
    gradient (norm-wise and element by element), the clamped parameters and what the Adam step moved."""
    Gd = golden("g14_wgan_clip_full_size.pt")
    V, B, T = 175, 128, 1024
    D, G, pd, _ = engines(1801, V)
    tokens, labels, lengths = common.batch(1810, B, T, vocab=V, min_len=64)
    g = common.gen(1820)
    noise = torch.randint(0, V, (B, 100), generator=g)
    U = torch.rand((B, 100, V), generator=g)
    fake = G.forward(noise.to(torch.int32).to(DEV), labels.to(DEV), 0.5, U=U.to(DEV))
    loss = D.critic_step(tokens.to(torch.int32).to(DEV).contiguous(), labels.to(DEV), lengths.to(torch.int32).to(DEV),
                         int(lengths.max()), fake, 2e-4, 0.5, 0.999, 0.1, masks_real=None, masks_fake=None)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(Gd["d_loss"])) < TOL * max(1.0, abs(float(Gd["d_loss"])))
    for n, gv in D.grads.views.items():
        assert_close(gv, Gd[f"grad.{n}"], TOL, f"grad {n}", atol=1e-9)
    for n, v in D.params.views.items():
        assert_close(v, Gd[f"D.{n}"], TOL, n)
        if n != "linear.bias":          # d(loss)/d(linear.bias) = -1 + 1 = 0: the reference never moves it
            assert_update_close(v, pd[n], Gd[f"D.{n}"], TOL, n)


def test_relaxed_embedding_and_attention_modules_match_reference_golden(golden):
    """RelaxedEmbedding.forward (both branches) and Attention.forward of wgan.py with their gradients (g15)."""
    import wgan as W
    Gd = golden("g15_wgan_modules.pt")
    V = 175
    pd, _ = common.wgan_clip_params(1901, V)
    emb = W.RelaxedEmbedding(V, 32).to(DEV)
    with torch.no_grad():
        emb.embed_layer.weight.copy_(pd["sequence_embedding.embed_layer.weight"])
    dy = Gd["emb.dy"].to(DEV)
    y = emb(Gd["emb.tok"].to(DEV))
    (gw,) = torch.autograd.grad(y, [emb.embed_layer.weight], dy)
    assert_close(y, Gd["emb.y_tok"], 1e-6, "lookup")
    assert_close(gw, Gd["emb.dW_tok"], 1e-5, "lookup d(table)", atol=1e-9)
    x = Gd["emb.soft"].to(DEV).requires_grad_(True)
    y = emb(x)
    gw, gx = torch.autograd.grad(y, [emb.embed_layer.weight, x], dy)
    assert_close(y, Gd["emb.y_soft"], TOL, "matmul branch")
    assert_close(gw, Gd["emb.dW_soft"], TOL, "matmul d(table)")
    assert_close(gx, Gd["emb.dx_soft"], TOL, "matmul d(x)")
    att = W.Attention(64).to(DEV)
    with torch.no_grad():
        att.attn.weight.copy_(pd["attention.attn.weight"]); att.attn.bias.copy_(pd["attention.attn.bias"])
        att.context_vector.weight.copy_(pd["attention.context_vector.weight"])
    hidden = Gd["att.hidden"].to(DEV).requires_grad_(True)
    lo = Gd["att.lstm_outputs"].to(DEV).requires_grad_(True)
    pooled, weights = att(hidden, lo)
    grads = torch.autograd.grad([pooled, weights], [hidden, lo, att.attn.weight, att.attn.bias, att.context_vector.weight],
                                [Gd["att.d_pooled"].to(DEV), Gd["att.d_weights"].to(DEV)])
    assert_close(pooled, Gd["att.pooled"], TOL, "pooled")
    assert_close(weights, Gd["att.weights"], TOL, "weights")
    for n, gv in zip(("d_hidden", "d_lstm_outputs", "d_attn_w", "d_attn_b", "d_context"), grads):
        assert_close(gv, Gd[f"att.{n}"], TOL, n)


def test_discriminator_forward_is_differentiable_and_train_anneals_like_the_reference(golden):
    """Discriminator.forward of wgan.py under autograd (token input: gradients of every parameter; relaxed input:
    also the input gradient) and CWGAN.train with tau annealing, against the reference's own run (g15)."""
    import argparse
    import contextlib
    import io
    import wgan as W
    Gd = golden("g15_wgan_modules.pt")
    V = 175
    pd, pg = common.wgan_clip_params(1901, V)
    D = W.Discriminator(V, 2, 2, dropout=0.0).to(DEV)
    D.load_state_dict(pd)
    tokens, labels, lengths = Gd["D.tokens"], Gd["D.labels"], Gd["D.lengths"]
    dval = Gd["D.dval"].to(DEV)
    val = D(tokens.to(DEV), labels.to(DEV), lengths)
    names = [n for n, _ in D.named_parameters()]
    gt = torch.autograd.grad(val, [q for _, q in D.named_parameters()], dval)
    assert_close(val, Gd["D.val_tok"], TOL, "D(tokens)")
    for n, gv in zip(names, gt):
        assert_close(gv, Gd[f"D.grad_tok.{n}"], TOL, f"token input, grad {n}", atol=1e-9)
    x = Gd["D.soft"].to(DEV).requires_grad_(True)
    full = torch.full((tokens.shape[0],), tokens.shape[1], dtype=torch.long)
    val2 = D(x, labels.to(DEV), full)
    gs = torch.autograd.grad(val2, [x] + [q for _, q in D.named_parameters()], dval)
    assert_close(val2, Gd["D.val_soft"], TOL, "D(relaxed)")
    assert_close(gs[0], Gd["D.dx_soft"], TOL, "d D / d relaxed input")
    for n, gv in zip(names, gs[1:]):
        assert_close(gv, Gd[f"D.grad_soft.{n}"], TOL, f"relaxed input, grad {n}", atol=1e-9)
    # CWGAN.train
    G2 = W.Generator(V, 2, 3, softmax="gumbel_softmax")
    D2 = W.Discriminator(V, 2, 2, dropout=0.0)
    G2.load_state_dict(pg); D2.load_state_dict(pd)
    opt = argparse.Namespace(lr=2e-4, beta1=0.5, beta2=0.999, clip_value=0.1, n_critic=5)
    model = W.CWGAN(G2, D2, opt)
    noises, Us = list(Gd["train.noise"]), list(Gd["train.U"])
    real_randint, real_rand = torch.randint, torch.rand
    torch.randint = lambda *a, **k: noises.pop(0).clone()
    torch.rand = lambda *a, **k: Us.pop(0).clone()
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            model.train([(tokens, labels, lengths)], epochs=3, tau=0.5, hard=False, anneal_tau=True, anneal_interval=1)
    finally:
        torch.randint, torch.rand = real_randint, real_rand
    got = torch.tensor(model.losses["d_loss"], dtype=torch.float64)
    assert torch.allclose(got, Gd["train.d_losses"], rtol=TOL, atol=1e-5), (got, Gd["train.d_losses"])
    assert model.losses["g_loss"] == [0.0, 0.0, 0.0]
    ref_lines, our_lines = Gd["train.printed"].strip().splitlines(), buf.getvalue().strip().splitlines()
    assert len(ref_lines) == len(our_lines) == 6
    for a, b in zip(our_lines, ref_lines):
        if a.startswith("epoch"):
            assert a == b, (a, b)          # the annealed tau, printed with full precision
        else:
            assert a.split("|")[0] == b.split("|")[0] and a.split("|")[2] == b.split("|")[2], (a, b)
            assert abs(float(a.split("Discriminator Loss:")[1].split("|")[0]) -
                       float(b.split("Discriminator Loss:")[1].split("|")[0])) <= 2e-5, (a, b)
    for n, v in D2.state_dict().items():
        assert_close(v, Gd[f"train.D.{n}"], TOL, f"after train: {n}")
    assert isinstance(model.eval(), W.CWGAN) and not model.training          # nn.Module.train(False) still works
