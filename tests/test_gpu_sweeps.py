"""HIP LSTM sweeps (through the C ABI) against the closed-form oracle sweeps, buffer by buffer.
Tolerance: fp32 path, 1e-3 relative (the north-star bound); observed errors are ~1e-5."""
import pytest
import torch
import torch.nn.functional as F

import common
from conftest import assert_close

pytestmark = pytest.mark.gpu

TOL = 1e-3


def to_internal(x):
    """[..., 200] torch gate order (q*50+u) -> internal order (4u+q)."""
    return x.reshape(*x.shape[:-1], 4, 50).transpose(-1, -2).reshape(*x.shape[:-1], 200)


def setup_case(seed, B, T, V, ragged, with_mask, scale):
    from oracle import lstm_sweeps as SW
    p64 = common.critic_params(seed, vocab=V, scale=scale, dtype=torch.float64)
    g = common.gen(seed + 1)
    x = torch.softmax(torch.randn((B, T, V), generator=g, dtype=torch.float64) * 2, -1)
    lengths = torch.tensor([T] + [max(1, T - 5 * k) for k in range(1, B)]) if ragged else torch.full((B,), T)
    mask = (torch.rand((B, T, 50), generator=g) > 0.3).double() / 0.7 if with_mask else None
    return SW, p64, x, lengths, mask


@pytest.mark.parametrize("B,T,ragged,with_mask,scale", [(3, 7, False, False, 1.0), (5, 33, True, True, 2.0),
                                                        (4, 64, True, False, 3.0)])
def test_four_sweeps_match_oracle(B, T, ragged, with_mask, scale):
    from s2g.critic import CriticEngine, make_desc
    from s2g.ops import ptr
    V = 175
    dev = torch.device("cuda:0")
    SW, p64, x, lengths, mask = setup_case(31, B, T, V, ragged, with_mask, scale)

    # ---- oracle (float64)
    b0 = p64["lstm.bias_ih_l0"] + p64["lstm.bias_hh_l0"]
    S, out = SW.sweep_forward(p64, x @ p64["lstm.weight_ih_l0"].t() + b0, lengths, mask)
    dout = torch.randn((B, 3), generator=common.gen(7), dtype=torch.float64)
    R = SW.sweep_backward(p64, S, lengths, mask, dout)
    vdir = torch.randn((B, T, V), generator=common.gen(8), dtype=torch.float64) * 0.1
    Tn = SW.sweep_tangent(p64, S, lengths, mask, vdir @ p64["lstm.weight_ih_l0"].t())
    e0 = torch.zeros_like(out); e0[:, 0] = 1
    R0 = SW.sweep_backward(p64, S, lengths, mask, e0)
    A = SW.sweep_second_reverse(p64, S, R0, Tn, lengths, mask)

    # ---- HIP
    eng = CriticEngine(V, dev)
    eng.load({k: v.float() for k, v in p64.items()})
    ws = eng.workspace(B, T, B)
    xf = x.float().to(dev).contiguous()
    P = torch.empty(B * T, 200, device=dev)
    eng.project(xf, P)
    assert_close(P.cpu().view(B, T, 200), to_internal(x @ p64["lstm.weight_ih_l0"].t()), TOL, "projection")
    mk = None
    if with_mask:
        mk = torch.zeros(B * T, 52, device=dev)
        mk[:, :50] = mask.float().reshape(B * T, 50).to(dev)
    desc = make_desc([dict(len=int(lengths[b]), p_off=b * T, p_len=T, ca=0.0, cp=1.0, ws_row=b * T,
                           mask_off=b * T if with_mask else -1) for b in range(B)], dev)
    eng.sweep_fwd(ws, ptr(desc), B, T, None, P, mk)
    eng.head_fwd(ws, B)
    torch.cuda.synchronize()
    assert_close(ws["out"].cpu(), out, TOL, "out")
    assert_close(ws["gates0"].cpu().view(B, T, 200), to_internal(S["gates0"]), TOL, "gates0")
    assert_close(ws["gates1"].cpu().view(B, T, 200), to_internal(S["gates1"]), TOL, "gates1")
    cell0 = ws["cell0"].cpu().view(B, T, 50, 4)
    assert_close(cell0[..., 0], S["c0"], TOL, "c0")
    assert_close(cell0[..., 2], S["h0"], TOL, "h0")
    assert_close(ws["cell1"].cpu().view(B, T, 50, 4)[..., 2], S["h1"], TOL, "h1")

    # sweep B with a random d(out)
    from s2g.lib import call
    from s2g.ops import stream_ptr
    ws["dout"].copy_(dout.float())
    call("s2g_head_bwd_dh", ptr(ws["dout"]), ptr(eng.params.views["fc.weight"]), ptr(ws["dhT"]), B, 3, stream_ptr())
    eng.sweep_bwd(ws, ptr(desc), B, T, mk)
    torch.cuda.synchronize()
    assert_close(ws["delta0"].cpu().view(B, T, 200), to_internal(R["delta0"]), TOL, "delta0")
    assert_close(ws["delta1"].cpu().view(B, T, 200), to_internal(R["delta1"]), TOL, "delta1")

    # sweep B with d(out) = e0, storing the adjoints; then sweeps C and D
    ws["dout"].copy_(e0.float())
    call("s2g_head_bwd_dh", ptr(ws["dout"]), ptr(eng.params.views["fc.weight"]), ptr(ws["dhT"]), B, 3, stream_ptr())
    eng.sweep_bwd(ws, ptr(desc), B, T, mk, adj_seq0=0)
    eng.project(vdir.float().to(dev).contiguous(), ws["Pv"])
    eng.sweep_tangent(ws, ptr(desc), B, ptr(mk), 0)
    eng.sweep_bwd2(ws, ptr(desc), B, T, ptr(mk), 0)
    torch.cuda.synchronize()
    adj0 = ws["adj0"].cpu().view(B, T, 50, 2)
    assert_close(adj0[..., 0], R0["dh0"], TOL, "dh0")
    assert_close(adj0[..., 1], R0["dc0"], TOL, "dc0")
    assert_close(ws["adot0"].cpu().view(B, T, 200), to_internal(Tn["adot0"]), TOL, "adot0")
    assert_close(ws["adot1"].cpu().view(B, T, 200), to_internal(Tn["adot1"]), TOL, "adot1")
    assert_close(ws["celld1"].cpu().view(B, T, 50, 4)[..., 1], Tn["hdot1"], TOL, "hdot1")
    assert_close(ws["hdotT"].cpu(), Tn["hdotT"], TOL, "hdotT")
    assert_close(ws["abar1"].cpu().view(B, T, 200), to_internal(A["abar1"]), TOL, "abar1")
    assert_close(ws["abar0"].cpu().view(B, T, 200), to_internal(A["abar0"]), TOL, "abar0")

    # the sweep scaling its own direction by the penalty coefficient of each sequence: the tangent
    # states are linear in the direction
    plain = {k: ws[k].clone() for k in ("adot0", "adot1", "celld1", "hdotT")}
    part = (torch.rand(B, 4, generator=common.gen(77)) * 2.0).to(dev)
    lam, nb = 10.0, 7.0
    eng.sweep_tangent(ws, ptr(desc), B, ptr(mk), 0, gp=(part, lam, nb))
    torch.cuda.synchronize()
    n = part.double().sum(1).sqrt()
    coef = (lam * 2 * (n - 1) / (nb * n)).cpu()
    assert_close(ws["hdotT"].cpu(), plain["hdotT"].cpu().double() * coef[:, None], TOL, "hdotT scaled")
    for k in ("adot0", "adot1", "celld1"):
        want = plain[k].cpu().double().view(B, -1) * coef[:, None]
        assert_close(ws[k].cpu().view(B, -1), want, TOL, k + " scaled")


@pytest.mark.parametrize("mode", [0, 1])
def test_reverse_sweep_with_inline_head_matches_head_kernels(mode):
    """s2g_lstm2_bwd(head=...) derives d loss / d h_T per sequence itself: same delta as the sweep fed
    the d(h_T) of the head + loss kernels (critic batch [real|fake|interp] and generator pass)."""
    from s2g.critic import CriticEngine, make_desc
    from s2g.lib import call
    from s2g.ops import ptr, stream_ptr
    V, B, T = 175, 4, 19
    dev = torch.device("cuda:0")
    nseq = 3 * B if mode == 0 else B
    SW, p64, x, lengths, _ = setup_case(33, nseq, T, V, True, False, 2.0)
    eng = CriticEngine(V, dev)
    eng.load({k: v.float() for k, v in p64.items()})
    ws = eng.fresh_workspace(nseq, T, 0)
    P = torch.empty(nseq * T, 200, device=dev)
    eng.project(x.float().to(dev).contiguous(), P)
    desc = make_desc([dict(len=int(lengths[b]), p_off=b * T, p_len=T, ca=0.0, cp=1.0, ws_row=b * T, mask_off=-1)
                      for b in range(nseq)], dev)
    eng.sweep_fwd(ws, ptr(desc), nseq, T, None, P, None)
    labels = torch.randint(0, 2, (B,), generator=common.gen(9)).to(dev)
    inv_n, inv_wsum, cw = 1.0 / B, 1.0 / float((1.0 + labels.float()).sum()), (1.0, 2.0)
    p, st = eng.params.views, stream_ptr()
    # reference path: head + loss kernels -> dhT -> sweep
    if mode == 0:
        gw, gb = torch.empty(3, 50, device=dev), torch.empty(3, device=dev)
        call("s2g_critic_head", ptr(ws["hT"]), ptr(p["fc.weight"]), ptr(p["fc.bias"]), ptr(labels), B, inv_n, inv_wsum,
             cw[0], cw[1], ptr(ws["out"]), ptr(ws["dout"]), ptr(ws["losses"]), ptr(ws["dhT"]), ptr(gw), ptr(gb), st)
    else:
        eng.head_fwd(ws, B)
        call("s2g_generator_loss", ptr(ws["out"]), ptr(labels), B, inv_n, inv_wsum, cw[0], cw[1], ptr(ws["dout"]),
             ptr(ws["losses"]), st)
        call("s2g_head_bwd_dh", ptr(ws["dout"]), ptr(p["fc.weight"]), ptr(ws["dhT"]), B, 3, st)
    eng.sweep_bwd(ws, ptr(desc), nseq, T, None)
    torch.cuda.synchronize()
    want0, want1 = ws["delta0"].clone(), ws["delta1"].clone()
    ws["delta0"].fill_(float("nan")); ws["delta1"].fill_(float("nan")); ws["dhT"].fill_(float("nan"))
    eng.sweep_bwd(ws, ptr(desc), nseq, T, None, head=eng.head_args(ws, labels, B, mode, inv_n, inv_wsum, cw))
    torch.cuda.synchronize()
    assert torch.isfinite(ws["delta0"]).all() and want0.abs().max() > 0
    assert_close(ws["delta0"].cpu(), want0.cpu().double(), 1e-5, "delta0")
    assert_close(ws["delta1"].cpu(), want1.cpu().double(), 1e-5, "delta1")
    # the wrapper rejects a head whose batch layout does not match the launch
    bad = eng.head_args(ws, labels, B + 1, mode, inv_n, inv_wsum, cw)
    with pytest.raises(Exception):
        eng.sweep_bwd(ws, ptr(desc), nseq, T, None, head=bad)


def test_token_and_interp_inputs():
    """xproj built in-kernel from tokens (real) and alpha*token + (1-alpha)*projection (interp)."""
    from oracle import wgan_gp_cpu as O
    from s2g.critic import CriticEngine, make_desc
    from s2g.ops import ptr
    V, B, T, Tf = 175, 4, 19, 11
    dev = torch.device("cuda:0")
    p = common.critic_params(41, scale=2.0)
    tokens, labels, lengths = common.batch(42, B, T, lengths=[19, 19, 8, 2])
    g = common.gen(43)
    fake = torch.softmax(torch.randn((B, Tf, V), generator=g) * 2, -1)
    alpha = torch.rand((B, 1, 1), generator=g)
    real = F.one_hot(tokens, V).float()
    xhat = alpha * real + (1 - alpha) * O.pad_or_crop_time(fake, T)
    ref_real = O.critic_forward(p, real, lengths)
    ref_fake = O.critic_forward(p, fake, None)
    ref_int = O.critic_forward(p, xhat, lengths)

    eng = CriticEngine(V, dev)
    eng.load(p)
    Tw = max(T, Tf)
    ws = eng.workspace(3 * B, Tw, 0)
    P = torch.empty(B * Tf, 200, device=dev)
    eng.project(fake.to(dev).contiguous(), P)
    tok = tokens.to(torch.int32).to(dev).contiguous()
    ent = []
    for b in range(B):
        ent.append(dict(len=int(lengths[b]), tok_off=b * T, ca=1.0, cp=0.0, ws_row=b * Tw))
    for b in range(B):
        ent.append(dict(len=Tf, p_off=b * Tf, p_len=Tf, ca=0.0, cp=1.0, ws_row=(B + b) * Tw))
    for b in range(B):
        a = float(alpha[b])
        ent.append(dict(len=int(lengths[b]), tok_off=b * T, p_off=b * Tf, p_len=Tf, ca=a, cp=1 - a,
                        ws_row=(2 * B + b) * Tw))
    desc = make_desc(ent, dev)
    eng.sweep_fwd(ws, ptr(desc), 3 * B, Tw, tok, P, None)
    eng.head_fwd(ws, 3 * B)
    torch.cuda.synchronize()
    out = ws["out"].cpu()
    assert_close(out[:B], ref_real, TOL, "real")
    assert_close(out[B:2 * B], ref_fake, TOL, "fake")
    assert_close(out[2 * B:], ref_int, TOL, "interp")

    # the device-built descriptors agree with the host-built ones
    from s2g.lib import call
    from s2g.ops import stream_ptr
    d2 = torch.zeros(4 * B, 8, dtype=torch.int32, device=dev)
    len_d = lengths.to(torch.int32).to(dev)
    alpha_d = alpha.view(-1).to(dev).contiguous()
    call("s2g_critic_batch_desc", ptr(len_d), ptr(alpha_d), B, T, Tf, 0, ptr(d2), stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(d2[:3 * B].cpu(), desc.cpu())


def test_out_of_range_tokens_and_lengths_stay_in_bounds():
    """Device tensors cannot be range-checked without a synchronisation, so the kernels themselves must not
    leave their buffers: token ids are clamped to the rows of the W_ih0 table and lengths to the rows a
    sequence owns.  A batch with ids -1 / 10^6 and lengths T+9 must give exactly the result of the batch
    with those values clamped (the reference raises for such a batch, wgan_gp.py:277-279)."""
    from s2g.trainer import WGANGPTrainer, class_weight_sum
    DEV = "cuda:0"
    B, T, Tf, V = 4, 16, 12, 175
    pd, pg = common.critic_params(61, scale=2.0), common.generator_params(62, V, Tf)
    tokens, labels, lengths = common.batch(63, B, T, lengths=[16, 16, 9, 3])
    nz = common.noise(64, B, Tf)
    bad_tok, bad_len = tokens.clone(), lengths.clone()
    bad_tok[0, 3], bad_tok[1, 0], bad_tok[2, 8] = -1, 10 ** 6, 175
    bad_len[0], bad_len[1] = T + 9, T + 1
    ok_tok = bad_tok.clamp(0, V - 1)
    out = []
    for tok, ln in ((ok_tok, lengths), (bad_tok, bad_len)):
        tr = WGANGPTrainer(V, Tf, DEV, drop_rate=0.0)
        tr.G.load(pg); tr.D.load(pd)
        tr.critic_step(tok.to(torch.int32).to(DEV).contiguous(), labels.to(DEV), ln.to(torch.int32).to(DEV),
                       class_weight_sum(labels), nz["z"].to(DEV), nz["U"].to(DEV), nz["alpha"].view(-1).to(DEV))
        torch.cuda.synchronize()
        out.append((tr.D.params.flat.clone(), tr.last["d_parts"].clone(), tr.last["pen"].clone()))
    for a, b in zip(*out):
        assert torch.isfinite(b).all()
        assert torch.equal(a, b)
