"""Pins oracle/wgan_gp_cpu.py (the CPU restatement) to golden vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only."""
import torch
import torch.nn.functional as F

import common
from conftest import assert_close
from oracle import wgan_gp_cpu as O


def test_g1_sampler(golden):
    G = golden("g1_sampler.pt")
    g = common.gen(101)
    logits = torch.randn((4, 8, 175), generator=g) * 2
    U = torch.rand((4, 8, 175), generator=g)
    dy = torch.randn((4, 8, 175), generator=g)
    for tau in (0.5, 1.0, 5.0):
        for hard in (False, True):
            lg = logits.clone().requires_grad_(True)
            y = O.gumbel_softmax(lg, U, tau, hard)
            (dl,) = torch.autograd.grad(y, lg, dy)
            assert_close(y, G[f"y_tau{tau}_hard{int(hard)}"], 1e-6, "y")
            assert_close(dl, G[f"dlogits_tau{tau}_hard{int(hard)}"], 1e-5, "dlogits")
    assert_close(O.relaxed_bernoulli(logits, 0.7), G["relaxed_tau0.7"], 1e-6)
    assert_close(F.softmax(logits, -1), G["plain"], 1e-6)
    an = torch.tensor([O.anneal_tau(5.0, 0.5, e, 100, 10) for e in (0, 7, 10, 50)])
    assert_close(an, G["anneal"], 1e-7)


def test_g2_generator(golden):
    G = golden("g2_generator.pt")
    for seq_len in (8, 120):
        p = {k: v.requires_grad_(True) for k, v in common.generator_params(202, 175, seq_len).items()}
        bn = common.bn_buffers()
        nz = common.noise(203, 6, seq_len)
        labels = torch.tensor([0, 1, 1, 0, 1, 0])
        y = O.generator_forward(p, bn, nz["z"], labels, nz["U"], 0.5, False, seq_len, 175, True)
        dy = torch.randn(y.shape, generator=common.gen(204))
        names = list(p)
        grads = dict(zip(names, torch.autograd.grad(y, [p[n] for n in names], dy)))
        tag = f"T{seq_len}"
        assert_close(y if seq_len == 8 else y[:, :4], G[f"{tag}.y"], 1e-5, "y")
        for n in names:
            if f"{tag}.grad.{n}" in G:
                assert_close(grads[n], G[f"{tag}.grad.{n}"], 2e-4, n, atol=1e-7)
            else:
                assert_close(grads[n].norm(), G[f"{tag}.grad_norm.{n}"], 2e-4, n)
                assert_close(grads[n][:64], G[f"{tag}.grad_head.{n}"], 2e-4, n)
        for b in ("model.3.running_mean", "model.3.running_var", "model.6.running_mean", "model.6.running_var"):
            assert_close(bn[b], G[f"{tag}.{b}"], 1e-5, b)
        with torch.no_grad():
            tok = O.generator_forward(p, bn, nz["z"], labels, nz["U2"], 0.5, True, seq_len, 175, False).argmax(-1)
        assert (tok == G[f"{tag}.eval_tokens"]).float().mean() > 0.999


def test_g3_critic(golden):
    G = golden("g3_critic.pt")
    g = common.gen(302)
    x = torch.softmax(torch.randn((4, 16, 175), generator=g) * 3, -1)
    dout = torch.randn((4, 3), generator=g)
    lengths = torch.tensor([16, 11, 7, 3])
    for tag, L in (("dense", None), ("packed", lengths)):
        p = {k: v.requires_grad_(True) for k, v in common.critic_params(301, scale=2.0).items()}
        xin = x.clone().requires_grad_(True)
        o = O.critic_forward(p, xin, L)
        names = list(p)
        grads = torch.autograd.grad(o, [xin] + [p[n] for n in names], dout)
        assert_close(o, G[f"{tag}.out"], 1e-5, "out")
        assert_close(grads[0], G[f"{tag}.dx"], 1e-4, "dx")
        for n, gr in zip(names, grads[1:]):
            assert_close(gr, G[f"{tag}.grad.{n}"], 1e-4, n)


def test_g4_gp(golden):
    G = golden("g4_gp.pt")
    for scale, tag in ((1.0, "s1"), (3.0, "s3")):
        tokens, labels, lengths = common.batch(402, 5, 12, lengths=[12, 12, 9, 4, 1])
        real = F.one_hot(tokens, 175).float()
        for Tf in (7, 12, 15):
            k = f"{tag}.Tf{Tf}"
            p = {n: v.requires_grad_(True) for n, v in common.critic_params(401, scale=scale).items()}
            gp, _ = O.gradient_penalty(p, real, G[f"{k}.fake"], G[f"{k}.alpha"], lengths)
            assert_close(gp, G[f"{k}.gp"], 1e-5, "gp")
            names = list(p)
            grads = torch.autograd.grad(10.0 * gp, [p[n] for n in names], allow_unused=True)
            for n, gr in zip(names, grads):
                if f"{k}.grad.{n}" in G:
                    assert_close(gr, G[f"{k}.grad.{n}"], 5e-4, f"{k} {n}")
                else:
                    assert gr is None or float(gr.abs().max()) == 0.0


def test_g5_ce(golden):
    G = golden("g5_ce.pt")
    lg = G["logits"].clone().requires_grad_(True)
    loss = O.weighted_ce_on_probs(F.softmax(lg, -1), G["labels"])
    (dl,) = torch.autograd.grad(loss, lg)
    assert_close(loss, G["loss"], 1e-6)
    assert_close(dl, G["dlogits"], 1e-5)


def test_g6_adam(golden):
    G = golden("g6_adam.pt")
    P = {"w": G["w0"].clone(), "w2": G["w20"].clone()}
    opt = O.Adam(P, 2e-4, 0.5, 0.999)
    for s in range(3):
        opt.step({"w": G[f"gw{s}"], "w2": G.get(f"gw2{s}")})
        assert_close(P["w"], G[f"w{s + 1}"], 1e-7)
        assert_close(P["w2"], G[f"w2{s + 1}"], 1e-7)


def test_g7_full_step(golden):
    G = golden("g7_full_step.pt")
    cfg = dict(vocab=175, tau=0.5, hard=False, lambda_gp=10.0, n_critic=5)
    for Tf in (4, 32):
        k = f"Tf{Tf}"
        cfg["gen_seq_len"] = Tf
        pg = common.generator_params(701, 175, Tf)
        pd = common.critic_params(702, scale=1.5)
        bn = common.bn_buffers()
        batches = [common.batch(710 + i, 8, 32, min_len=3) for i in range(6)]
        noises = [common.noise(720 + i, 8, Tf) for i in range(6)]
        og, od = O.Adam(pg, 2e-4, 0.5, 0.999), O.Adam(pd, 2e-4, 0.5, 0.999)
        d_losses, g_losses = O.full_step(pg, bn, pd, og, od, batches, noises, cfg)
        printed = G[f"{k}.printed_losses"]          # rows: (d_loss, g_loss) at i = 0 and i = 5
        assert abs(d_losses[0] - float(printed[0, 0])) < 2e-5 * max(1, abs(d_losses[0]))
        assert abs(d_losses[5] - float(printed[1, 0])) < 2e-4 * max(1, abs(d_losses[5]))
        assert abs(g_losses[0] - float(printed[0, 1])) < 2e-5 * max(1, abs(g_losses[0]))
        assert abs(g_losses[1] - float(printed[1, 1])) < 2e-4 * max(1, abs(g_losses[1]))
        for n in pd:
            assert_close(pd[n], G[f"{k}.D.{n}"], 1e-4, f"D {n}")
        for n in pg:
            if f"{k}.G.{n}" in G:
                assert_close(pg[n], G[f"{k}.G.{n}"], 1e-4, f"G {n}", atol=1e-5)
            else:
                assert_close(pg[n][:32], G[f"{k}.G.head.{n}"], 1e-4, f"G {n}")
        for b in ("model.3.running_mean", "model.3.running_var", "model.6.running_mean", "model.6.running_var"):
            assert_close(bn[b], G[f"{k}.G.{b}"], 1e-4, b)
