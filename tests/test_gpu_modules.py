"""The drop-in module surface (top-level wgan_gp.py) on the GPU against golden vectors recorded
from the reference's own classes (tests/golden/g1, g2, g3, g4, g7).  These tests read like the
reference's usage: construct the module, load a state_dict, call forward, call backward."""
import argparse
import contextlib
import io
import os

import pytest
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence

import common
from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-3
DEV = "cuda:0"


class Queue:
    def __init__(self, *items):
        self.items = list(items)

    def pop(self, *a, **k):
        return self.items.pop(0).clone()


@contextlib.contextmanager
def injected(mod, q_torch_rand=None, q_rand=None, q_randn=None):
    old = (torch.rand, mod.rand, mod.randn)
    if q_torch_rand is not None:
        torch.rand = q_torch_rand.pop
    if q_rand is not None:
        mod.rand = q_rand.pop
    if q_randn is not None:
        mod.randn = q_randn.pop
    try:
        yield
    finally:
        torch.rand, mod.rand, mod.randn = old


def test_custom_softmax(golden):
    import wgan_gp as M
    G = golden("g1_sampler.pt")
    g = common.gen(101)
    logits = torch.randn((4, 8, 175), generator=g) * 2
    U = torch.rand((4, 8, 175), generator=g)
    dy = torch.randn((4, 8, 175), generator=g)
    sm = M.CustomSoftmax()
    for tau in (0.5, 1.0, 5.0):
        for hard in (False, True):
            lg = logits.clone().to(DEV).requires_grad_(True)
            with injected(M, Queue(U)):
                y = sm(lg, tau, hard)
            (dl,) = torch.autograd.grad(y, lg, dy.to(DEV))
            assert_close(y.cpu(), G[f"y_tau{tau}_hard{int(hard)}"], TOL, "y")
            assert_close(dl.cpu(), G[f"dlogits_tau{tau}_hard{int(hard)}"], TOL, "dlogits")
    assert_close(M.CustomSoftmax("relaxed_bernoulli")(logits.to(DEV), 0.7, False).cpu(), G["relaxed_tau0.7"], TOL)
    assert_close(M.CustomSoftmax("softmax")(logits.to(DEV), 0.7, False).cpu(), G["plain"], TOL)
    an = torch.tensor([M.CustomSoftmax.anneal_tau_(5.0, 0.5, e, 100, 10) for e in (0, 7, 10, 50)])
    assert_close(an, G["anneal"], 1e-7)


def test_generator_module_state_dict_and_eval(golden):
    import wgan_gp as M
    G = golden("g2_generator.pt")
    gen = M.Generator(175, 2)
    assert gen.seq_len == 120 and gen.latent_dim == 100
    expect = ["model.0.weight", "model.0.bias", "model.2.weight", "model.2.bias", "model.3.weight", "model.3.bias",
              "model.3.running_mean", "model.3.running_var", "model.3.num_batches_tracked", "model.5.weight",
              "model.5.bias", "model.6.weight", "model.6.bias", "model.6.running_mean", "model.6.running_var",
              "model.6.num_batches_tracked", "model.8.weight", "model.8.bias", "label_embedding.weight"]
    assert list(gen.state_dict().keys()) == expect        # generate_fake_data.py:28-29 loads these keys
    sd = dict(common.generator_params(202, 175, 120)); sd.update(common.bn_buffers())
    gen.load_state_dict(sd)
    gen = gen.to(DEV)
    nz = common.noise(203, 6, 120)
    labels = torch.tensor([0, 1, 1, 0, 1, 0], device=DEV)
    gen.train()
    with injected(M, Queue(nz["U"])):
        y = gen(nz["z"].to(DEV), labels, 0.5, False)
    dy = torch.randn(y.shape, generator=common.gen(204)).to(DEV)
    y.backward(dy)
    assert_close(y[:, :4].detach().cpu(), G["T120.y"], TOL, "y")
    for n, p in gen.named_parameters():
        if f"T120.grad.{n}" in G:
            assert_close(p.grad.cpu(), G[f"T120.grad.{n}"], TOL, n, atol=1e-6)
    assert_close(gen.state_dict()["model.6.running_var"].cpu(), G["T120.model.6.running_var"], TOL)
    assert int(gen.state_dict()["model.3.num_batches_tracked"]) == 1
    gen.eval()
    with injected(M, Queue(nz["U2"])), torch.no_grad():
        tok = gen(nz["z"].to(DEV), labels, 0.5, True).argmax(-1)
    assert (tok.cpu() == G["T120.eval_tokens"]).float().mean() > 0.999
    with pytest.raises(Exception):
        M.Generator(175, 2)(nz["z"], labels.cpu(), 0.5, False)      # CPU tensors: loud failure, no fallback


def test_discriminator_module(golden):
    import wgan_gp as M
    G = golden("g3_critic.pt")
    g = common.gen(302)
    x = torch.softmax(torch.randn((4, 16, 175), generator=g) * 3, -1)
    dout = torch.randn((4, 3), generator=g)
    lengths = torch.tensor([16, 11, 7, 3])
    for tag, use_len in (("dense", False), ("packed", True)):
        D = M.Discriminator(175, 50, 2, 2, drop_rate=0.0)
        D.load_state_dict(common.critic_params(301, scale=2.0))
        D = D.to(DEV).train()
        xin = x.clone().to(DEV).requires_grad_(True)
        inp = pack_padded_sequence(xin, lengths, batch_first=True, enforce_sorted=False) if use_len else xin
        o = D(inp)
        o.backward(dout.to(DEV))
        assert_close(o.detach().cpu(), G[f"{tag}.out"], TOL, "out")
        assert_close(xin.grad.cpu(), G[f"{tag}.dx"], TOL, "dx")
        for n, p in D.named_parameters():
            assert_close(p.grad.cpu(), G[f"{tag}.grad.{n}"], TOL, n)


def test_compute_gradient_penalty(golden):
    import wgan_gp as M
    G = golden("g4_gp.pt")
    for scale, tag in ((1.0, "s1"), (3.0, "s3")):
        tokens, labels, lengths = common.batch(402, 5, 12, lengths=[12, 12, 9, 4, 1])
        real = F.one_hot(tokens, 175).float().to(DEV)
        for Tf in (7, 12, 15):
            k = f"{tag}.Tf{Tf}"
            D = M.Discriminator(175, 50, 2, 2, drop_rate=0.0)
            D.load_state_dict(common.critic_params(401, scale=scale))
            D = D.to(DEV).train()
            with injected(M, None, Queue(G[f"{k}.alpha"])):
                gp = M.compute_gradient_penalty(D, real, G[f"{k}.fake"].to(DEV), labels.to(DEV), lengths)
            assert_close(gp.detach().cpu(), G[f"{k}.gp"], TOL, "gp")
            (10.0 * gp).backward()
            for n, p in D.named_parameters():
                if f"{k}.grad.{n}" in G:
                    assert_close(p.grad.cpu(), G[f"{k}.grad.{n}"], TOL, f"{k} {n}", atol=1e-7)


@pytest.mark.parametrize("Tf", [4, 32])
def test_train_wgan_matches_reference_train_wgan(golden, Tf, tmp_path):
    """Our train_wgan over the same 6 batches and injected noise as the reference's own train_wgan
    (which produced g7): printed losses and final weights agree."""
    import wgan_gp as M
    G = golden("g7_full_step.pt")
    k = f"Tf{Tf}"
    ns = argparse.Namespace(vocab_size=175, epochs=1, batch_size=8, lr=2e-4, b1=0.5, b2=0.999, latent_dim=100,
                            seq_len=32, n_critic=5, clip_value=0.01, lambda_gp=10.0, hard=False, tau=0.5)
    gen = M.Generator(175, 2, seq_len=Tf)
    sd = dict(common.generator_params(701, 175, Tf)); sd.update(common.bn_buffers())
    gen.load_state_dict(sd)
    D = M.Discriminator(175, 50, 2, 2, drop_rate=0.0)
    D.load_state_dict(common.critic_params(702, scale=1.5))
    gen, D = gen.to(DEV), D.to(DEV)
    oG = torch.optim.Adam(gen.parameters(), lr=ns.lr, betas=(ns.b1, ns.b2))
    oD = torch.optim.Adam(D.parameters(), lr=ns.lr, betas=(ns.b1, ns.b2))
    batches = [common.batch(710 + i, 8, 32, min_len=3) for i in range(6)]
    noises = [common.noise(720 + i, 8, Tf) for i in range(6)]
    qU, qA, qZ = Queue(), Queue(), Queue()
    for i, nz in enumerate(noises):
        qZ.items.append(nz["z"]); qU.items.append(nz["U"]); qA.items.append(nz["alpha"])
        if i % 5 == 0:
            qU.items.append(nz["U2"])
    buf = io.StringIO()
    with injected(M, qU, qA, qZ), contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        M.train_wgan(gen, D, oG, oD, batches, [], ns, torch.device(DEV))
    import re
    printed = [(float(a), float(b)) for a, b in
               re.findall(r"\[D loss: ([-\d.eE+]+)\] \[G loss: ([-\d.eE+]+)\]", buf.getvalue())]
    ref = G[f"{k}.printed_losses"]
    assert len(printed) == 2
    for (d, g_), r in zip(printed, ref):
        assert abs(d - float(r[0])) < TOL * max(1.0, abs(d)) and abs(g_ - float(r[1])) < TOL * max(1.0, abs(g_))
    for n, v in D.state_dict().items():
        assert_close(v.cpu(), G[f"{k}.D.{n}"], TOL, f"D {n}")
    for n, v in gen.state_dict().items():
        if f"{k}.G.{n}" in G and v.dtype == torch.float32:
            assert_close(v.cpu(), G[f"{k}.G.{n}"], TOL, f"G {n}", atol=1e-5)
    assert int(gen.state_dict()["model.3.num_batches_tracked"]) == int(G[f"{k}.G.model.3.num_batches_tracked"])
    # the optimizers expose the fused Adam state
    st = oD.state[dict(D.named_parameters())["fc.weight"]]
    assert float(st["step"]) == 6.0 and st["exp_avg"].shape == (3, 50)
    # checkpoint format: what generate_fake_data.py:28-29 does with the saved file
    path = os.path.join(tmp_path, "wgan_generator_5.pth")
    torch.save(gen.state_dict(), path)
    g2 = M.Generator(vocab_size=175, seq_len=Tf)
    g2.load_state_dict(torch.load(path, weights_only=True))


def test_user_level_double_backward_through_discriminator(golden):
    """The reference's own penalty code (wgan_gp.py:228-256) written against OUR Discriminator: interpolate, pack,
    D(packed)[:, 0], autograd.grad(..., create_graph=True, retain_graph=True), per-sample norm, mean; then the backward
    of 10 * gp.  Discriminator.forward's backward is differentiable (tangent + second reverse sweep), so the
    second-order parameter gradients must match what the reference's autograd produced (g4), as the fused
    compute_gradient_penalty does."""
    import wgan_gp as M
    G = golden("g4_gp.pt")
    tokens, labels, lengths = common.batch(402, 5, 12, lengths=[12, 12, 9, 4, 1])
    real = F.one_hot(tokens, 175).float().to(DEV)
    for scale, tag in ((1.0, "s1"), (3.0, "s3")):
        for Tf in (7, 12, 15):
            k = f"{tag}.Tf{Tf}"
            D = M.Discriminator(175, 50, 2, 2, drop_rate=0.0)
            D.load_state_dict(common.critic_params(401, scale=scale))
            D = D.to(DEV).train()
            fake, alpha = G[f"{k}.fake"].to(DEV), G[f"{k}.alpha"].to(DEV)
            padded_fake = F.pad(fake, (0, 0, 0, real.size(1) - fake.size(1)))                      # :231-233
            interpolates = (alpha * real + (1 - alpha) * padded_fake).requires_grad_(True)           # :239
            packed = pack_padded_sequence(interpolates, lengths, batch_first=True, enforce_sorted=False)
            d_interpolates = D(packed)[:, 0]                                                         # :244
            gradients = torch.autograd.grad(outputs=d_interpolates, inputs=interpolates,
                                            grad_outputs=torch.ones(d_interpolates.size(), device=DEV),
                                            create_graph=True, retain_graph=True)[0]                 # :247-254
            gradients = gradients.view(gradients.size(0), -1)
            gp = ((gradients.norm(2, dim=1) - 1) ** 2).mean()                                        # :255-256
            assert_close(gp.detach().cpu(), G[f"{k}.gp"], TOL, "gp")
            (10.0 * gp).backward()
            for n, p in D.named_parameters():
                if f"{k}.grad.{n}" in G:
                    assert_close(p.grad.cpu(), G[f"{k}.grad.{n}"], TOL, f"{k} {n}", atol=1e-7)
