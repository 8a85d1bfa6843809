"""The four closed-form sweeps (oracle/lstm_sweeps.py) equal autograd on the restated critic
(oracle/wgan_gp_cpu.py) in float64 to ~1e-10, with ragged lengths and a dropout mask.  CPU only."""
import pytest
import torch

import common
from conftest import assert_close
from oracle import lstm_sweeps as SW
from oracle import wgan_gp_cpu as O


def _setup(seed, B, T, V, scale, with_mask, ragged):
    p = common.critic_params(seed, vocab=V, scale=scale, dtype=torch.float64)
    g = common.gen(seed + 1)
    x = torch.softmax(torch.randn((B, T, V), generator=g, dtype=torch.float64) * 2, -1)
    lengths = torch.tensor([T] + [max(1, T - 3 * k) for k in range(1, B)]) if ragged else torch.full((B,), T)
    mask = None
    if with_mask:
        mask = (torch.rand((B, T, 50), generator=g) > 0.3).double() / 0.7
    return p, x, lengths, mask


@pytest.mark.parametrize("with_mask,ragged", [(False, False), (True, True)])
def test_forward_backward_match_autograd(with_mask, ragged):
    p, x, lengths, mask = _setup(11, 4, 9, 23, 2.0, with_mask, ragged)
    pl = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xin = x.clone().requires_grad_(True)
    out = O.critic_forward(pl, xin, lengths, mask)
    dout = torch.randn((4, 3), generator=common.gen(5), dtype=torch.float64)
    names = list(pl)
    grads = torch.autograd.grad(out, [xin] + [pl[n] for n in names], dout)

    b0 = p["lstm.bias_ih_l0"] + p["lstm.bias_hh_l0"]
    S, out2 = SW.sweep_forward(p, x @ p["lstm.weight_ih_l0"].t() + b0, lengths, mask)
    assert_close(out2, out, 1e-12, "out")
    R = SW.sweep_backward(p, S, lengths, mask, dout)
    assert_close(R["delta0"] @ p["lstm.weight_ih_l0"], grads[0], 1e-10, "dx")
    G = SW.first_order_grads(p, S, R, x, mask, dout)
    for n, gr in zip(names, grads[1:]):
        assert_close(G[n], gr, 1e-10, n)


@pytest.mark.parametrize("with_mask,ragged,scale", [(False, False, 1.0), (True, True, 3.0)])
def test_gradient_penalty_second_order(with_mask, ragged, scale):
    p, x, lengths, mask = _setup(21, 5, 8, 17, scale, with_mask, ragged)
    pl = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    # O.gradient_penalty interpolates; alpha = 1 with real := x leaves x_hat = x
    alpha = torch.ones(5, 1, 1, dtype=torch.float64)
    gp, g = O.gradient_penalty(pl, x, x, alpha, lengths, mask)
    names = list(pl)
    grads = torch.autograd.grad(7.0 * gp, [pl[n] for n in names], allow_unused=True)

    gp2, g2, G, aux = SW.gp_by_sweeps(p, x, lengths, mask, upstream=7.0)
    assert_close(gp2, gp, 1e-12, "gp")
    assert_close(g2, g, 1e-10, "g")
    # <v, g> must equal the tangent output (consistency of sweep C)
    d_dot = aux["Tn"]["hdotT"] @ p["fc.weight"][0]
    assert_close(d_dot, (aux["v"] * g2).sum((1, 2)), 1e-9, "tangent output")
    for n, gr in zip(names, grads):
        if gr is None:
            assert float(G[n].abs().max()) == 0.0, n
        else:
            assert_close(G[n], gr, 1e-8, n)
