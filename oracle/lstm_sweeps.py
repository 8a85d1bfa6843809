"""Closed-form sweeps for the critic (TEST INFRASTRUCTURE ONLY; see oracle/wgan_gp_cpu.py).

The reference obtains the critic gradients, including the gradient-penalty's second-order
term, from torch autograd (wgan_gp.py:244-256, 317).  The HIP path computes the same
quantities with four explicit time sweeps over the 2-layer LSTM of wgan_gp.py:189-223.
This module states those sweeps without autograd, in plain tensor arithmetic, so each HIP
sweep can be checked buffer by buffer; tests/test_sweeps_cpu.py proves the sweeps equal
autograd on the restated critic (oracle/wgan_gp_cpu.py), which is itself pinned to the
reference's outputs.

  A  forward            states h,c and gates of both layers
  B  reverse            dgates ("delta") for a given d(out); gives dL/dx and first-order dW
  C  tangent forward    directional derivative of every state along an input direction v
  D  second reverse     adjoints of the PRIMAL states for the loss  sum_b <v_b, dD/dx_b>,
                        i.e. the parameter gradient of the gradient penalty

With g_b = dD(x_b)[0]/dx_b, n_b = |g_b| and gp = mean_b (n_b - 1)^2 the chain rule gives
d gp/d theta = sum_b d/d theta <v_b, g_b(theta)> with v_b = 2 (n_b - 1) / (B n_b) g_b held
constant, and <v_b, g_b> is the forward-mode derivative of D along v_b: sweep C.  The
adjoints of the tangent states in sweep D coincide with the adjoints of sweep B (the tangent
map is linear with the same Jacobians), so D reuses B's delta/dh/dc instead of recomputing.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

Tensor = torch.Tensor


def _sig(x):
    return torch.sigmoid(x)


def split4(a: Tensor, H: int):
    return a[..., 0:H], a[..., H:2 * H], a[..., 2 * H:3 * H], a[..., 3 * H:4 * H]


def sweep_forward(p: Dict[str, Tensor], xproj: Tensor, lengths: Tensor, mask: Optional[Tensor]):
    """A. xproj [B,T,4H] = x W_ih0^T + b_ih0 + b_hh0.  Returns dict of saved tensors
    (zero beyond each sample's length) and out [B,n_out]."""
    B, T, H4 = xproj.shape
    H = H4 // 4
    dt = xproj.dtype
    Whh0, Wih1, Whh1 = p["lstm.weight_hh_l0"], p["lstm.weight_ih_l1"], p["lstm.weight_hh_l1"]
    b1 = p["lstm.bias_ih_l1"] + p["lstm.bias_hh_l1"]
    S = {k: xproj.new_zeros(B, T, H4) for k in ("gates0", "gates1")}
    for k in ("c0", "c1", "h0", "h1", "tc0", "tc1"):
        S[k] = xproj.new_zeros(B, T, H)
    h0 = xproj.new_zeros(B, H); c0 = h0.clone(); h1 = h0.clone(); c1 = h0.clone()
    for t in range(T):
        live = (lengths > t).to(dt).unsqueeze(1)
        a0 = xproj[:, t] + h0 @ Whh0.t()
        i, f, g, o = split4(a0, H)
        i, f, g, o = _sig(i), _sig(f), torch.tanh(g), _sig(o)
        nc0 = f * c0 + i * g; tc = torch.tanh(nc0); nh0 = o * tc
        S["gates0"][:, t] = torch.cat([i, f, g, o], 1) * live
        S["c0"][:, t] = nc0 * live; S["tc0"][:, t] = tc * live; S["h0"][:, t] = nh0 * live
        y = nh0 if mask is None else nh0 * mask[:, t]
        a1 = y @ Wih1.t() + b1 + h1 @ Whh1.t()
        i, f, g, o = split4(a1, H)
        i, f, g, o = _sig(i), _sig(f), torch.tanh(g), _sig(o)
        nc1 = f * c1 + i * g; tc = torch.tanh(nc1); nh1 = o * tc
        S["gates1"][:, t] = torch.cat([i, f, g, o], 1) * live
        S["c1"][:, t] = nc1 * live; S["tc1"][:, t] = tc * live; S["h1"][:, t] = nh1 * live
        h0 = live * nh0 + (1 - live) * h0; c0 = live * nc0 + (1 - live) * c0
        h1 = live * nh1 + (1 - live) * h1; c1 = live * nc1 + (1 - live) * c1
    S["hT"] = h1
    out = h1 @ p["fc.weight"].t() + p["fc.bias"]
    return S, out


def _prev(x: Tensor, t: int):
    return x[:, t - 1] if t > 0 else torch.zeros_like(x[:, 0])


def sweep_backward(p, S, lengths, mask, dout: Tensor):
    """B. dout [B,n_out].  Returns dict: delta0/delta1 [B,T,4H] (gradient wrt gate
    pre-activations), dh0/dh1/dc0/dc1 [B,T,H] (total adjoints of h_t and c_t)."""
    B, T, H4 = S["gates0"].shape
    H = H4 // 4
    dt = S["gates0"].dtype
    Whh0, Wih1, Whh1 = p["lstm.weight_hh_l0"], p["lstm.weight_ih_l1"], p["lstm.weight_hh_l1"]
    R = {k: S["gates0"].new_zeros(B, T, H4) for k in ("delta0", "delta1")}
    for k in ("dh0", "dh1", "dc0", "dc1"):
        R[k] = S["c0"].new_zeros(B, T, H)
    dhT = dout @ p["fc.weight"]                                   # [B,H]
    dh1n = torch.zeros_like(dhT); dc1 = torch.zeros_like(dhT)
    dh0n = torch.zeros_like(dhT); dc0 = torch.zeros_like(dhT)

    def cell_bwd(gates, cprev, tc, dh, dc):
        i, f, g, o = split4(gates, H)
        dc = dc + dh * o * (1 - tc * tc)
        delta = torch.cat([dc * g * i * (1 - i), dc * cprev * f * (1 - f), dc * i * (1 - g * g),
                           dh * tc * o * (1 - o)], 1)
        return delta, dc, dc * f

    for t in range(T - 1, -1, -1):
        live = (lengths > t).to(dt).unsqueeze(1)
        last = (lengths == t + 1).to(dt).unsqueeze(1)
        dh = (dh1n + last * dhT) * live
        d1, dct, dc1 = cell_bwd(S["gates1"][:, t], _prev(S["c1"], t), S["tc1"][:, t], dh, dc1 * live)
        R["delta1"][:, t] = d1; R["dh1"][:, t] = dh; R["dc1"][:, t] = dct
        dh1n = d1 @ Whh1
        dy = d1 @ Wih1
        dh = ((dy if mask is None else dy * mask[:, t]) + dh0n) * live
        d0, dct, dc0 = cell_bwd(S["gates0"][:, t], _prev(S["c0"], t), S["tc0"][:, t], dh, dc0 * live)
        R["delta0"][:, t] = d0; R["dh0"][:, t] = dh; R["dc0"][:, t] = dct
        dh0n = d0 @ Whh0
    return R


def first_order_grads(p, S, R, x: Tensor, mask, dout: Tensor):
    """Parameter gradients from sweep B (x [B,T,V] is the critic input)."""
    B, T, H4 = S["gates0"].shape
    H = H4 // 4
    z = S["h0"].new_zeros(B, 1, H)
    h0p = torch.cat([z, S["h0"][:, :-1]], 1); h1p = torch.cat([z, S["h1"][:, :-1]], 1)
    y = S["h0"] if mask is None else S["h0"] * mask
    f2 = lambda a: a.reshape(-1, a.shape[-1])
    G = {
        "lstm.weight_ih_l0": f2(R["delta0"]).t() @ f2(x),
        "lstm.weight_hh_l0": f2(R["delta0"]).t() @ f2(h0p),
        "lstm.bias_ih_l0": f2(R["delta0"]).sum(0), "lstm.bias_hh_l0": f2(R["delta0"]).sum(0),
        "lstm.weight_ih_l1": f2(R["delta1"]).t() @ f2(y),
        "lstm.weight_hh_l1": f2(R["delta1"]).t() @ f2(h1p),
        "lstm.bias_ih_l1": f2(R["delta1"]).sum(0), "lstm.bias_hh_l1": f2(R["delta1"]).sum(0),
        "fc.weight": dout.t() @ S["hT"], "fc.bias": dout.sum(0),
    }
    return G


def sweep_tangent(p, S, lengths, mask, xproj_dot: Tensor):
    """C. xproj_dot [B,T,4H] = v W_ih0^T (no bias).  Returns tangents adot0/adot1 [B,T,4H]
    (pre-activations), cdot0/1, hdot0/1 [B,T,H], hdotT [B,H]."""
    B, T, H4 = xproj_dot.shape
    H = H4 // 4
    dt = xproj_dot.dtype
    Whh0, Wih1, Whh1 = p["lstm.weight_hh_l0"], p["lstm.weight_ih_l1"], p["lstm.weight_hh_l1"]
    Tn = {k: xproj_dot.new_zeros(B, T, H4) for k in ("adot0", "adot1")}
    for k in ("cdot0", "cdot1", "hdot0", "hdot1"):
        Tn[k] = xproj_dot.new_zeros(B, T, H)
    hd0 = xproj_dot.new_zeros(B, H); cd0 = hd0.clone(); hd1 = hd0.clone(); cd1 = hd0.clone()

    def cell_tan(gates, cprev, tc, adot, cdprev):
        i, f, g, o = split4(gates, H)
        ai, af, ag, ao = split4(adot, H)
        id_, fd, gd, od = i * (1 - i) * ai, f * (1 - f) * af, (1 - g * g) * ag, o * (1 - o) * ao
        cd = fd * cprev + f * cdprev + id_ * g + i * gd
        hd = od * tc + o * (1 - tc * tc) * cd
        return cd, hd

    for t in range(T):
        live = (lengths > t).to(dt).unsqueeze(1)
        ad0 = (xproj_dot[:, t] + hd0 @ Whh0.t()) * live
        ncd0, nhd0 = cell_tan(S["gates0"][:, t], _prev(S["c0"], t), S["tc0"][:, t], ad0, cd0)
        yd = nhd0 if mask is None else nhd0 * mask[:, t]
        ad1 = (yd @ Wih1.t() + hd1 @ Whh1.t()) * live
        ncd1, nhd1 = cell_tan(S["gates1"][:, t], _prev(S["c1"], t), S["tc1"][:, t], ad1, cd1)
        Tn["adot0"][:, t] = ad0; Tn["adot1"][:, t] = ad1
        Tn["cdot0"][:, t] = ncd0 * live; Tn["cdot1"][:, t] = ncd1 * live
        Tn["hdot0"][:, t] = nhd0 * live; Tn["hdot1"][:, t] = nhd1 * live
        hd0 = live * nhd0 + (1 - live) * hd0; cd0 = live * ncd0 + (1 - live) * cd0
        hd1 = live * nhd1 + (1 - live) * hd1; cd1 = live * ncd1 + (1 - live) * cd1
    Tn["hdotT"] = hd1
    return Tn


def sweep_second_reverse(p, S, R, Tn, lengths, mask):
    """D. Returns abar0/abar1 [B,T,4H]: adjoints of the PRIMAL gate pre-activations for the
    loss sum_b fc.weight[0] . hdot1_b(final).  R must come from sweep B with dout = e_0."""
    B, T, H4 = S["gates0"].shape
    H = H4 // 4
    dt = S["gates0"].dtype
    Whh0, Wih1, Whh1 = p["lstm.weight_hh_l0"], p["lstm.weight_ih_l1"], p["lstm.weight_hh_l1"]
    A = {k: S["gates0"].new_zeros(B, T, H4) for k in ("abar0", "abar1")}
    zero = S["c0"].new_zeros(B, H)
    hb1 = zero.clone(); cb1 = zero.clone(); hb0n = zero.clone(); cb0 = zero.clone()

    def cell(gates, cprev, tc, adot, cdprev, cd, Hd, Cd, hb, cb):
        i, f, g, o = split4(gates, H)
        ai, af, ag, ao = split4(adot, H)
        si, sf, sg, so = i * (1 - i), f * (1 - f), (1 - g * g), o * (1 - o)
        id_, fd, gd, od = si * ai, sf * af, sg * ag, so * ao
        u = 1 - tc * tc
        tcb = Hd * (od - 2 * o * tc * cd) + hb * o
        ob = Hd * u * cd + hb * tc + (Hd * tc) * (1 - 2 * o) * ao
        cbt = cb + tcb * u
        fb = Cd * cdprev + cbt * cprev + (Cd * cprev) * (1 - 2 * f) * af
        ib = Cd * gd + cbt * g + (Cd * g) * (1 - 2 * i) * ai
        gb = Cd * id_ + cbt * i + (Cd * i) * (-2 * g) * ag
        cb_prev = cbt * f + Cd * fd
        abar = torch.cat([ib * si, fb * sf, gb * sg, ob * so], 1)
        return abar, cb_prev

    for t in range(T - 1, -1, -1):
        live = (lengths > t).to(dt).unsqueeze(1)
        a1, cb1 = cell(S["gates1"][:, t], _prev(S["c1"], t), S["tc1"][:, t], Tn["adot1"][:, t],
                       _prev(Tn["cdot1"], t), Tn["cdot1"][:, t], R["dh1"][:, t], R["dc1"][:, t],
                       hb1 * live, cb1 * live)
        a1 = a1 * live; cb1 = cb1 * live
        A["abar1"][:, t] = a1
        hb1 = a1 @ Whh1
        yb = a1 @ Wih1
        hb0 = ((yb if mask is None else yb * mask[:, t]) + hb0n) * live
        a0, cb0 = cell(S["gates0"][:, t], _prev(S["c0"], t), S["tc0"][:, t], Tn["adot0"][:, t],
                       _prev(Tn["cdot0"], t), Tn["cdot0"][:, t], R["dh0"][:, t], R["dc0"][:, t],
                       hb0, cb0 * live)
        a0 = a0 * live; cb0 = cb0 * live
        A["abar0"][:, t] = a0
        hb0n = a0 @ Whh0
    return A


def second_order_grads(p, S, R, Tn, A, x: Tensor, v: Tensor, mask):
    """Parameter gradient of sum_b <v_b, dD/dx_b> (x = interpolates, v = input direction)."""
    B, T, H4 = S["gates0"].shape
    H = H4 // 4
    z = S["h0"].new_zeros(B, 1, H)
    sh = lambda a: torch.cat([z, a[:, :-1]], 1)
    m = (lambda a: a) if mask is None else (lambda a: a * mask)
    f2 = lambda a: a.reshape(-1, a.shape[-1])
    a0, a1, d0, d1 = f2(A["abar0"]), f2(A["abar1"]), f2(R["delta0"]), f2(R["delta1"])
    G = {
        "lstm.weight_ih_l0": a0.t() @ f2(x) + d0.t() @ f2(v),
        "lstm.weight_hh_l0": a0.t() @ f2(sh(S["h0"])) + d0.t() @ f2(sh(Tn["hdot0"])),
        "lstm.bias_ih_l0": a0.sum(0), "lstm.bias_hh_l0": a0.sum(0),
        "lstm.weight_ih_l1": a1.t() @ f2(m(S["h0"])) + d1.t() @ f2(m(Tn["hdot0"])),
        "lstm.weight_hh_l1": a1.t() @ f2(sh(S["h1"])) + d1.t() @ f2(sh(Tn["hdot1"])),
        "lstm.bias_ih_l1": a1.sum(0), "lstm.bias_hh_l1": a1.sum(0),
    }
    fcw = torch.zeros_like(p["fc.weight"])
    fcw[0] = Tn["hdotT"].sum(0)
    G["fc.weight"] = fcw
    G["fc.bias"] = torch.zeros_like(p["fc.bias"])
    return G


def gp_by_sweeps(p, x_hat: Tensor, lengths: Tensor, mask, upstream: float = 1.0):
    """Gradient penalty value, input gradient g and parameter gradients of upstream*gp,
    all by the four sweeps (no autograd)."""
    B = x_hat.shape[0]
    Wih0 = p["lstm.weight_ih_l0"]
    b0 = p["lstm.bias_ih_l0"] + p["lstm.bias_hh_l0"]
    S, out = sweep_forward(p, x_hat @ Wih0.t() + b0, lengths, mask)
    e0 = torch.zeros_like(out); e0[:, 0] = 1
    R = sweep_backward(p, S, lengths, mask, e0)
    g = R["delta0"] @ Wih0                                         # [B,T,V]
    n = g.reshape(B, -1).norm(2, dim=1)
    gp = ((n - 1) ** 2).mean()
    coef = upstream * 2 * (n - 1) / (B * n)
    v = coef.view(B, 1, 1) * g
    Tn = sweep_tangent(p, S, lengths, mask, v @ Wih0.t())
    A = sweep_second_reverse(p, S, R, Tn, lengths, mask)
    G = second_order_grads(p, S, R, Tn, A, x_hat, v, mask)
    return gp, g, G, dict(S=S, R=R, Tn=Tn, A=A, out=out, v=v)
