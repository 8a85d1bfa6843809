"""CPU restatement of the reference WGAN-GP training step (TEST INFRASTRUCTURE ONLY).

This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  It restates, in plain per-timestep PyTorch
(so every random draw and every dropout mask is an explicit argument), the algorithm of
the reference's wgan_gp.py.  It is pinned against golden vectors produced by importing
the reference itself (tests/golden/make_golden.py, tests/test_oracle_golden.py).

Each function cites the reference lines it follows.  All arithmetic runs in the dtype of
the tensors handed in (float32 for parity with the reference, float64 for tight checks).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

LEAKY_SLOPE = 0.2      # wgan_gp.py:151  nn.LeakyReLU(0.2)
BN_EPS = 0.8           # wgan_gp.py:150  nn.BatchNorm1d(out_feat, 0.8) -> second positional arg is eps
BN_MOMENTUM = 0.1      # torch default
CLASS_WEIGHT = (1.0, 2.0)  # wgan_gp.py:267


# --------------------------------------------------------------------------- sampler
def gumbel_noise(U: Tensor, eps: float = 1e-10) -> Tensor:
    """wgan_gp.py:56-61 with the uniform draw `torch.rand(...)` replaced by the argument."""
    U = eps + (1 - eps) * U
    return -torch.log(-torch.log(U + 1e-20) + 1e-20)


def gumbel_softmax(logits: Tensor, U: Tensor, tau: float, hard: bool) -> Tensor:
    """wgan_gp.py:63-86.  Straight-through when `hard`: forward one-hot, backward soft."""
    y = F.softmax((logits + gumbel_noise(U)) / tau, dim=-1)
    if hard:
        y_hard = F.one_hot(y.argmax(dim=-1), num_classes=logits.size(-1)).to(y.dtype)
        y = (y_hard - y).detach() + y
    return y


def relaxed_bernoulli(logits: Tensor, tau: float = 1.0) -> Tensor:
    """wgan_gp.py:94-104."""
    return F.softmax(logits / tau, dim=-1)


def anneal_tau(initial_tau, final_tau, current_epoch, total_epochs, anneal_interval):
    """wgan_gp.py:88-92."""
    if current_epoch % anneal_interval == 0:
        return initial_tau - (initial_tau - final_tau) * (current_epoch / total_epochs)
    return initial_tau


# --------------------------------------------------------------------------- generator
def generator_forward(p: Params, bn: Dict[str, Tensor], z: Tensor, labels: Tensor, U: Tensor,
                      tau: float, hard: bool, seq_len: int, vocab: int,
                      training: bool = True) -> Tensor:
    """wgan_gp.py:165-171 (+ layer stack 154-160).

    `p` uses the reference state_dict names (model.0.weight ... label_embedding.weight);
    `bn` holds model.{3,6}.running_{mean,var} and is updated in place when training, the
    way nn.BatchNorm1d does (biased variance to normalise, unbiased for the running stat).
    """
    x = z * p["label_embedding.weight"][labels]
    x = F.leaky_relu(x @ p["model.0.weight"].t() + p["model.0.bias"], LEAKY_SLOPE)
    for lin, nb in (("model.2", "model.3"), ("model.5", "model.6")):
        x = x @ p[lin + ".weight"].t() + p[lin + ".bias"]
        if training:
            mean = x.mean(0)
            var = x.var(0, unbiased=False)
            with torch.no_grad():
                n = x.shape[0]
                bn[nb + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
                bn[nb + ".running_var"].mul_(1 - BN_MOMENTUM).add_(
                    BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
                if nb + ".num_batches_tracked" in bn:
                    bn[nb + ".num_batches_tracked"] += 1
        else:
            mean, var = bn[nb + ".running_mean"], bn[nb + ".running_var"]
        x = (x - mean) / torch.sqrt(var + BN_EPS) * p[nb + ".weight"] + p[nb + ".bias"]
        x = F.leaky_relu(x, LEAKY_SLOPE)
    logits = x @ p["model.8.weight"].t() + p["model.8.bias"]
    logits = logits.view(z.shape[0], seq_len, vocab)
    return gumbel_softmax(logits, U, tau, hard)


# --------------------------------------------------------------------------- critic
def lstm_cell(x_proj: Tensor, h: Tensor, c: Tensor, w_hh: Tensor) -> Tuple[Tensor, Tensor]:
    """One step of torch.nn.LSTM (gate order i, f, g, o) given the input projection."""
    gates = x_proj + h @ w_hh.t()
    H = h.shape[1]
    i = torch.sigmoid(gates[:, 0:H])
    f = torch.sigmoid(gates[:, H:2 * H])
    g = torch.tanh(gates[:, 2 * H:3 * H])
    o = torch.sigmoid(gates[:, 3 * H:4 * H])
    c = f * c + i * g
    h = o * torch.tanh(c)
    return h, c


def critic_forward(p: Params, x: Tensor, lengths: Optional[Tensor] = None,
                   drop_mask: Optional[Tensor] = None) -> Tensor:
    """wgan_gp.py:194-223: 2-layer LSTM -> last layer's final hidden state -> fc.

    x: [B,T,V] dense.  `lengths` [B] gives pack_padded_sequence semantics (state frozen
    after each sample's own length, so the "final" state is the one at len-1); None means
    every sample runs all T steps (the dense call of wgan_gp.py:302,333).
    drop_mask: [B,T,H] multiplicative mask (entries 0 or 1/(1-p)) applied to the layer-0
    output sequence before it feeds layer 1, which is what nn.LSTM(dropout=p) does in
    train mode; None = no dropout (eval mode or p = 0).
    """
    B, T, _ = x.shape
    H = p["lstm.weight_hh_l0"].shape[1]
    h0 = x.new_zeros(B, H); c0 = x.new_zeros(B, H)
    h1 = x.new_zeros(B, H); c1 = x.new_zeros(B, H)
    b0 = p["lstm.bias_ih_l0"] + p["lstm.bias_hh_l0"]
    b1 = p["lstm.bias_ih_l1"] + p["lstm.bias_hh_l1"]
    xp = x @ p["lstm.weight_ih_l0"].t() + b0            # [B,T,4H]
    for t in range(T):
        nh0, nc0 = lstm_cell(xp[:, t], h0, c0, p["lstm.weight_hh_l0"])
        y0 = nh0 if drop_mask is None else nh0 * drop_mask[:, t]
        nh1, nc1 = lstm_cell(y0 @ p["lstm.weight_ih_l1"].t() + b1, h1, c1, p["lstm.weight_hh_l1"])
        if lengths is None:
            h0, c0, h1, c1 = nh0, nc0, nh1, nc1
        else:
            live = (lengths > t).to(x.dtype).unsqueeze(1)
            h0 = live * nh0 + (1 - live) * h0
            c0 = live * nc0 + (1 - live) * c0
            h1 = live * nh1 + (1 - live) * h1
            c1 = live * nc1 + (1 - live) * c1
    return h1 @ p["fc.weight"].t() + p["fc.bias"]


def pad_or_crop_time(fake: Tensor, T: int) -> Tensor:
    """wgan_gp.py:231-233: F.pad(fake, (0,0,0,T-Tf)); a negative pad crops."""
    Tf = fake.shape[1]
    if T >= Tf:
        return F.pad(fake, (0, 0, 0, T - Tf))
    return fake[:, :T]


def gradient_penalty(p: Params, real: Tensor, fake: Tensor, alpha: Tensor, lengths: Tensor,
                     drop_mask: Optional[Tensor] = None, create_graph: bool = True):
    """wgan_gp.py:225-257.  alpha: [B,1,1] (the `rand` draw of line 228, injected).

    Returns (gp, gradients [B,T,V]).  gp carries the graph for the second-order backward.
    """
    fake_p = pad_or_crop_time(fake, real.shape[1])
    x_hat = (alpha * real + (1 - alpha) * fake_p).detach().requires_grad_(True)
    d = critic_forward(p, x_hat, lengths, drop_mask)[:, 0]
    (g,) = torch.autograd.grad(d, x_hat, grad_outputs=torch.ones_like(d),
                               create_graph=create_graph, retain_graph=True)
    gp = ((g.reshape(g.shape[0], -1).norm(2, dim=1) - 1) ** 2).mean()
    return gp, g


def weighted_ce_on_probs(probs: Tensor, labels: Tensor) -> Tensor:
    """wgan_gp.py:267-268,299,310: CrossEntropyLoss(weight=[1,2]) applied to inputs that
    are ALREADY softmax probabilities (so a second log_softmax happens inside)."""
    w = torch.tensor(CLASS_WEIGHT, dtype=probs.dtype, device=probs.device)
    return F.cross_entropy(probs, labels, weight=w)


def critic_loss(p: Params, real: Tensor, lengths: Tensor, fake: Tensor, labels: Tensor,
                alpha: Tensor, lambda_gp: float, masks=(None, None, None)):
    """wgan_gp.py:297-315.  Returns (d_loss, parts dict)."""
    out_r = critic_forward(p, real, lengths, masks[0])
    out_f = critic_forward(p, fake, None, masks[1])
    gp, g = gradient_penalty(p, real, fake, alpha, lengths, masks[2])
    ce = 0.5 * (weighted_ce_on_probs(F.softmax(out_r[:, 1:], -1), labels)
                + weighted_ce_on_probs(F.softmax(out_f[:, 1:], -1), labels))
    d_loss = -out_r[:, 0].mean() + out_f[:, 0].mean() + lambda_gp * gp + ce
    return d_loss, dict(out_real=out_r, out_fake=out_f, gp=gp, grad=g, ce=ce)


def generator_loss(p_d: Params, fake: Tensor, labels: Tensor, mask=None):
    """wgan_gp.py:333-340."""
    out = critic_forward(p_d, fake, None, mask)
    g_class = 0.5 * weighted_ce_on_probs(F.softmax(out[:, 1:], -1), labels)
    return -out[:, 0].mean() + 0.5 * g_class, out


# --------------------------------------------------------------------------- optimiser
def adam_update(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
                b1: float, b2: float, eps: float = 1e-8) -> None:
    """torch.optim.Adam single-tensor rule (no weight decay / amsgrad), in place.
    wgan_gp.py:405-406 constructs Adam(lr, betas=(b1,b2)); `step` counts from 1."""
    m.mul_(b1).add_(grad, alpha=1 - b1)
    v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-(lr / bc1))


class Adam:
    """Minimal stateful wrapper over adam_update for a dict of leaf tensors."""

    def __init__(self, params: Params, lr: float, b1: float, b2: float):
        self.params, self.lr, self.b1, self.b2 = params, lr, b1, b2
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = {k: 0 for k in params}

    def step(self, grads: Dict[str, Optional[Tensor]]) -> None:
        with torch.no_grad():
            for k, g in grads.items():
                if g is None:
                    continue
                self.t[k] += 1
                adam_update(self.params[k], g, self.m[k], self.v[k], self.t[k], self.lr,
                            self.b1, self.b2)


# --------------------------------------------------------------------------- the step
def one_hot_real(tokens: Tensor, vocab: int, dtype=torch.float32) -> Tensor:
    """wgan_gp.py:277."""
    return F.one_hot(tokens, vocab).to(dtype)


def critic_iteration(pg: Params, bn: Dict[str, Tensor], pd: Params, opt_d: Adam, tokens: Tensor,
                     labels: Tensor, lengths: Tensor, z: Tensor, U: Tensor, alpha: Tensor,
                     cfg: dict, masks=(None, None, None)):
    """wgan_gp.py:277-318: one critic update on one batch.  Returns d_loss (float) and the
    fake batch used (detached)."""
    dtype = z.dtype
    real = one_hot_real(tokens, cfg["vocab"], dtype)
    with torch.no_grad():
        fake = generator_forward(pg, bn, z, labels, U, cfg["tau"], cfg["hard"],
                                 cfg["gen_seq_len"], cfg["vocab"], training=True)
    leaves = {k: v.detach().requires_grad_(True) for k, v in pd.items()}
    d_loss, parts = critic_loss(leaves, real, lengths, fake, labels, alpha, cfg["lambda_gp"], masks)
    names = list(leaves)
    grads = torch.autograd.grad(d_loss, [leaves[k] for k in names], allow_unused=True)
    gd = dict(zip(names, grads))
    opt_d.step(gd)
    return float(d_loss.detach()), fake, gd, parts


def generator_iteration(pg: Params, bn: Dict[str, Tensor], pd: Params, opt_g: Adam, labels: Tensor,
                        z: Tensor, U: Tensor, cfg: dict, mask=None):
    """wgan_gp.py:330-343: regenerate with the same z/labels and fresh Gumbel noise, update G."""
    leaves = {k: v.detach().requires_grad_(True) for k, v in pg.items()}
    fake = generator_forward(leaves, bn, z, labels, U, cfg["tau"], cfg["hard"],
                             cfg["gen_seq_len"], cfg["vocab"], training=True)
    g_loss, _ = generator_loss(pd, fake, labels, mask)
    names = list(leaves)
    grads = torch.autograd.grad(g_loss, [leaves[k] for k in names], allow_unused=True)
    gg = dict(zip(names, grads))
    opt_g.step(gg)
    return float(g_loss.detach()), gg


def full_step(pg, bn, pd, opt_g, opt_d, batches, noise, cfg):
    """n_critic consecutive batches = one "full WGAN-GP step" (wgan_gp.py:274-343): every
    batch updates the critic; the generator is updated on batch i when i % n_critic == 0,
    AFTER that batch's critic update, with the same z/labels and a second Gumbel draw.

    batches[i] = (tokens [B,T] int64, labels [B] int64, lengths [B] int64)
    noise[i]   = dict(z, U, alpha[, U2][, masks][, mask_g])
    Returns lists of d_losses and g_losses.
    """
    d_losses, g_losses = [], []
    for i, ((tokens, labels, lengths), nz) in enumerate(zip(batches, noise)):
        dl, _, _, _ = critic_iteration(pg, bn, pd, opt_d, tokens, labels, lengths, nz["z"], nz["U"],
                                       nz["alpha"], cfg, nz.get("masks", (None, None, None)))
        d_losses.append(dl)
        if i % cfg["n_critic"] == 0:
            gl, _ = generator_iteration(pg, bn, pd, opt_g, labels, nz["z"], nz["U2"], cfg,
                                        nz.get("mask_g"))
            g_losses.append(gl)
    return d_losses, g_losses
