"""torch.autograd bridges: the reference's module surface (CustomSoftmax / Generator /
Discriminator / compute_gradient_penalty, wgan_gp.py:22-257) expressed as autograd Functions
whose forward and backward enqueue libs2g kernels.  Parameters of an nn.Module are "adopted":
their storage is moved into the engine's flat buffer so that the module, torch optimisers and
the kernels all see the same memory."""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd.function import once_differentiable

from . import ops
from .critic import CriticEngine, _wgrad, make_desc
from .generator import GeneratorEngine
from .lib import S2GError, call
from .ops import G4, H, MASKW, ptr, stream_ptr


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise S2GError(f"{what}: this implementation runs on an MI355X only (got a {t.device} tensor); "
                       "there is deliberately no CPU path")


def adopt(module: torch.nn.Module, flat, names=None) -> None:
    """Move the module's parameters into the flat buffer's views (once, or again after .to())."""
    named = dict(module.named_parameters())
    named.update(dict(module.named_buffers()))
    for name, view in flat.views.items():
        p = named[name]
        if p.data_ptr() != view.data_ptr():
            view.copy_(p.data.to(view.device, torch.float32))
            p.data = view


# ----------------------------------------------------------------------------- sampler
class SamplerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, U, tau, mode, hard):
        _require_gpu(logits, "CustomSoftmax")
        lg = logits.contiguous().float()
        y, y_soft = ops.gumbel_softmax_fwd(lg, None if U is None else U.contiguous().float(), tau, mode, hard)
        ctx.save_for_backward(y_soft)
        ctx.scale = 1.0 if mode == 2 else 1.0 / tau
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y_soft,) = ctx.saved_tensors
        return ops.gumbel_softmax_bwd(y_soft, dy.contiguous(), ctx.scale), None, None, None, None


# ----------------------------------------------------------------------------- generator
class GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng: GeneratorEngine, noise, labels, U, tau, hard, training, act, *params):
        y = eng.forward(noise.contiguous().float(), labels.contiguous(), U, tau, hard, training=training,
                        save=True, act=act)
        ctx.eng = eng
        ctx.saved = eng.saved
        ctx.names = list(eng.params.views)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        eng = ctx.eng
        eng.saved = ctx.saved
        eng.backward(dy)
        grads = [eng.grads.views[n].clone() for n in ctx.names]
        return (None,) * 8 + tuple(grads)


# ----------------------------------------------------------------------------- critic
def _mask52(mask: Optional[torch.Tensor], rows: int, device):
    if mask is None:
        return None
    m = torch.zeros(rows, MASKW, device=device)
    m[:, :H] = mask.reshape(rows, H)
    return m


class CriticFn(torch.autograd.Function):
    """out = fc(LSTM(x)[final hidden of layer 2])  (Discriminator.forward, wgan_gp.py:194-223).

    Its backward is itself differentiable (CriticBackwardFn), so a caller can write the reference's own
    penalty, autograd.grad(D(x)[:, 0], x, create_graph=True) followed by a backward of a function of that
    gradient (wgan_gp.py:247-256): the first backward is the reverse sweep, the backward of THAT is the tangent
    sweep along the upstream gradient and the second reverse sweep."""

    @staticmethod
    def forward(ctx, eng: CriticEngine, x, lengths_cpu, mask, *params):
        _require_gpu(x, "Discriminator")
        B, T, V = x.shape
        x = x.contiguous().float()
        eng.repack()
        # an input that requires a gradient may be differentiated twice (the gradient penalty): the workspace then
        # also holds the buffers of the second-order pass
        ws = eng.fresh_workspace(B, T, B if ctx.needs_input_grad[1] else 0)
        lens = [T] * B if lengths_cpu is None else [int(v) for v in lengths_cpu]
        m52 = _mask52(mask, B * T, x.device)
        desc = make_desc([dict(len=lens[b], p_off=b * T, p_len=T, ca=0.0, cp=1.0, ws_row=b * T,
                               mask_off=(b * T if m52 is not None else -1)) for b in range(B)], x.device)
        P = torch.empty(B * T, G4, device=x.device)
        eng.project(x, P)
        eng.sweep_fwd(ws, ptr(desc), B, T, None, P, m52)
        eng.head_fwd(ws, B)
        ctx.state = dict(eng=eng, ws=ws, desc=desc, m52=m52, x=x, dims=(B, T, V), names=list(eng.params.views))
        ctx.save_for_backward(*params)
        return ws["out"].clone()

    @staticmethod
    def backward(ctx, dout):
        res = CriticBackwardFn.apply(ctx.state, ctx.needs_input_grad[1], dout, *ctx.saved_tensors)
        return (None, res[0], None, None) + tuple(res[1:])


class CriticBackwardFn(torch.autograd.Function):
    """(dx, parameter gradients) of sum(out * dout) for a CriticFn forward: the reverse sweep (BPTT) and the batched
    weight-gradient products.  Its own backward, for an upstream gradient v on dx: d/d(theta) <v, dx(theta)> is the
    forward-mode derivative of the critic along v, differentiated once more -- the tangent sweep seeded by
    v W_ih0^T and the second reverse sweep, the same kernels the fused penalty uses (with d(out) general instead
    of e_0).  Also returned: the gradient with respect to dout (the tangent of the outputs).  NOT provided: the
    gradient of dx with respect to the input x itself (returned as None; the reference's penalty never uses it),
    and upstream gradients on the parameter-gradient outputs (refused)."""

    @staticmethod
    def forward(ctx, state, want_dx, dout, *params):
        eng, ws, (B, T, V) = state["eng"], state["ws"], state["dims"]
        st = stream_ptr()
        p = eng.params.views
        second = "adj0" in ws
        ws["dout"].copy_(dout)
        call("s2g_head_bwd_dh", ptr(ws["dout"]), ptr(p["fc.weight"]), ptr(ws["dhT"]), B, 3, st)
        eng.sweep_bwd(ws, ptr(state["desc"]), B, T, state["m52"], adj_seq0=0 if second else -1)
        g = {n: torch.empty_like(v) for n, v in p.items()}
        call("s2g_head_bwd_w", ptr(ws["dout"]), ptr(ws["hT"]), ptr(g["fc.weight"]), ptr(g["fc.bias"]), B, 3, 0, st)
        d0, d1, c0, c1 = ws["delta0"], ws["delta1"], ws["cell0"], ws["cell1"]
        n = B * T
        _wgrad(eng, g["lstm.weight_ih_l0"], V, [(ptr(d0), n, ptr(state["x"]), V, 1, 0, 0, 0, False)])
        _wgrad(eng, g["lstm.weight_hh_l0"], H, [(ptr(d0), n, ptr(c0, 2), G4, 4, T, T, T, True)])
        _wgrad(eng, g["lstm.weight_ih_l1"], H, [(ptr(d1), n, ptr(c0, 3), G4, 4, 0, 0, 0, False)])
        _wgrad(eng, g["lstm.weight_hh_l1"], H, [(ptr(d1), n, ptr(c1, 2), G4, 4, T, T, T, True)])
        ops.colsum(d0, n, G4, G4, g["lstm.bias_ih_l0"], eng.scratch, permute_rows=True)
        ops.colsum(d1, n, G4, G4, g["lstm.bias_ih_l1"], eng.scratch, permute_rows=True)
        g["lstm.bias_hh_l0"].copy_(g["lstm.bias_ih_l0"])
        g["lstm.bias_hh_l1"].copy_(g["lstm.bias_ih_l1"])
        dx = None
        if want_dx:
            dx = torch.empty(B, T, V, device=dout.device)
            ops.gemm_tall(ptr(d0), G4, ptr(eng.W0p), eng.Vp, ptr(dx), V, n, V, G4)
        ctx.state, ctx.second = state, second
        ctx.dout = ws["dout"].clone()
        ctx.set_materialize_grads(False)
        return (dx,) + tuple(g[k] for k in state["names"])

    @staticmethod
    @once_differentiable
    def backward(ctx, v, *gparams):
        state = ctx.state
        eng, ws, (B, T, V) = state["eng"], state["ws"], state["dims"]
        if any(t is not None for t in gparams):
            raise S2GError("Discriminator: differentiating through its parameter gradients is not supported "
                           "(only through the input gradient, as the gradient penalty does)")
        if not ctx.second:
            raise S2GError("Discriminator: a second-order pass needs an input that requires a gradient at forward time")
        names = state["names"]
        if v is None:
            return (None, None, None) + (None,) * len(names)
        v = v.contiguous().float().view(B * T, V)
        eng.project(v, ws["Pv"])                              # the direction in gate space: v W_ih0^T
        mp = ptr(state["m52"])
        eng.sweep_tangent(ws, ptr(state["desc"]), B, mp, 0)
        eng.sweep_bwd2(ws, ptr(state["desc"]), B, T, mp, 0)
        p = eng.params.views
        g = {n: torch.zeros_like(x) for n, x in p.items()}
        d0, d1, c0, c1 = ws["delta0"], ws["delta1"], ws["cell0"], ws["cell1"]
        a0, a1, cd0, cd1 = ws["abar0"], ws["abar1"], ws["celld0"], ws["celld1"]
        n = B * T
        _wgrad(eng, g["lstm.weight_ih_l0"], V, [(ptr(a0), n, ptr(state["x"]), V, 1, 0, 0, 0, False),
                                                (ptr(d0), n, ptr(v), V, 1, 0, 0, 0, False)])
        _wgrad(eng, g["lstm.weight_hh_l0"], H, [(ptr(a0), n, ptr(c0, 2), G4, 4, T, T, T, True),
                                                (ptr(d0), n, ptr(cd0, 1), G4, 4, T, T, T, True)])
        _wgrad(eng, g["lstm.weight_ih_l1"], H, [(ptr(a1), n, ptr(c0, 3), G4, 4, 0, 0, 0, False),
                                                (ptr(d1), n, ptr(cd0, 2), G4, 4, 0, 0, 0, False)])
        _wgrad(eng, g["lstm.weight_hh_l1"], H, [(ptr(a1), n, ptr(c1, 2), G4, 4, T, T, T, True),
                                                (ptr(d1), n, ptr(cd1, 1), G4, 4, T, T, T, True)])
        ops.colsum(a0, n, G4, G4, g["lstm.bias_ih_l0"], eng.scratch, permute_rows=True)
        ops.colsum(a1, n, G4, G4, g["lstm.bias_ih_l1"], eng.scratch, permute_rows=True)
        g["lstm.bias_hh_l0"].copy_(g["lstm.bias_ih_l0"])
        g["lstm.bias_hh_l1"].copy_(g["lstm.bias_ih_l1"])
        # fc.weight[j] += sum_n dout[n, j] hdot_final[n]; d/d(dout)[n, j] = fc.weight[j] . hdot_final[n]
        ops.gemm_raw(ptr(ctx.dout), 1, 3, ptr(ws["hdotT"]), H, 1, ptr(g["fc.weight"]), H, 3, H, B)
        ddout = torch.empty(B, 3, device=v.device)
        ops.gemm_raw(ptr(ws["hdotT"]), H, 1, ptr(p["fc.weight"]), 1, H, ptr(ddout), 3, B, 3, H)
        return (None, None, ddout) + tuple(g[k] for k in names)


class GradientPenaltyFn(torch.autograd.Function):
    """gp = mean_b (|d D(x_hat_b)[0] / d x_hat_b| - 1)^2 with x_hat = alpha real + (1-alpha) pad(fake)
    (compute_gradient_penalty, wgan_gp.py:225-257).  Backward = the second-order sweeps."""

    @staticmethod
    def forward(ctx, eng: CriticEngine, real, fake, alpha, lengths_cpu, mask, *params):
        _require_gpu(real, "compute_gradient_penalty")
        B, T, V = real.shape
        Tf = fake.shape[1]
        dev = real.device
        st = stream_ptr()
        eng.repack()
        ws = eng.fresh_workspace(B, T, B)
        lens_d = lengths_cpu.to(torch.int32).to(dev)
        real = real.contiguous().float(); fake = fake.contiguous().float()
        alpha = alpha.reshape(-1).contiguous().float()
        xhat = torch.empty(B * T, V, device=dev)
        call("s2g_gp_interp_dense", ptr(real), ptr(lens_d), ptr(fake), ptr(alpha), B, T, Tf, V, ptr(xhat), st)
        m52 = _mask52(mask, B * T, dev)
        desc = make_desc([dict(len=int(lengths_cpu[b]), p_off=b * T, p_len=T, ca=0.0, cp=1.0, ws_row=b * T,
                               mask_off=(b * T if m52 is not None else -1)) for b in range(B)], dev)
        P = torch.empty(B * T, G4, device=dev)
        eng.project(xhat, P)
        eng.sweep_fwd(ws, ptr(desc), B, T, None, P, m52)
        ws["dout"].zero_(); ws["dout"][:, 0] = 1.0
        call("s2g_head_bwd_dh", ptr(ws["dout"]), ptr(eng.params.views["fc.weight"]), ptr(ws["dhT"]), B, 3, st)
        eng.sweep_bwd(ws, ptr(desc), B, T, m52, adj_seq0=0)
        ops.gemm_tall(ptr(ws["delta0"]), G4, ptr(eng.W0p), eng.Vp, ptr(ws["G"]), V, B * T, V, G4)
        call("s2g_gp_norm_penalty", ptr(ws["G"]), B, T * V, 1.0, float(B), ptr(ws["pen"]), ptr(ws["norms"]), 0, st)
        gp = torch.empty(1, device=dev)
        ops.reduce_slabs(ws["pen"], 1, B, 1, 1, gp, scale=1.0 / B)
        ctx.eng, ctx.ws, ctx.desc, ctx.m52, ctx.xhat, ctx.dims = eng, ws, desc, m52, xhat, (B, T, V)
        ctx.names = list(eng.params.views)
        return gp[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, ggp):
        eng, ws, (B, T, V) = ctx.eng, ctx.ws, ctx.dims
        up = ggp.reshape(1).contiguous().float()
        v = torch.empty(B * T, V, device=up.device)
        call("s2g_gp_scale", ptr(ws["G"]), ptr(ws["norms"]), ptr(up), float(B), T * V, ptr(v), B, stream_ptr())
        eng.project(v, ws["Pv"])
        mp = ptr(ctx.m52)
        eng.sweep_tangent(ws, ptr(ctx.desc), B, mp, 0)
        eng.sweep_bwd2(ws, ptr(ctx.desc), B, T, mp, 0)
        p = eng.params.views
        g = {n: torch.zeros_like(x) for n, x in p.items()}
        d0, d1, c0, c1 = ws["delta0"], ws["delta1"], ws["cell0"], ws["cell1"]
        a0, a1, cd0, cd1 = ws["abar0"], ws["abar1"], ws["celld0"], ws["celld1"]
        n = B * T
        _wgrad(eng, g["lstm.weight_ih_l0"], V, [(ptr(a0), n, ptr(ctx.xhat), V, 1, 0, 0, 0, False),
                                                (ptr(d0), n, ptr(v), V, 1, 0, 0, 0, False)])
        _wgrad(eng, g["lstm.weight_hh_l0"], H, [(ptr(a0), n, ptr(c0, 2), G4, 4, T, T, T, True),
                                                (ptr(d0), n, ptr(cd0, 1), G4, 4, T, T, T, True)])
        _wgrad(eng, g["lstm.weight_ih_l1"], H, [(ptr(a1), n, ptr(c0, 3), G4, 4, 0, 0, 0, False),
                                                (ptr(d1), n, ptr(cd0, 2), G4, 4, 0, 0, 0, False)])
        _wgrad(eng, g["lstm.weight_hh_l1"], H, [(ptr(a1), n, ptr(c1, 2), G4, 4, T, T, T, True),
                                                (ptr(d1), n, ptr(cd1, 1), G4, 4, T, T, T, True)])
        ops.colsum(a0, n, G4, G4, g["lstm.bias_ih_l0"], eng.scratch, permute_rows=True)
        ops.colsum(a1, n, G4, G4, g["lstm.bias_ih_l1"], eng.scratch, permute_rows=True)
        g["lstm.bias_hh_l0"].copy_(g["lstm.bias_ih_l0"])
        g["lstm.bias_hh_l1"].copy_(g["lstm.bias_ih_l1"])
        ops.colsum(ws["hdotT"], B, H, H, g["fc.weight"][0], eng.scratch)
        return (None,) * 6 + tuple(g[k] for k in ctx.names)
