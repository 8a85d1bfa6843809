"""Host side of the attention path (reference: MaskedMultiheadAttention, transformer.py:196-282): the fused
in-projection of concat(q, k, v) with a [3E, 3E] weight, the attention core, the output projection, and the
hand-derived backward of all three, every product in libs2g kernels (torch provides memory and autograd
plumbing only)."""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch
from torch.autograd.function import once_differentiable

from . import ops
from .lib import S2GAttn, S2GError, call
from .ops import ptr, stream_ptr


def _dims(N, H, T, D, E):
    return S2GAttn(N, H, T, D, 3 * E, E, 1.0 / math.sqrt(E))      # scaled by the embedding size (transformer.py:267)


class MultiheadAttentionFn(torch.autograd.Function):
    """out = fc_out(attention(in_proj(concat(q, k, v)))) for [N, T, E] inputs (dim 0 is what the reference calls
    N, transformer.py:228).  attn_mask float [T, T] or None, pad_mask float [N, T] or None."""

    @staticmethod
    def forward(ctx, values, keys, queries, attn_mask, pad_mask, in_w, in_b, out_w, out_b, num_heads):
        for t in (values, keys, queries, in_w, in_b, out_w, out_b):
            if not t.is_cuda:
                raise S2GError("MaskedMultiheadAttention needs GPU tensors (MI355X-only implementation, no CPU path)")
        N, T, E = queries.shape
        if keys.shape != queries.shape or values.shape != queries.shape:
            raise S2GError("MaskedMultiheadAttention: q, k, v must share one shape (the reference concatenates them, "
                           "transformer.py:232)")
        D = E // num_heads
        if D * num_heads != E or D % 16 or D > 64:
            raise S2GError("attention kernels need head_dim = embed_size / num_heads to be 16, 32, 48 or 64")
        if T > 2048:
            raise S2GError("attention kernels keep one row of scores per query in LDS: sequence length <= 2048")
        dev = queries.device
        x3 = torch.cat([queries, keys, values], dim=-1).contiguous().float()          # transformer.py:232
        in_w, in_b = in_w.contiguous().float(), in_b.contiguous().float()
        out_w, out_b = out_w.contiguous().float(), out_b.contiguous().float()
        M = N * T
        qkv = torch.empty(M, 3 * E, device=dev)
        # F.linear(qkv, in_proj_weight.T, bias) = qkv @ in_proj_weight + bias (transformer.py:235)
        ops.gemm_raw(ptr(x3), 3 * E, 1, ptr(in_w), 3 * E, 1, ptr(qkv), 3 * E, M, 3 * E, 3 * E, bias=ptr(in_b))
        am = None if attn_mask is None else attn_mask.to(dev, torch.float32).contiguous()
        pm = None if pad_mask is None else pad_mask.to(dev, torch.float32).contiguous()
        if am is not None and tuple(am.shape) != (T, T):
            raise S2GError(f"attn_mask must be [{T}, {T}]")
        if pm is not None and tuple(pm.shape) != (N, T):
            raise S2GError(f"key padding mask must be [{N}, {T}]")
        heads = torch.empty(M, E, device=dev)
        P = torch.empty(N, num_heads, T, T, device=dev)
        dm = _dims(N, num_heads, T, D, E)
        call("s2g_attention_fwd", ptr(qkv), ptr(qkv, E), ptr(qkv, 2 * E), ptr(am), ptr(pm), ptr(heads), ptr(P),
             ctypes.byref(dm), stream_ptr())
        y = torch.empty(M, E, device=dev)
        ops.linear_fwd(heads, out_w, out_b, y)                                          # transformer.py:277
        ctx.save_for_backward(x3, qkv, heads, P, in_w, out_w)
        ctx.masks, ctx.dims = (am, pm), (N, T, E, num_heads, D)
        return y.view(N, T, E)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x3, qkv, heads, P, in_w, out_w = ctx.saved_tensors
        am, pm = ctx.masks
        N, T, E, H, D = ctx.dims
        dev, M = dy.device, N * T
        dy = dy.contiguous().float().view(M, E)
        sc = ops.Scratch(dev)
        dheads = torch.empty(M, E, device=dev)
        ops.linear_bwd_dx(dy, out_w, dheads, sc)
        d_out_w = torch.empty_like(out_w)
        ops.linear_bwd_dw(dy, heads, d_out_w, sc)
        d_out_b = torch.empty(E, device=dev)
        ops.colsum(dy, M, E, E, d_out_b, sc)
        dqkv = torch.empty(M, 3 * E, device=dev)
        dS = torch.empty_like(P)
        dm = _dims(N, H, T, D, E)
        call("s2g_attention_bwd", ptr(qkv), ptr(qkv, E), ptr(qkv, 2 * E), ptr(am), ptr(pm), ptr(P), ptr(dheads), ptr(dS),
             ptr(dqkv), ptr(dqkv, E), ptr(dqkv, 2 * E), ctypes.byref(dm), stream_ptr())
        # in-projection: qkv = x3 @ W + b  ->  dW = x3^T dqkv, db = colsum(dqkv), dx3 = dqkv @ W^T
        d_in_w = torch.empty_like(in_w)
        ops.linear_bwd_dw(x3.view(M, 3 * E), dqkv, d_in_w, sc)
        d_in_b = torch.empty(3 * E, device=dev)
        ops.colsum(dqkv, M, 3 * E, 3 * E, d_in_b, sc)
        dx3 = torch.empty(M, 3 * E, device=dev)
        ops.gemm_raw(ptr(dqkv), 3 * E, 1, ptr(in_w), 1, 3 * E, ptr(dx3), 3 * E, M, 3 * E, 3 * E)
        dx3 = dx3.view(N, T, 3 * E)
        dq, dk, dv = dx3[..., :E], dx3[..., E:2 * E], dx3[..., 2 * E:]
        return dv, dk, dq, None, None, d_in_w, d_in_b, d_out_w, d_out_b, None


# ----------------------------------------------------------------------------------------------------------------
# The pieces CustomTransformerEncoderLayer.forward puts around the attention module (transformer.py:298-317):
# residual + LayerNorm, the feed-forward Linears with ReLU, the dropout multipliers.  Each is an autograd Function
# over libs2g kernels; the layer composes them in the reference's order.
# ----------------------------------------------------------------------------------------------------------------
class AddLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + r * rmask) over the last dimension (rmask: the dropout multiplier of r, or None)."""

    @staticmethod
    def forward(ctx, x, r, rmask, gamma, beta, eps):
        if not x.is_cuda:
            raise S2GError("CustomTransformerEncoderLayer needs GPU tensors (MI355X-only implementation, no CPU path)")
        E = x.shape[-1]
        if E > 1024:
            raise S2GError("the LayerNorm kernel takes up to 1024 features")
        rows = x.numel() // E
        xc, rc = x.contiguous().float(), r.contiguous().float()
        mc = None if rmask is None else rmask.contiguous().float()
        g, b = gamma.contiguous().float(), beta.contiguous().float()
        y, xhat, rstd = torch.empty_like(xc), torch.empty_like(xc), torch.empty(rows, device=x.device)
        call("s2g_add_layernorm_fwd", ptr(xc), ptr(rc), ptr(mc), ptr(g), ptr(b), float(eps), ptr(y), ptr(xhat), ptr(rstd),
             rows, E, stream_ptr())
        ctx.save_for_backward(xhat, rstd, g, mc)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        xhat, rstd, g, mc = ctx.saved_tensors
        E = xhat.shape[-1]
        rows = xhat.numel() // E
        dy = dy.contiguous().float()
        dz, dyxh = torch.empty_like(xhat), torch.empty_like(xhat)
        call("s2g_add_layernorm_bwd", ptr(dy), ptr(xhat), ptr(rstd), ptr(g), ptr(dz), ptr(dyxh), rows, E, stream_ptr())
        sc = ops.Scratch(dy.device)
        dgamma, dbeta = torch.empty(E, device=dy.device), torch.empty(E, device=dy.device)
        ops.colsum(dyxh, rows, E, E, dgamma, sc)
        ops.colsum(dy, rows, E, E, dbeta, sc)
        dr = dz
        if mc is not None:
            dr = dz.clone()
            call("s2g_mul_inplace", ptr(dr), ptr(mc), dr.numel(), stream_ptr())
        return dz, dr, None, dgamma, dbeta, None


class LinearActFn(torch.autograd.Function):
    """y = act(x W^T + b) over the last dimension; act 0: none, 2: ReLU (fused into the product's epilogue)."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        if not x.is_cuda:
            raise S2GError("CustomTransformerEncoderLayer needs GPU tensors (MI355X-only implementation, no CPU path)")
        K, N = x.shape[-1], W.shape[0]
        M = x.numel() // K
        xc, Wc, bc = x.contiguous().float().view(M, K), W.contiguous().float(), b.contiguous().float()
        y = torch.empty(M, N, device=x.device)
        ops.linear_fwd(xc, Wc, bc, y, act=act)
        ctx.save_for_backward(xc, Wc, y)
        ctx.act, ctx.lead = act, tuple(x.shape[:-1])
        return y.view(*ctx.lead, N)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        xc, Wc, y = ctx.saved_tensors
        M, K = xc.shape
        N = Wc.shape[0]
        dy = dy.contiguous().float().view(M, N)
        if ctx.act == 2:
            dy = dy.clone()
            call("s2g_relu_bwd", ptr(y), ptr(dy), dy.numel(), stream_ptr())
        sc = ops.Scratch(dy.device)
        dx, dW, db = torch.empty(M, K, device=dy.device), torch.empty_like(Wc), torch.empty(N, device=dy.device)
        ops.linear_bwd_dx(dy, Wc, dx, sc)
        ops.linear_bwd_dw(dy, xc, dW, sc)
        ops.colsum(dy, M, N, N, db, sc)
        return dx.view(*ctx.lead, K), dW, db, None


class MulMaskFn(torch.autograd.Function):
    """y = x * mask (a dropout multiplier: 0 or 1 / (1 - p))."""

    @staticmethod
    def forward(ctx, x, mask):
        y = x.contiguous().float().clone()
        m = mask.contiguous().float()
        call("s2g_mul_inplace", ptr(y), ptr(m), y.numel(), stream_ptr())
        ctx.save_for_backward(m)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (m,) = ctx.saved_tensors
        dx = dy.contiguous().float().clone()
        call("s2g_mul_inplace", ptr(dx), ptr(m), dx.numel(), stream_ptr())
        return dx, None


def dropout_multiplier(shape, p: float, device, rng: ops.DeviceRng) -> Optional[torch.Tensor]:
    """A train-mode dropout multiplier of the given shape drawn by the library's counter-based generator
    (None for p = 0: nothing is dropped)."""
    if p <= 0.0:
        return None
    width = shape[-1]
    rows = 1
    for d in shape[:-1]:
        rows *= d
    if width % 4:
        raise S2GError("dropout multipliers are drawn four columns at a time: the last dimension must be a multiple of 4")
    return ops.dropout_mask_rng(rows, p, device, rng, width=width, live=width).view(*shape)
