"""MaskedMultiheadAttention on the HIP kernels (s2g_attention_fwd / _bwd + the fp32 MFMA GEMMs) through the
module surface of transformer.py: against outputs and gradients recorded from the reference module (g11,
1e-3 relative = the fp32 bound), and at the TenGAN shape (N=16, T=512, E=64, 4 heads) against the float64
CPU restatement."""
import pytest
import torch

from conftest import assert_close
from common import attention_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3
CASES = {"small": (1301, 3, 20, 32, 2), "selftest": (1302, 8, 50, 64, 4), "ragged": (1303, 2, 37, 64, 1)}


def build(c, E, H, batch_first=False):
    import transformer as T
    m = T.MaskedMultiheadAttention(E, H, batch_first=batch_first).to(DEV)
    with torch.no_grad():
        m.in_proj_weight.copy_(c["in_w"]); m.in_proj_bias.copy_(c["in_b"])
        m.fc_out.weight.copy_(c["out_w"]); m.fc_out.bias.copy_(c["out_b"])
    return T, m


@pytest.mark.parametrize("tag", list(CASES))
@pytest.mark.parametrize("mode", ["nomask", "masks", "batch_first"])
def test_module_matches_reference_golden(golden, tag, mode):
    G = golden("g11_attention.pt")
    seed, N, T_, E, H = CASES[tag]
    c = attention_case(seed, N, T_, E, H)
    T, m = build(c, E, H, batch_first=(mode == "batch_first"))
    q, k, v = (c[n].to(DEV).requires_grad_(True) for n in "qkv")
    am = T.create_causal_mask(T_) if mode != "nomask" else None
    pm = T.create_padding_mask(c["tokens"]).to(DEV) if mode != "nomask" else None
    if mode == "batch_first":
        y = m(v.transpose(0, 1), k.transpose(0, 1), q.transpose(0, 1), attn_mask=am, src_key_padding_mask=pm).transpose(0, 1)
    else:
        y = m(v, k, q, attn_mask=am, src_key_padding_mask=pm)
    grads = torch.autograd.grad(y, [q, k, v, m.in_proj_weight, m.in_proj_bias, m.fc_out.weight, m.fc_out.bias],
                                c["dy"].to(DEV))
    torch.cuda.synchronize()
    key = f"{tag}.{mode}"
    assert_close(y, G[f"{key}.y"], TOL, "y")
    for n, gv in zip(("dq", "dk", "dv", "d_in_w", "d_in_b", "d_out_w", "d_out_b"), grads):
        assert_close(gv, G[f"{key}.{n}"], TOL, n, atol=1e-7)


@pytest.mark.parametrize("masks", [False, True])
def test_tengan_shape_matches_float64_oracle(masks):
    from oracle.attention_cpu import attention_forward
    N, T_, E, H = 16, 512, 64, 4
    c = attention_case(1310, N, T_, E, H)
    T, m = build(c, E, H)
    q, k, v = (c[n].to(DEV).requires_grad_(True) for n in "qkv")
    am = T.create_causal_mask(T_) if masks else None
    pm_cpu = T.create_padding_mask(c["tokens"]) if masks else None
    y = m(v, k, q, attn_mask=am, src_key_padding_mask=None if pm_cpu is None else pm_cpu.to(DEV))
    grads = torch.autograd.grad(y, [q, k, v, m.in_proj_weight, m.in_proj_bias, m.fc_out.weight, m.fc_out.bias],
                                c["dy"].to(DEV))
    torch.cuda.synchronize()
    leaves = [c[n].double().requires_grad_(True) for n in ("q", "k", "v", "in_w", "in_b", "out_w", "out_b")]
    q6, k6, v6, *w6 = leaves
    y6 = attention_forward(v6, k6, q6, *w6, H, am, pm_cpu)
    g6 = torch.autograd.grad(y6, leaves, c["dy"].double())
    assert_close(y, y6, 1e-4, "y")
    for n, gv, gw in zip(("dq", "dk", "dv", "d_in_w", "d_in_b", "d_out_w", "d_out_b"), grads, g6):
        assert_close(gv, gw, 1e-4, n, atol=1e-8)


def test_rejects_cpu_tensors_and_odd_head_sizes():
    import transformer as T
    from s2g.lib import S2GError
    m = T.MaskedMultiheadAttention(32, 2)
    x = torch.zeros(2, 5, 32)
    with pytest.raises(S2GError):
        m(x, x, x)
    m = T.MaskedMultiheadAttention(24, 2).to(DEV)            # head_dim 12
    x = torch.zeros(2, 5, 24, device=DEV)
    with pytest.raises(S2GError):
        m(x, x, x)


USED = ("attn.in_proj_weight", "attn.in_proj_bias", "attn.fc_out.weight", "attn.fc_out.bias", "linear1.weight",
        "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


@pytest.mark.parametrize("mode", ["plain", "batch_first"])
def test_encoder_layer_matches_reference_golden(golden, mode):
    """CustomTransformerEncoderLayer (reference transformer.py:285-317, dropout 0) against the imported reference
    (g16): same state_dict keys (the parent's unused self_attn.* included), output, input gradient and the gradient
    of every parameter the forward uses; 1e-3 relative, the fp32 bound."""
    import transformer as T
    G = golden("g16_encoder_layer.pt")
    N, T_, E, H, DFF = 4, 24, 64, 4, 128
    layer = T.CustomTransformerEncoderLayer(E, H, DFF, dropout=0.0, batch_first=(mode == "batch_first"))
    assert set(layer.state_dict()) == set(G[f"{mode}.state"])
    layer.load_state_dict(G[f"{mode}.state"])
    layer.to(DEV).train()
    x = G[f"{mode}.x"].to(DEV).requires_grad_(True)
    am, pm = T.create_causal_mask(T_), T.create_padding_mask(G[f"{mode}.tokens"]).to(DEV)
    y = layer(x, src_mask=am, src_key_padding_mask=pm)
    params = dict(layer.named_parameters())
    grads = torch.autograd.grad(y, [x] + [params[n] for n in USED], G[f"{mode}.dy"].to(DEV))
    torch.cuda.synchronize()
    assert_close(y, G[f"{mode}.y"], TOL, "y")
    assert_close(grads[0], G[f"{mode}.dx"], TOL, "dx")
    for n, gv in zip(USED, grads[1:]):
        assert_close(gv, G[f"{mode}.grad.{n}"], TOL, f"grad {n}", atol=1e-7)


def test_encoder_stack_matches_reference_golden(golden):
    """nn.TransformerEncoder over two CustomTransformerEncoderLayers in train mode, the arrangement of the
    reference's self-test (transformer.py:340-348), against the imported reference (g16)."""
    import transformer as T
    G = golden("g16_encoder_layer.pt")
    N, T_, E, H, DFF = 4, 24, 64, 4, 128
    enc = torch.nn.TransformerEncoder(T.CustomTransformerEncoderLayer(E, H, DFF, dropout=0.0, batch_first=False), num_layers=2)
    enc.load_state_dict(G["stack.state"])
    enc.to(DEV).train()
    x = G["stack.x"].to(DEV).requires_grad_(True)
    y = enc(x, mask=T.create_causal_mask(T_), src_key_padding_mask=T.create_padding_mask(G["stack.tokens"]).to(DEV))
    (dx,) = torch.autograd.grad(y, [x], G["stack.dy"].to(DEV))
    torch.cuda.synchronize()
    assert_close(y, G["stack.y"], TOL, "stack y")
    assert_close(dx, G["stack.dx"], TOL, "stack dx")


def test_encoder_layer_with_dropout_masks_matches_float64_restatement():
    """The three dropout multipliers (after attention, inside the feed-forward block, after it) injected into the
    layer and into the float64 restatement of its forward at the TenGAN shape (batch 16, L = 512, E = 64, 4 heads);
    and the train-mode multipliers the layer draws itself have the right rate and scale."""
    import transformer as T
    from oracle.attention_cpu import encoder_layer_forward
    N, T_, E, H, DFF, P = 16, 512, 64, 4, 256, 0.25
    g = torch.Generator().manual_seed(1650)
    torch.manual_seed(1651)
    layer = T.CustomTransformerEncoderLayer(E, H, DFF, dropout=P).to(DEV).train()
    x = torch.randn((N, T_, E), generator=g)
    tokens = torch.randint(0, 4, (N, T_), generator=g)
    dy = torch.randn((N, T_, E), generator=g)
    masks = [((torch.rand(s, generator=g) >= P).float() / (1 - P)) for s in ((N, T_, E), (N, T_, DFF), (N, T_, E))]
    layer.dropout_masks = tuple(m.to(DEV) for m in masks)
    xin = x.to(DEV).requires_grad_(True)
    am, pm = T.create_causal_mask(T_), T.create_padding_mask(tokens)
    y = layer(xin, src_mask=am, src_key_padding_mask=pm.to(DEV))
    params = dict(layer.named_parameters())
    grads = torch.autograd.grad(y, [xin] + [params[n] for n in USED], dy.to(DEV))
    torch.cuda.synchronize()
    p64 = {n: params[n].detach().cpu().double().requires_grad_(True) for n in USED}
    x64 = x.double().requires_grad_(True)
    y64 = encoder_layer_forward(x64, p64, H, am.double(), pm.double(), tuple(m.double() for m in masks))
    g64 = torch.autograd.grad(y64, [x64] + [p64[n] for n in USED], dy.double())
    assert_close(y, y64, TOL, "y")
    assert_close(grads[0], g64[0], TOL, "dx")
    for n, a, b in zip(USED, grads[1:], g64[1:]):
        assert_close(a, b, TOL, f"grad {n}", atol=1e-7)
    layer.dropout_masks = None
    m1, mf, m2 = layer._masks(xin, DFF)
    torch.cuda.synchronize()
    for m, shape in ((m1, (N, T_, E)), (mf, (N, T_, DFF)), (m2, (N, T_, E))):
        assert tuple(m.shape) == shape
        assert abs(float((m != 0).float().mean()) - (1 - P)) < 5e-3 and abs(float(m.max()) - 1 / (1 - P)) < 1e-6
    layer.eval()
    assert layer._masks(xin, DFF) == (None, None, None)
