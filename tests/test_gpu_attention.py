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
