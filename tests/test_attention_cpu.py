"""The CPU restatement of MaskedMultiheadAttention (oracle/attention_cpu.py) against outputs and gradients
recorded from the imported reference module (g11)."""
import pytest
import torch

from conftest import assert_close
from common import attention_case
from oracle.attention_cpu import attention_forward

CASES = {"small": (1301, 3, 20, 32, 2), "selftest": (1302, 8, 50, 64, 4), "ragged": (1303, 2, 37, 64, 1)}


def masks_for(c, T, mode):
    if mode == "nomask":
        return None, None
    return torch.tril(torch.ones(T, T), diagonal=1).float(), (c["tokens"] == 0).float()


@pytest.mark.parametrize("tag", list(CASES))
@pytest.mark.parametrize("mode", ["nomask", "masks", "batch_first"])
def test_oracle_matches_reference_golden(golden, tag, mode):
    G = golden("g11_attention.pt")
    seed, N, T, E, H = CASES[tag]
    c = attention_case(seed, N, T, E, H)
    am, pm = masks_for(c, T, mode)
    leaves = [c[n].clone().requires_grad_(True) for n in ("q", "k", "v", "in_w", "in_b", "out_w", "out_b")]
    q, k, v, in_w, in_b, out_w, out_b = leaves
    if mode == "batch_first":
        y = attention_forward(v.transpose(0, 1), k.transpose(0, 1), q.transpose(0, 1), in_w, in_b, out_w, out_b, H, am, pm,
                              batch_first=True).transpose(0, 1)
    else:
        y = attention_forward(v, k, q, in_w, in_b, out_w, out_b, H, am, pm)
    grads = torch.autograd.grad(y, leaves, c["dy"])
    key = f"{tag}.{mode}"
    assert_close(y, G[f"{key}.y"], 1e-5, "y")
    for n, gv in zip(("dq", "dk", "dv", "d_in_w", "d_in_b", "d_out_w", "d_out_b"), grads):
        assert_close(gv, G[f"{key}.{n}"], 1e-4, n, atol=1e-7)


def test_encoder_layer_restatement_matches_reference_golden(golden):
    """oracle.attention_cpu.encoder_layer_forward against the imported reference's CustomTransformerEncoderLayer
    (g16, dropout 0): output and input gradient, in float64."""
    import torch
    from oracle.attention_cpu import encoder_layer_forward
    G = golden("g16_encoder_layer.pt")
    T_ = 24
    for mode in ("plain", "batch_first"):
        p = {k: v.double() for k, v in G[f"{mode}.state"].items()}
        x = G[f"{mode}.x"].double().requires_grad_(True)
        am = torch.tril(torch.ones(T_, T_, dtype=torch.float64), diagonal=1)
        pm = (G[f"{mode}.tokens"] == 0).double()
        y = encoder_layer_forward(x, p, 4, am, pm)
        (dx,) = torch.autograd.grad(y, [x], G[f"{mode}.dy"].double())
        assert float((y - G[f"{mode}.y"].double()).abs().max()) < 1e-5
        assert float((dx - G[f"{mode}.dx"].double()).abs().max()) < 1e-5
