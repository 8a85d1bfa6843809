"""TEST INFRASTRUCTURE: CPU restatement (plain torch ops, any dtype) of the reference's
MaskedMultiheadAttention.forward, transformer.py:222-282, parameters passed explicitly.  Pinned by
tests/golden/g11_attention.pt, recorded from the imported reference module (tests/test_oracle_golden.py)."""
import torch


def attention_forward(values, keys, queries, in_w, in_b, out_w, out_b, num_heads, attn_mask=None,
                      key_padding_mask=None, batch_first=False):
    if batch_first:                                                            # transformer.py:223-226
        values, keys, queries = values.transpose(0, 1), keys.transpose(0, 1), queries.transpose(0, 1)
    N, T, E = queries.shape
    D = E // num_heads
    qkv = torch.cat([queries, keys, values], dim=-1) @ in_w + in_b            # :232-235 (F.linear with W.T)
    q, k, v = qkv.chunk(3, dim=-1)                                            # :236
    q, k, v = (t.reshape(N, T, num_heads, D) for t in (q, k, v))              # :239-241
    energy = torch.einsum("nqhd,nkhd->nhqk", q, k)                            # :244
    if attn_mask is not None:                                                 # :251-257
        energy = energy.masked_fill(attn_mask.to(energy.dtype).reshape(1, 1, T, T) == 0, -1e3)
    if key_padding_mask is not None:                                          # :259-264
        energy = energy.masked_fill(key_padding_mask.reshape(N, 1, 1, T) == 0, -1e3)
    attention = torch.softmax(energy / (E ** 0.5), dim=3)                     # :267
    out = torch.einsum("nhql,nlhd->nqhd", attention, v).reshape(N, T, E)      # :272-274
    out = out @ out_w.T + out_b                                               # :277
    if batch_first:                                                           # :279-280
        out = out.transpose(0, 1)
    return out


def encoder_layer_forward(x, p, num_heads, attn_mask=None, pad_mask=None, masks=(None, None, None), eps=1e-5):
    """CustomTransformerEncoderLayer.forward (reference transformer.py:298-317) for x [N, T, E] (dim 0 = what the
    attention module treats as the batch) in plain torch ops of x's dtype; p: the layer's parameters under their
    state_dict names; masks = (m1, m_ffn, m2): the three dropout multipliers (None: no dropout).
    Test infrastructure only: the product path never imports this module."""
    import torch.nn.functional as F
    m1, mf, m2 = masks
    src2 = attention_forward(x, x, x, p["attn.in_proj_weight"], p["attn.in_proj_bias"], p["attn.fc_out.weight"],
                             p["attn.fc_out.bias"], num_heads, attn_mask, pad_mask)
    if m1 is not None:
        src2 = src2 * m1
    E = x.shape[-1]
    src = F.layer_norm(x + src2, (E,), p["norm1.weight"], p["norm1.bias"], eps)
    h = F.relu(F.linear(src, p["linear1.weight"], p["linear1.bias"]))
    if mf is not None:
        h = h * mf
    src2 = F.linear(h, p["linear2.weight"], p["linear2.bias"])
    if m2 is not None:
        src2 = src2 * m2
    return F.layer_norm(src + src2, (E,), p["norm2.weight"], p["norm2.bias"], eps)
