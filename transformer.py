"""MI355X-native attention with the module surface of the reference's transformer.py: the mask helpers,
PositionalEncoding, MaskedMultiheadAttention and CustomTransformerEncoderLayer (reference transformer.py:9-64,
196-317) -- same names,
constructor arguments, parameter names (in_proj_weight [3E, 3E], in_proj_bias, fc_out.*) and conventions,
including the unusual ones (the fused projection of concat(q, k, v), the finite -1e3 fill, the
tril(diagonal=1) "causal" mask, scaling by sqrt(embed_size), the padding mask that fills the non-pad keys,
batch_first=True meaning "transpose first").  All arithmetic of MaskedMultiheadAttention runs in libs2g
kernels (include/s2g.h: s2g_attention_fwd / _bwd and the fp32 MFMA GEMM); it needs GPU tensors."""
import math
import os
import sys

import torch
import torch.nn as nn

_PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sequence-to-graph-gan_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from s2g import ops  # noqa: E402
from s2g.attention import AddLayerNormFn, LinearActFn, MulMaskFn, MultiheadAttentionFn, dropout_multiplier  # noqa: E402


def create_causal_mask(seq_len):
    """Reference :9-13: tril(ones, diagonal=1) as floats (query q may attend to keys k <= q + 1)."""
    return torch.tril(torch.ones(seq_len, seq_len), diagonal=1).float()


def create_padding_mask(sequences, pad_token=0):
    """Reference :21-34: 1.0 where the token IS the pad token."""
    return (sequences == pad_token).float()


def generate_masks(seq_len, batch_size, padding_mask=None):
    """Reference :128-140."""
    attn_mask = torch.tril(torch.ones((batch_size, seq_len, seq_len))).unsqueeze(0).unsqueeze(0)
    if padding_mask is not None:
        return attn_mask * padding_mask.unsqueeze(1).unsqueeze(2)
    return attn_mask


class PositionalEncoding(nn.Module):
    """Reference :37-64 (a constant table added to the embeddings)."""

    def __init__(self, d_model, max_len):
        super(PositionalEncoding, self).__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        return x + self.pe[:, :x.size(1), :]


class MaskedMultiheadAttention(nn.Module):
    """Reference :196-282."""

    def __init__(self, embed_size, num_heads, batch_first=False):
        super(MaskedMultiheadAttention, self).__init__()
        self.embed_size = embed_size
        self.num_heads = num_heads
        self.head_dim = embed_size // num_heads
        self.batch_first = batch_first
        self._qkv_same_embed_dim = True
        assert self.head_dim * num_heads == embed_size, "Embedding size needs to be divisible by heads"
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_size, 3 * embed_size))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_size))
        self.fc_out = nn.Linear(self.num_heads * self.head_dim, embed_size)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.)
        nn.init.xavier_uniform_(self.fc_out.weight)
        nn.init.constant_(self.fc_out.bias, 0.)

    def forward(self, values, keys, queries, attn_mask=None, src_key_padding_mask=None):
        if self.batch_first:
            values, keys, queries = values.transpose(0, 1), keys.transpose(0, 1), queries.transpose(0, 1)
        if attn_mask is not None:
            # the reference repeats the mask over (N, H) (:254): any leading singleton dimensions are dropped
            attn_mask = attn_mask.reshape(attn_mask.shape[-2], attn_mask.shape[-1])
        out = MultiheadAttentionFn.apply(values.contiguous(), keys.contiguous(), queries.contiguous(), attn_mask,
                                         src_key_padding_mask, self.in_proj_weight, self.in_proj_bias,
                                         self.fc_out.weight, self.fc_out.bias, self.num_heads)
        if self.batch_first:
            out = out.transpose(0, 1)
        return out


class CustomTransformerEncoderLayer(nn.TransformerEncoderLayer):
    """Reference :285-317: a subclass of nn.TransformerEncoderLayer (so nn.TransformerEncoder stacks it and the
    state_dict keeps the parent's unused self_attn.* entries beside attn.*) whose forward is
    src = norm1(src + dropout1(attn(src, src, src))); src = norm2(src + dropout2(linear2(dropout(relu(linear1(src)))))).
    With batch_first the layer transposes its input AND the attention module transposes it back (:299-300 and
    :223-226): kept.  Every product, the LayerNorms and the dropout multipliers are libs2g kernels; in train mode
    the multipliers come from the library's counter-based generator (torch's dropout stream cannot be reproduced)."""

    def __init__(self, embed_size, num_heads, dim_feedforward=2048, dropout=0.1, batch_first=False):
        super(CustomTransformerEncoderLayer, self).__init__(embed_size, num_heads, dim_feedforward, dropout,
                                                            batch_first=batch_first, activation=nn.functional.relu)
        self.attn = MaskedMultiheadAttention(embed_size, num_heads, batch_first=batch_first)
        self.linear1 = nn.Linear(embed_size, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, embed_size)
        self.norm1 = nn.LayerNorm(embed_size)
        self.norm2 = nn.LayerNorm(embed_size)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.rng = None                 # ops.DeviceRng of the dropout multipliers (made on first use)
        self.dropout_masks = None       # optional (mask1, mask_ffn, mask2) used instead of drawn multipliers

    def _masks(self, src, dff):
        if self.dropout_masks is not None:
            return self.dropout_masks
        if not self.training:
            return None, None, None
        if self.rng is None:
            self.rng = ops.DeviceRng()
        dev = src.device
        return (dropout_multiplier(tuple(src.shape), self.dropout1.p, dev, self.rng),
                dropout_multiplier(tuple(src.shape[:-1]) + (dff,), self.dropout.p, dev, self.rng),
                dropout_multiplier(tuple(src.shape), self.dropout2.p, dev, self.rng))

    def forward(self, src, src_mask=None, src_key_padding_mask=None, **kwargs):
        if self.batch_first:
            src = src.transpose(0, 1)
        src = src.contiguous()
        m1, mf, m2 = self._masks(src, self.linear1.out_features)
        # self-attention block
        src2 = self.attn(src, src, src, attn_mask=src_mask, src_key_padding_mask=src_key_padding_mask)
        src = AddLayerNormFn.apply(src, src2, m1, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        # feed-forward block
        h = LinearActFn.apply(src, self.linear1.weight, self.linear1.bias, 2)
        if mf is not None:
            h = MulMaskFn.apply(h, mf)
        src2 = LinearActFn.apply(h, self.linear2.weight, self.linear2.bias, 0)
        src = AddLayerNormFn.apply(src, src2, m2, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        if self.batch_first:
            src = src.transpose(0, 1)
        return src
