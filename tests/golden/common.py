"""Deterministic synthetic weights / batches / noise shared by the golden-vector generator
and the tests.  Everything is drawn from a seeded CPU torch.Generator, so the inputs are
regenerated identically wherever the same torch build runs and only the EXPECTED OUTPUTS
have to be stored as fixtures."""
from __future__ import annotations

import math

import torch


def gen(seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def uniform(g, shape, k, dtype=torch.float32):
    return ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * k).to(dtype)


def critic_params(seed: int, vocab: int = 175, hidden: int = 50, n_out: int = 3, scale: float = 1.0,
                  dtype=torch.float32):
    """Same names/shapes as the reference Discriminator state_dict (wgan_gp.py:189-192);
    U(-k,k) with k = scale/sqrt(hidden) (torch's LSTM init range when scale = 1)."""
    g = gen(seed)
    k = scale / math.sqrt(hidden)
    H4 = 4 * hidden
    p = {
        "lstm.weight_ih_l0": uniform(g, (H4, vocab), k, dtype),
        "lstm.weight_hh_l0": uniform(g, (H4, hidden), k, dtype),
        "lstm.bias_ih_l0": uniform(g, (H4,), k, dtype),
        "lstm.bias_hh_l0": uniform(g, (H4,), k, dtype),
        "lstm.weight_ih_l1": uniform(g, (H4, hidden), k, dtype),
        "lstm.weight_hh_l1": uniform(g, (H4, hidden), k, dtype),
        "lstm.bias_ih_l1": uniform(g, (H4,), k, dtype),
        "lstm.bias_hh_l1": uniform(g, (H4,), k, dtype),
        "fc.weight": uniform(g, (n_out, hidden), k, dtype),
        "fc.bias": uniform(g, (n_out,), k, dtype),
    }
    return p


def generator_params(seed: int, vocab: int = 175, seq_len: int = 120, n_classes: int = 2,
                     dtype=torch.float32):
    """Same names/shapes as the reference Generator state_dict (wgan_gp.py:154-163)."""
    g = gen(seed)
    dims = [(100, 128, "model.0"), (128, 256, "model.2"), (256, 512, "model.5"),
            (512, vocab * seq_len, "model.8")]
    p = {}
    for fin, fout, name in dims:
        k = 1.0 / math.sqrt(fin)
        p[name + ".weight"] = uniform(g, (fout, fin), k, dtype)
        p[name + ".bias"] = uniform(g, (fout,), k, dtype)
    for name, n in (("model.3", 256), ("model.6", 512)):
        p[name + ".weight"] = (1.0 + uniform(g, (n,), 0.2, dtype))
        p[name + ".bias"] = uniform(g, (n,), 0.2, dtype)
    p["label_embedding.weight"] = torch.randn((n_classes, 100), generator=g, dtype=torch.float64).to(dtype)
    return p


def bn_buffers(dtype=torch.float32):
    return {
        "model.3.running_mean": torch.zeros(256, dtype=dtype),
        "model.3.running_var": torch.ones(256, dtype=dtype),
        "model.3.num_batches_tracked": torch.zeros((), dtype=torch.int64),
        "model.6.running_mean": torch.zeros(512, dtype=dtype),
        "model.6.running_var": torch.ones(512, dtype=dtype),
        "model.6.num_batches_tracked": torch.zeros((), dtype=torch.int64),
    }


def batch(seed: int, B: int, T: int, vocab: int = 175, lengths=None, min_len: int = 1):
    """A collate_fn-shaped batch (custom_dataset.py:30-47): tokens int64 [B,Tmax] zero padded,
    labels int64 [B], lengths int64 [B] sorted descending."""
    g = gen(seed)
    if lengths is None:
        lengths = torch.randint(min_len, T + 1, (B,), generator=g)
        lengths[0] = T
    lengths = torch.sort(torch.as_tensor(lengths, dtype=torch.int64), descending=True).values
    Tmax = int(lengths.max())
    tokens = torch.randint(0, vocab, (B, Tmax), generator=g)
    ar = torch.arange(Tmax).unsqueeze(0)
    tokens = torch.where(ar < lengths.unsqueeze(1), tokens, torch.zeros_like(tokens))
    labels = torch.randint(0, 2, (B,), generator=g)
    return tokens, labels, lengths


def noise(seed: int, B: int, Tf: int, vocab: int = 175, latent: int = 100, dtype=torch.float32):
    g = gen(seed)
    return {
        "z": torch.randn((B, latent), generator=g, dtype=torch.float64).to(dtype),
        "U": torch.rand((B, Tf, vocab), generator=g, dtype=torch.float64).to(dtype),
        "alpha": torch.rand((B, 1, 1), generator=g, dtype=torch.float64).to(dtype),
        "U2": torch.rand((B, Tf, vocab), generator=g, dtype=torch.float64).to(dtype),
    }


def adfa_lengths(seed: int, B: int, L: int = 512):
    """ADFA-shaped stand-in: clip(round(lognormal(ln 330, 0.8)), 75, L), sorted descending."""
    g = gen(seed)
    ln = torch.exp(math.log(330.0) + 0.8 * torch.randn((B,), generator=g, dtype=torch.float64))
    return torch.sort(ln.round().clamp(75, L).to(torch.int64), descending=True).values
