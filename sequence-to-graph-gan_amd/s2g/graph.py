"""Sequence -> graph node features on the device (reference: GraphEncoder.encode, graph_encoder.py:57-133).

`graph_features` is the launch wrapper over s2g_graph_features: a batch of token sequences in, the four
centralities the reference takes from networkx (graph_encoder.py:86-91) out, one workgroup per sequence.
`check_status` turns the kernel's per-sequence status into the exceptions the reference's calls raise.
"""
from __future__ import annotations

import numpy as np
import torch

from . import lib
from .lib import S2GError
from .ops import ptr, stream_ptr

MAX_NODES = 256            # node bitmasks of the kernel
MAX_LENGTH = 65535         # transition counts are 16-bit
_WORK_BYTES = 1 << 30      # Katz workspace per launch; longer batches go in chunks


class PowerIterationFailedConvergence(Exception):
    """networkx.PowerIterationFailedConvergence when networkx is installed (see _convergence_error)."""


def _convergence_error(iterations):
    try:
        import networkx as nx
        return nx.PowerIterationFailedConvergence(iterations)
    except ImportError:
        return PowerIterationFailedConvergence(f"power iteration failed to converge within {iterations} iterations")


def graph_features(tokens: torch.Tensor, lengths: torch.Tensor | None = None, vocab: int = 175):
    """tokens int32 (or int64, as Generator.sample_tokens returns them) [N, L] on the device (symbol ids in
    [0, vocab)), lengths int32 [N] or None ->
    (feat float64 [N, vocab, 4] = katz, betweenness, closeness, pagerank by symbol id; info int32 [N, 4] = nodes,
    status, pagerank iterations, distinct edges).  Stays on the device and on the current stream."""
    if not tokens.is_cuda:
        raise S2GError("graph_features needs device tensors: the HIP kernel is the only implementation")
    if tokens.dtype == torch.int64:
        tokens = tokens.to(torch.int32)
    if tokens.dtype != torch.int32 or tokens.dim() != 2 or not tokens.is_contiguous():
        raise S2GError("tokens must be a contiguous int32 (or int64) [N, L] tensor")
    N, L = tokens.shape
    if not 0 < vocab <= MAX_NODES:
        raise S2GError(f"graph_features supports at most {MAX_NODES} distinct symbols, got vocab={vocab}")
    if not 0 < L <= MAX_LENGTH:
        raise S2GError(f"graph_features supports sequences of 1..{MAX_LENGTH} symbols, got {L}")
    if lengths is not None and (lengths.dtype != torch.int32 or lengths.shape != (N,) or not lengths.is_cuda):
        raise S2GError("lengths must be an int32 [N] device tensor")
    feat = torch.empty(N, vocab, 4, dtype=torch.float64, device=tokens.device)
    info = torch.empty(N, 4, dtype=torch.int32, device=tokens.device)
    per = vocab * vocab * 8
    chunk = max(1, min(N, _WORK_BYTES // per))
    work = torch.empty(chunk * per, dtype=torch.uint8, device=tokens.device)
    for s in range(0, N, chunk):
        m = min(chunk, N - s)
        lib.call("s2g_graph_features", ptr(tokens[s:]), ptr(lengths[s:]) if lengths is not None else None, m, L, vocab,
                 ptr(feat[s:]), ptr(info[s:]), ptr(work), work.numel(), stream_ptr())
    return feat, info


def check_status(info_row) -> None:
    """Raise what the reference's networkx / numpy calls raise for this sequence's status."""
    status = int(info_row[1])
    if status & 1:
        raise _convergence_error(100)
    if status & 2:
        raise np.linalg.LinAlgError("Singular matrix")
