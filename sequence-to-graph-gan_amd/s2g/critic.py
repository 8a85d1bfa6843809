"""Host orchestration of the critic (reference: Discriminator + compute_gradient_penalty +
the critic half of train_wgan, wgan_gp.py:175-257, 277-318).

CriticEngine owns the critic parameters as one flat fp32 buffer (torch-layout views per
parameter), their packed copies, and the workspaces of the four LSTM sweeps; every number is
produced by kernels of libs2g (see include/s2g.h), torch only provides the memory."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .lib import S2GError, call
from .ops import G4, H, MASKW, ptr, stream_ptr

CRITIC_SHAPES = lambda V: [  # noqa: E731  (state_dict order of the reference Discriminator)
    ("lstm.weight_ih_l0", (G4, V)), ("lstm.weight_hh_l0", (G4, H)), ("lstm.bias_ih_l0", (G4,)),
    ("lstm.bias_hh_l0", (G4,)), ("lstm.weight_ih_l1", (G4, H)), ("lstm.weight_hh_l1", (G4, H)),
    ("lstm.bias_ih_l1", (G4,)), ("lstm.bias_hh_l1", (G4,)), ("fc.weight", (3, H)), ("fc.bias", (3,)),
]


class FlatBuffer:
    """One contiguous fp32 buffer carved into named views (each view starts 16-byte aligned)."""

    def __init__(self, shapes, device):
        self.offsets = {}
        off = 0
        for name, shape in shapes:
            n = 1
            for s in shape:
                n *= s
            self.offsets[name] = (off, n, tuple(shape))
            off += (n + 3) // 4 * 4
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.views = {k: self.flat[o:o + n].view(shape) for k, (o, n, shape) in self.offsets.items()}

    def like(self):
        fb = FlatBuffer.__new__(FlatBuffer)
        fb.offsets, fb.total = self.offsets, self.total
        fb.flat = torch.zeros_like(self.flat)
        fb.views = {k: fb.flat[o:o + n].view(shape) for k, (o, n, shape) in self.offsets.items()}
        return fb


class CriticEngine:
    def __init__(self, vocab: int, device, n_out: int = 3):
        if n_out != 3:
            raise S2GError("critic head is Linear(50, num_classes + 1 = 3) (wgan_gp.py:192)")
        self.V = vocab
        self.device = device
        self.params = FlatBuffer(CRITIC_SHAPES(vocab), device)
        self.grads = self.params.like()
        self.WT0 = torch.empty(vocab, G4, device=device)
        self.W0p = torch.empty(G4, vocab, device=device)
        self.b0 = torch.empty(G4, device=device)
        self.b1 = torch.empty(G4, device=device)
        self.scratch = ops.Scratch(device)
        self._ws_key = None
        self.ws: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ parameters
    def load(self, state: Dict[str, torch.Tensor]) -> None:
        for k, v in self.params.views.items():
            v.copy_(state[k].to(self.device, torch.float32))
        self.repack()

    def repack(self) -> None:
        p = self.params.views
        call("s2g_pack_critic", ptr(p["lstm.weight_ih_l0"]), ptr(p["lstm.bias_ih_l0"]), ptr(p["lstm.bias_hh_l0"]),
             ptr(p["lstm.bias_ih_l1"]), ptr(p["lstm.bias_hh_l1"]), ptr(self.WT0), ptr(self.W0p), ptr(self.b0),
             ptr(self.b1), self.V, stream_ptr())

    # ------------------------------------------------------------------ workspaces
    def workspace(self, nseq: int, Tw: int, so_seqs: int) -> Dict[str, torch.Tensor]:
        """Buffers for `nseq` sequences of up to Tw steps; the last `so_seqs` of them also get the
        second-order buffers.  Zero-initialised once: rows beyond a sequence's length keep finite
        stale values and are only ever multiplied by zero-filled gate gradients."""
        key = (nseq, Tw, so_seqs)
        if self._ws_key == key:
            return self.ws
        dev = self.device
        rows, so_rows = nseq * Tw, so_seqs * Tw
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)  # noqa: E731
        ws = dict(gates0=z(rows, G4), cell0=z(rows, G4), gates1=z(rows, G4), cell1=z(rows, G4),
                  delta0=z(rows, G4), delta1=z(rows, G4), hT=z(nseq, H), dhT=z(nseq, H), out=z(nseq, 3),
                  dout=z(nseq, 3), losses=z(4), desc=torch.zeros(nseq + so_seqs, 8, dtype=torch.int32, device=dev))
        if so_seqs:
            ws.update(adj0=z(so_rows, 2 * H), adj1=z(so_rows, 2 * H), adot0=z(so_rows, G4), adot1=z(so_rows, G4),
                      celld0=z(so_rows, G4), celld1=z(so_rows, G4), abar0=z(so_rows, G4), abar1=z(so_rows, G4),
                      hdotT=z(so_seqs, H), G=z(so_rows, self.V), Pv=z(so_rows, G4), pen=z(so_seqs),
                      norms=z(so_seqs))
        self.ws, self._ws_key = ws, key
        return ws

    # ------------------------------------------------------------------ sweeps
    def _w(self):
        p = self.params.views
        return ptr(p["lstm.weight_hh_l0"]), ptr(p["lstm.weight_ih_l1"]), ptr(p["lstm.weight_hh_l1"])

    def project(self, x: torch.Tensor, out: torch.Tensor) -> None:
        """out[rows,200] = x[rows,V] @ W_ih0^T (internal gate order, no bias)."""
        rows = x.numel() // self.V
        ops.gemm_raw(ptr(x), self.V, 1, ptr(self.WT0), G4, 1, ptr(out), G4, rows, G4, self.V)

    def sweep_fwd(self, ws, desc_ptr, nseq, max_len, tokens, P, mask) -> None:
        whh0, wih1, whh1 = self._w()
        call("s2g_lstm2_fwd", desc_ptr, nseq, max_len, ptr(tokens), ptr(self.WT0), ptr(P), ptr(mask), ptr(self.b0),
             ptr(self.b1), whh0, wih1, whh1, ptr(ws["gates0"]), ptr(ws["cell0"]), ptr(ws["gates1"]),
             ptr(ws["cell1"]), ptr(ws["hT"]), stream_ptr())

    def head_fwd(self, ws, nseq) -> None:
        p = self.params.views
        call("s2g_head_fwd", ptr(ws["hT"]), ptr(p["fc.weight"]), ptr(p["fc.bias"]), ptr(ws["out"]), nseq, 3,
             stream_ptr())

    def sweep_bwd(self, ws, desc_ptr, nseq, Tw, mask, adj_seq0: int = -1) -> None:
        """adj_seq0 >= 0: sequences [adj_seq0, nseq) also store (dh, dc) into the compact adj0/adj1
        buffers (their row 0 = first workspace row of sequence adj_seq0)."""
        whh0, wih1, whh1 = self._w()
        store = adj_seq0 >= 0
        call("s2g_lstm2_bwd", desc_ptr, nseq, Tw, ptr(mask), whh0, wih1, whh1, ptr(ws["gates0"]), ptr(ws["cell0"]),
             ptr(ws["gates1"]), ptr(ws["cell1"]), ptr(ws["dhT"]), ptr(ws["delta0"]), ptr(ws["delta1"]),
             ptr(ws["adj0"]) if store else None, ptr(ws["adj1"]) if store else None,
             max(adj_seq0, 0), max(adj_seq0, 0) * Tw, stream_ptr())

    def sweep_tangent(self, ws, desc_ptr, nseq, mask_ptr, row0: int) -> None:
        """Sequences of `desc` count rows from workspace row `row0` (their descriptors are relative)."""
        whh0, wih1, whh1 = self._w()
        o = row0 * G4
        call("s2g_lstm2_tangent", desc_ptr, nseq, ptr(ws["Pv"]), mask_ptr, whh0, wih1, whh1,
             ptr(ws["gates0"], o), ptr(ws["cell0"], o), ptr(ws["gates1"], o), ptr(ws["cell1"], o),
             ptr(ws["adot0"]), ptr(ws["celld0"]), ptr(ws["adot1"]), ptr(ws["celld1"]), ptr(ws["hdotT"]),
             stream_ptr())

    def sweep_bwd2(self, ws, desc_ptr, nseq, Tw, mask_ptr, row0: int) -> None:
        whh0, wih1, whh1 = self._w()
        o = row0 * G4
        call("s2g_lstm2_bwd2", desc_ptr, nseq, Tw, mask_ptr, whh0, wih1, whh1, ptr(ws["gates0"], o),
             ptr(ws["cell0"], o), ptr(ws["gates1"], o), ptr(ws["cell1"], o), ptr(ws["adot0"]), ptr(ws["celld0"]),
             ptr(ws["adot1"]), ptr(ws["celld1"]), ptr(ws["adj0"]), ptr(ws["adj1"]), ptr(ws["abar0"]),
             ptr(ws["abar1"]), stream_ptr())


def make_desc(entries, device) -> torch.Tensor:
    """entries: list of dicts(len, tok_off, p_off, p_len, ca, cp, mask_off, ws_row) -> int32 [n,8]
    tensor with the two floats bit-cast (host-built descriptors for the module-level API/tests)."""
    import numpy as np
    arr = np.zeros((len(entries), 8), dtype=np.int32)
    for i, e in enumerate(entries):
        arr[i, 0] = e["len"]; arr[i, 1] = e.get("tok_off", -1); arr[i, 2] = e.get("p_off", -1)
        arr[i, 3] = e.get("p_len", 0)
        arr[i, 4:6] = np.array([e.get("ca", 0.0), e.get("cp", 0.0)], dtype=np.float32).view(np.int32)
        arr[i, 6] = e.get("mask_off", -1); arr[i, 7] = e["ws_row"]
    return torch.from_numpy(arr).to(device)
