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
        self.timed_events = None      # list -> (start, end) HIP event pairs around the 3B-sequence forward sweep
        self._gemm_tables = {}        # cached device-side problem tables of the grouped weight-gradient GEMM
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
        self.ws, self._ws_key = self.fresh_workspace(nseq, Tw, so_seqs), key
        return self.ws

    def fresh_workspace(self, nseq: int, Tw: int, so_seqs: int) -> Dict[str, torch.Tensor]:
        """A private set of buffers (the autograd bridges keep one per outstanding forward)."""
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
        return ws

    def workspace_g(self, B: int, Tf: int) -> Dict[str, torch.Tensor]:
        """Buffers of the generator-side D(fake) pass (first-order only)."""
        key = (B, Tf)
        if getattr(self, "_wsg_key", None) == key:
            return self.wsg
        dev = self.device
        rows = B * Tf
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)  # noqa: E731
        self.wsg = dict(gates0=z(rows, G4), cell0=z(rows, G4), gates1=z(rows, G4), cell1=z(rows, G4),
                        delta0=z(rows, G4), delta1=z(rows, G4), hT=z(B, H), dhT=z(B, H), out=z(B, 3),
                        dout=z(B, 3), losses=z(4), Pf=z(rows, G4), dfake=z(rows, self.V))
        self._wsg_key = key
        return self.wsg

    # ------------------------------------------------------------------ sweeps
    def _w(self):
        p = self.params.views
        return ptr(p["lstm.weight_hh_l0"]), ptr(p["lstm.weight_ih_l1"]), ptr(p["lstm.weight_hh_l1"])

    def project(self, x: torch.Tensor, out: torch.Tensor) -> None:
        """out[rows,200] = x[rows,V] @ W_ih0^T (internal gate order, no bias)."""
        rows = x.numel() // self.V
        ops.gemm_raw(ptr(x), self.V, 1, ptr(self.WT0), G4, 1, ptr(out), G4, rows, G4, self.V)

    def sweep_fwd(self, ws, desc_ptr, nseq, max_len, tokens, P, mask, timed: bool = False) -> None:
        whh0, wih1, whh1 = self._w()
        te = self.timed_events if timed else None
        if te is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        call("s2g_lstm2_fwd", desc_ptr, nseq, max_len, ptr(tokens), ptr(self.WT0), ptr(P), ptr(mask), ptr(self.b0),
             ptr(self.b1), whh0, wih1, whh1, ptr(ws["gates0"]), ptr(ws["cell0"]), ptr(ws["gates1"]),
             ptr(ws["cell1"]), ptr(ws["hT"]), stream_ptr())
        if te is not None:
            e1.record()
            te.append((e0, e1))

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


# ---------------------------------------------------------------------------------------------
# Weight gradients: every dW is sum over rows of (gate gradient row)^T (operand row), a TN GEMM
# over the saved buffers; several such products land in consecutive split-K slabs and one reduce
# writes the torch-layout gradient.
# ---------------------------------------------------------------------------------------------
def _wgrad(eng: "CriticEngine", out: torch.Tensor, N: int, terms) -> None:
    """One weight gradient.  terms: (A_ptr, K, B_ptr, sbk, sbn, segT, a_seg, b_seg, shift); A is
    [rows,200] row-major."""
    _wgrad_grouped(eng, [(out, N, terms)], "single")


def _wgrad_grouped(eng: "CriticEngine", jobs, cache_key) -> None:
    """jobs: list of (out tensor [200,N] torch layout, N, terms).  Every product of every job runs
    in ONE grouped GEMM launch (split-K slabs), then one reduce per job.  The device-side problem
    table is cached per `cache_key` and rebuilt only when a pointer or size changes."""
    import ctypes
    from .lib import S2GGemm
    plan, sig = [], []
    total_slabs, blocks = 0, 0
    job_slabs = []
    for out, N, terms in jobs:
        slab = G4 * N
        first = total_slabs
        base_elems = sum(js[1] * js[2] for js in job_slabs)
        off = 0
        nsl = 0
        for (A, K, B, sbk, sbn, segT, a_seg, b_seg, shift) in terms:
            if K <= 0:
                continue
            ns = max(1, min(128, (K + 512) // 1024))
            kchunk = ((K + ns - 1) // ns + 31) // 32 * 32
            tiles = (N + 63) // 64                               # (all 200 rows) x 64 output tiles
            plan.append((A, K, B, sbk, sbn, segT, a_seg, b_seg, shift, N, ns, kchunk, base_elems + off, slab, blocks))
            sig.append((A, K, B, sbk, sbn, segT, a_seg, b_seg, shift, N))
            blocks += tiles * ns
            off += ns * slab
            nsl += ns
        job_slabs.append((out, nsl, slab, base_elems))
        total_slabs += nsl
    need = sum(js[1] * js[2] for js in job_slabs)
    slabs = eng.scratch.get(need)
    sig = (tuple(sig), slabs.data_ptr())
    cache = eng._gemm_tables.get(cache_key)
    if cache is None or cache[0] != sig:
        arr = (S2GGemm * len(plan))()
        prefix = []
        for i, (A, K, B, sbk, sbn, segT, a_seg, b_seg, shift, N, ns, kchunk, soff, slab, b0) in enumerate(plan):
            arr[i] = S2GGemm(A, 1, G4, B, sbk, sbn, ptr(slabs, soff), N, slab, None, G4, N, K, kchunk, segT, a_seg,
                             b_seg, int(shift), 0, 0, 1.0)
            prefix.append(b0)
        prefix.append(blocks)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(eng.device)
        pre = torch.tensor(prefix, dtype=torch.int32, device=eng.device)
        cache = (sig, raw, pre, len(plan), blocks)
        eng._gemm_tables[cache_key] = cache
    _, raw, pre, nprob, nblocks = cache
    call("s2g_gemm_grouped", ptr(raw), ptr(pre), nprob, nblocks, stream_ptr())
    for out, nsl, slab, base in job_slabs:
        if nsl:
            N = slab // G4
            call("s2g_reduce_slabs", ptr(slabs, base), slab, nsl, G4, N, ptr(out), 1, 0, 1.0, stream_ptr())


def critic_iteration(eng: CriticEngine, tokens: torch.Tensor, lengths: torch.Tensor, labels: torch.Tensor,
                     fake: torch.Tensor, alpha: torch.Tensor, lambda_gp: float, n_global: int, inv_wsum: float,
                     mask: Optional[torch.Tensor] = None, class_w=(1.0, 2.0)):
    """One critic loss + gradient evaluation (wgan_gp.py:297-317) on the batch
    [real B | fake B | interp B], all three passes sharing each of the four sweeps.

    tokens int32 [B,T] (zero padded), lengths int32 [B], labels int64 [B], fake fp32 [B,Tf,V],
    alpha fp32 [B] (the draw of wgan_gp.py:228), mask fp32 [3*B*Tw, 52] or None.
    n_global: global batch size (means and the penalty's 1/B use it; ranks then SUM gradients).
    Fills eng.grads; returns (losses[3] = (-mean D(real), mean D(fake), class loss), pen[B])."""
    B, T = tokens.shape
    Tf, V = fake.shape[1], fake.shape[2]
    if V != eng.V:
        raise S2GError("critic_iteration: vocab mismatch")
    for t, dt, nm in ((tokens, torch.int32, "tokens"), (lengths, torch.int32, "lengths"),
                      (labels, torch.int64, "labels"), (fake, torch.float32, "fake"), (alpha, torch.float32, "alpha")):
        ops._chk(t, dt, nm)
    if alpha.numel() != B or lengths.numel() != B or labels.numel() != B or fake.shape[0] != B:
        raise S2GError("critic_iteration: batch size mismatch")
    Tw = max(T, Tf)
    ws = eng.workspace(3 * B, Tw, B)
    if mask is not None:
        ops._chk(mask, name="mask")
        if tuple(mask.shape) != (3 * B * Tw, MASKW):
            raise S2GError("critic_iteration: mask must be [3*B*Tw, 52]")
    p, g, st = eng.params.views, eng.grads.views, stream_ptr()
    desc = ws["desc"]
    dptr = ptr(desc)
    call("s2g_critic_batch_desc", ptr(lengths), ptr(alpha), B, T, Tf, int(mask is not None), dptr, st)
    if "Pf" not in ws or ws["Pf"].shape[0] != B * Tf:
        ws["Pf"] = torch.empty(B * Tf, G4, device=eng.device)
        ws["xreal"] = torch.empty(B * T, V, device=eng.device)
        ws["xhat"] = torch.empty(B * T, V, device=eng.device)
    # sweep A on 3B sequences
    eng.project(fake, ws["Pf"])
    eng.sweep_fwd(ws, dptr, 3 * B, Tw, tokens, ws["Pf"], mask, timed=True)
    eng.head_fwd(ws, 3 * B)
    inv_n = 1.0 / float(n_global)
    call("s2g_critic_loss", ptr(ws["out"]), ptr(labels), B, inv_n, inv_wsum, class_w[0], class_w[1],
         ptr(ws["dout"]), ptr(ws["losses"]), st)
    call("s2g_head_bwd_dh", ptr(ws["dout"]), ptr(p["fc.weight"]), ptr(ws["dhT"]), 3 * B, 3, st)
    call("s2g_head_bwd_w", ptr(ws["dout"]), ptr(ws["hT"]), ptr(g["fc.weight"]), ptr(g["fc.bias"]), 2 * B, 3, 0, st)
    # sweep B on 3B sequences (the interp third with d(out) = e0 and stored adjoints)
    eng.sweep_bwd(ws, dptr, 3 * B, Tw, mask, adj_seq0=2 * B)
    # input gradient of the interp third, its norm, the penalty and the direction v (in place)
    r_int = 2 * B * Tw
    ops.gemm_raw(ptr(ws["delta0"], r_int * G4), G4, 1, ptr(eng.W0p), V, 1, ptr(ws["G"]), V, B * Tw, V, G4)
    call("s2g_gp_norm_penalty", ptr(ws["G"]), B, Tw * V, float(lambda_gp), float(n_global), ptr(ws["pen"]),
         ptr(ws["norms"]), 1, st)
    eng.project(ws["G"], ws["Pv"])
    # sweeps C and D on the interp third (group-relative descriptors = 4th block of desc)
    drel = ptr(desc, 3 * B * 8)
    mrel = ptr(mask, r_int * MASKW) if mask is not None else None
    eng.sweep_tangent(ws, drel, B, mrel, r_int)
    eng.sweep_bwd2(ws, drel, B, Tw, mrel, r_int)
    # operands of the weight-gradient products
    call("s2g_one_hot", ptr(tokens), ptr(lengths), T, ptr(ws["xreal"]), B * T, V, st)
    call("s2g_gp_interp", ptr(tokens), ptr(lengths), ptr(fake), ptr(alpha), B, T, Tf, V, ptr(ws["xhat"]), st)
    d0, d1, c0, c1 = ws["delta0"], ws["delta1"], ws["cell0"], ws["cell1"]
    a0, a1, cd0, cd1 = ws["abar0"], ws["abar1"], ws["celld0"], ws["celld1"]
    oi = r_int * G4
    n12, n3 = 2 * B * Tw, B * Tw
    jobs = [(g["lstm.weight_ih_l0"], V, [
        (ptr(d0), B * T, ptr(ws["xreal"]), V, 1, T, Tw, T, False),
        (ptr(d0, B * Tw * G4), B * Tf, ptr(fake), V, 1, Tf, Tw, Tf, False),
        (ptr(a0), B * T, ptr(ws["xhat"]), V, 1, T, Tw, T, False),
        (ptr(d0, oi), n3, ptr(ws["G"]), V, 1, 0, 0, 0, False)]),
            (g["lstm.weight_hh_l0"], H, [
        (ptr(d0), n12, ptr(c0, 2), G4, 4, Tw, Tw, Tw, True),
        (ptr(a0), n3, ptr(c0, oi + 2), G4, 4, Tw, Tw, Tw, True),
        (ptr(d0, oi), n3, ptr(cd0, 1), G4, 4, Tw, Tw, Tw, True)]),
            (g["lstm.weight_ih_l1"], H, [
        (ptr(d1), n12, ptr(c0, 3), G4, 4, 0, 0, 0, False),
        (ptr(a1), n3, ptr(c0, oi + 3), G4, 4, 0, 0, 0, False),
        (ptr(d1, oi), n3, ptr(cd0, 2), G4, 4, 0, 0, 0, False)]),
            (g["lstm.weight_hh_l1"], H, [
        (ptr(d1), n12, ptr(c1, 2), G4, 4, Tw, Tw, Tw, True),
        (ptr(a1), n3, ptr(c1, oi + 2), G4, 4, Tw, Tw, Tw, True),
        (ptr(d1, oi), n3, ptr(cd1, 1), G4, 4, Tw, Tw, Tw, True)])]
    _wgrad_grouped(eng, jobs, ("critic", B, T, Tf))
    sc = eng.scratch
    ops.colsum(d0, n12, G4, G4, g["lstm.bias_ih_l0"], sc, permute_rows=True)
    ops.colsum(a0, n3, G4, G4, g["lstm.bias_ih_l0"], sc, permute_rows=True, accumulate=True)
    ops.colsum(d1, n12, G4, G4, g["lstm.bias_ih_l1"], sc, permute_rows=True)
    ops.colsum(a1, n3, G4, G4, g["lstm.bias_ih_l1"], sc, permute_rows=True, accumulate=True)
    g["lstm.bias_hh_l0"].copy_(g["lstm.bias_ih_l0"])
    g["lstm.bias_hh_l1"].copy_(g["lstm.bias_ih_l1"])
    # second-order part of fc.weight[0]: sum_b hdot1_final(b)
    ops.colsum(ws["hdotT"], B, H, H, g["fc.weight"][0], sc, accumulate=True)
    return ws["losses"], ws["pen"]


def critic_forward_backward_fake(eng: CriticEngine, fake: torch.Tensor, labels: torch.Tensor, n_global: int,
                                 inv_wsum: float, mask: Optional[torch.Tensor] = None, class_w=(1.0, 2.0)):
    """Generator-side pass (wgan_gp.py:333-342): D(fake) dense, generator loss, and d loss/d fake.
    The critic's own parameter gradients of this pass are discarded by the reference
    (optimizer_D.zero_grad at the next iteration, wgan_gp.py:288) and are not computed.
    Returns (losses[2] = (-mean validity, 0.25*CE), dfake [B,Tf,V])."""
    B, Tf, V = fake.shape
    ops._chk(fake, name="fake"); ops._chk(labels, torch.int64, "labels")
    ws = eng.workspace_g(B, Tf)
    st = stream_ptr()
    p = eng.params.views
    desc = make_desc([dict(len=Tf, p_off=b * Tf, p_len=Tf, ca=0.0, cp=1.0, ws_row=b * Tf,
                           mask_off=(b * Tf if mask is not None else -1)) for b in range(B)], eng.device) \
        if ws.get("desc_key") != (B, Tf, mask is not None) else ws["desc"]
    ws["desc"], ws["desc_key"] = desc, (B, Tf, mask is not None)
    eng.project(fake, ws["Pf"])
    eng.sweep_fwd(ws, ptr(desc), B, Tf, None, ws["Pf"], mask)
    eng.head_fwd(ws, B)
    call("s2g_generator_loss", ptr(ws["out"]), ptr(labels), B, 1.0 / float(n_global), inv_wsum, class_w[0],
         class_w[1], ptr(ws["dout"]), ptr(ws["losses"]), st)
    call("s2g_head_bwd_dh", ptr(ws["dout"]), ptr(p["fc.weight"]), ptr(ws["dhT"]), B, 3, st)
    eng.sweep_bwd(ws, ptr(desc), B, Tf, mask)
    ops.gemm_raw(ptr(ws["delta0"]), G4, 1, ptr(eng.W0p), V, 1, ptr(ws["dfake"]), V, B * Tf, V, G4)
    return ws["losses"], ws["dfake"].view(B, Tf, V)
