"""Host orchestration of the critic (reference: Discriminator + compute_gradient_penalty +
the critic half of train_wgan, wgan_gp.py:175-257, 277-318).

CriticEngine owns the critic parameters as one flat fp32 buffer (torch-layout views per
parameter), their packed copies, and the workspaces of the four LSTM sweeps; every number is
produced by kernels of libs2g (see include/s2g.h), torch only provides the memory."""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

from . import ops
from .lib import S2GError, S2GHead, call
from .ops import G4, H, MASKW, ptr, stream_ptr

CRITIC_SHAPES = lambda V: [  # noqa: E731  (state_dict order of the reference Discriminator)
    ("lstm.weight_ih_l0", (G4, V)), ("lstm.weight_hh_l0", (G4, H)), ("lstm.bias_ih_l0", (G4,)),
    ("lstm.bias_hh_l0", (G4,)), ("lstm.weight_ih_l1", (G4, H)), ("lstm.weight_hh_l1", (G4, H)),
    ("lstm.bias_ih_l1", (G4,)), ("lstm.bias_hh_l1", (G4,)), ("fc.weight", (3, H)), ("fc.bias", (3,)),
]


# (device index, comm stream reserved) -> (CU-masked stream, unmasked stream, second CU-masked stream or None) | False
_SIDE_STREAMS: Dict = {}


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
        self.Vp = (vocab + 3) // 4 * 4          # row stride of W0p: 16-byte rows for the tall products
        self.W0p = torch.zeros(G4, self.Vp, device=device)
        self.b0 = torch.empty(G4, device=device)
        self.b1 = torch.empty(G4, device=device)
        self.M2 = torch.empty(G4, G4, device=device)
        self._scalars = {}
        self.scratch = ops.Scratch(device)
        self.timed_events = None      # dict kernel name -> [(start, end) HIP event pairs] (benchmark hook)
        self.timed_every = 1          # ... filled for every n-th critic iteration only (event pairs are not free:
        self._timed_iter = 0          # 14 per iteration cost ~3 % of the step and pile up in the runtime's pools)
        self._timed_now = True
        self._gemm_tables = {}        # cached device-side problem tables of the grouped weight-gradient GEMM
        self._wgrad_slabs = None      # split-K slabs of the weight-gradient products
        self._pending_built = None    # weight-gradient plan whose final reduce was left to fused_update
        self._m2_due = False          # M2 must be rebuilt from the packed weights before its next reader
        self.cu_masked = 0            # CUs the masked side streams can reach (0 until side() has made them)
        self.overlap = True           # run the work that does not depend on sweeps C/D beside them (critic_iteration)
        self._side = None             # (stream, fork event, join event, scratch) of that second stream
        self.stream_g = None          # CU-masked stream of the prefetch hook (created with _side)
        self.bf16 = False             # tall products of the input layer and the weight-gradient products on bf16 MFMA
                                      # (fp32 accumulation; the 2e-2 path); default exact fp32
        self.reserve_comm_stream = False   # data-parallel runs: the collective library owns a stream (and a hardware
                                           # queue) of its own, so the prefetch hook shares the second stream instead
        self._m2_events = None        # (weights packed, M2 built) when M2 is built on a side stream
        self._m2_pending = False
        self._ws_cache = {}           # (nseq, Tw, so_seqs) -> workspace; batches of a few shapes alternate in an epoch
        self.ws: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ parameters
    def load(self, state: Dict[str, torch.Tensor]) -> None:
        for k, v in self.params.views.items():
            v.copy_(state[k].to(self.device, torch.float32))
        self.repack()

    def fused_update(self, m: torch.Tensor, v: torch.Tensor, lr: float, b1: float, b2: float, eps: float,
                     step: int) -> None:
        """optimizer_D.step() (wgan_gp.py:318) for one process, straight from the split-K slabs a
        critic_iteration(defer_reduce=True) left behind: final slab reduction (the gradients still land in
        self.grads), Adam on every parameter and the packed copies of repack() in one launch; then M2."""
        from .lib import S2GCriticUpdate, S2GReduce
        built, self._pending_built = self._pending_built, None
        if built is None:
            raise S2GError("fused_update: no critic_iteration(defer_reduce=True) pending")
        _, job_slabs, slabs = built[:3]
        names = ("lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.weight_ih_l1", "lstm.weight_hh_l1")
        off = self.params.offsets
        u = S2GCriticUpdate()
        for k, (name, (out, n_early, n_late, slab, base)) in enumerate(zip(names, job_slabs)):
            if out.data_ptr() != self.grads.views[name].data_ptr() or n_early + n_late == 0:
                raise S2GError("fused_update: unexpected weight-gradient jobs")
            first = base + n_early * slab if n_early else base
            u.job[k] = S2GReduce(ptr(slabs, first), slab, ptr(out), (1 if n_early else 0) + n_late, G4, slab // G4)
            u.poff[k] = off[name][0]
        u.p, u.g, u.m, u.v = ptr(self.params.flat), ptr(self.grads.flat), ptr(m), ptr(v)
        u.o_bih0, u.o_bhh0 = off["lstm.bias_ih_l0"][0], off["lstm.bias_hh_l0"][0]
        u.o_bih1, u.o_bhh1 = off["lstm.bias_ih_l1"][0], off["lstm.bias_hh_l1"][0]
        u.o_fcw, u.o_fcb = off["fc.weight"][0], off["fc.bias"][0]
        u.n_fcw, u.n_fcb = off["fc.weight"][1], off["fc.bias"][1]
        u.V, u.Vp = self.V, self.Vp
        u.WT0, u.W0p, u.b0, u.b1 = ptr(self.WT0), ptr(self.W0p), ptr(self.b0), ptr(self.b1)
        u.lr, u.b1c, u.b2c, u.eps, u.step = lr, b1, b2, eps, step
        call("s2g_critic_update", ctypes.byref(u), stream_ptr())
        self.repack(packed=True)

    def repack(self, packed: bool = False) -> None:
        """packed: WT0 / W0p / b0 / b1 are already current (fused_update wrote them); only M2 is rebuilt."""
        p = self.params.views
        if not packed:
            call("s2g_pack_critic", ptr(p["lstm.weight_ih_l0"]), ptr(p["lstm.bias_ih_l0"]), ptr(p["lstm.bias_hh_l0"]),
                 ptr(p["lstm.bias_ih_l1"]), ptr(p["lstm.bias_hh_l1"]), ptr(self.WT0), ptr(self.W0p), ptr(self.b0),
                 ptr(self.b1), self.V, stream_ptr())
        # M2 = W_ih0 W_ih0^T (internal gate order): maps layer-0 gate gradients straight to the projected
        # input gradient (critic_iteration).  Its first reader comes after sweeps A and B of the next
        # iteration, so with the side streams up it is built there, off the caller's stream.
        side = self._side if self.overlap else None
        if side:
            # Not right here, beside the next fake projection (256 workgroups that want a CU each):
            # kick_m2() starts it once that projection is on its way, beside sweep A.
            self._m2_due = True
        else:
            ops.gemm_raw(ptr(self.W0p), self.Vp, 1, ptr(self.WT0), G4, 1, ptr(self.M2), G4, G4, G4, self.V)

    def kick_m2(self) -> None:
        """Starts a due rebuild of M2 on the fourth stream, ordered behind everything enqueued on the
        current stream so far (the packed weights it reads are among that)."""
        if not self._m2_due:
            return
        self._m2_due = False
        side = self._side
        if self._m2_events is None:
            self._m2_events = (torch.cuda.Event(), torch.cuda.Event())
        packed, ready = self._m2_events
        ops.record(packed)
        with ops.on_stream(side[1]):
            side[1].wait_event(packed)
            ops.gemm_raw(ptr(self.W0p), self.Vp, 1, ptr(self.WT0), G4, 1, ptr(self.M2), G4, G4, G4, self.V)
            ops.record(ready)
        self._m2_pending = True

    def m2(self) -> torch.Tensor:
        """M2 for a reader on the current stream (ordered behind a build that runs on the side stream)."""
        self.kick_m2()
        if self._m2_pending:
            ops.cur_stream().wait_event(self._m2_events[1])
            self._m2_pending = False
        return self.M2

    def device_scalar(self, value: float) -> torch.Tensor:
        """A cached 1-element device tensor holding `value` (kernel arguments read on the device)."""
        t = self._scalars.get(value)
        if t is None:
            t = self._scalars[value] = torch.full((1,), value, dtype=torch.float32, device=self.device)
        return t

    # ------------------------------------------------------------------ workspaces
    def workspace(self, nseq: int, Tw: int, so_seqs: int) -> Dict[str, torch.Tensor]:
        """Buffers for `nseq` sequences of up to Tw steps; the last `so_seqs` of them also get the
        second-order buffers.  Zero-initialised once: rows beyond a sequence's length keep finite
        stale values and are only ever multiplied by zero-filled gate gradients."""
        key = (nseq, Tw, so_seqs)
        ws = self._ws_cache.get(key)
        if ws is None:
            # a ragged last batch or a batch whose longest sequence is shorter comes back every epoch: keep
            # a few shapes (most recently used last) instead of re-allocating and zero-filling ~0.8 GB each time
            while len(self._ws_cache) >= 4:
                self._ws_cache.pop(next(iter(self._ws_cache)))
            ws = self.fresh_workspace(nseq, Tw, so_seqs)
        else:
            self._ws_cache.pop(key)
        self._ws_cache[key] = ws
        self.ws = ws
        return ws

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
                      norms=z(so_seqs), gp_partial=z(so_seqs, 4), bsumB=z(2, nseq, G4), bsumD=z(2, so_seqs, G4))
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
        ops.gemm_tall(ptr(x), self.V, ptr(self.WT0), G4, ptr(out), G4, rows, G4, self.V, bf16=self.bf16)

    def side(self):
        """The second HIP stream of critic_iteration, confined to 3/4 of the CUs: sweeps C and D run
        one workgroup per sequence of the interp third (B workgroups, each filling a CU's register
        file) and find the reserved quarter of the chip free, while the weight-gradient products that
        only need sweep B fill the rest.  The mask enables the first 3/4 of the CU bits; the driver
        deals mask bits round-robin to the 8 XCDs, so every XCD keeps a quarter of its CUs free.
        Returns None (no overlap, one stream) where the runtime refuses the mask: an unmasked second
        stream would let the GEMM occupy every CU and starve the sweeps' workgroups."""
        if self._side is None:
            import ctypes
            n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
            self.cu_masked = n_cu * 3 // 4
            # The streams are shared by every engine of the process on this device: a process has FOUR hardware
            # queues, and a second trainer with streams of its own would push the count to eight (measured: a
            # second train_wgan() call in one process ran at 9.4 instead of 7.1 ms per step).
            key = (self.device.index if self.device.index is not None else torch.cuda.current_device(),
                   bool(self.reserve_comm_stream))
            shared = _SIDE_STREAMS.get(key)
            if shared is None:
                words = (n_cu + 31) // 32
                mask = (ctypes.c_uint32 * words)()
                for i in range(n_cu * 3 // 4):
                    mask[i // 32] |= 1 << (i % 32)
                handle = ctypes.c_void_p()
                try:
                    call("s2g_stream_create_cu_mask", mask, words, ctypes.byref(handle))
                except S2GError:
                    _SIDE_STREAMS[key] = False
                    self._side = False
                    return None
                stream = torch.cuda.ExternalStream(handle.value, device=self.device)
                handles = [handle.value]
                # a second stream under the same mask for the caller's prefetch hook, so that it runs beside
                # (not behind) this iteration's own work on the first one
                # (not with several ranks: four hardware queues per process, and RCCL's stream takes one)
                stream_g = None
                handle_g = ctypes.c_void_p()
                try:
                    if self.reserve_comm_stream:
                        raise S2GError("no fifth stream")
                    call("s2g_stream_create_cu_mask", mask, words, ctypes.byref(handle_g))
                    stream_g = torch.cuda.ExternalStream(handle_g.value, device=self.device)
                    handles.append(handle_g.value)
                except S2GError:
                    stream_g = None
                import atexit

                def close(hs=tuple(handles), dev=self.device):
                    # destroy the streams while the HIP runtime is still up (leaving it to process teardown
                    # crashes in a runtime finalizer)
                    torch.cuda.synchronize(dev)
                    for h in hs:
                        call("s2g_stream_destroy", h)
                atexit.register(close)
                shared = _SIDE_STREAMS[key] = (stream, torch.cuda.Stream(device=self.device), stream_g)
            elif shared is False:
                self._side = False
                return None
            stream, stream_c, self.stream_g = shared
            # The masked stream is a blocking stream (the call takes no flags): it synchronises
            # implicitly with the NULL stream, torch's default.  The sweeps it must run beside are
            # therefore put on a third, non-blocking stream for the duration of the overlap.
            ev = [torch.cuda.Event() for _ in range(4)]      # fork, after sweep C, join (stream2), join (stream_c)
            self._iter_events = (torch.cuda.Event(), torch.cuda.Event())   # small pieces done, after sweep D
            self._side = (stream, stream_c, *ev, ops.Scratch(self.device))
        return self._side or None

    def timed(self, name: str, fn) -> None:
        """Runs fn(); when the benchmark hook is set, brackets it with HIP events on the current stream
        (the one every libs2g launch of this engine goes to)."""
        te = self.timed_events
        if te is None or not self._timed_now:
            fn()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.record(e0)
        fn()
        ops.record(e1)
        te.setdefault(name, []).append((e0, e1))

    def sweep_fwd(self, ws, desc_ptr, nseq, max_len, tokens, P, mask) -> None:
        whh0, wih1, whh1 = self._w()
        call("s2g_lstm2_fwd", desc_ptr, nseq, max_len, self.V, ptr(tokens), ptr(self.WT0), ptr(P), ptr(mask), ptr(self.b0),
             ptr(self.b1), whh0, wih1, whh1, ptr(ws["gates0"]), ptr(ws["cell0"]), ptr(ws["gates1"]),
             ptr(ws["cell1"]), ptr(ws["hT"]), stream_ptr())

    def head_fwd(self, ws, nseq) -> None:
        p = self.params.views
        call("s2g_head_fwd", ptr(ws["hT"]), ptr(p["fc.weight"]), ptr(p["fc.bias"]), ptr(ws["out"]), nseq, 3,
             stream_ptr())

    def head_args(self, ws, labels, B, mode, inv_n, inv_wsum, class_w) -> S2GHead:
        """The fc head + loss derivative sweep B evaluates per sequence (mode 0: critic batch, 1: generator pass)."""
        p = self.params.views
        return S2GHead(ptr(ws["hT"]), ptr(p["fc.weight"]), ptr(p["fc.bias"]), ptr(labels), B, mode, inv_n, inv_wsum,
                       class_w[0], class_w[1])

    def sweep_bwd(self, ws, desc_ptr, nseq, Tw, mask, adj_seq0: int = -1, head: Optional[S2GHead] = None,
                  bsum: Optional[torch.Tensor] = None) -> None:
        """adj_seq0 >= 0: sequences [adj_seq0, nseq) also store (dh, dc) into the compact adj0/adj1
        buffers (their row 0 = first workspace row of sequence adj_seq0).
        head: the sweep derives d loss / d h_T itself from ws["hT"] (head_args) instead of reading ws["dhT"].
        bsum [2, nseq, 200]: receives every sequence's sums over time of delta0 / delta1."""
        whh0, wih1, whh1 = self._w()
        store = adj_seq0 >= 0
        call("s2g_lstm2_bwd", desc_ptr, nseq, Tw, ptr(mask), whh0, wih1, whh1, ptr(ws["gates0"]), ptr(ws["cell0"]),
             ptr(ws["gates1"]), ptr(ws["cell1"]), ptr(ws["dhT"]), ptr(ws["delta0"]), ptr(ws["delta1"]),
             ptr(ws["adj0"]) if store else None, ptr(ws["adj1"]) if store else None,
             max(adj_seq0, 0), max(adj_seq0, 0) * Tw, ctypes.byref(head) if head is not None else None,
             ptr(bsum), stream_ptr())

    def sweep_tangent(self, ws, desc_ptr, nseq, mask_ptr, row0: int, gp=None) -> None:
        """Sequences of `desc` count rows from workspace row `row0` (their descriptors are relative).
        gp = (partial [nseq,4], upstream, n_batch): the sweep scales sequence b's direction by the penalty
        coefficient it derives from the partial sums of |g_b|^2; None: ws["Pv"] is the direction as is."""
        whh0, wih1, whh1 = self._w()
        o = row0 * G4
        part, up, nb = gp if gp is not None else (None, 0.0, 1.0)
        call("s2g_lstm2_tangent", desc_ptr, nseq, ptr(ws["Pv"]), mask_ptr, whh0, wih1, whh1,
             ptr(ws["gates0"], o), ptr(ws["cell0"], o), ptr(ws["gates1"], o), ptr(ws["cell1"], o),
             ptr(ws["adot0"]), ptr(ws["celld0"]), ptr(ws["adot1"]), ptr(ws["celld1"]), ptr(ws["hdotT"]),
             ptr(part), float(up), float(nb), stream_ptr())

    def sweep_bwd2(self, ws, desc_ptr, nseq, Tw, mask_ptr, row0: int, bsum: Optional[torch.Tensor] = None) -> None:
        """bsum [2, nseq, 200]: receives every sequence's sums over time of abar0 / abar1."""
        whh0, wih1, whh1 = self._w()
        o = row0 * G4
        call("s2g_lstm2_bwd2", desc_ptr, nseq, Tw, mask_ptr, whh0, wih1, whh1, ptr(ws["gates0"], o),
             ptr(ws["cell0"], o), ptr(ws["gates1"], o), ptr(ws["cell1"], o), ptr(ws["adot0"]), ptr(ws["celld0"]),
             ptr(ws["adot1"]), ptr(ws["celld1"]), ptr(ws["adj0"]), ptr(ws["adj1"]), ptr(ws["abar0"]),
             ptr(ws["abar1"]), ptr(bsum), stream_ptr())


def batch_desc(lengths: torch.Tensor, alpha: torch.Tensor, B: int, T: int, Tf: int, use_mask: bool,
               desc: torch.Tensor) -> None:
    """Fills the 4B sequence descriptors of a critic batch [real | fake | interp | interp, group-relative]
    (the padding / packing logic of wgan_gp.py:233-244) on the current stream.  A caller that knows the
    next batch can do this ahead of time and hand the tensor to critic_iteration(desc=...)."""
    call("s2g_critic_batch_desc", ptr(lengths), ptr(alpha), B, T, Tf, int(use_mask), ptr(desc), stream_ptr())


def make_desc(entries, device) -> torch.Tensor:
    """entries: list of dicts(len, tok_off, p_off, p_len, ca, cp, mask_off, ws_row) -> int32 [n,8]
    tensor with the two floats bit-cast (host-built descriptors for the module-level API/tests)."""
    import numpy as np
    arr = np.zeros((len(entries), 8), dtype=np.int32)
    for i, e in enumerate(entries):
        if e["len"] < 0 or e["ws_row"] < 0 or (e.get("p_off", -1) >= 0 and e["len"] > max(e.get("p_len", 0), 0)
                                               and e.get("tok_off", -1) < 0):
            raise S2GError("make_desc: a sequence needs 0 <= len (<= p_len when it has no tokens) and ws_row >= 0")
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


def _wgrad_build(eng: "CriticEngine", jobs, cache_key):
    """jobs: list of (out tensor [200,N] torch layout, N, terms[, options]); a term is
    (A_ptr, K, B_ptr, sbk, sbn, segT, a_seg, b_seg, shift[, phase[, bscale_ptr]]) -- bscale: one factor per sequence
    (segT > 0) on the rows of B.  All products of one phase run in ONE grouped GEMM launch; every product owns its
    split-K slabs.  The slabs of a job are laid out [phases before the last | extra | one slot for their sum | last
    phase], so the early reduce (everything in front of the slot -> the slot) runs while the last phase is still
    being computed and the final reduce reads the slot and the last phase only.
    options: {"extra": n} reserves n more slabs in front of the slot that the CALLER fills (contributions that are
    not products of this launch, in the job's internal row order) before the early reduce;
    {"manual": True}: the job takes no part in the reduces here (the caller sums its slabs itself);
    {"kchunk": rows}: rows per split-K slab of this job's products (a multiple of 32) instead of the phase's.
    Returns (tables {phase: (table, prefix, nprob, nblocks)}, job_slabs, slabs, info) with
    info[job index] = dict(extra=[element offsets of the extra slabs], base, nslabs, slab);
    the device-side tables are cached per `cache_key` and rebuilt only when a pointer or size changes."""
    from .lib import S2GGemm
    jobs = [j if len(j) > 3 else (*j, {}) for j in jobs]
    last = max((term[9] if len(term) > 9 else 0) for _, _, terms, o in jobs for term in terms if not o.get("manual"))
    # K-chunk per phase: 1024 rows unless that leaves the launch with fewer than ~3 blocks per CU
    # (the kernel hides its loads behind the MFMAs of the other blocks on the CU)
    chunk = {}
    for _, N, terms, _o in jobs:
        for term in terms:
            ph = term[9] if len(term) > 9 else 0
            chunk.setdefault(ph, []).append(((N + 63) // 64, term[1]))
    for ph, lst in chunk.items():
        c = 1024
        while c > 256 and sum(t * ((K + c - 1) // c) for t, K in lst if K > 0) < 700:
            c //= 2
        chunk[ph] = c
    plan, sig, job_slabs, info = [], [], [], {}
    base = 0
    for ji, (out, N, terms, opts) in enumerate(jobs):
        slab = G4 * N
        tiles = (N + 63) // 64                               # (all 200 rows) x 64 output tiles
        manual, n_extra = bool(opts.get("manual")), int(opts.get("extra", 0))
        early, late = [], []
        for term in terms:
            A, K, B, sbk, sbn, segT, a_seg, b_seg, shift = term[:9]
            phase = term[9] if len(term) > 9 else 0
            bscale = term[10] if len(term) > 10 else None
            if K <= 0:
                continue
            if opts.get("kchunk"):                         # the caller fixes the rows per slab (a multiple of 32)
                kchunk = int(opts["kchunk"])
            else:
                c = chunk[phase]
                ns = max(1, min(256, (K + c // 2) // c))
                kchunk = ((K + ns - 1) // ns + 31) // 32 * 32
            ns = (K + kchunk - 1) // kchunk
            (late if (phase == last and not manual) else early).append(
                (phase, A, K, B, sbk, sbn, segT, a_seg, b_seg, shift, N, bscale, kchunk, ns))
        n_early = sum(e[-1] for e in early)
        n_late = sum(e[-1] for e in late)
        pos = 0
        for e in early:
            plan.append(e[:-1] + (base + pos * slab, slab, tiles * e[-1]))
            sig.append(e[:12])
            pos += e[-1]
        extra = [base + (pos + k) * slab for k in range(n_extra)]
        pos += n_extra
        if manual:
            info[ji] = dict(extra=extra, base=base, nslabs=pos, slab=slab)
            base += pos * slab
            continue
        n_front = pos                                      # slabs the early reduce sums into the slot
        pos = n_front + (1 if n_front else 0)              # the slot for the early sum
        for e in late:
            plan.append(e[:-1] + (base + pos * slab, slab, tiles * e[-1]))
            sig.append(e[:12])
            pos += e[-1]
        job_slabs.append((out, n_front, n_late, slab, base))
        info[ji] = dict(extra=extra, base=base, nslabs=pos, slab=slab)
        base += pos * slab
    if eng._wgrad_slabs is None or eng._wgrad_slabs.numel() < base:
        eng._wgrad_slabs = torch.empty(max(base, 1), dtype=torch.float32, device=eng.device)
    slabs = eng._wgrad_slabs
    sig = (tuple(sig), slabs.data_ptr())
    cache = eng._gemm_tables.get(cache_key)
    if cache is None or cache[0] != sig:
        tables = {}
        for ph in sorted({e[0] for e in plan}):
            sel = [e for e in plan if e[0] == ph]
            arr = (S2GGemm * len(sel))()
            prefix, blocks = [], 0
            # every product of the phase cut into the same number of K-chunks: blocks ordered K-chunk
            # first (prefix counts output tiles), so products sharing an operand meet its rows in L2
            nzs = {(e[2] + e[12] - 1) // e[12] for e in sel}
            nz = nzs.pop() if len(nzs) == 1 else 0
            for i, (_, A, K, B, sbk, sbn, segT, a_seg, b_seg, shift, N, bscale, kchunk, soff, slab, nblk) in enumerate(sel):
                arr[i] = S2GGemm(A, 1, G4, B, sbk, sbn, ptr(slabs, soff), N, slab, None, G4, N, K, kchunk, segT,
                                 a_seg, b_seg, int(shift), 0, 0, 1.0, bscale)
                prefix.append(blocks)
                blocks += (N + 63) // 64 if nz else nblk
            prefix.append(blocks)
            if nz:
                blocks *= nz
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(eng.device)
            pre = torch.tensor(prefix, dtype=torch.int32, device=eng.device)
            tables[ph] = (raw, pre, len(sel), blocks, nz)
        cache = (sig, tables)
        eng._gemm_tables[cache_key] = cache
    return cache[1], job_slabs, slabs, info


def _wgrad_launch(eng: "CriticEngine", built, phase: int) -> None:
    t = built[0].get(phase)
    if t is not None:
        raw, pre, nprob, nblocks, nz = t
        fn = "s2g_gemm_grouped_bf16" if eng.bf16 else "s2g_gemm_grouped"
        eng.timed(f"s2g_k_gemm_grouped#{phase}",
                  lambda: call(fn, ptr(raw), ptr(pre), nprob, nblocks, nz, stream_ptr()))


def _wgrad_reduce(eng: "CriticEngine", built, early: bool = False) -> None:
    """early: sums every job's slabs of the phases before the last into the job's slot (internal row
    order, may run beside the last phase); otherwise: slot + last phase -> the torch-layout gradient."""
    from .lib import S2GReduce
    _, job_slabs, slabs = built[:3]
    live = []
    for out, n_early, n_late, slab, base in job_slabs:
        if early:
            if n_early:
                live.append((ptr(slabs, base), slab, ptr(slabs, base + n_early * slab), n_early))
        elif n_early + n_late:
            first = base + n_early * slab if n_early else base
            live.append((ptr(slabs, first), slab, ptr(out), (1 if n_early else 0) + n_late))
    for i in range(0, len(live), 8):                    # one launch per 8 parameters
        part = live[i:i + 8]
        arr = (S2GReduce * len(part))()
        for k, (src, slab, dst, nsl) in enumerate(part):
            arr[k] = S2GReduce(src, slab, dst, nsl, G4, slab // G4)
        call("s2g_reduce_slabs_multi", arr, len(part), 0 if early else 1, 0, 1.0, stream_ptr())


def _wgrad_grouped(eng: "CriticEngine", jobs, cache_key) -> None:
    built = _wgrad_build(eng, jobs, cache_key)
    phases = sorted(built[0])
    for ph in phases[:-1]:
        _wgrad_launch(eng, built, ph)
    _wgrad_reduce(eng, built, early=True)
    _wgrad_launch(eng, built, phases[-1])
    _wgrad_reduce(eng, built)


def critic_iteration(eng: CriticEngine, tokens: torch.Tensor, lengths: torch.Tensor, labels: torch.Tensor,
                     fake: torch.Tensor, alpha: torch.Tensor, lambda_gp: float, n_global: int, inv_wsum: float,
                     mask: Optional[torch.Tensor] = None, class_w=(1.0, 2.0), prefetch=None, idle=None,
                     desc: Optional[torch.Tensor] = None, defer_reduce: bool = False):
    """One critic loss + gradient evaluation (wgan_gp.py:297-317) on the batch
    [real B | fake B | interp B], all three passes sharing each of the four sweeps.

    prefetch: optional callable launched on the second stream after this iteration's own work there
    (the trainer generates the next iteration's fake batch in it); it is NOT waited for here, the
    caller synchronises with whatever it enqueued.
    desc: optional descriptors prepared with batch_desc for exactly these lengths / alpha / T / Tf / mask.
    defer_reduce: leave the final slab reduction of the four LSTM weight gradients to
    CriticEngine.fused_update (those four gradients are then NOT yet in eng.grads on return).
    idle: optional callable run on the caller's stream while it would otherwise wait for the
    second-order sweeps (the trainer draws the next iteration's noise and dropout mask in it).

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
    if B > 1024:
        raise S2GError("critic_iteration: at most 1024 sequences per GPU and batch (the loss kernels are one workgroup)")
    Tw = max(T, Tf)
    ws = eng.workspace(3 * B, Tw, B)
    eng._timed_iter += 1
    eng._timed_now = eng._timed_iter % eng.timed_every == 0
    if mask is not None:
        ops._chk(mask, name="mask")
        if tuple(mask.shape) != (3 * B * Tw, MASKW):
            raise S2GError("critic_iteration: mask must be [3*B*Tw, 52]")
    p, g, st = eng.params.views, eng.grads.views, stream_ptr()
    if desc is None:
        desc = ws["desc"]
        batch_desc(lengths, alpha, B, T, Tf, mask is not None, desc)
    elif desc.dtype != torch.int32 or desc.numel() < 4 * B * 8 or not desc.is_cuda:
        raise S2GError("critic_iteration: desc must be an int32 device tensor of 4*B descriptors")
    dptr = ptr(desc)
    if "Pf" not in ws or ws["Pf"].shape[0] != B * Tf:
        from .lib import get
        ws["Pf"] = torch.empty(B * Tf, G4, device=eng.device)
        ws["xhat"] = torch.empty(B * T, V, device=eng.device)
        ws["coef"] = torch.empty(B, device=eng.device)
        ws["gram"] = torch.empty(G4, G4, device=eng.device)
        ws["bucket_part"] = torch.empty(int(get().s2g_embed_grad_tokens_scratch(B * T, V)) * ((G4 + 31) // 32),
                                        device=eng.device)
    # sweep A on 3B sequences
    eng.project(fake, ws["Pf"])
    eng.kick_m2()          # a rebuild of M2 left over from the last update: beside sweep A, whose 192 workgroups leave CUs free
    eng.timed("s2g_k_lstm2_fwd", lambda: eng.sweep_fwd(ws, dptr, 3 * B, Tw, tokens, ws["Pf"], mask))
    inv_n = 1.0 / float(n_global)

    def head_losses():
        """Loss values, d(out) and the fc gradients (wgan_gp.py:221, 299-313).  Sweep B derives its own
        d loss / d h_T per sequence (head_args), so this bookkeeping runs off the critical path."""
        # head, loss, d(out), d(hT) and the fc gradients in one single-workgroup launch
        call("s2g_critic_head", ptr(ws["hT"]), ptr(p["fc.weight"]), ptr(p["fc.bias"]), ptr(labels), B, inv_n,
             inv_wsum, class_w[0], class_w[1], ptr(ws["out"]), ptr(ws["dout"]), ptr(ws["losses"]),
             ptr(ws["dhT"]), ptr(g["fc.weight"]), ptr(g["fc.bias"]), stream_ptr())

    # sweep B on 3B sequences (the interp third with d(out) = e0 and stored adjoints)
    head = eng.head_args(ws, labels, B, 0, inv_n, inv_wsum, class_w)
    eng.timed("s2g_k_lstm2_bwd", lambda: eng.sweep_bwd(ws, dptr, 3 * B, Tw, mask, adj_seq0=2 * B, head=head,
                                                       bsum=ws["bsumB"]))
    # Penalty of the interp third (wgan_gp.py:247-256).  With g_t = W_ih0^T delta0_t the norm is
    # |g_b|^2 = sum_t delta0_t . (M2 delta0_t), M2 = W_ih0 W_ih0^T, and the projected direction the
    # tangent sweep consumes is coef_b * M2 delta0_t: one 200x200 product on the critical path.  The
    # 175-wide direction v = coef_b * g itself feeds one weight-gradient product only and is built
    # beside sweep C (first_order below).
    r_int = 2 * B * Tw
    ops.gemm_tall(ptr(ws["delta0"], r_int * G4), G4, ptr(eng.m2()), G4, ptr(ws["Pv"]), G4, B * Tw, G4, G4,
                  bf16=eng.bf16)
    # only the reduction stays on the critical path: sweep C scales its own seed by coef_b, and the
    # penalty values / norms come out beside it (first_order)
    def gp_dots():
        call("s2g_gp_dots", ptr(ws["delta0"], r_int * G4), ptr(ws["Pv"]), B, Tw * G4, ptr(ws["gp_partial"]),
             stream_ptr())
    gp = (ws["gp_partial"], float(lambda_gp), float(n_global))
    d0, d1, c0, c1 = ws["delta0"], ws["delta1"], ws["cell0"], ws["cell1"]
    a0, a1, cd0, cd1 = ws["abar0"], ws["abar1"], ws["celld0"], ws["celld1"]
    oi = r_int * G4
    n12, n3 = 2 * B * Tw, B * Tw
    # Weight-gradient products; the 10th field is the phase = the last sweep the product needs:
    # 0 = A/B, 1 = C (tangent states), 2 = D (adjoints abar).  dW_ih0 (wgan_gp.py:277, 239, 317) in five parts:
    #   real third   x is one-hot: sum over rows of delta0[row]^T e_token = delta0 rows bucketed by token
    #                (s2g_bucket_rows_by_token_t, an extra slab: no one-hot operand, no product with zeros);
    #   fake third   delta0^T fake;
    #   penalty, first-order factor: sum_rows delta0_int[row]^T v[row] with v = coef_b (delta0_int W_ih0), i.e.
    #                (sum_b coef_b sum_t delta0_int[b,t]^T delta0_int[b,t]) W_ih0: a 200-wide Gram product and one
    #                200 x 200 x V product (an extra slab) -- the 175-wide direction v is never formed.  With one
    #                split-K slab per sequence the Gram product needs no coefficient (phase 0) and the slab sum
    #                applies coef_b; when a sequence's rows are no multiple of 32 the factor rides on one operand
    #                instead (phase 1, when coef is known);
    #   penalty, second-order factor: abar0^T x_hat (phase 2).
    coef = ws["coef"]
    jobs = [(g["lstm.weight_ih_l0"], V, [
        (ptr(d0, B * Tw * G4), B * Tf, ptr(fake), V, 1, Tf, Tw, Tf, False, 0),
        (ptr(a0), B * T, ptr(ws["xhat"]), V, 1, T, Tw, T, False, 2)], {"extra": 2}),
            (g["lstm.weight_hh_l0"], H, [
        (ptr(d0), n3, ptr(c0, 2), G4, 4, Tw, Tw, Tw, True, 0),
        (ptr(d0, oi // 2), n3, ptr(c0, oi // 2 + 2), G4, 4, Tw, Tw, Tw, True, 0),
        (ptr(a0), n3, ptr(c0, oi + 2), G4, 4, Tw, Tw, Tw, True, 2),
        (ptr(d0, oi), n3, ptr(cd0, 1), G4, 4, Tw, Tw, Tw, True, 1)]),
            (g["lstm.weight_ih_l1"], H, [
        (ptr(d1), n3, ptr(c0, 3), G4, 4, 0, 0, 0, False, 0),
        (ptr(d1, oi // 2), n3, ptr(c0, oi // 2 + 3), G4, 4, 0, 0, 0, False, 0),
        (ptr(a1), n3, ptr(c0, oi + 3), G4, 4, 0, 0, 0, False, 2),
        (ptr(d1, oi), n3, ptr(cd0, 2), G4, 4, 0, 0, 0, False, 1)]),
            (g["lstm.weight_hh_l1"], H, [
        (ptr(d1), n3, ptr(c1, 2), G4, 4, Tw, Tw, Tw, True, 0),
        (ptr(d1, oi // 2), n3, ptr(c1, oi // 2 + 2), G4, 4, Tw, Tw, Tw, True, 0),
        (ptr(a1), n3, ptr(c1, oi + 2), G4, 4, Tw, Tw, Tw, True, 2),
        (ptr(d1, oi), n3, ptr(cd1, 1), G4, 4, Tw, Tw, Tw, True, 1)]),
            (ws["gram"], G4, [
        (ptr(d0, oi), n3, ptr(d0, oi), G4, 1, Tw, Tw, Tw, False, 0)], {"manual": True, "kchunk": Tw}) if Tw % 32 == 0 else
            (ws["gram"], G4, [
        (ptr(d0, oi), n3, ptr(d0, oi), G4, 1, Tw, Tw, Tw, False, 1, ptr(coef))], {"manual": True})]
    built = _wgrad_build(eng, jobs, ("critic", B, T, Tf, fake.data_ptr()))
    slabs, info = built[2], built[3]
    slab_real, slab_gram = info[0]["extra"]
    side = eng.side() if eng.overlap else None

    def first_order(sc):
        """Everything that needs sweeps A and B only: operands of the products and their phase-0 part."""
        s_ = stream_ptr()
        call("s2g_gp_interp", ptr(tokens), ptr(lengths), ptr(fake), ptr(alpha), B, T, Tf, V, ptr(ws["xhat"]), s_)
        _wgrad_launch(eng, built, 0)

    def real_third():
        """dW_ih0 of the real third, [gate row][token], straight into its slab"""
        call("s2g_bucket_rows_by_token_t", ptr(tokens), ptr(lengths), T, B * T, V, ptr(d0), G4, Tw, G4,
             ptr(slabs, slab_real), V, ptr(ws["bucket_part"]), stream_ptr())

    def after_tangent():
        """What needs the penalty's reduction (ordered behind it by the caller): penalty values, norms and the
        per-sample coefficient (wgan_gp.py:255-256), the phase-1 products with the Gram product among them, its
        slab sum, and its product with W_ih0 into dW_ih0's second extra slab."""
        call("s2g_gp_penalty_final", ptr(ws["gp_partial"]), B, float(lambda_gp), float(n_global), ptr(ws["pen"]),
             ptr(ws["norms"]), ptr(coef), stream_ptr())
        _wgrad_launch(eng, built, 1)
        gi = info[4]
        if Tw % 32 == 0:        # one slab per sequence: the slab sum applies the per-sample coefficient
            call("s2g_reduce_slabs_weighted", ptr(slabs, gi["base"]), gi["slab"], gi["nslabs"], ptr(coef), ptr(ws["gram"]),
                 G4 * G4, stream_ptr())
        else:
            ops.reduce_slabs(slabs[gi["base"]:], gi["slab"], gi["nslabs"], G4, G4, ws["gram"])
        ops.gemm_raw(ptr(ws["gram"]), G4, 1, ptr(eng.W0p), eng.Vp, 1, ptr(slabs, slab_gram), V, G4, V, G4)

    def bias_sums(sums, nseq, hdot, accumulate):
        """bias gradients of both layers from the per-sequence sums the reverse sweeps leave behind
        (sequences [0, nseq) of sums [2, *, 200]); one small launch"""
        call("s2g_bias_sums_final", ptr(sums), nseq, sums.shape[1] * G4, ptr(hdot), B, ptr(g["lstm.bias_ih_l0"]),
             ptr(g["lstm.bias_hh_l0"]), ptr(g["lstm.bias_ih_l1"]), ptr(g["lstm.bias_hh_l1"]), ptr(g["fc.weight"]),
             int(accumulate), stream_ptr())

    def first_order_sums(sc):
        bias_sums(ws["bsumB"], 2 * B, None, False)         # real + fake sequences; the interp third only seeds the penalty

    drel = ptr(desc, 3 * B * 8)          # group-relative descriptors of the interp third = 4th block of desc
    mrel = ptr(mask, r_int * MASKW) if mask is not None else None

    sc = eng.scratch

    def second_order_sums():
        """bias gradients of the penalty term and its part of fc.weight[0] = sum_b hdot1_final(b)"""
        bias_sums(ws["bsumD"], B, ws["hdotT"], True)

    if side is not None:
        # stream_c: sweep C, sweep D (B workgroups on the reserved quarter of the CUs), then the
        #           second-order column sums, which run beside the last products on the caller's stream
        # stream2 : products of phase 0 beside sweep C, products of phase 1 beside sweep D
        stream2, stream_c, fork, after_c, join, join_c, sc2 = side
        cur = ops.cur_stream()
        # The chain A -> B -> penalty -> C -> D -> last products is the critical path: it stays on ONE
        # stream (every hop to another stream costs ~10 us of event latency), the caller's if that is a
        # non-blocking stream.  On the NULL stream the sweeps move to stream_c instead, because the
        # masked streams are blocking streams and would serialise with every launch there.
        own = cur.cuda_stream != 0
        sweeps = cur if own else stream_c
        if own:
            # fork BEFORE the penalty reduction: no event record sits between it and sweep C on this stream
            ops.record(fork)
            gp_dots()
        else:
            gp_dots()
            ops.record(fork)
        with ops.on_stream(stream2):
            stream2.wait_event(fork)
            first_order(sc2)
        with ops.on_stream(sweeps):
            if not own:
                stream_c.wait_event(fork)
            eng.timed("s2g_k_lstm2_tangent", lambda: eng.sweep_tangent(ws, drel, B, mrel, r_int, gp))
            ops.record(after_c)
        # the small launch-bound pieces (loss values + fc gradients, first-order bias sums) go beside the
        # operand / product chain of stream2, which must be through before sweep D is
        small = eng.stream_g if eng.stream_g is not None else stream2
        join_g, after_d = eng._iter_events
        with ops.on_stream(small):
            small.wait_event(fork)
            real_third()
            head_losses()
            first_order_sums(sc2)
            ops.record(join_g)
        with ops.on_stream(stream2):
            stream2.wait_event(after_c)         # (sweep C is queued behind the penalty's reduction)
            after_tangent()
            if small is not stream2:
                stream2.wait_event(join_g)      # the bucketed real third is one of the slabs summed next
            _wgrad_reduce(eng, built, early=True)
            ops.record(join)
            if prefetch is not None and eng.stream_g is None:
                prefetch()          # not part of the join: it may run on beside this iteration's tail
        if prefetch is not None and eng.stream_g is not None:
            # The hook's work (not part of the join) goes to the masked stream of its own.  Its light part
            # starts right away; its heavy, HBM-bound part (the generator's output layer, 183 MB of weights)
            # waits for the phase-1 products (`join`): beside the phase-0 products it slows them past the end of
            # sweep C (308 against 217 us); beside the phase-1 products both crawl (134 + 144 us under sweep D
            # instead of 55 + 80 one after the other) -- the step time is the same either way, but one after the
            # other they finish well before sweep D does and leave the last products alone on the chip.
            with ops.on_stream(eng.stream_g):
                prefetch(lambda: eng.stream_g.wait_event(join))
        if idle is not None and own:
            with ops.on_stream(stream_c):      # free while the sweeps run on the caller's stream
                stream_c.wait_event(fork)
                idle()
        with ops.on_stream(sweeps):
            eng.timed("s2g_k_lstm2_bwd2", lambda: eng.sweep_bwd2(ws, drel, B, Tw, mrel, r_int, bsum=ws["bsumD"]))
            ops.record(after_d)
        if idle is not None and not own:
            idle()
        if not own:
            cur.wait_event(after_d)
        cur.wait_event(join)
        _wgrad_launch(eng, built, 2)          # (join_g is covered by join_c below: stream_c waits for it first)
        with ops.on_stream(stream_c):
            if own:
                stream_c.wait_event(after_d)
            stream_c.wait_event(join_g)      # the first-order bias sums and fc gradients these accumulate onto
            second_order_sums()
            ops.record(join_c)
        if not defer_reduce:
            _wgrad_reduce(eng, built)
        cur.wait_event(join_c)
    else:
        gp_dots()
        eng.timed("s2g_k_lstm2_tangent", lambda: eng.sweep_tangent(ws, drel, B, mrel, r_int, gp))
        eng.timed("s2g_k_lstm2_bwd2", lambda: eng.sweep_bwd2(ws, drel, B, Tw, mrel, r_int, bsum=ws["bsumD"]))
        head_losses()
        first_order(sc)
        real_third()
        first_order_sums(sc)
        after_tangent()
        _wgrad_reduce(eng, built, early=True)
        if prefetch is not None:
            prefetch()
        _wgrad_launch(eng, built, 2)
        if not defer_reduce:
            _wgrad_reduce(eng, built)
        second_order_sums()
        if idle is not None:
            idle()
    eng._pending_built = built if defer_reduce else None
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
    # the reverse sweep derives d loss / d h_T per sequence itself; the loss values follow behind it
    head = eng.head_args(ws, labels, B, 1, 1.0 / float(n_global), inv_wsum, class_w)

    def loss_values():
        eng.head_fwd(ws, B)
        call("s2g_generator_loss", ptr(ws["out"]), ptr(labels), B, 1.0 / float(n_global), inv_wsum, class_w[0],
             class_w[1], ptr(ws["dout"]), ptr(ws["losses"]), stream_ptr())

    side = eng.side() if eng.overlap else None
    if side is not None:
        fwd_done, vals_done = torch.cuda.Event(), torch.cuda.Event()
        ops.record(fwd_done)
        with ops.on_stream(side[1]):          # beside the reverse sweep
            side[1].wait_event(fwd_done)
            loss_values()
            ops.record(vals_done)
    eng.sweep_bwd(ws, ptr(desc), B, Tf, mask, head=head)
    ops.gemm_tall(ptr(ws["delta0"]), G4, ptr(eng.W0p), eng.Vp, ptr(ws["dfake"]), V, B * Tf, V, G4, bf16=eng.bf16)
    if side is not None:
        ops.cur_stream().wait_event(vals_done)
    else:
        loss_values()
    return ws["losses"], ws["dfake"].view(B, Tf, V)
