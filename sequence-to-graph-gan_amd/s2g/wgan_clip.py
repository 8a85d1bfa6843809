"""Host orchestration of the weight-clipping WGAN variant (reference wgan.py): the LSTM generator's forward
(:46-71), the LSTM + attention-pooling critic's forward and hand-derived backward (:92-122), and the critic
update of CWGAN.train_step (:174-198: Wasserstein loss, Adam, weight clipping).  Every product is a libs2g
kernel; one LSTM layer = one batched input product + one sweep over time; weight gradients are batched
products over the saved gate gradients."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import torch

from . import ops
from .critic import FlatBuffer
from .lib import S2GError, S2GLstm64Stack, S2GLstm64StackBwd, call, get
from .ops import ptr, stream_ptr

HID, G4X, EMB = 64, 256, 32
LATENT_LEN = 100          # Generator.seq_len = latent_dim = 100 (wgan.py:52-53)


def _lstm_shapes(prefix: str, layers: int):
    out = []
    for l in range(layers):
        out += [(f"{prefix}.weight_ih_l{l}", (G4X, HID)), (f"{prefix}.weight_hh_l{l}", (G4X, HID)),
                (f"{prefix}.bias_ih_l{l}", (G4X,)), (f"{prefix}.bias_hh_l{l}", (G4X,))]
    return out


def critic_shapes(vocab: int, layers: int, n_classes: int = 2):
    """state_dict order of the reference Discriminator (wgan.py:93-103)."""
    return ([("label_embedding.weight", (n_classes, EMB)), ("sequence_embedding.embed_layer.weight", (vocab, EMB))]
            + _lstm_shapes("lstm", layers)
            + [("attention.attn.weight", (HID, HID)), ("attention.attn.bias", (HID,)),
               ("attention.context_vector.weight", (1, HID)), ("linear.weight", (1, HID)), ("linear.bias", (1,))])


def generator_shapes(vocab: int, layers: int, n_classes: int = 2):
    """state_dict order of the reference Generator (wgan.py:47-61)."""
    return ([("label_embedding.weight", (n_classes, EMB)), ("noise_embedding.weight", (vocab, EMB))]
            + _lstm_shapes("lstm", layers) + [("linear.weight", (vocab, HID)), ("linear.bias", (vocab,))])


def _tn(A: int, lda: int, M: int, B: int, ldb: int, N: int, K: int, out: torch.Tensor, scratch: ops.Scratch,
        accumulate: bool, segT: int = 0, shift: bool = False) -> None:
    """out[M, N] (+)= sum_k A[k][m] B[k][n] over K rows (A, B row-major with row strides lda, ldb); with
    shift, B's row k - 1 is used and the first row of every segment of segT rows contributes nothing."""
    ns = ops.pick_split(M, N, K)
    if ns == 1:
        ops.gemm_raw(A, 1, lda, B, ldb, 1, ptr(out), N, M, N, K, accumulate=accumulate, segT=segT, a_seg=segT,
                     b_seg=segT, shift=shift)
    else:
        slabs = scratch.get(ns * M * N)
        ops.gemm_raw(A, 1, lda, B, ldb, 1, ptr(slabs), N, M, N, K, nsplit=ns, slab=M * N, segT=segT, a_seg=segT,
                     b_seg=segT, shift=shift)
        ops.reduce_slabs(slabs, M * N, ns, M, N, out, accumulate=accumulate)


class _Fan:
    """Two more streams, each with split-K scratch of its own, for the weight-gradient products of a long pass
    that nothing later in the pass waits for: they leave its critical chain of sweeps."""

    def __init__(self, stream_b: torch.cuda.Stream, scratch_b: ops.Scratch, device):
        self.b, self.scratch_b = stream_b, scratch_b
        self.c, self.scratch_c = torch.cuda.Stream(device=device), ops.Scratch(device)
        self.fork_ev = [torch.cuda.Event() for _ in range(2)]
        self.join_ev = [torch.cuda.Event() for _ in range(2)]
        self.keep: List[torch.Tensor] = []

    def hold(self, *tensors) -> None:
        """Tensors allocated on the caller's stream that the two streams read: kept referenced until join().
        (Dropped earlier, the caching allocator hands their blocks to the caller stream's next allocation,
        which may be written while the products queued here still read them.)"""
        self.keep.extend(t for t in tensors if t is not None)

    def join(self) -> None:
        """The current stream waits for everything queued on the two streams so far; what hold() kept is
        released behind that wait (a later allocation on the current stream is ordered after the readers)."""
        cur = ops.cur_stream()
        for stream, ev in zip((self.b, self.c), self.join_ev):
            ev.record(stream)
            cur.wait_event(ev)
        self.keep.clear()


class _LstmStack:
    """Shared by both networks: the layers of an nn.LSTM(64, 64, layers, batch_first=True)."""

    def __init__(self, params: FlatBuffer, layers: int, device):
        self.p, self.layers, self.device = params, layers, device
        self.bsum = [torch.empty(G4X, device=device) for _ in range(layers)]

    def forward(self, x: Optional[torch.Tensor], B: int, T: int, lengths: Optional[torch.Tensor],
                masks: Optional[List[Optional[torch.Tensor]]], save: bool, tok=None):
        """tok (2- and 3-layer stacks only, x unused): (tokens int32 [B, T], labels, embedding [V, 32], label
        embedding [n_classes, 32]) -- layer 0's input part becomes a table lookup inside the sweep."""
        p, dev, rows = self.p.views, self.device, B * T
        if self.layers in (2, 3):
            return self._forward_fused(x, B, T, lengths, masks, save, tok)
        if tok is not None:
            raise S2GError("the token-table input needs a stack of 2 or 3 layers")
        inp, saved = x, []
        for l in range(self.layers):
            # b_ih + b_hh (adjacent views of the flat buffer) -> one bias for the batched input product
            ops.reduce_slabs(p[f"lstm.bias_ih_l{l}"], G4X, 2, 1, G4X, self.bsum[l])
            pin = torch.empty(rows, G4X, device=dev)
            ops.linear_fwd(inp, p[f"lstm.weight_ih_l{l}"], self.bsum[l], pin)
            gates, cst, hout = (torch.empty(rows, G4X, device=dev), torch.empty(rows, HID, device=dev),
                                torch.empty(rows, HID, device=dev))
            mask = masks[l] if (masks is not None and l + 1 < self.layers) else None
            hdrop = torch.empty(rows, HID, device=dev) if mask is not None else None
            call("s2g_lstm64_fwd", ptr(pin), ptr(p[f"lstm.weight_hh_l{l}"]), ptr(lengths), B, T, ptr(gates), ptr(cst),
                 ptr(hout), ptr(mask), ptr(hdrop), stream_ptr())
            if save:
                saved.append(dict(inp=inp, gates=gates, cst=cst, hout=hout, mask=mask))
            inp = hdrop if hdrop is not None else hout
        return hout, saved

    def _forward_fused(self, x: Optional[torch.Tensor], B: int, T: int, lengths: Optional[torch.Tensor],
                       masks: Optional[List[Optional[torch.Tensor]]], save: bool, tok=None):
        """All layers in one sweep (s2g_lstm64_stack_fwd): only layer 0 takes a batched input product, or, for
        token inputs, two small tables: x_t W_ih0^T = (emb W_ih0[:, :32]^T)[token_t] + (lab_emb W_ih0[:, 32:]^T)[label]."""
        p, dev, rows, NL = self.p.views, self.device, B * T, self.layers
        for l in range(NL):
            ops.reduce_slabs(p[f"lstm.bias_ih_l{l}"], G4X, 2, 1, G4X, self.bsum[l])
        st, saved, inp = S2GLstm64Stack(), [], x
        W0 = p["lstm.weight_ih_l0"]
        if tok is not None:
            tokens, labels, emb, lab_emb = tok
            V, NC = emb.shape[0], lab_emb.shape[0]
            tab, tabl = torch.empty(V, G4X, device=dev), torch.empty(NC, G4X, device=dev)
            ops.gemm_raw(ptr(emb), EMB, 1, ptr(W0), 1, 2 * EMB, ptr(tab), G4X, V, G4X, EMB, bias=ptr(self.bsum[0]))
            ops.gemm_raw(ptr(lab_emb), EMB, 1, ptr(W0, EMB), 1, 2 * EMB, ptr(tabl), G4X, NC, G4X, EMB)
            st.tokens, st.tab, st.labels, st.tabl, st.V, st.n_classes = ptr(tokens), ptr(tab), ptr(labels), ptr(tabl), V, NC
            keep = (tab, tabl)            # referenced until the launch below is queued
        else:
            pin = torch.empty(rows, G4X, device=dev)
            ops.linear_fwd(x, W0, self.bsum[0], pin)
            st.pin0 = ptr(pin)
        for l in range(NL):
            top = l + 1 == NL
            mask = masks[l] if (masks is not None and not top) else None
            gates = torch.empty(rows, G4X, device=dev) if save else None
            cst = torch.empty(rows, HID, device=dev) if save else None
            hout = torch.empty(rows, HID, device=dev) if (save or top) else None
            hdrop = torch.empty(rows, HID, device=dev) if mask is not None else None
            st.whh[l] = ptr(p[f"lstm.weight_hh_l{l}"])
            if l:
                st.wih[l], st.bsum[l] = ptr(p[f"lstm.weight_ih_l{l}"]), ptr(self.bsum[l])
            st.gates[l], st.cst[l], st.hout[l], st.mask[l], st.hdrop[l] = (ptr(gates), ptr(cst), ptr(hout), ptr(mask),
                                                                          ptr(hdrop))
            if save:
                saved.append(dict(inp=inp, gates=gates, cst=cst, hout=hout, mask=mask))
            inp = hdrop if hdrop is not None else hout
        call("s2g_lstm64_stack_fwd", ctypes.byref(st), NL, ptr(lengths), B, T, stream_ptr())
        return hout, saved

    def backward(self, dh: torch.Tensor, saved, B: int, T: int, lengths: Optional[torch.Tensor], g: Dict[str, torch.Tensor],
                 scratch: ops.Scratch, accumulate: bool, fan: Optional[_Fan] = None, tok: bool = False):
        """dh [rows, 64]: gradient of the top layer's outputs.  Accumulates the LSTM parameter gradients into g
        and returns the gradient of the stack's input [rows, 64].  fan (two-layer stacks): the weight-gradient
        products run on its streams, layer 1's on one and layer 0's on the other; the caller joins them.
        tok (the forward took token tables): layer 0's input-side gradients are the caller's (they follow from
        token- and label-bucketed sums of its gate gradients); returns (gate gradients of layer 0 [rows, 256],
        their per-sequence sums [B, 256]) instead."""
        p, dev, rows = self.p.views, self.device, B * T

        def weight_grads(l, delta, sc, bias_part=None):
            s = saved[l]
            if not (tok and l == 0):
                _tn(ptr(delta), G4X, G4X, ptr(s["inp"]), HID, HID, rows, g[f"lstm.weight_ih_l{l}"], sc, accumulate)
            _tn(ptr(delta), G4X, G4X, ptr(s["hout"]), HID, HID, rows, g[f"lstm.weight_hh_l{l}"], sc, accumulate,
                segT=T, shift=True)
            if bias_part is not None:      # the sweep already summed delta over time: B rows instead of B T
                ops.colsum(bias_part, B, G4X, G4X, g[f"lstm.bias_ih_l{l}"], sc, accumulate=accumulate)
            else:
                ops.colsum(delta, rows, G4X, G4X, g[f"lstm.bias_ih_l{l}"], sc, accumulate=accumulate)

        fused = None
        if self.layers == 2:
            # both reverse sweeps in one launch; layer 0's incoming gradient never leaves the chip
            fused = [torch.empty(rows, G4X, device=dev) for _ in range(2)]
            bparts = [torch.empty(B, G4X, device=dev) for _ in range(2)]
            st = S2GLstm64StackBwd()
            st.dh_top, st.wih1, st.mask0 = ptr(dh), ptr(p["lstm.weight_ih_l1"]), ptr(saved[0]["mask"])
            for l in range(2):
                st.whh[l], st.gates[l], st.cst[l], st.delta[l] = (ptr(p[f"lstm.weight_hh_l{l}"]), ptr(saved[l]["gates"]),
                                                                ptr(saved[l]["cst"]), ptr(fused[l]))
                st.bias_part[l] = ptr(bparts[l])
            call("s2g_lstm64_stack_bwd", ctypes.byref(st), ptr(lengths), B, T, stream_ptr())
            if fan is not None:
                fan.hold(*fused, *bparts, dh)
                ops.record(fan.fork_ev[1])
                for l, stream, sc in ((1, fan.b, fan.scratch_b), (0, fan.c, fan.scratch_c)):
                    with ops.on_stream(stream):
                        stream.wait_event(fan.fork_ev[1])
                        weight_grads(l, fused[l], sc, bparts[l])
                if tok:
                    return fused[0], bparts[0]
                dinp = torch.empty(rows, HID, device=dev)
                ops.linear_bwd_dx(fused[0], p["lstm.weight_ih_l0"], dinp, scratch)
                return dinp
        for l in reversed(range(self.layers)):
            s = saved[l]
            bpart = None
            if fused is not None:
                delta, bpart = fused[l], bparts[l]
            else:
                delta = torch.empty(rows, G4X, device=dev)
                call("s2g_lstm64_bwd", ptr(dh), ptr(p[f"lstm.weight_hh_l{l}"]), ptr(lengths), B, T, ptr(s["gates"]),
                     ptr(s["cst"]), ptr(delta), stream_ptr())
            weight_grads(l, delta, scratch, bpart)
            if fused is not None and l > 0:
                continue                  # the sweep below already took delta W_ih (times the mask) on chip
            if tok:
                return fused[0], bparts[0]
            dinp = torch.empty(rows, HID, device=dev)
            ops.linear_bwd_dx(delta, p[f"lstm.weight_ih_l{l}"], dinp, scratch)
            if l > 0 and saved[l - 1]["mask"] is not None:
                call("s2g_mul_inplace", ptr(dinp), ptr(saved[l - 1]["mask"]), dinp.numel(), stream_ptr())
            dh = dinp
        return dh


class ClipGeneratorEngine:
    """Generator.forward of wgan.py:63-71 (forward only: the critic update detaches its output, :185)."""

    def __init__(self, vocab: int, layers: int, device, n_classes: int = 2):
        self.V, self.layers, self.n_classes, self.device = vocab, layers, n_classes, device
        self.params = FlatBuffer(generator_shapes(vocab, layers, n_classes), device)
        self.stack = _LstmStack(self.params, layers, device)
        self.rng = ops.DeviceRng()

    def load(self, state: Dict[str, torch.Tensor]) -> None:
        for k, v in self.params.views.items():
            v.copy_(state[k].to(self.device, torch.float32))

    def forward(self, noise: torch.Tensor, labels: torch.Tensor, tau: float, U: Optional[torch.Tensor] = None,
                act: str = "gumbel_softmax", hard: bool = False) -> torch.Tensor:
        """noise int32 [B, 100] token ids (torch.randint(0, V, ...), wgan.py:183), labels int64 [B] -> [B, 100, V]."""
        ops._chk(noise, torch.int32, "noise"); ops._chk(labels, torch.int64, "labels")
        B, T = noise.shape
        p, dev = self.params.views, self.device
        x = torch.empty(B * T, 2 * EMB, device=dev)
        call("s2g_embed_concat", ptr(noise), ptr(p["noise_embedding.weight"]), self.V, ptr(labels),
             ptr(p["label_embedding.weight"]), self.n_classes, T, B * T, 0, ptr(x), stream_ptr())
        hout, _ = self.stack.forward(x, B, T, None, None, save=False)
        logits = torch.empty(B * T, self.V, device=dev)
        ops.linear_fwd(hout, p["linear.weight"], p["linear.bias"], logits)
        mode = ops.MODES.get(act, 2)
        y, _ = ops.gumbel_softmax_fwd(logits.view(B, T, self.V), U if mode == 0 else None, tau, mode, hard, rng=self.rng)
        return y


class ClipCriticEngine:
    """Discriminator of wgan.py:92-122 with its backward, and the critic update of CWGAN.train_step (:174-198)."""

    def __init__(self, vocab: int, layers: int, device, dropout: float = 0.1, n_classes: int = 2):
        self.V, self.layers, self.n_classes, self.device, self.dropout = vocab, layers, n_classes, device, dropout
        self.params = FlatBuffer(critic_shapes(vocab, layers, n_classes), device)
        self.grads = self.params.like()
        self.m, self.v = self.params.like(), self.params.like()
        self.step = 0
        self.stack = _LstmStack(self.params, layers, device)
        self.scratch = ops.Scratch(device)
        self.rng = ops.DeviceRng()
        self.overlap = True           # critic_step: the fake pass on a second stream beside the real pass
        self._side = None

    def load(self, state: Dict[str, torch.Tensor]) -> None:
        for k, v in self.params.views.items():
            v.copy_(state[k].to(self.device, torch.float32))

    # ------------------------------------------------------------------ one pass (real tokens or soft fakes)
    def forward(self, seq: torch.Tensor, labels: torch.Tensor, lengths: Optional[torch.Tensor], Tp: int,
                masks=None, dval: Optional[torch.Tensor] = None, save: bool = True):
        """seq: int32 [B, T] tokens (the lookup branch of RelaxedEmbedding) or fp32 [B, T, V] (the matmul branch);
        lengths int32 [B] on the device or None (all T); Tp = the longest length (pad_packed_sequence's width).
        Returns (D values [B], saved state)."""
        p, dev = self.params.views, self.device
        soft = seq.dtype != torch.int32
        B, T = seq.shape[0], seq.shape[1]
        rows = B * T
        if not 1 <= Tp <= T:
            raise S2GError("critic: the longest length must lie in [1, T]")
        tok = (not soft) and self.layers in (2, 3)
        if tok:
            # tokens: embedding lookup, concat and layer 0's input product collapse into two small tables
            x = None
            out, saved = self.stack.forward(None, B, T, lengths, masks, save,
                                            tok=(seq, labels, p["sequence_embedding.embed_layer.weight"],
                                                 p["label_embedding.weight"]))
        else:
            x = torch.empty(rows, 2 * EMB, device=dev)
            if soft:
                ops._chk(seq, name="sequence")
                ops.gemm_raw(ptr(seq), self.V, 1, ptr(p["sequence_embedding.embed_layer.weight"]), EMB, 1, ptr(x),
                             2 * EMB, rows, EMB, self.V)
            call("s2g_embed_concat", None if soft else ptr(seq), ptr(p["sequence_embedding.embed_layer.weight"]),
                 self.V, ptr(labels), ptr(p["label_embedding.weight"]), self.n_classes, T, rows, int(soft), ptr(x),
                 stream_ptr())
            out, saved = self.stack.forward(x, B, T, lengths, masks, save)
        pre = torch.empty(rows, HID, device=dev)
        ops.linear_fwd(out, p["attention.attn.weight"], p["attention.attn.bias"], pre)
        weights, pooled = torch.empty(B, T, device=dev), torch.empty(B, HID, device=dev)
        if dval is None:
            dval = torch.empty(B, device=dev)
        call("s2g_attnpool_fwd", ptr(pre), ptr(out), ptr(p["attention.context_vector.weight"]), ptr(p["linear.weight"]),
             ptr(p["linear.bias"]), B, T, Tp, ptr(weights), ptr(pooled), ptr(dval), stream_ptr())
        state = dict(seq=seq, soft=soft, tok=tok, labels=labels, lengths=lengths, B=B, T=T, Tp=Tp, x=x, out=out, a=pre,
                     weights=weights, pooled=pooled, lstm=saved) if save else None
        return dval, state

    def backward(self, st, dD: torch.Tensor, accumulate: bool, grads=None, scratch=None,
                 fan: Optional[_Fan] = None, want_dseq: bool = False) -> Optional[torch.Tensor]:
        """Adds (or writes, accumulate=False) the parameter gradients of sum_n dD[n] * D_n into self.grads
        (or `grads`, with `scratch` for the split-K slabs: a pass that runs beside another needs both of its own).
        fan: streams for the weight-gradient products off the chain of sweeps; joined before this returns.
        want_dseq (relaxed input only): also returns the gradient with respect to the input [B, T, V]
        (through the matmul branch of RelaxedEmbedding, wgan.py:45)."""
        p, g, dev, sc = self.params.views, (grads or self.grads).views, self.device, scratch or self.scratch
        B, T, Tp, rows = st["B"], st["T"], st["Tp"], st["B"] * st["T"]
        dout = torch.empty(rows, HID, device=dev)
        dwc = torch.empty(B, HID, device=dev)
        call("s2g_attnpool_bwd", ptr(st["a"]), ptr(st["out"]), ptr(p["attention.context_vector.weight"]),
             ptr(p["linear.weight"]), ptr(st["weights"]), ptr(dD), B, T, Tp, ptr(dout), ptr(dwc), None, None,
             stream_ptr())
        dpre = st["a"]

        def head_grads(sc_):
            ops.colsum(dwc, B, HID, HID, g["attention.context_vector.weight"], sc_, accumulate=accumulate)
            _tn(ptr(dD), 1, 1, ptr(st["pooled"]), HID, HID, B, g["linear.weight"], sc_, accumulate)
            ops.colsum(dD, B, 1, 1, g["linear.bias"], sc_, accumulate=accumulate)
            _tn(ptr(dpre), HID, HID, ptr(st["out"]), HID, HID, rows, g["attention.attn.weight"], sc_, accumulate)
            ops.colsum(dpre, rows, HID, HID, g["attention.attn.bias"], sc_, accumulate=accumulate)

        if fan is not None and self.layers == 2:
            ops.record(fan.fork_ev[0])
            with ops.on_stream(fan.b):
                fan.b.wait_event(fan.fork_ev[0])
                head_grads(fan.scratch_b)
        else:
            fan = None
            head_grads(sc)
        # d(out) += d(pre) Wa
        ops.gemm_raw(ptr(dpre), HID, 1, ptr(p["attention.attn.weight"]), HID, 1, ptr(dout), HID, rows, HID, HID,
                     accumulate=True)
        ge = g["sequence_embedding.embed_layer.weight"]
        if st["tok"]:
            # S [V, 256] / Sl [n_classes, 256]: layer 0's gate gradients summed per token / per label; the
            # gradients of W_ih0, of the embedding and of the label embedding are small products of them
            delta0, bpart0 = self.stack.backward(dout, st["lstm"], B, T, st["lengths"], g, sc, accumulate, fan, tok=True)
            lens = st["lengths"]
            if lens is None:
                lens = torch.full((B,), T, dtype=torch.int32, device=dev)
            V, NC = self.V, self.n_classes
            part = sc.get(int(get().s2g_embed_grad_tokens_scratch(rows, V)) * (G4X // 32))
            S, Sl = torch.empty(V, G4X, device=dev), torch.empty(NC, G4X, device=dev)
            call("s2g_bucket_rows_by_token", ptr(st["seq"]), ptr(lens), T, rows, V, ptr(delta0), G4X, G4X, ptr(S),
                 ptr(part), stream_ptr())
            call("s2g_bucket_rows_by_label", ptr(st["labels"]), B, NC, ptr(bpart0), G4X, ptr(Sl), stream_ptr())
            emb, lab, W0 = p["sequence_embedding.embed_layer.weight"], p["label_embedding.weight"], p["lstm.weight_ih_l0"]
            gW0 = g["lstm.weight_ih_l0"]
            ops.gemm_raw(ptr(S), 1, G4X, ptr(emb), EMB, 1, ptr(gW0), 2 * EMB, G4X, EMB, V, accumulate=accumulate)
            ops.gemm_raw(ptr(Sl), 1, G4X, ptr(lab), EMB, 1, ptr(gW0, EMB), 2 * EMB, G4X, EMB, NC, accumulate=accumulate)
            ops.gemm_raw(ptr(S), G4X, 1, ptr(W0), 2 * EMB, 1, ptr(ge), EMB, V, EMB, G4X, accumulate=accumulate)
            ops.gemm_raw(ptr(Sl), G4X, 1, ptr(W0, EMB), 2 * EMB, 1, ptr(g["label_embedding.weight"]), EMB, NC, EMB, G4X,
                         accumulate=accumulate)
            if fan is not None:
                fan.join()
            return
        dx = self.stack.backward(dout, st["lstm"], B, T, st["lengths"], g, sc, accumulate, fan)
        dseq = None
        if st["soft"]:
            _tn(ptr(st["seq"]), self.V, self.V, ptr(dx), 2 * EMB, EMB, rows, ge, sc, accumulate)
            if want_dseq:       # d(seq) = d(x)[:, :32] E^T
                dseq = torch.empty(B, T, self.V, device=dev)
                ops.gemm_raw(ptr(dx), 2 * EMB, 1, ptr(p["sequence_embedding.embed_layer.weight"]), 1, EMB, ptr(dseq),
                             self.V, rows, self.V, EMB)
        else:
            lens = st["lengths"]
            if lens is None:
                lens = torch.full((B,), T, dtype=torch.int32, device=dev)
            part = sc.get(int(get().s2g_embed_grad_tokens_scratch(rows, self.V)))
            call("s2g_embed_grad_tokens", ptr(st["seq"]), ptr(lens), T, rows, self.V, ptr(dx), ptr(ge), int(accumulate),
                 ptr(part), stream_ptr())
        part = sc.get(B * 32)
        call("s2g_embed_grad_labels", ptr(st["labels"]), ptr(st["lengths"]), T, B, self.n_classes, ptr(dx),
             ptr(g["label_embedding.weight"]), int(accumulate), ptr(part), stream_ptr())
        if fan is not None:
            fan.join()
        return dseq

    def dropout_masks(self, rows: int):
        if self.dropout <= 0.0:
            return None
        return [ops.dropout_mask_rng(rows, self.dropout, self.device, self.rng, width=HID, live=HID)
                for _ in range(self.layers - 1)] + [None]

    # ------------------------------------------------------------------ the critic update
    def critic_step(self, real: torch.Tensor, labels: torch.Tensor, lengths: torch.Tensor, Tp: int, fake,
                    lr: float, b1: float, b2: float, clip: float, masks_real="draw", masks_fake="draw",
                    eps: float = 1e-8, update: bool = True, fake_len: Optional[int] = None) -> torch.Tensor:
        """wgan.py:174-198 for one batch: D(real tokens), D(fake soft sequences of length 100), d_loss =
        -(mean D(real) - mean D(fake)), Adam, clamp to [-clip, clip].  Returns d_loss (1-element device tensor);
        self.grads holds the gradients.  masks_*: lists of [rows, 64] dropout multipliers per layer gap,
        None (no dropout) or "draw" (device draws at self.dropout).
        fake: the generator's output [B, Tf, V] (detached, wgan.py:185), or a callable returning it together with
        fake_len = Tf: the generator's forward then runs on the second stream too, beside the real pass."""
        B = real.shape[0]
        dev = self.device
        make_fake = fake if callable(fake) else None
        if make_fake is not None:
            if fake_len is None:
                raise S2GError("critic_step: a callable `fake` needs fake_len")
            if not self.overlap:
                fake, make_fake = make_fake(), None
        Tf = fake_len if make_fake is not None else fake.shape[1]
        if masks_real == "draw":
            masks_real = self.dropout_masks(B * real.shape[1])
        if masks_fake == "draw":
            masks_fake = self.dropout_masks(B * Tf)
        dval = torch.empty(2 * B, device=dev)
        dD, loss = torch.empty(2 * B, device=dev), torch.empty(1, device=dev)
        if self.overlap:
            # The real pass is one workgroup per sequence for 1024 steps: half of the chip idles under it.  The fake
            # pass (100 steps) runs beside it on a second stream, forward and backward, with gradients and split-K
            # scratch of its own; the two gradient buffers are added at the end (one fixed order of additions).
            if self._side is None:
                self._side = (torch.cuda.Stream(device=dev), self.params.like(), ops.Scratch(dev),
                              [torch.cuda.Event() for _ in range(4)])
                self._fan = _Fan(self._side[0], self._side[2], dev)
            side, grads2, scratch2, (e_in, e_fwd, e_loss, e_bwd) = self._side
            cur = ops.cur_stream()
            ops.record(e_in)
            with ops.on_stream(side):
                side.wait_event(e_in)
                if make_fake is not None:
                    fake = make_fake()
                _, s_fake = self.forward(fake, labels, None, fake.shape[1], masks_fake, dval[B:])
                ops.record(e_fwd)
            _, s_real = self.forward(real, labels, lengths, Tp, masks_real, dval[:B])
            cur.wait_event(e_fwd)
            call("s2g_wasserstein_loss", ptr(dval), B, ptr(dD), ptr(loss), stream_ptr())
            ops.record(e_loss)
            with ops.on_stream(side):
                side.wait_event(e_loss)
                self.backward(s_fake, dD[B:], accumulate=False, grads=grads2, scratch=scratch2)
                ops.record(e_bwd)
            # the real pass's weight-gradient products follow the (short) fake pass on the second stream and on a third
            self.backward(s_real, dD[:B], accumulate=False, fan=self._fan)
            cur.wait_event(e_bwd)
            call("s2g_add_inplace", ptr(self.grads.flat), ptr(grads2.flat), self.grads.total, stream_ptr())
        else:
            _, s_real = self.forward(real, labels, lengths, Tp, masks_real, dval[:B])
            _, s_fake = self.forward(fake, labels, None, fake.shape[1], masks_fake, dval[B:])
            call("s2g_wasserstein_loss", ptr(dval), B, ptr(dD), ptr(loss), stream_ptr())
            self.backward(s_real, dD[:B], accumulate=False)
            self.backward(s_fake, dD[B:], accumulate=True)
        g = self.grads.views
        for l in range(self.layers):                    # b_ih and b_hh enter the gates as a sum: one gradient
            g[f"lstm.bias_hh_l{l}"].copy_(g[f"lstm.bias_ih_l{l}"])
        if update:
            self.step += 1
            ops.adam(self.params.flat, self.grads.flat, self.m.flat, self.v.flat, lr, b1, b2, eps, self.step)
            call("s2g_clamp", ptr(self.params.flat), self.params.total, float(clip), stream_ptr())
        self.last_values = dval
        return loss
