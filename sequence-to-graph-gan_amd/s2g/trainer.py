"""The fused WGAN-GP step (reference: train_wgan's inner loop, wgan_gp.py:274-343): critic
update on every batch, generator update when i % n_critic == 0 (after that batch's critic
update, same z and labels, fresh Gumbel noise).  All arithmetic runs in libs2g kernels; this
file only sequences launches, owns the Adam state and talks to the data-parallel communicator."""
from __future__ import annotations

import contextlib
import math
from typing import Dict, Optional

import torch

from . import ops
from .critic import CriticEngine, batch_desc, critic_forward_backward_fake, critic_iteration
from .generator import LATENT, GeneratorEngine, NoComm
from .lib import S2GError, call
from .ops import MASKW, H, ptr, stream_ptr

CLASS_W = (1.0, 2.0)      # wgan_gp.py:267


def class_weight_sum(labels_cpu: torch.Tensor) -> float:
    """sum_b w[y_b], the denominator of the weighted CrossEntropyLoss mean (wgan_gp.py:267-268)."""
    n1 = int((labels_cpu == 1).sum())
    return CLASS_W[0] * (labels_cpu.numel() - n1) + CLASS_W[1] * n1


def dropout_mask(rows: int, p: float, device, rng: ops.DeviceRng) -> Optional[torch.Tensor]:
    """[rows, 52] multiplier (0 or 1/(1-p)) for the layer-0 outputs, as nn.LSTM(dropout=p) applies
    in train mode (wgan_gp.py:189); columns 50,51 are padding."""
    if p <= 0.0:
        return None
    return ops.dropout_mask_rng(rows, p, device, rng)


_OWN_STREAMS: Dict[int, torch.cuda.Stream] = {}


class WGANGPTrainer:
    def __init__(self, vocab: int, gen_seq_len: int, device, lr: float = 2e-4, b1: float = 0.5, b2: float = 0.999,
                 lambda_gp: float = 10.0, tau: float = 0.5, hard: bool = False, n_critic: int = 5,
                 drop_rate: float = 0.3, comm=None, gen_engine=None, critic_engine=None, eps: float = 1e-8,
                 adam_g=None, seed: Optional[int] = None, dtype: str = "f32"):
        """lr, b1, b2, eps: Adam hyper-parameters of the critic's optimizer (wgan_gp.py:406) and, unless
        adam_g = (lr, b1, b2, eps) says otherwise, of the generator's (wgan_gp.py:405).
        seed: seed of the device-side draws (z, Gumbel uniforms, alpha, dropout masks) that are not injected."""
        self.device = torch.device(device)
        self.comm = comm or NoComm()
        self.G = gen_engine or GeneratorEngine(vocab, gen_seq_len, self.device, comm=self.comm)
        self.G.comm = self.comm
        self.D = critic_engine or CriticEngine(vocab, self.device)
        self.D.reserve_comm_stream = self.comm.split
        if dtype not in ("f32", "bf16"):
            raise S2GError("dtype must be 'f32' (exact fp32 MFMA everywhere) or 'bf16' (bf16 operands in the critic's "
                           "input-layer products, fp32 accumulation and fp32 recurrence)")
        self.D.bf16 = dtype == "bf16"
        self.D.repack()
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.adam_g = tuple(adam_g) if adam_g is not None else (lr, b1, b2, eps)
        self.lambda_gp, self.tau, self.hard, self.n_critic, self.drop_rate = lambda_gp, tau, hard, n_critic, drop_rate
        self.V, self.L = vocab, gen_seq_len
        rank = getattr(self.comm, "rank", 0)
        if seed is not None:
            # rank r draws its Gumbel uniforms, alpha and dropout masks from a key of its own (rank 0: `seed` itself)
            self.G.rng = ops.DeviceRng((int(seed) + rank * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        self.rng = self.G.rng           # one stream of draws: z, Gumbel uniforms, alpha, dropout masks
        # several ranks: z of the GLOBAL batch is drawn identically on every rank (the generator's small layers
        # run replicated on the global batch), from a key all ranks share: rank 0's, handed round once
        self.rng_shared = None
        if self.comm.split:
            base = int(seed) if seed is not None else torch.initial_seed()
            self.rng_shared = ops.DeviceRng(self.comm.broadcast_int(base, self.device) ^ 0x6A09E667F3BCC908)
        self._labels_global: Dict[int, tuple] = {}    # id(local labels) -> (local labels, global labels)
        self.mD_named, self.vD_named = self.D.params.like(), self.D.params.like()   # Adam moments (named views)
        self.mG_named, self.vG_named = self.G.params.like(), self.G.params.like()
        self.mD, self.vD = self.mD_named.flat, self.vD_named.flat
        self.mG, self.vG = self.mG_named.flat, self.vG_named.flat
        self.tD = 0
        self.tG = 0
        self._fake = [None, None]     # two persistent fake batches: the one in use and the prefetched next one
        self._fake_cur = 0
        self._prefetched = None       # dict(z, labels, done event) of the fake batch the last critic_step prepared
        self._gen_prefetched = None   # dict(z, labels, U, fake, done event) of a generator forward made ahead
        self._stream = None           # non-blocking stream full_step runs on (see full_step)
        self.fuse_update = True       # critic update as one launch when there is no all-reduce to wait for
        self._g_pending = None        # (done event, tensors kept alive) of a generator update on the side stream
        self._desc = [None, None]     # descriptor buffers of the batch in flight and of the next one
        self._desc_cur = 0
        self._g_scratch = ops.Scratch(self.device)
        self.last: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ updates
    def _adam(self, eng, m, v, step, lo: int = 0, hi: Optional[int] = None, reduce: bool = True):
        """Gradient all-reduce + Adam on [lo, hi) of the engine's flat buffers (everything by default).
        reduce=False: the gradients are global already (the generator's replicated small layers)."""
        hi = eng.params.total if hi is None else hi
        if hi <= lo:
            return
        g = eng.grads.flat[lo:hi]
        if reduce:
            self.comm.allreduce(g)
        lr, b1, b2, eps = self.adam_g if eng is self.G else (self.lr, self.b1, self.b2, self.eps)
        ops.adam(eng.params.flat[lo:hi], g, m[lo:hi], v[lo:hi], lr, b1, b2, eps, step)

    def _out_layer_update_gathered(self, dl2: torch.Tensor, a3: torch.Tensor, update: bool, pieces: int = 4) -> None:
        """Several ranks: gradients (and, with `update`, the Adam step) of the generator's output Linear from the
        factors of the GLOBAL batch -- dl2 [B, L*V] of this rank, a3 [world * B, 512] of all.  The columns of dl2
        travel in `pieces` all-gathers started back to back; the product, the bias sums and the Adam step of one
        piece (a block of rows of W) run while the next piece is still on the links."""
        G = self.G
        NV, K = dl2.shape[1], a3.shape[1]
        ow, nw, _ = G.params.offsets["model.8.weight"]
        ob, nb, _ = G.params.offsets["model.8.bias"]
        gW, gb = G.grads.views["model.8.weight"], G.grads.views["model.8.bias"]
        step = (NV + pieces - 1) // pieces
        step = (step + 3) // 4 * 4
        cuts = [(c, min(c + step, NV)) for c in range(0, NV, step)]
        started = [self.comm.allgather_rows_start(dl2[:, c0:c1]) for c0, c1 in cuts]
        for (c0, c1), (rows, wait) in zip(cuts, started):
            wait()
            ops.linear_bwd_dw(rows, a3, gW[c0:c1], self._g_scratch)
            ops.colsum(rows, rows.shape[0], c1 - c0, c1 - c0, gb[c0:c1], self._g_scratch)
            if update:
                self._adam(G, self.mG, self.vG, self.tG, ow + c0 * K, ow + c1 * K, reduce=False)
        if update:
            self._adam(G, self.mG, self.vG, self.tG, ob, ob + (nb + 3) // 4 * 4, reduce=False)

    def _gen_inputs(self, z: Optional[torch.Tensor], labels: torch.Tensor, B: int):
        """(noise, labels) as GeneratorEngine.forward takes them.  One process: this batch's z (drawn here when
        None) and labels.  Several ranks: the GLOBAL batch's -- z drawn from the key all ranks share (an
        injected shard is gathered instead: tests), labels gathered once per labels tensor and remembered."""
        dev = self.device
        if not self.comm.split:
            if z is None:
                z = ops.rng_fill(torch.empty(B, LATENT, device=dev), 1, self.rng)
            return z, labels
        hit = self._labels_global.get(id(labels))
        if hit is None or hit[0] is not labels:
            while len(self._labels_global) >= 256:
                self._labels_global.pop(next(iter(self._labels_global)))
            hit = self._labels_global[id(labels)] = (labels, self.comm.allgather_rows(labels))
        if z is None:
            z = ops.rng_fill(torch.empty(B * self.comm.world, LATENT, device=dev), 1, self.rng_shared)
        else:
            z = self.comm.allgather_rows(z)
        return z, hit[1]

    def _fake_buf(self, which: int, B: int) -> torch.Tensor:
        # persistent output buffers: stable pointers let the grouped-GEMM problem table be reused
        if self._fake[which] is None or self._fake[which].shape[0] != B:
            self._fake[which] = torch.empty(B, self.L, self.V, device=self.device)
        return self._fake[which]

    @contextlib.contextmanager
    def _own_stream(self):
        """Moves the enclosed launches onto a non-blocking stream of the trainer when the caller sits on
        the NULL stream: the critic's CU-masked second stream is a blocking stream and would otherwise
        synchronise implicitly with every launch there.  The caller's stream waits for the work on exit."""
        cur = ops.cur_stream()
        if not (self.D.overlap and cur.cuda_stream == 0):
            yield
            return
        if self._stream is None:
            # one non-blocking stream per device for every trainer of the process (four hardware queues in all)
            idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
            if idx not in _OWN_STREAMS:
                _OWN_STREAMS[idx] = torch.cuda.Stream(device=self.device)
            self._stream = _OWN_STREAMS[idx]
        own = self._stream
        own.wait_stream(cur)
        with ops.on_stream(own):
            yield
        cur.wait_stream(own)

    def stream(self):
        """Context for a caller's whole loop: everything enqueued inside (the steps, but also the caller's own
        host-to-device copies) goes to the trainer's non-blocking stream.  A copy issued on the NULL stream
        would wait for the CU-masked streams (blocking streams) to drain, i.e. stall the host behind the step
        that is still running."""
        return self._own_stream()

    def critic_step(self, *args, **kwargs):
        """One critic update (wgan_gp.py:277-318); arguments of _critic_step."""
        with self._own_stream():
            return self._critic_step(*args, **kwargs)

    def generator_step(self, *args, **kwargs):
        """One generator update (wgan_gp.py:330-343); arguments of _generator_step."""
        with self._own_stream():
            return self._generator_step(*args, **kwargs)

    def _critic_step(self, tokens: torch.Tensor, labels: torch.Tensor, lengths: torch.Tensor, wsum: float,
                    z: Optional[torch.Tensor] = None, U: Optional[torch.Tensor] = None,
                    alpha: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                    update: bool = True, next_batch=None, gen_next=None):
        """tokens int32 [B,T], labels int64 [B], lengths int32 [B]: device tensors of this rank's shard.
        wsum: global sum of class weights.  Noise defaults to fresh device draws.
        next_batch: optional dict(labels=..., z=..., U=...) of the NEXT critic_step (same batch size,
        generator unchanged in between): its fake batch is generated on the second stream beside this
        step's second-order sweeps and tail, and the next call picks it up.
        gen_next: optional dict(U=..., mask=...) announcing that generator_step follows with this call's z and
        labels (wgan_gp.py:323-333): its generator forward depends on the generator only, so it runs beside
        this step's sweeps as well and generator_step picks it up.
        Returns the loss parts (device tensors; nothing synchronises)."""
        B, T = tokens.shape
        dev = self.device
        n_global = B * self.comm.world
        desc = None
        pre = self._prefetched
        self._prefetched = None
        if pre is not None and pre["labels"] is labels and (z is None or z is pre["z_in"]):
            self._join_generator()
            zg, lg, fake = pre["z"], pre["labels_gen"], self._fake_buf(self._fake_cur, B)
            ops.cur_stream().wait_event(pre["done"])
            if alpha is None:
                alpha = pre.get("alpha")
                # the descriptors were built with or without dropout-mask rows: reuse them only if that
                # matches whether a mask (the caller's or the prefetched one) is used now
                if pre.get("desc_key") == (B, T, self.L, mask is not None or pre.get("mask") is not None):
                    desc = pre["desc"]
            if mask is None and pre.get("mask") is not None and pre["mask"].shape[0] == 3 * B * max(T, self.L):
                mask = pre["mask"]
        else:
            if pre is not None:
                ops.cur_stream().wait_event(pre["done"])     # keep the generator's buffer updates ordered
            zg, lg = self._gen_inputs(z, labels, B)      # what the generator takes (the global batch's with several ranks)
            # this batch's own draws and sequence descriptors BEFORE the generator forward: after a generator
            # update that forward waits (just before its last layer) for the output layer's update on the side
            # stream, and whatever is enqueued ahead of the wait runs meanwhile instead of after it
            if alpha is None:
                alpha = ops.rng_fill(torch.empty(B, device=dev), 0, self.rng)
            if mask is None and self.drop_rate > 0.0:
                mask = dropout_mask(3 * B * max(T, self.L), self.drop_rate, dev, self.rng)
            self._desc_cur ^= 1
            if self._desc[self._desc_cur] is None or self._desc[self._desc_cur].numel() < 4 * B * 8:
                self._desc[self._desc_cur] = torch.zeros(4 * B, 8, dtype=torch.int32, device=dev)
            desc = self._desc[self._desc_cur]
            batch_desc(lengths, alpha.reshape(-1), B, T, self.L, mask is not None, desc)
            fake = self.G.forward(zg, lg, U, self.tau, self.hard, training=True, save=False,
                                  out=self._fake_buf(self._fake_cur, B))    # .detach(): no graph is kept
            self._join_generator(waited=True)     # the forward waited for a pending output-layer update
        if alpha is None:
            alpha = ops.rng_fill(torch.empty(B, device=dev), 0, self.rng)
        Tw = max(T, self.L)
        if mask is None and self.drop_rate > 0.0:
            mask = dropout_mask(3 * B * Tw, self.drop_rate, dev, self.rng)
        prefetch = None
        # (with several ranks the prefetched forwards issue their BatchNorm all-reduces from the second stream;
        # every rank runs the same host code, so all ranks issue their collectives in the same order)
        if next_batch is not None and self.D.overlap:
            nxt = dict(labels=next_batch["labels"], z_in=next_batch.get("z"), done=torch.cuda.Event())
            other = self._fake_buf(self._fake_cur ^ 1, B)

            def prefetch(heavy_after=None):
                # (z drawn on the prefetch stream: nothing of it sits on the caller's)
                nxt["z"], nxt["labels_gen"] = self._gen_inputs(nxt["z_in"], nxt["labels"], B)
                self.G.forward(nxt["z"], nxt["labels_gen"], next_batch.get("U"), self.tau, self.hard, training=True,
                               save=False, out=other, heavy_after=heavy_after, workgroups=self.D.cu_masked)
                ops.record(nxt["done"])

            def idle():
                # the next step's own draws, made while this stream waits for the second-order sweeps
                if next_batch.get("alpha") is None:
                    nxt["alpha"] = ops.rng_fill(torch.empty(B, device=dev), 0, self.rng)
                if next_batch.get("mask") is None and self.drop_rate > 0.0 and next_batch.get("T"):
                    nxt["mask"] = dropout_mask(3 * B * max(next_batch["T"], self.L), self.drop_rate, dev, self.rng)
                if nxt.get("alpha") is not None and next_batch.get("lengths") is not None and next_batch.get("T"):
                    # the next batch's sequence descriptors too (two buffers: this batch's are still in use)
                    self._desc_cur ^= 1
                    if self._desc[self._desc_cur] is None or self._desc[self._desc_cur].numel() < 4 * B * 8:
                        self._desc[self._desc_cur] = torch.zeros(4 * B, 8, dtype=torch.int32, device=dev)
                    use_mask = next_batch.get("mask") is not None or nxt.get("mask") is not None
                    batch_desc(next_batch["lengths"], nxt["alpha"].reshape(-1), B, next_batch["T"], self.L, use_mask,
                               self._desc[self._desc_cur])
                    nxt["desc"] = self._desc[self._desc_cur]
                    nxt["desc_key"] = (B, next_batch["T"], self.L, use_mask)
        elif gen_next is not None and self.D.overlap:
            gp = dict(z=zg, labels=labels, U=gen_next.get("U"), done=torch.cuda.Event())

            def prefetch(heavy_after=None):
                gp["fake"] = self.G.forward(zg, lg, gp["U"], self.tau, self.hard, training=True, save=True,
                                            heavy_after=heavy_after, workgroups=self.D.cu_masked)
                ops.record(gp["done"])

            def idle():
                if gen_next.get("mask") is None and self.drop_rate > 0.0:
                    gp["mask"] = dropout_mask(B * self.L, self.drop_rate, dev, self.rng)
        else:
            idle = None
        # one process: the final slab reduction, Adam and the weight repack run as one launch
        fused = update and self.fuse_update and not self.comm.split
        losses, pen = critic_iteration(self.D, tokens, lengths, labels, fake, alpha.reshape(-1).contiguous(),
                                       self.lambda_gp, n_global, 1.0 / wsum, mask, CLASS_W, prefetch=prefetch,
                                       idle=idle, desc=desc, defer_reduce=fused)
        if prefetch is not None and next_batch is not None:
            self._prefetched = nxt
            self._fake_cur ^= 1
        elif prefetch is not None:
            self._gen_prefetched = gp
        if update:
            self.tD += 1
            if fused:
                self.D.fused_update(self.mD, self.vD, self.lr, self.b1, self.b2, self.eps, self.tD)
            else:
                self._adam(self.D, self.mD, self.vD, self.tD)
                self.D.repack()
        # z / labels_gen: the generator's inputs of this batch (the global batch's with several ranks)
        self.last = dict(z=zg, labels=labels, labels_gen=lg, d_parts=losses, pen=pen, n_global=n_global, wsum=wsum)
        return losses, pen

    def d_loss(self) -> torch.Tensor:
        """d_loss of the last critic_step as a 0-dim device tensor (wgan_gp.py:313-315)."""
        l, pen = self.last["d_parts"], self.last["pen"]
        t = (l[0] + l[1] + l[2] + self.lambda_gp * pen.sum() / self.last["n_global"]).reshape(1).clone()
        self.comm.allreduce(t)
        return t[0]

    def _generator_step(self, labels: Optional[torch.Tensor] = None, z: Optional[torch.Tensor] = None,
                       U: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                       wsum: Optional[float] = None, update: bool = True, defer_join: bool = False,
                       keep_grads: bool = True):
        """wgan_gp.py:330-343 with the z / labels of the preceding critic_step by default.
        defer_join: return while the output Linear's update may still run on the side stream; the next
        generator forward waits for it just before that layer (full_step uses this).
        keep_grads: False lets the output Linear's weight gradient stay in registers (one process: its
        product and its Adam step are one kernel); G.grads then holds every gradient but that one."""
        if labels is None and z is None:
            labels, z, lg = self.last["labels"], self.last["z"], self.last["labels_gen"]
        else:
            labels = self.last["labels"] if labels is None else labels
            if z is None and not self.comm.split:
                z = self.last["z"]
            elif z is None:
                raise S2GError("generator_step: with several ranks pass z together with labels (or neither)")
            z, lg = self._gen_inputs(z, labels, labels.shape[0])
        wsum = self.last["wsum"] if wsum is None else wsum
        if self._prefetched is not None:          # a prefetched forward nobody consumed: stay ordered behind it
            ops.cur_stream().wait_event(self._prefetched["done"])
            self._prefetched = None
        B = labels.shape[0]
        n_global = B * self.comm.world
        gp, self._gen_prefetched = self._gen_prefetched, None
        if gp is not None:
            ops.cur_stream().wait_event(gp["done"])
        if gp is not None and gp["z"] is z and gp["labels"] is labels and gp["U"] is U and "fake" in gp:
            fake = gp["fake"]                     # forward (and its saved activations) made beside the critic step
            if mask is None:
                mask = gp.get("mask")
        else:
            fake = self.G.forward(z, lg, U, self.tau, self.hard, training=True, save=True)
        if mask is None and self.drop_rate > 0.0:
            mask = dropout_mask(B * self.L, self.drop_rate, self.device, self.rng)
        losses, dfake = critic_forward_backward_fake(self.D, fake, labels, n_global, 1.0 / wsum, mask, CLASS_W)
        side = self.D.side() if (update and self.D.overlap) else None
        lo, hi = self.G.last_layer_range()
        # several ranks: every generator gradient is global and identical on every rank already -- the small layers'
        # (computed on the gathered global batch) and the output Linear's (formed from the gathered rows of dl2)
        small_reduce = not self.G.small_layers_replicated()
        out_reduce = not self.G.output_layer_factors_gathered()
        # one process and a batch the kernel takes: the output Linear's weight gradient and its Adam step in ONE
        # pass over W, m, v (6 passes over 184 MB instead of 10); the kernel takes up to 64 rows, so with several
        # ranks (the rows of the GLOBAL batch) the product and the flat Adam stay separate
        fused_out = update and not self.comm.split and ops.linear_wgrad_adam_ok(B, self.G.params.views["model.8.weight"].shape[1])

        def last_layer_update(dl2, a3, workgroups=0, gather=False):
            """gradients and Adam step of the output Linear (weight and bias) on the current stream;
            gather: the rows of dl2 are all-gathered over the ranks first (a3 is the global batch's then)"""
            if gather:
                self._out_layer_update_gathered(dl2, a3, update)
                return
            if not fused_out:
                self.G.last_layer_grads(dl2, a3, self._g_scratch)
                self._adam(self.G, self.mG, self.vG, self.tG, lo, hi, reduce=out_reduce)
                return
            ow, nw, _ = self.G.params.offsets["model.8.weight"]
            lr, b1, b2, eps = self.adam_g
            ops.linear_wgrad_adam(dl2, a3, self.G.params.flat[ow:ow + nw], self.mG[ow:ow + nw], self.vG[ow:ow + nw],
                                  lr, b1, b2, eps, self.tG,
                                  grad=self.G.grads.flat[ow:ow + nw] if keep_grads else None, workgroups=workgroups)
            self.G.last_layer_bias_grad(dl2, self._g_scratch)
            self._adam(self.G, self.mG, self.vG, self.tG, ow + nw, hi)

        if side is None:
            if update:
                self.tG += 1
                self.G.backward(dfake, last_layer=lambda dl2, a3, gather=False: last_layer_update(dl2, a3, 0, gather))
                self._adam(self.G, self.mG, self.vG, self.tG, 0, lo, reduce=small_reduce)
                self._adam(self.G, self.mG, self.vG, self.tG, hi, self.G.params.total, reduce=small_reduce)
            else:
                self.G.backward(dfake)
        else:
            # The output Linear holds 99.6 % of the parameters: its weight gradient and Adam update are
            # HBM-bound and independent of the rest of the backward (a chain of small kernels), so they
            # run beside it on the critic's CU-masked stream (idle now), which leaves a quarter of the
            # CUs to the chain: the grid-stride Adam would otherwise hold every wave slot of the chip.
            self._join_generator()
            self.tG += 1
            stream_s = side[0]
            done, keep = torch.cuda.Event(), []

            def last_layer(dl2, a3, gather=False):
                ready = torch.cuda.Event()
                ops.record(ready)
                keep.extend((dl2, a3))        # allocated on this stream, read on the other: alive until the join
                with ops.on_stream(stream_s):
                    stream_s.wait_event(ready)
                    last_layer_update(dl2, a3, self.D.cu_masked, gather)
                    ops.record(done)

            self.G.backward(dfake, last_layer=last_layer)
            self._adam(self.G, self.mG, self.vG, self.tG, 0, lo, reduce=small_reduce)
            self._adam(self.G, self.mG, self.vG, self.tG, hi, self.G.params.total, reduce=small_reduce)
            self.G.last_layer_ready = done
            self._g_pending = (done, keep)
            if not defer_join:
                self._join_generator()
        self.last["g_parts"] = losses
        return losses

    def join_generator(self) -> None:
        """Order the trainer's stream behind a generator update that generator_step(defer_join=True) left running
        on the side stream (before anything else reads the generator's parameters or Adam state)."""
        with self._own_stream():
            self._join_generator()

    def _join_generator(self, waited: bool = False) -> None:
        """Orders the current stream behind a generator update still running on the side stream
        (waited: the stream already waited for it, only the bookkeeping is left)."""
        if self._g_pending is not None:
            if not waited:
                ops.cur_stream().wait_event(self._g_pending[0])
            self._g_pending = None
            self.G.last_layer_ready = None

    def g_loss(self) -> torch.Tensor:
        l = self.last["g_parts"]
        t = (l[0] + l[1]).reshape(1).clone()
        self.comm.allreduce(t)
        return t[0]

    def losses(self) -> torch.Tensor:
        """(d_loss, g_loss) of the last critic / generator step as one 2-element device tensor: the values
        train_wgan prints (wgan_gp.py:345-348), summed over the ranks with ONE all-reduce."""
        l, pen, gl = self.last["d_parts"], self.last["pen"], self.last["g_parts"]
        t = torch.stack((l[0] + l[1] + l[2] + self.lambda_gp * pen.sum() / self.last["n_global"], gl[0] + gl[1]))
        self.comm.allreduce(t)
        return t

    # ------------------------------------------------------------------ the full step
    def full_step(self, batches, noises=None, lookahead=None, lookahead_noise=None):
        """batches: n_critic device batches (tokens i32, labels i64, lengths i32, wsum).  Generator
        update after the first batch's critic update (i % n_critic == 0, wgan_gp.py:323).

        The step runs on a non-blocking stream of the trainer (the caller's stream waits for it on
        return): the critic's CU-masked second stream is a blocking stream and would otherwise
        synchronise with every launch when the caller sits on the NULL stream.  While the generator
        stays fixed (between two generator updates) the next batch's fake sequences are prefetched
        beside the current batch's second-order sweeps.
        lookahead: the batch the NEXT call starts with, if the caller's loader can peek (the generator
        does not change between this call's last batch and that one, so its fake sequences are
        prefetched too); lookahead_noise: its injected noise dict, as in `noises`."""
        with self._own_stream():
            n = len(batches)
            for i, (tokens, labels, lengths, wsum) in enumerate(batches):
                nz = noises[i] if noises is not None else {}
                nxt = None
                nb = batches[i + 1] if i + 1 < n else lookahead
                if nb is not None and i % self.n_critic != 0 and nb[0].shape[0] == tokens.shape[0]:
                    nn_ = (noises[i + 1] if noises is not None else {}) if i + 1 < n else (lookahead_noise or {})
                    nxt = dict(labels=nb[1], z=nn_.get("z"), U=nn_.get("U"), alpha=nn_.get("alpha"),
                               mask=nn_.get("mask"), T=nb[0].shape[1], lengths=nb[2])
                gen = dict(U=nz.get("U2"), mask=nz.get("mask_g")) if i % self.n_critic == 0 else None
                self._critic_step(tokens, labels, lengths, wsum, nz.get("z"), nz.get("U"), nz.get("alpha"),
                                  nz.get("mask"), next_batch=nxt, gen_next=gen)
                if i % self.n_critic == 0:
                    self._generator_step(U=nz.get("U2"), mask=nz.get("mask_g"), defer_join=i + 1 < n, keep_grads=False)
            self._join_generator()
