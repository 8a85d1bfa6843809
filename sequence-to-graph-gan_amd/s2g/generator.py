"""Host orchestration of the generator (reference: Generator, wgan_gp.py:141-171): label
embedding product -> Linear+LeakyReLU -> 2 x (Linear, BatchNorm1d(eps=0.8), LeakyReLU) ->
Linear(512, V*seq_len) -> Gumbel-softmax.  Forward and hand-derived backward, every product an
fp32 MFMA GEMM of libs2g; parameters live in one flat buffer (one fused Adam launch, one
all-reduce bucket)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .critic import FlatBuffer
from .lib import S2GError, call
from .ops import ptr, stream_ptr

LATENT = 100
BN_EPS = 0.8          # nn.BatchNorm1d(out_feat, 0.8): the second positional argument is eps
BN_MOMENTUM = 0.1


def generator_shapes(vocab: int, seq_len: int, n_classes: int = 2):
    """state_dict order of the reference Generator (parameters only)."""
    return [("model.0.weight", (128, LATENT)), ("model.0.bias", (128,)),
            ("model.2.weight", (256, 128)), ("model.2.bias", (256,)),
            ("model.3.weight", (256,)), ("model.3.bias", (256,)),
            ("model.5.weight", (512, 256)), ("model.5.bias", (512,)),
            ("model.6.weight", (512,)), ("model.6.bias", (512,)),
            ("model.8.weight", (vocab * seq_len, 512)), ("model.8.bias", (vocab * seq_len,)),
            ("label_embedding.weight", (n_classes, LATENT))]


class NoComm:
    world = 1
    rank = 0
    split = False          # True: take the multi-rank code paths (replicated small layers, all-reduced buckets)

    def allreduce(self, t: torch.Tensor) -> None:  # sum over ranks, in place
        return

    def allgather_rows(self, t: torch.Tensor) -> torch.Tensor:
        return t


class GeneratorEngine:
    def __init__(self, vocab: int, seq_len: int, device, n_classes: int = 2, comm=None):
        self.V, self.L, self.n_classes = vocab, seq_len, n_classes
        self.device = device
        self.comm = comm or NoComm()
        self.params = FlatBuffer(generator_shapes(vocab, seq_len, n_classes), device)
        self.grads = self.params.like()
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=device)  # noqa: E731
        self.buffers = {"model.3.running_mean": z(256), "model.3.running_var": torch.ones(256, device=device),
                        "model.6.running_mean": z(512), "model.6.running_var": torch.ones(512, device=device)}
        self.num_batches_tracked = 0
        self.scratch = ops.Scratch(device)
        self.saved: Dict[str, torch.Tensor] = {}
        self.last_layer_ready: Optional[torch.cuda.Event] = None   # set while that layer's update is in flight
        self.rng = ops.DeviceRng()      # Gumbel uniforms of forwards that are not handed a tensor of draws

    def load(self, state: Dict[str, torch.Tensor]) -> None:
        for k, v in self.params.views.items():
            v.copy_(state[k].to(self.device, torch.float32))
        for k in self.buffers:
            if k in state:
                self.buffers[k].copy_(state[k].to(self.device, torch.float32))

    # ------------------------------------------------------------------ forward
    def _bn(self, pre, name, training, update_running, B, F_):
        """BatchNorm1d(eps=0.8) + LeakyReLU on all B rows of `pre`.  Training: batch statistics + normalise in
        one launch (with several ranks `pre` holds the GLOBAL batch, so the statistics are exact without an
        exchange); eval: the running statistics."""
        p = self.params.views
        y = torch.empty_like(pre)
        xhat = torch.empty_like(pre) if training else None
        invstd = torch.empty(F_, device=self.device) if training else None
        if training:
            call("s2g_bn_lrelu_fwd_fused", ptr(pre), ptr(p[name + ".weight"]), ptr(p[name + ".bias"]),
                 ptr(self.buffers[name + ".running_mean"]), ptr(self.buffers[name + ".running_var"]), BN_EPS,
                 BN_MOMENTUM, int(update_running), ptr(y), ptr(xhat), ptr(invstd), B, F_, stream_ptr())
            return y, xhat, invstd
        call("s2g_bn_lrelu_apply", ptr(pre), None, float(B), ptr(p[name + ".weight"]), ptr(p[name + ".bias"]),
             ptr(self.buffers[name + ".running_mean"]), ptr(self.buffers[name + ".running_var"]), BN_EPS,
             BN_MOMENTUM, 0, ptr(y), None, None, B, F_, stream_ptr())
        return y, xhat, invstd

    def shard(self, rows: int):
        """(first row, rows) of this rank's samples within the `rows` the small layers ran on: everything for
        one process; with several ranks the small layers take the global batch and the output Linear, the
        sampler and everything behind them this rank's contiguous share of it."""
        if not self.comm.split:
            return 0, rows
        if rows % self.comm.world:
            raise S2GError("generator: the global batch must divide evenly over the ranks")
        per = rows // self.comm.world
        return self.comm.rank * per, per

    def forward(self, z: torch.Tensor, labels: torch.Tensor, U: Optional[torch.Tensor], tau: float, hard: bool,
                training: bool = True, save: bool = True, act: str = "gumbel_softmax",
                out: Optional[torch.Tensor] = None, heavy_after=None, workgroups: int = 0) -> torch.Tensor:
        """heavy_after: optional callable run between the small layers and the output Linear; a caller
        that runs this forward beside other work makes the current stream wait for an event in it, so
        that the HBM-bound part (183 MB of weights) does not collide with its own heavy kernels.
        workgroups: CUs the current stream can reach when it is CU-masked (sizes the output layer's grid).
        With several ranks z and labels are the GLOBAL batch's (identical on every rank; the small layers run
        on all of it), while U, out and the returned sample are this rank's share (see shard())."""
        ops._chk(z, name="noise")
        ops._chk(labels, torch.int64, "labels")
        Bt = z.shape[0]
        if z.shape[1] != LATENT or labels.shape[0] != Bt:
            raise S2GError("generator: noise must be [B,100] and labels [B]")
        lo, B = self.shard(Bt)
        p = self.params.views
        dev = self.device
        x0 = torch.empty(Bt, LATENT, device=dev)
        call("s2g_embed_mul", ptr(z), ptr(labels), ptr(p["label_embedding.weight"]), ptr(x0), Bt, LATENT,
             self.n_classes, stream_ptr())
        a1 = torch.empty(Bt, 128, device=dev)
        ops.linear_fwd(x0, p["model.0.weight"], p["model.0.bias"], a1, act=1)
        pre2 = torch.empty(Bt, 256, device=dev)
        ops.linear_fwd(a1, p["model.2.weight"], p["model.2.bias"], pre2)
        a2, xhat2, inv2 = self._bn(pre2, "model.3", training, True, Bt, 256)
        pre3 = torch.empty(Bt, 512, device=dev)
        ops.linear_fwd(a2, p["model.5.weight"], p["model.5.bias"], pre3)
        a3_all, xhat3, inv3 = self._bn(pre3, "model.6", training, True, Bt, 512)
        a3 = a3_all[lo:lo + B]                 # this rank's rows (a contiguous view)
        if training:
            self.num_batches_tracked += 1
        logits = torch.empty(B, self.L * self.V, device=dev)
        mode = ops.MODES.get(act, 2)
        if heavy_after is not None:
            heavy_after()
        if self.last_layer_ready is not None:
            # the output Linear's update may still be running on another stream (trainer._generator_step)
            ops.cur_stream().wait_event(self.last_layer_ready)
        ops.linear_fwd_small_batch(a3, p["model.8.weight"], p["model.8.bias"], logits, workgroups)
        # (U None in the Gumbel mode: the sampler draws its uniforms itself, wgan_gp.py:58)
        y, y_soft = ops.gumbel_softmax_fwd(logits.view(B, self.L, self.V), U if mode == 0 else None, tau, mode,
                                           hard, out=out, rng=self.rng)
        if save:
            self.saved = dict(z=z, labels=labels, x0=x0, a1=a1, a2=a2, xhat2=xhat2, inv2=inv2, a3=a3, a3_all=a3_all,
                              xhat3=xhat3, inv3=inv3, y_soft=y_soft, scale=(1.0 if mode == 2 else 1.0 / tau), B=B,
                              Bt=Bt)
        return y

    def hidden(self, z: torch.Tensor, labels: torch.Tensor, training: bool) -> torch.Tensor:
        """The activations in front of the output Linear (wgan_gp.py:166-167, model[0..7]); with `training`
        the BatchNorm layers use (and update) batch statistics, as Generator.forward in train mode."""
        B = z.shape[0]
        p, dev = self.params.views, self.device
        x0 = torch.empty(B, LATENT, device=dev)
        call("s2g_embed_mul", ptr(z), ptr(labels), ptr(p["label_embedding.weight"]), ptr(x0), B, LATENT,
             self.n_classes, stream_ptr())
        a1 = torch.empty(B, 128, device=dev)
        ops.linear_fwd(x0, p["model.0.weight"], p["model.0.bias"], a1, act=1)
        pre2 = torch.empty(B, 256, device=dev)
        ops.linear_fwd(a1, p["model.2.weight"], p["model.2.bias"], pre2)
        a2, _, _ = self._bn(pre2, "model.3", training, True, B, 256)
        pre3 = torch.empty(B, 512, device=dev)
        ops.linear_fwd(a2, p["model.5.weight"], p["model.5.bias"], pre3)
        a3, _, _ = self._bn(pre3, "model.6", training, True, B, 512)
        if training:
            self.num_batches_tracked += 1
        return a3

    def generate_tokens(self, z: torch.Tensor, labels: torch.Tensor, U: Optional[torch.Tensor] = None,
                        training: bool = False) -> torch.Tensor:
        """generator(noise, labels, tau, hard).argmax(-1) of generate_fake_data.py:64-66 as int64 [B, L], with
        the output layer, the Gumbel perturbation and the argmax in one kernel (no [B, L, V] tensor exists).
        Equal to the argmax of forward()'s sample for every tau and for hard and soft sampling.
        U [B, L, V]: injected uniforms; None: drawn in the kernel."""
        ops._chk(z, name="noise")
        ops._chk(labels, torch.int64, "labels")
        B = z.shape[0]
        if z.shape[1] != LATENT or labels.shape[0] != B:
            raise S2GError("generator: noise must be [B,100] and labels [B]")
        if U is not None:
            ops._chk(U, name="U")
            if tuple(U.shape) != (B, self.L, self.V):
                raise S2GError("generate_tokens: U must be [B, seq_len, vocab]")
        p = self.params.views
        tokens = torch.empty(B, self.L, dtype=torch.int64, device=self.device)
        if training:
            a3 = self.hidden(z, labels, True)          # batch statistics couple the samples: one pass
        for lo in range(0, B, 64):                     # the kernel takes up to 64 samples per launch
            hi = min(B, lo + 64)
            a = a3[lo:hi] if training else self.hidden(z[lo:hi].contiguous(), labels[lo:hi].contiguous(), False)
            seed, off = (0, 0) if U is not None else self.rng.take()
            call("s2g_generate_tokens", ptr(p["model.8.weight"]), ptr(a.contiguous()), ptr(p["model.8.bias"]),
                 ptr(U[lo:hi].contiguous()) if U is not None else None, ptr(tokens, lo * self.L), self.L, self.V,
                 hi - lo, 512, seed, off, stream_ptr())
        return tokens

    # ------------------------------------------------------------------ backward
    def _bn_bwd(self, a, da, xhat, inv, name, B, F_):
        g = self.grads.views
        dpre = torch.empty_like(da)
        call("s2g_bn_lrelu_bwd_fused", ptr(a), ptr(da), ptr(xhat), ptr(self.params.views[name + ".weight"]),
             ptr(inv), ptr(g[name + ".weight"]), ptr(g[name + ".bias"]), ptr(dpre), B, F_, stream_ptr())
        return dpre

    def last_layer_grads(self, dl2: torch.Tensor, a3: torch.Tensor, scratch: ops.Scratch) -> None:
        """Gradients of the output Linear (99.6 % of the generator's parameters) on the current stream."""
        g = self.grads.views
        ops.linear_bwd_dw(dl2, a3, g["model.8.weight"], scratch)
        ops.colsum(dl2, dl2.shape[0], self.L * self.V, self.L * self.V, g["model.8.bias"], scratch)

    def last_layer_bias_grad(self, dl2: torch.Tensor, scratch: ops.Scratch) -> None:
        """Bias gradient of the output Linear alone (its weight gradient is consumed inside the fused update)."""
        ops.colsum(dl2, dl2.shape[0], self.L * self.V, self.L * self.V, self.grads.views["model.8.bias"], scratch)

    def small_layers_replicated(self) -> bool:
        """True when the small layers run on the global batch on every rank (several ranks): their gradients
        come out global and identical everywhere and are not summed over the ranks."""
        return bool(self.comm.split)

    def output_layer_factors_gathered(self) -> bool:
        """True when backward() forms the output Linear's gradients from the GLOBAL batch's factors (several
        ranks): dW = dl2^T a3 has rank <= batch, so the ranks exchange their rows of dl2 [B, L*V] (one all-gather,
        world * B * L*V floats received) instead of summing the 512 x L*V products (an all-reduce of the whole
        matrix: 2 (world - 1) / world of 184 MB per rank at L*V = 89 600); a3 of the global batch is on every rank
        already.  The gradient comes out global and bit-identical everywhere."""
        return bool(self.comm.split)

    def last_layer_range(self):
        """[lo, hi) of the output Linear's weight and bias in the flat parameter buffer (adjacent views)."""
        lo = self.params.offsets["model.8.weight"][0]
        ob, nb, _ = self.params.offsets["model.8.bias"]
        return lo, ob + (nb + 3) // 4 * 4

    def backward(self, dy: torch.Tensor, last_layer=None) -> None:
        """Fills self.grads from d loss/d y (this rank's samples, losses normalised by the GLOBAL batch).
        One process: the gradient.  Several ranks: d(a3) of all ranks is gathered and the small layers' backward
        runs on the global batch; the output Linear's gradients are formed from the gathered rows of dl2 and the
        global a3 (output_layer_factors_gathered()).  Every entry is then the global gradient, identical on every
        rank, and must not be summed over the ranks again.

        last_layer: optional callable(dl2, a3, gather) that takes over the output Linear's gradients (the
        trainer runs them, and that layer's update, on a second stream beside the rest of this
        backward: a chain of small launch-bound kernels); it must keep dl2 alive until its work is
        ordered before the caller's stream again.  gather (bool; several ranks): the callee all-gathers the rows of
        dl2 over the ranks on ITS stream before it forms the gradients -- a3 is the global batch's then.  The gather
        of d(a3), which the rest of this backward waits for, is issued first: collectives of one communicator run
        in issue order, and the small message must not queue behind the large one."""
        s = self.saved
        if not s:
            raise S2GError("generator.backward called without a saved forward")
        B, Bl = s["Bt"], s["B"]           # rows of the small layers (the global batch with several ranks), of this rank
        p, g, sc, dev = self.params.views, self.grads.views, self.scratch, self.device
        dlogits = ops.gumbel_softmax_bwd(s["y_soft"], dy.contiguous().view_as(s["y_soft"]), s["scale"])
        dl2 = dlogits.view(Bl, self.L * self.V)
        da3 = torch.empty(Bl, 512, device=dev)
        ops.linear_bwd_dx(dl2, p["model.8.weight"], da3, sc)
        if B != Bl:
            da3 = self.comm.allgather_rows(da3)     # [B_global, 512], rank order = row order of the global batch
        gather = self.output_layer_factors_gathered()
        a3 = s["a3_all"] if gather else s["a3"]
        if last_layer is None:
            self.last_layer_grads(self.comm.allgather_rows(dl2) if gather else dl2, a3, sc)
        else:
            last_layer(dl2, a3, gather)
        # the chain d(a3) -> d(x0) first (every link waits for the one before), then the weight and bias
        # gradients of the three small layers, which hang off it, in ONE launch
        dpre3 = self._bn_bwd(s["a3_all"], da3, s["xhat3"], s["inv3"], "model.6", B, 512)
        da2 = torch.empty(B, 256, device=dev)
        ops.linear_bwd_dx(dpre3, p["model.5.weight"], da2, sc)
        dpre2 = self._bn_bwd(s["a2"], da2, s["xhat2"], s["inv2"], "model.3", B, 256)
        da1 = torch.empty(B, 128, device=dev)
        ops.linear_bwd_dx(dpre2, p["model.2.weight"], da1, sc)
        call("s2g_lrelu_bwd", ptr(s["a1"]), ptr(da1), da1.numel(), stream_ptr())
        dx0 = torch.empty(B, LATENT, device=dev)
        ops.linear_bwd_dx(da1, p["model.0.weight"], dx0, sc)
        jobs = [(dpre3, s["a2"], g["model.5.weight"], g["model.5.bias"]),
                (dpre2, s["a1"], g["model.2.weight"], g["model.2.bias"]),
                (da1, s["x0"], g["model.0.weight"], g["model.0.bias"])]
        if all(x.shape[1] % 4 == 0 for _, x, _, _ in jobs):
            ops.small_linear_grads(jobs)
        else:                                   # a latent size that is no multiple of 4: the generic products
            for dyj, xj, dWj, dbj in jobs:
                ops.linear_bwd_dw(dyj, xj, dWj, sc)
                ops.colsum(dyj, B, dyj.shape[1], dyj.shape[1], dbj, sc)
        call("s2g_embed_mul_bwd", ptr(s["z"]), ptr(s["labels"]), ptr(dx0), ptr(g["label_embedding.weight"]), B,
             LATENT, self.n_classes, stream_ptr())
