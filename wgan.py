"""MI355X-native weight-clipping conditional WGAN with the module surface of the reference's wgan.py:
RelaxedEmbedding, Generator (LSTM), Attention, Discriminator (LSTM + attention pooling) and CWGAN -- same
class names, constructor arguments and state_dict keys (reference wgan.py:23-160), so checkpoints written by
`torch.save(model.discriminator.state_dict(), ...)` (:374-375) and generate_fake_data.py's `.pt` branch
(:31-33) keep their format.  Forward passes and the critic update (CWGAN.train_step lines 174-198) run in the
HIP kernels of libs2g (include/s2g.h); they need GPU tensors.

Two statements of the reference's train_step cannot be mirrored as working code and are kept as they are:
  * `if i % self.critic_iterations:` (:206) enters the generator branch on every batch whose index is NOT a
    multiple of n_critic (the inverse of WGAN-GP's test); kept.
  * inside that branch `self.loss(...)` (:214) is undefined: the reference raises AttributeError on the second
    batch of every epoch.  CWGAN.train_step raises the same AttributeError at the same point (after that
    batch's critic update) unless opt.skip_generator_update is set, in which case only the critic trains.
"""
import os
import sys

import torch
import torch.nn as nn
from torch.optim import Adam

_PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sequence-to-graph-gan_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from torch.autograd.function import once_differentiable  # noqa: E402

from s2g import ops  # noqa: E402
from s2g.lib import S2GError, call, get  # noqa: E402
from s2g.modules import adopt  # noqa: E402
from s2g.ops import ptr, stream_ptr  # noqa: E402
from s2g.wgan_clip import ClipCriticEngine, ClipGeneratorEngine, _tn  # noqa: E402
from wgan_gp import CustomSoftmax  # noqa: E402


def _gpu(t, what):
    if not t.is_cuda:
        raise S2GError(f"{what} needs GPU tensors (MI355X-only implementation, no CPU path)")


class _EmbeddingFn(torch.autograd.Function):
    """RelaxedEmbedding.forward (reference :36-45): table lookup for integer input, x @ table for relaxed samples."""

    @staticmethod
    def forward(ctx, x, W):
        _gpu(x, "RelaxedEmbedding.forward")
        V, E = W.shape
        Wc = W.contiguous().float()
        ctx.lookup = x.dtype in (torch.int64, torch.int32)
        lead = tuple(x.shape) if ctx.lookup else tuple(x.shape[:-1])
        rows = 1
        for d in lead:
            rows *= d
        y = torch.empty(rows, E, device=x.device)
        if ctx.lookup:
            if E != 32:
                raise S2GError("RelaxedEmbedding: the lookup kernels take embedding_dim = 32 (wgan.py:49,95)")
            tok = x.reshape(-1).to(torch.int32).contiguous()
            if rows and (int(tok.min()) < 0 or int(tok.max()) >= V):
                raise S2GError(f"RelaxedEmbedding: token ids must lie in [0, {V})")
            # the concat kernel of the critic's input (token embedding | label embedding) with an all-zero label part
            x64 = torch.empty(rows, 64, device=x.device)
            zl, ze = torch.zeros(rows, dtype=torch.int64, device=x.device), torch.zeros(1, 32, device=x.device)
            call("s2g_embed_concat", ptr(tok), ptr(Wc), V, ptr(zl), ptr(ze), 1, 1, rows, 0, ptr(x64), stream_ptr())
            y = x64[:, :32].contiguous()
            ctx.save_for_backward(tok, Wc)
        else:
            if x.shape[-1] != V:
                raise S2GError("RelaxedEmbedding: relaxed input must be [..., vocab_size]")
            xs = x.reshape(rows, V).contiguous().float()
            ops.gemm_raw(ptr(xs), V, 1, ptr(Wc), E, 1, ptr(y), E, rows, E, V)
            ctx.save_for_backward(xs, Wc)
        ctx.lead = lead
        return y.view(*lead, E)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        a, W = ctx.saved_tensors
        V, E = W.shape
        rows = a.shape[0]
        dy2 = dy.reshape(rows, E).contiguous().float()
        dW = torch.zeros_like(W)
        sc = ops.Scratch(dy.device)
        if ctx.lookup:
            lens = torch.ones(rows, dtype=torch.int32, device=dy.device)         # every row live: sequences of one step
            part = sc.get(int(get().s2g_embed_grad_tokens_scratch(rows, V)))
            call("s2g_bucket_rows_by_token", ptr(a), ptr(lens), 1, rows, V, ptr(dy2), E, E, ptr(dW), ptr(part), stream_ptr())
            return None, dW
        _tn(ptr(a), V, V, ptr(dy2), E, E, rows, dW, sc, False)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(rows, V, device=dy.device)
            ops.gemm_raw(ptr(dy2), E, 1, ptr(W), 1, E, ptr(dx), V, rows, V, E)
            dx = dx.view(*ctx.lead, V)
        return dx, dW


class RelaxedEmbedding(nn.Module):
    """Reference :23-45 (lookup for integer input, matmul with the table for relaxed samples)."""

    def __init__(self, vocab_size, embedding_dim):
        super(RelaxedEmbedding, self).__init__()
        self.embed_layer = nn.Embedding(num_embeddings=vocab_size, embedding_dim=embedding_dim)

    def forward(self, x):
        return _EmbeddingFn.apply(x, self.embed_layer.weight)


class Generator(nn.Module):
    """Reference :46-71."""

    def __init__(self, vocab_size, num_classes, num_layers, softmax="relaxed_bernoulli"):
        super(Generator, self).__init__()
        self.embedding_dim = 32
        self.hidden_dim = 64
        self.seq_len = 100
        self.latent_dim = 100
        self.vocab_size = vocab_size
        self.num_classes, self.num_layers = num_classes, num_layers
        self.act = softmax
        self.label_embedding = nn.Embedding(num_classes, self.embedding_dim)
        self.noise_embedding = nn.Embedding(vocab_size, self.embedding_dim)
        self.lstm = nn.LSTM(self.embedding_dim * 2, self.hidden_dim, num_layers=num_layers, batch_first=True)
        self.linear = nn.Linear(self.hidden_dim, vocab_size)
        self._eng = None

    def engine(self, device) -> ClipGeneratorEngine:
        if self._eng is None or self._eng.device != device:
            self._eng = ClipGeneratorEngine(self.vocab_size, self.num_layers, device, self.num_classes)
        adopt(self, self._eng.params)
        return self._eng

    def forward(self, noise, labels, tau=0.2, anneal_tau=False, hard=False):
        if not noise.is_cuda:
            raise S2GError("Generator.forward needs GPU tensors (MI355X-only implementation, no CPU path)")
        eng = self.engine(noise.device)
        U = None
        if self.act == "gumbel_softmax":            # the draw of CustomSoftmax.sample_gumbel (wgan_gp.py:58)
            U = torch.rand((noise.shape[0], noise.shape[1], self.vocab_size)).to(noise.device)
        with torch.no_grad():
            return eng.forward(noise.to(torch.int32).contiguous(), labels.contiguous(), float(tau), U=U, act=self.act,
                               hard=bool(hard))


class _AttentionFn(torch.autograd.Function):
    """Attention.forward (reference :82-89): scores = context_vector(tanh(attn(lstm_outputs))), softmax over time,
    weighted sum of `hidden`; returns (pooled [B, 64], weights [B, T]).  Forward and backward are the critic's
    pooling kernels (s2g_attnpool_fwd / _bwd) with the output Linear left out."""

    @staticmethod
    def forward(ctx, hidden, lstm_outputs, Wa, ba, wc):
        _gpu(hidden, "Attention.forward")
        B, T, Hd = lstm_outputs.shape
        if Hd != 64 or tuple(hidden.shape) != (B, T, 64):
            raise S2GError("Attention: the pooling kernels take hidden_dim = 64 (wgan.py:50,96) and [B, T, 64] inputs")
        dev = hidden.device
        hid = hidden.contiguous().float().view(B * T, 64)
        lo = lstm_outputs.contiguous().float().view(B * T, 64)
        Wa, ba, wc = Wa.contiguous().float(), ba.contiguous().float(), wc.contiguous().float()
        a = torch.empty(B * T, 64, device=dev)
        ops.linear_fwd(lo, Wa, ba, a)
        weights, pooled, dval = torch.empty(B, T, device=dev), torch.empty(B, 64, device=dev), torch.empty(B, device=dev)
        zero = torch.zeros(65, device=dev)                     # an all-zero output Linear: its value is not used
        call("s2g_attnpool_fwd", ptr(a), ptr(hid), ptr(wc), ptr(zero), ptr(zero, 64), B, T, T, ptr(weights), ptr(pooled),
             ptr(dval), stream_ptr())
        ctx.save_for_backward(a, hid, lo, weights, Wa, wc)
        ctx.dims = (B, T)
        return pooled, weights

    @staticmethod
    @once_differentiable
    def backward(ctx, dpooled, dweights):
        a, hid, lo, weights, Wa, wc = ctx.saved_tensors
        B, T = ctx.dims
        dev, rows = a.device, B * T
        dpre = a.clone()                                       # tanh(pre) in, d(pre) out
        dhid, dwc_part = torch.empty(rows, 64, device=dev), torch.empty(B, 64, device=dev)
        call("s2g_attnpool_bwd", ptr(dpre), ptr(hid), ptr(wc), None, ptr(weights), None, B, T, T, ptr(dhid),
             ptr(dwc_part), ptr(dpooled.contiguous().float()), ptr(dweights.contiguous().float()), stream_ptr())
        sc = ops.Scratch(dev)
        dlo = torch.empty(rows, 64, device=dev)
        ops.gemm_raw(ptr(dpre), 64, 1, ptr(Wa), 64, 1, ptr(dlo), 64, rows, 64, 64)       # d(lstm_outputs) = d(pre) Wa
        dWa, dba, dwc = torch.empty(64, 64, device=dev), torch.empty(64, device=dev), torch.empty(1, 64, device=dev)
        _tn(ptr(dpre), 64, 64, ptr(lo), 64, 64, rows, dWa, sc, False)
        ops.colsum(dpre, rows, 64, 64, dba, sc)
        ops.colsum(dwc_part, B, 64, 64, dwc, sc)
        return dhid.view(B, T, 64), dlo.view(B, T, 64), dWa, dba, dwc


class Attention(nn.Module):
    """Reference :75-89.  (Inside Discriminator the pooling runs fused with the output Linear.)"""

    def __init__(self, hidden_dim):
        super(Attention, self).__init__()
        self.hidden_dim = hidden_dim
        self.attn = nn.Linear(hidden_dim, hidden_dim)
        self.context_vector = nn.Linear(hidden_dim, 1, bias=False)

    def forward(self, hidden, lstm_outputs):
        return _AttentionFn.apply(hidden, lstm_outputs, self.attn.weight, self.attn.bias, self.context_vector.weight)


class _CriticFn(torch.autograd.Function):
    """Discriminator.forward (reference :106-122) with its hand-derived backward: gradients with respect to every
    parameter and, for a relaxed input, to the input."""

    @staticmethod
    def forward(ctx, eng, seq, labels, lengths_dev, Tp, masks, *params):
        val, st = eng.forward(seq, labels, lengths_dev, Tp, masks, save=True)
        ctx.eng, ctx.st, ctx.names = eng, st, list(eng.params.views)
        return val.view(-1, 1).clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, dval):
        eng = ctx.eng
        g = eng.params.like()
        want = bool(ctx.st["soft"] and ctx.needs_input_grad[1])
        dseq = eng.backward(ctx.st, dval.contiguous().float().view(-1), accumulate=False, grads=g, want_dseq=want)
        for l in range(eng.layers):                 # b_ih and b_hh enter the gates as a sum: one gradient
            g.views[f"lstm.bias_hh_l{l}"].copy_(g.views[f"lstm.bias_ih_l{l}"])
        return (None, dseq, None, None, None, None) + tuple(g.views[n] for n in ctx.names)


class Discriminator(nn.Module):
    """Reference :92-122."""

    def __init__(self, vocab_size, num_classes, num_layers, dropout=0.1, use_gpu=True):
        super(Discriminator, self).__init__()
        self.embedding_dim = 32
        self.hidden_dim = 64
        self.vocab_size, self.num_classes, self.num_layers, self.dropout = vocab_size, num_classes, num_layers, dropout
        self.label_embedding = nn.Embedding(num_classes, self.embedding_dim)
        self.sequence_embedding = RelaxedEmbedding(vocab_size, self.embedding_dim)
        self.lstm = nn.LSTM(self.embedding_dim * 2, self.hidden_dim, num_layers=num_layers, dropout=dropout, batch_first=True)
        self.attention = Attention(self.hidden_dim)
        self.linear = nn.Linear(self.hidden_dim, 1)
        self._eng = None

    def engine(self, device) -> ClipCriticEngine:
        if self._eng is None or self._eng.device != device:
            self._eng = ClipCriticEngine(self.vocab_size, self.num_layers, device, self.dropout, self.num_classes)
        adopt(self, self._eng.params)
        return self._eng

    def forward(self, seq, labels, lengths):
        """Critic values [B, 1], differentiable with respect to the parameters and to a relaxed input
        (CWGAN.train_step takes the fused path of critic_train_step instead)."""
        if not seq.is_cuda:
            raise S2GError("Discriminator.forward needs GPU tensors (MI355X-only implementation, no CPU path)")
        eng = self.engine(seq.device)
        lengths = torch.as_tensor(lengths)
        T = seq.shape[1]
        if int(lengths.min()) < 1 or int(lengths.max()) > T:
            raise S2GError(f"sequence lengths must lie in [1, {T}]")
        tokens = seq.dtype in (torch.int64, torch.int32)
        if tokens and (int(seq.min()) < 0 or int(seq.max()) >= self.vocab_size):
            raise S2GError(f"token ids must lie in [0, {self.vocab_size})")
        s = seq.to(torch.int32).contiguous() if tokens else seq.contiguous().float()
        masks = eng.dropout_masks(seq.shape[0] * T) if self.training else None
        lens_dev = lengths.to(torch.int32).to(seq.device)
        if not torch.is_grad_enabled():
            val, _ = eng.forward(s, labels.contiguous(), lens_dev, int(lengths.max()), masks, save=False)
            return val.view(-1, 1)
        params = [dict(self.named_parameters())[n] for n in eng.params.views]
        return _CriticFn.apply(eng, s, labels.contiguous(), lens_dev, int(lengths.max()), masks, *params)


class CWGAN(nn.Module):
    """Reference :126-246."""

    def __init__(self, generator, discriminator, opt):
        super(CWGAN, self).__init__()
        if not torch.cuda.is_available():
            raise S2GError("CWGAN needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.generator = generator.to(self.device)
        self.discriminator = discriminator.to(self.device)
        self.d_optimizer = Adam(self.discriminator.parameters(), lr=opt.lr, betas=(opt.beta1, opt.beta2))
        self.g_optimizer = Adam(self.generator.parameters(), lr=opt.lr, betas=(opt.beta1, opt.beta2))
        self.clip_value = opt.clip_value
        self.critic_iterations = opt.n_critic
        self.skip_generator_update = bool(getattr(opt, "skip_generator_update", False))
        self.losses = {"d_loss": [], "g_loss": []}

    def critic_train_step(self, sequences, labels, lengths, tau, hard=False):
        """Lines 174-198 for one batch; returns d_loss as a 1-element device tensor (no synchronisation)."""
        dev = self.device
        D, G = self.discriminator.engine(dev), self.generator.engine(dev)
        lengths = torch.as_tensor(lengths)
        T = sequences.shape[1]
        if int(lengths.min()) < 1 or int(lengths.max()) > T:
            raise S2GError(f"sequence lengths must lie in [1, {T}]")
        lo, hi = int(sequences.min()), int(sequences.max())
        if lo < 0 or hi >= self.discriminator.vocab_size:
            raise S2GError(f"token ids must lie in [0, {self.discriminator.vocab_size})")
        labels = labels.to(dev)
        B = labels.size(0)
        noise = torch.randint(0, self.generator.vocab_size, size=(B, self.generator.latent_dim)).to(dev)     # :183
        # the reference passes `hard` in the position of `anneal_tau` (:185 against :63): hard sampling is never on
        gen_data = lambda: self.generator(noise, labels, tau, hard)      # noqa: E731  (runs beside the real pass)
        g = self.d_optimizer.param_groups[0]
        return D.critic_step(sequences.to(torch.int32).to(dev).contiguous(), labels, lengths.to(torch.int32).to(dev),
                             int(lengths.max()), gen_data, g["lr"], g["betas"][0], g["betas"][1], self.clip_value,
                             masks_real="draw" if self.discriminator.training else None,
                             masks_fake="draw" if self.discriminator.training else None, eps=g["eps"],
                             fake_len=noise.shape[1])

    def train_step(self, train_dataloader, tau, hard):
        from tqdm import tqdm
        d_loss_batch, g_loss_batch = 0.0, 0.0
        number_of_batchs = len(train_dataloader)
        for i, (sequences, labels, lengths) in enumerate(tqdm(train_dataloader)):
            d_loss_batch += self.critic_train_step(sequences, labels, lengths, tau, hard).item()
            if i % self.critic_iterations and not self.skip_generator_update:
                # wgan.py:214 calls self.loss, which no line of the reference defines
                raise AttributeError("'CWGAN' object has no attribute 'loss'")
        d_loss_batch /= number_of_batchs
        g_loss_batch /= number_of_batchs
        self.losses["d_loss"].append(d_loss_batch)
        self.losses["g_loss"].append(g_loss_batch)
        return d_loss_batch, g_loss_batch

    def train(self, train_dataloader=True, epochs=None, tau=None, hard=False, anneal_tau=False, anneal_interval=5):
        """Reference :230-246: `epochs` calls of train_step, tau annealed from 5.0 towards 0.1 by
        CustomSoftmax.anneal_tau_ when anneal_tau is set.  The name shadows nn.Module.train, as in the reference;
        a call without `epochs` (model.train() / model.train(False), which nn.Module.eval() makes) keeps its
        nn.Module meaning."""
        if epochs is None:
            return super(CWGAN, self).train(bool(train_dataloader))
        initial_tau = 5.0 if anneal_tau else tau
        final_tau = 0.1 if anneal_tau else tau
        for epoch in range(epochs):
            d_loss, g_loss = self.train_step(train_dataloader, tau=initial_tau, hard=hard)
            print(f"Epoch {epoch + 1} / {epochs}| Discriminator Loss: {d_loss:.5f} | Generator Loss: {g_loss:.5f}")
            if anneal_tau:
                initial_tau = CustomSoftmax.anneal_tau_(initial_tau, final_tau, epoch, epochs, anneal_interval)
                print("epoch = ", epoch + 1, "| tau = ", initial_tau)

    def validate(self, val_dataloader):
        """Reference :244-246: a placeholder there as well."""
        pass
