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

from s2g.lib import S2GError  # noqa: E402
from s2g.modules import adopt  # noqa: E402
from s2g.wgan_clip import ClipCriticEngine, ClipGeneratorEngine  # noqa: E402


class RelaxedEmbedding(nn.Module):
    """Reference :23-45 (lookup for integer input, matmul with the table for relaxed samples)."""

    def __init__(self, vocab_size, embedding_dim):
        super(RelaxedEmbedding, self).__init__()
        self.embed_layer = nn.Embedding(num_embeddings=vocab_size, embedding_dim=embedding_dim)


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


class Attention(nn.Module):
    """Reference :75-89 (parameters only; the pooling runs inside the critic's kernels)."""

    def __init__(self, hidden_dim):
        super(Attention, self).__init__()
        self.hidden_dim = hidden_dim
        self.attn = nn.Linear(hidden_dim, hidden_dim)
        self.context_vector = nn.Linear(hidden_dim, 1, bias=False)


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
        """Critic values [B, 1] (forward only; training goes through CWGAN.train_step)."""
        if not seq.is_cuda:
            raise S2GError("Discriminator.forward needs GPU tensors (MI355X-only implementation, no CPU path)")
        eng = self.engine(seq.device)
        lengths = torch.as_tensor(lengths)
        T = seq.shape[1]
        if int(lengths.min()) < 1 or int(lengths.max()) > T:
            raise S2GError(f"sequence lengths must lie in [1, {T}]")
        s = seq.to(torch.int32).contiguous() if seq.dtype in (torch.int64, torch.int32) else seq.contiguous().float()
        masks = eng.dropout_masks(seq.shape[0] * T) if self.training else None
        with torch.no_grad():
            val, _ = eng.forward(s, labels.contiguous(), lengths.to(torch.int32).to(seq.device), int(lengths.max()), masks,
                                 save=False)
        return val.view(-1, 1)


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
