"""MI355X-native WGAN-GP trainer with the module surface of the reference's wgan_gp.py:
CustomSoftmax, Generator, Discriminator, compute_gradient_penalty, train_wgan, main -- same
names, argument meaning, state_dict keys and checkpoint files (reference wgan_gp.py:22-447), so
`from wgan_gp import Generator` (generate_fake_data.py:8) and `from wgan_gp import CustomSoftmax`
(wgan.py:16) keep working.  All arithmetic runs in the HIP kernels of libs2g (include/s2g.h); the
modules need a GPU tensor and raise on CPU tensors (there is deliberately no CPU path).

    python3 wgan_gp.py --vocab_size=175 --seq_len=512 --batch_size=64 --lr=0.0002 \
        --lambda_gp=10.0 --tau=0.5 --epochs=500
"""
import os
import sys

import torch
import torch.nn as nn
from torch import rand, randn  # noqa: F401  (module-level names, as in the reference, so callers can patch them)
from torch.nn.utils.rnn import PackedSequence, pad_packed_sequence
from torch.utils.data import DataLoader

_PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sequence-to-graph-gan_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from config import get_args  # noqa: E402
from s2g import data as s2g_data  # noqa: E402
from s2g.critic import CriticEngine  # noqa: E402
from s2g.generator import GeneratorEngine  # noqa: E402
from s2g.lib import S2GError  # noqa: E402
from s2g.modules import CriticFn, GeneratorFn, GradientPenaltyFn, SamplerFn, adopt  # noqa: E402
from s2g.trainer import WGANGPTrainer, class_weight_sum  # noqa: E402


class CustomSoftmax(nn.Module):
    """Gumbel-softmax / relaxed (temperature) softmax / plain softmax sampler (reference :22-104)."""

    def __init__(self, act="gumbel_softmax"):
        super().__init__()
        self.act = act

    def forward(self, logits, tau, hard):
        if self.act == "gumbel_softmax":
            return self.gumbel_softmax(logits, tau, hard)
        elif self.act == "relaxed_bernoulli":
            return self.relaxed_bernoulli(logits, tau)
        return SamplerFn.apply(logits, None, 1.0, 2, False)

    def _uniform(self, logits):
        # the reference draws on the CPU generator and copies (:58): same seed -> same noise
        return torch.rand(logits.size()).to(logits.device)

    def sample_gumbel(self, logits, eps=1e-10):
        """Gumbel(0,1) noise as a tensor (reference :56-61).  Auxiliary accessor only: the sampling
        path fuses this transform into the kernel and never materialises it."""
        U = eps + (1 - eps) * self._uniform(logits)
        return -torch.log(-torch.log(U + 1e-20) + 1e-20)

    def gumbel_softmax_sample(self, logits, tau):
        return SamplerFn.apply(logits, self._uniform(logits), float(tau), 0, False)

    def gumbel_softmax(self, logits, tau, hard=False):
        return SamplerFn.apply(logits, self._uniform(logits), float(tau), 0, bool(hard))

    @staticmethod
    def anneal_tau_(initial_tau, final_tau, current_epoch, total_epochs, anneal_interval):
        if current_epoch % anneal_interval == 0:
            return initial_tau - (initial_tau - final_tau) * (current_epoch / total_epochs)
        return initial_tau

    def relaxed_bernoulli(self, logits, tau=1, dim=-1):
        if dim not in (-1, logits.dim() - 1):
            raise S2GError("relaxed_bernoulli: only the last dimension is supported")
        return SamplerFn.apply(logits, None, float(tau), 1, False)


class Generator(nn.Module):
    """Reference :141-171.  `seq_len` is an extra keyword; the default 120 is the reference's
    hard-coded value, so checkpoints stay loadable by generate_fake_data.py:28-29."""

    def __init__(self, vocab_size, n_classes=2, seq_len=120):
        super(Generator, self).__init__()
        self.latent_dim = 100
        self.seq_len = seq_len
        self.vocab_size = vocab_size
        self.n_classes = n_classes

        def block(in_feat, out_feat, normalize=True):
            layers = [nn.Linear(in_feat, out_feat)]
            if normalize:
                layers.append(nn.BatchNorm1d(out_feat, 0.8))
            layers.append(nn.LeakyReLU(0.2, inplace=True))
            return layers

        self.model = nn.Sequential(
            *block(self.latent_dim, 128, normalize=False),
            *block(128, 256),
            *block(256, 512),
            nn.Linear(512, self.vocab_size * self.seq_len)
        )
        self.label_embedding = nn.Embedding(n_classes, self.latent_dim)
        self.gumbel_softmax = CustomSoftmax()
        self._eng = None

    def engine(self, device) -> GeneratorEngine:
        if self._eng is None or self._eng.device != device:
            self._eng = GeneratorEngine(self.vocab_size, self.seq_len, device, self.n_classes)
        eng = self._eng
        adopt(self, eng.params)
        named = dict(self.named_buffers())
        for k, buf in eng.buffers.items():
            if named[k].data_ptr() != buf.data_ptr():
                buf.copy_(named[k].data.to(device))
                named[k].data = buf
        return eng

    def forward(self, noise, labels, tau, hard):
        if not noise.is_cuda:
            raise S2GError("Generator.forward needs GPU tensors (MI355X-only implementation, no CPU path)")
        s2g_data.check_batch(labels=labels, n_classes=self.n_classes)
        eng = self.engine(noise.device)
        U = None
        if self.gumbel_softmax.act == "gumbel_softmax":
            U = torch.rand((noise.shape[0], self.seq_len, self.vocab_size)).to(noise.device)
        y = GeneratorFn.apply(eng, noise, labels, U, float(tau), bool(hard), self.training,
                              self.gumbel_softmax.act, *[p for _, p in self.named_parameters()])
        if self.training:
            for m in (self.model[3], self.model[6]):
                m.num_batches_tracked += 1
        return y


def _generator_sample_tokens(self, noise, labels, uniforms=None):
    """tokens = self(noise, labels, tau, hard).argmax(-1) (generate_fake_data.py:64-66) as int64 [B, seq_len],
    computed by one fused kernel (output layer + Gumbel perturbation + argmax); the result does not depend
    on tau or hard.  uniforms: optional [B, seq_len, vocab] draws; default: torch.rand on the CPU generator,
    as the reference's sampler draws them (:58)."""
    if not noise.is_cuda:
        raise S2GError("Generator.sample_tokens needs GPU tensors (MI355X-only implementation, no CPU path)")
    s2g_data.check_batch(labels=labels, n_classes=self.n_classes)
    eng = self.engine(noise.device)
    if uniforms is None and self.gumbel_softmax.act == "gumbel_softmax":
        uniforms = torch.rand((noise.shape[0], self.seq_len, self.vocab_size)).to(noise.device)
    if self.gumbel_softmax.act != "gumbel_softmax":
        raise S2GError("Generator.sample_tokens: only the Gumbel-softmax sampler is fused")
    with torch.no_grad():
        tok = eng.generate_tokens(noise.contiguous().float(), labels.contiguous(), uniforms.contiguous().float(),
                                  training=self.training)
    if self.training:
        for m in (self.model[3], self.model[6]):
            m.num_batches_tracked += 1
    return tok


Generator.sample_tokens = _generator_sample_tokens


class Discriminator(nn.Module):
    """Reference :175-223: nn.LSTM(input, hidden=50, layers=2, dropout) -> last layer's final hidden
    state -> Linear(50, num_classes + 1).  Accepts a PackedSequence or a dense [B,T,V] tensor."""

    def __init__(self, input_size, hidden_size, num_layers, num_classes, drop_rate=0.3):
        super(Discriminator, self).__init__()
        if hidden_size != 50 or num_layers != 2 or num_classes != 2:
            raise S2GError("the HIP critic is specialised to hidden_size=50, num_layers=2, num_classes=2 "
                           "(the configuration wgan_gp.main builds, reference :393-400)")
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.input_size = input_size
        self.drop_rate = drop_rate
        self.lstm = nn.LSTM(input_size, hidden_size, num_layers, batch_first=True, dropout=drop_rate)
        self.fc = nn.Linear(self.hidden_size, num_classes + 1)
        self._eng = None

    def engine(self, device) -> CriticEngine:
        if self._eng is None or self._eng.device != device:
            self._eng = CriticEngine(self.input_size, device)
        adopt(self, self._eng.params)
        return self._eng

    def dropout_mask(self, B, T, device):
        if not self.training or self.drop_rate <= 0.0:
            return None
        return (torch.rand(B, T, 50, device=device) >= self.drop_rate).float() / (1.0 - self.drop_rate)

    def forward(self, packed_input, lengths=None):
        if isinstance(packed_input, PackedSequence):
            x, lens = pad_packed_sequence(packed_input, batch_first=True)
        else:
            x, lens = packed_input, None
        if not x.is_cuda:
            raise S2GError("Discriminator.forward needs GPU tensors (MI355X-only implementation, no CPU path)")
        if x.dim() != 3 or x.shape[2] != self.input_size:
            raise S2GError(f"Discriminator.forward: expected [B, T, {self.input_size}] input, got {tuple(x.shape)}")
        s2g_data.check_batch(lengths=lens, T=x.shape[1])
        eng = self.engine(x.device)
        mask = self.dropout_mask(x.shape[0], x.shape[1], x.device)
        return CriticFn.apply(eng, x, lens, mask, *[p for _, p in self.named_parameters()])


def compute_gradient_penalty(discriminator, encoded_padded_real_data, fake_data, labels, lengths):
    """Gradient penalty of WGAN-GP (reference :225-257): returns a 0-dim tensor attached to the
    graph; its backward yields the critic-parameter gradient of the penalty (second order)."""
    alpha = rand((fake_data.size(0), 1, 1)).to(fake_data.device)
    eng = discriminator.engine(fake_data.device)
    B, T = encoded_padded_real_data.shape[0], encoded_padded_real_data.shape[1]
    lengths = torch.as_tensor(lengths)
    if lengths.numel() != B:
        raise S2GError("compute_gradient_penalty: one length per sample expected")
    s2g_data.check_batch(lengths=lengths.cpu(), T=T)
    mask = discriminator.dropout_mask(B, T, fake_data.device)
    return GradientPenaltyFn.apply(eng, encoded_padded_real_data, fake_data.detach(), alpha, torch.as_tensor(lengths),
                                   mask, *[p for _, p in discriminator.named_parameters()])


def _adam_hparams(optimizer):
    """(lr, beta1, beta2, eps) of a torch.optim.Adam as built at reference :405-406; options the fused
    Adam kernel does not implement are refused instead of being ignored."""
    if not isinstance(optimizer, torch.optim.Adam) or type(optimizer) is not torch.optim.Adam:
        raise S2GError(f"train_wgan drives torch.optim.Adam only (got {type(optimizer).__name__})")
    if len(optimizer.param_groups) != 1:
        raise S2GError("train_wgan: one parameter group per optimizer expected")
    g = optimizer.param_groups[0]
    if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
        raise S2GError("train_wgan: weight_decay / amsgrad / maximize are not supported by the fused Adam")
    return float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])


def _import_adam(optimizer, module, flat_m, flat_v):
    """Seeds the fused Adam's moments and step count from the torch optimizer's state (a second
    train_wgan call or a resumed run continues where the optimizer stands).  Returns the step count."""
    steps = set()
    for name, p in module.named_parameters():
        st = optimizer.state.get(p)
        if not st or name not in flat_m.views:
            continue
        flat_m.views[name].copy_(st["exp_avg"].to(flat_m.views[name].device))
        flat_v.views[name].copy_(st["exp_avg_sq"].to(flat_v.views[name].device))
        steps.add(int(st["step"]))
    if len(steps) > 1:
        raise S2GError("train_wgan: parameters of one optimizer carry different step counts")
    return steps.pop() if steps else 0


def _export_adam(optimizer, module, flat_m, flat_v, step):
    """Expose the fused Adam's moments through the torch optimizer's state_dict."""
    for name, p in module.named_parameters():
        if name in flat_m.views:
            optimizer.state[p] = {"step": torch.tensor(float(step)), "exp_avg": flat_m.views[name],
                                  "exp_avg_sq": flat_v.views[name]}


def train_wgan(generator, discriminator, optimizer_G, optimizer_D, train_dataloader, val_loader, opt, device):
    """Reference :261-382.  Same loop, prints and checkpoints; each batch's critic update and the
    generator update (when i % n_critic == 0) run as the fused HIP step, scheduled like
    WGANGPTrainer.full_step: the loader is read one batch ahead, so the next fake batch (or the generator
    step's own forward) is computed beside the current batch's second-order sweeps.
    opt.device_rng (flag --device_rng, default off): draw z / Gumbel uniforms / alpha on the device instead
    of the CPU generator; faster (no 23 MB of CPU draws and copies per batch at L = 512), but a given seed
    then no longer reproduces the reference's noise."""
    device = torch.device(device)
    if device.type != "cuda":
        raise S2GError("train_wgan needs an MI355X (device 'cuda'); there is no CPU path")
    lr, b1, b2, eps = _adam_hparams(optimizer_D)
    tr = WGANGPTrainer(opt.vocab_size, generator.seq_len, device, lr=lr, b1=b1, b2=b2, eps=eps,
                       adam_g=_adam_hparams(optimizer_G), lambda_gp=opt.lambda_gp,
                       tau=opt.tau, hard=bool(opt.hard), n_critic=opt.n_critic,
                       drop_rate=getattr(discriminator, "drop_rate", 0.3),
                       gen_engine=generator.engine(device), critic_engine=discriminator.engine(device))
    tr.tD = _import_adam(optimizer_D, discriminator, tr.mD_named, tr.vD_named)
    tr.tG = _import_adam(optimizer_G, generator, tr.mG_named, tr.vG_named)
    tr.G.num_batches_tracked = int(generator.model[3].num_batches_tracked)

    def sync_counters():
        for m in (generator.model[3], generator.model[6]):
            m.num_batches_tracked.fill_(tr.G.num_batches_tracked)
    from tqdm import tqdm
    from sklearn.metrics import f1_score

    device_rng = bool(getattr(opt, "device_rng", False))
    V, Lg = generator.vocab_size, generator.seq_len

    staging = {}      # (shape, dtype) -> [slots, next]: page-locked buffers reused round-robin, an event per slot

    def to_dev(t):
        # page-locked staging + an asynchronous copy on the current (the trainer's) stream: the host does not
        # wait for the step that is still running.  The buffers are kept (allocating page-locked memory is a slow,
        # synchronising driver call); a slot is reused once the copy out of it has completed.
        key = (tuple(t.shape), t.dtype)
        ring = staging.setdefault(key, [[], 0])
        if len(ring[0]) < 4:
            ring[0].append((torch.empty(t.shape, dtype=t.dtype).pin_memory(), torch.cuda.Event()))
            host, done = ring[0][-1]
        else:
            host, done = ring[0][ring[1] % 4]
            ring[1] += 1
            done.synchronize()
        host.copy_(t)
        out = host.to(device, non_blocking=True)
        done.record()
        return out

    def upload(batch):
        """Range checks on the host, then the batch as the kernels take it (int32 tokens / lengths)."""
        if batch is None:
            return None
        padded_sequences, labels, lengths = batch
        s2g_data.check_batch(padded_sequences, labels, lengths, vocab=opt.vocab_size,
                             n_classes=generator.n_classes, T=padded_sequences.shape[1])
        return dict(tok=to_dev(padded_sequences.to(torch.int32).contiguous()), lab=to_dev(labels),
                    ln=to_dev(lengths.to(torch.int32)), wsum=class_weight_sum(labels), B=padded_sequences.shape[0],
                    T=padded_sequences.shape[1])

    # The reference draws z / U / alpha / U2 on the CPU generator and copies them (:291, :58, :228, :330).
    # These helpers make the same draws in the same order -- z_i, U_i, alpha_i, [U2_i], z_i+1, U_i+1, ... --
    # only EARLIER than the reference needs them, so that the fake batch of step i+1 and the generator
    # forward of a generator step can be computed beside step i's second-order sweeps.  With
    # opt.device_rng they return nothing and the trainer draws on the device (another stream of numbers).
    def draw_zu(b):
        if device_rng or b is None:
            return {}
        return dict(z=to_dev(randn(b["B"], generator.latent_dim)), U=to_dev(torch.rand((b["B"], Lg, V))))

    def draw_alpha(b):
        return None if device_rng else to_dev(rand((b["B"], 1, 1))).view(-1)

    def draw_u2(b):
        return None if device_rng else to_dev(torch.rand((b["B"], Lg, V)))

    for epoch in range(1, opt.epochs + 1):
        generator.train()
        discriminator.train()
        # host-to-device copies and the steps all go to the trainer's non-blocking stream (see WGANGPTrainer.stream)
        with tr.stream():
            n_batches = len(train_dataloader)
            it = iter(tqdm(train_dataloader))
            cur = upload(next(it, None))
            zu = draw_zu(cur)
            i = 0
            after_next, have_after_next = None, False
            while cur is not None:
                # the loader is read one batch ahead (two around a loss print, see below)
                nxt = after_next if have_after_next else upload(next(it, None))
                have_after_next = False
                gen_now = i % opt.n_critic == 0
                alpha = draw_alpha(cur)
                U2 = draw_u2(cur) if gen_now else None
                zu_next = draw_zu(nxt)
                # the generator is unchanged between this batch and the next unless it is updated now: then the
                # next batch's fake sequences can be made ahead (same batch size: they land in a buffer of this shape)
                ahead = None
                if nxt is not None and not gen_now and nxt["B"] == cur["B"]:
                    ahead = dict(labels=nxt["lab"], T=nxt["T"], lengths=nxt["ln"], z=zu_next.get("z"), U=zu_next.get("U"))
                tr.critic_step(cur["tok"], cur["lab"], cur["ln"], cur["wsum"], z=zu.get("z"), U=zu.get("U"), alpha=alpha,
                               next_batch=ahead, gen_next=dict(U=U2) if gen_now else None)
                if gen_now:
                    # (the output layer's update may still run on the side stream: the next forward waits for it)
                    tr.generator_step(U=U2, defer_join=True, keep_grads=False)
                    # reading the losses waits for the GPU; whatever the host does between that moment and its next
                    # launch is time the GPU idles, so the batch after next is checked and staged BEFORE the wait
                    after_next, have_after_next = upload(next(it, None)), True
                    d_loss, g_loss = tr.losses().tolist()
                    print(
                        "[Epoch %d/%d] [Batch %d/%d] [D loss: %f] [G loss: %f]"
                        % (epoch, opt.epochs, i, n_batches, d_loss, g_loss)
                    )
                    if epoch % 5 == 0:
                        tr.join_generator()
                        sync_counters()
                        torch.save(generator.state_dict(), f"saved_models/wgan_generator_{epoch}.pth")
                    if epoch % 5 == 0:
                        with torch.no_grad():
                            discriminator.eval()
                            f1_scores = []
                            for _, (vp, vl, vlen) in enumerate(val_loader):
                                real = torch.nn.functional.one_hot(vp, opt.vocab_size).float().to(device)
                                out = discriminator(torch.nn.utils.rnn.pack_padded_sequence(
                                    real, vlen, batch_first=True, enforce_sorted=False), vlen)
                                preds = torch.argmax(out[:, 1:], dim=1).detach().cpu().tolist()
                                f1_scores.append(f1_score(preds, vl.cpu().tolist()))
                            avg_f1_score = torch.mean(torch.tensor(f1_scores, dtype=torch.float32))
                            print(f"f1 score: {avg_f1_score:.4f}")
                            discriminator.train()
                cur, zu, i = nxt, zu_next, i + 1
            tr.join_generator()
    sync_counters()
    _export_adam(optimizer_D, discriminator, tr.mD_named, tr.vD_named, tr.tD)
    _export_adam(optimizer_G, generator, tr.mG_named, tr.vG_named, tr.tG)
    return tr


def main():
    opt = get_args()
    vocabs = s2g_data.load_vocabulary()
    input_size = len(vocabs)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if device.type != "cuda":
        raise S2GError("wgan_gp.py needs an MI355X (torch.cuda.is_available() is False)")

    discriminator = Discriminator(input_size, 50, 2, 2, drop_rate=opt.drop_rate).to(device)
    generator = Generator(opt.vocab_size, 2, seq_len=opt.gen_seq_len).to(device)
    optimizer_G = torch.optim.Adam(generator.parameters(), lr=opt.lr, betas=(opt.b1, opt.b2))
    optimizer_D = torch.optim.Adam(discriminator.parameters(), lr=opt.lr, betas=(opt.b1, opt.b2))

    s2g_data.set_random_seeds(seed_value=42)
    dataset_folder = os.path.join(os.getcwd(), "data")
    train_data, train_labels = s2g_data.load_dataset(os.path.join(dataset_folder, "train_dataset.json"))
    test_data, test_labels = s2g_data.load_dataset(os.path.join(dataset_folder, "test_dataset.json"))
    train_data = s2g_data.encode_sequences(train_data, vocabs)
    test_data = s2g_data.encode_sequences(test_data, vocabs)
    train_labels, test_labels = s2g_data.encode_labels(train_labels, test_labels)
    train_data, train_labels = s2g_data.sample_equal_data(train_data, train_labels)

    train_dataset = s2g_data.CustomSequenceDataset(train_data, train_labels, length=opt.seq_len)
    test_dataset = s2g_data.CustomSequenceDataset(test_data, test_labels, opt.seq_len)
    train_loader = DataLoader(train_dataset, batch_size=opt.batch_size, collate_fn=s2g_data.collate_fn, shuffle=True)
    val_loader = DataLoader(test_dataset, batch_size=opt.batch_size, collate_fn=s2g_data.collate_fn, shuffle=False)
    os.makedirs("saved_models", exist_ok=True)
    train_wgan(generator, discriminator, optimizer_G, optimizer_D, train_loader, val_loader, opt, device)


if __name__ == "__main__":
    main()
