"""TEST INFRASTRUCTURE: CPU restatement (plain torch ops, explicit time loops) of the critic update of the
reference's weight-clipping WGAN, wgan.py:46-122 (modules) and :174-198 (the critic half of CWGAN.train_step).
Parameters are dicts under the reference's state_dict names; every random draw (noise tokens, Gumbel
uniforms, dropout masks) is an explicit argument.  Pinned by tests/golden/g12_wgan_clip.pt, recorded by
running the imported reference classes (tests/test_wgan_clip_cpu.py)."""
import torch

H = 64


def lstm(p, prefix, x, layers, lengths=None, masks=None):
    """nn.LSTM(64, 64, layers, batch_first=True, dropout) on x [B, T, 64]; lengths: pack_padded / pad_packed
    semantics (outputs are zero beyond a sequence's length, wgan.py:112-114); masks[l]: the dropout multiplier
    applied to layer l's outputs before layer l + 1 (train mode)."""
    B, T, _ = x.shape
    inp = x
    for l in range(layers):
        wi, wh = p[f"{prefix}.weight_ih_l{l}"], p[f"{prefix}.weight_hh_l{l}"]
        b = p[f"{prefix}.bias_ih_l{l}"] + p[f"{prefix}.bias_hh_l{l}"]
        h = torch.zeros(B, H, dtype=x.dtype)
        c = torch.zeros(B, H, dtype=x.dtype)
        outs = []
        for t in range(T):
            gates = inp[:, t] @ wi.T + h @ wh.T + b
            i, f, g, o = gates.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            if lengths is not None:
                live = (t < lengths).to(x.dtype).unsqueeze(1)
                outs.append(h * live)               # the state itself is never read again beyond the length
            else:
                outs.append(h)
        out = torch.stack(outs, dim=1)
        if masks is not None and l + 1 < layers and masks[l] is not None:
            out_next = out * masks[l]
        else:
            out_next = out
        inp = out_next
    return out


def generator_forward(pg, noise, labels, U, tau, layers=3):
    """Generator.forward (wgan.py:63-71) with the Gumbel-softmax sampler (wgan_gp.py:56-66), soft sample."""
    B, T = noise.shape
    lab = pg["label_embedding.weight"][labels]
    x = torch.cat([pg["noise_embedding.weight"][noise], lab.unsqueeze(1).repeat(1, T, 1)], -1)
    out = lstm(pg, "lstm", x, layers)
    logits = out @ pg["linear.weight"].T + pg["linear.bias"]
    u = 1e-10 + (1 - 1e-10) * U
    g = -torch.log(-torch.log(u + 1e-20) + 1e-20)
    return torch.softmax((logits + g) / tau, dim=-1)


def critic_forward(pd, seq, labels, lengths, layers=2, masks=None):
    """Discriminator.forward (wgan.py:106-122): seq int64 [B, T] (lookup) or float [B, T, V] (matmul)."""
    E = pd["sequence_embedding.embed_layer.weight"]
    emb = E[seq] if seq.dtype == torch.long else seq @ E                     # RelaxedEmbedding, :39-45
    T = emb.shape[1]
    lab = pd["label_embedding.weight"][labels]
    x = torch.cat([emb, lab.unsqueeze(1).repeat(1, T, 1)], -1)
    out = lstm(pd, "lstm", x, layers, lengths, masks)[:, :int(lengths.max())]   # pad_packed_sequence: longest length
    a = torch.tanh(out @ pd["attention.attn.weight"].T + pd["attention.attn.bias"])
    scores = (a @ pd["attention.context_vector.weight"].T).squeeze(-1)       # :84
    w = torch.softmax(scores, dim=1)                                         # :86 (padded steps take part)
    pooled = (out * w.unsqueeze(-1)).sum(1)                                  # :88
    return pooled @ pd["linear.weight"].T + pd["linear.bias"]                # :118


class Adam:
    def __init__(self, params, lr, b1, b2, eps=1e-8):
        self.p, self.lr, self.b1, self.b2, self.eps, self.t = params, lr, b1, b2, eps, 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, grads):
        self.t += 1
        for k, g in grads.items():
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            mh = self.m[k] / (1 - self.b1 ** self.t)
            vh = self.v[k] / (1 - self.b2 ** self.t)
            self.p[k] = self.p[k] - self.lr * mh / (vh.sqrt() + self.eps)


def critic_step(pd, opt, pg, real, labels, lengths, noise, U, tau, clip, masks_real=None, masks_fake=None):
    """wgan.py:174-198: D(real) on tokens, G(noise).detach(), D(fake) with lengths 100, Wasserstein loss, Adam,
    clamp.  Returns (d_loss, grads)."""
    leaves = {k: v.clone().requires_grad_(True) for k, v in pd.items()}
    real_validity = critic_forward(leaves, real, labels, lengths, masks=masks_real)
    with torch.no_grad():
        fake = generator_forward(pg, noise, labels, U, tau)
    fake_lengths = torch.full((labels.shape[0],), noise.shape[1], dtype=torch.long)
    fake_validity = critic_forward(leaves, fake, labels, fake_lengths, masks=masks_fake)
    d_loss = -(real_validity.mean() - fake_validity.mean())
    names = list(leaves)
    grads = dict(zip(names, torch.autograd.grad(d_loss, [leaves[n] for n in names])))
    opt.step(grads)
    for k in pd:
        opt.p[k] = opt.p[k].clamp(-clip, clip)
        pd[k] = opt.p[k]
    return d_loss.detach(), grads
