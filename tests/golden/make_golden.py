"""Generate the golden vectors in tests/golden/*.pt by IMPORTING THE REFERENCE itself.

Runs only in the build container (the reference lives at /root/reference and never travels);
the fixtures it writes are data: inputs are regenerated from seeds by tests/golden/common.py,
expected outputs come from the unmodified reference classes/functions.

    python tests/golden/make_golden.py            # all small fixtures (~1 min)
    python tests/golden/make_golden.py --full     # also the B=64, L=512 records g8 (~2 min) and g9 (~5 min)

torch_geometric (an unrelated import of the reference's dataset module) is absent here, so an
empty stand-in module is registered before the import; no reference code is changed.
The reference draws its noise with torch.rand / rand / randn on the CPU generator
(wgan_gp.py:58,228,291); those three names are patched to pop pre-drawn tensors so the same
noise can be replayed by the oracle and by the HIP path.
"""
from __future__ import annotations

import argparse
import contextlib
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import common  # noqa: E402


def import_reference():
    for name in ("torch_geometric", "torch_geometric.data", "torch_geometric.utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torch_geometric.data"].Dataset = object
    sys.modules["torch_geometric.data"].Data = object
    sys.modules["torch_geometric.utils"].dropout_edge = lambda *a, **k: None
    sys.path.insert(0, "/root/reference")
    import wgan_gp as ref
    return ref


class Queue:
    def __init__(self):
        self.items = []

    def push(self, *ts):
        self.items.extend(ts)

    def pop(self, *shape_args, **kw):
        t = self.items.pop(0)
        return t.clone()


@contextlib.contextmanager
def injected(ref, q_rand_torch: Queue, q_rand: Queue, q_randn: Queue):
    """torch.rand -> Gumbel U (wgan_gp.py:58); ref.rand -> alpha (:228); ref.randn -> z (:291)."""
    old = (torch.rand, ref.rand, ref.randn)
    torch.rand = q_rand_torch.pop
    ref.rand = q_rand.pop
    ref.randn = q_randn.pop
    try:
        yield
    finally:
        torch.rand, ref.rand, ref.randn = old


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def load_critic(ref, p, vocab=175, drop=0.0):
    D = ref.Discriminator(vocab, 50, 2, 2, drop_rate=drop)
    D.load_state_dict(p)
    return D


def load_generator(ref, p, vocab, seq_len):
    G = ref.Generator(vocab, 2)
    if seq_len != G.seq_len:   # only reference code paths are used for the forward
        G.seq_len = seq_len
        G.model[-1] = torch.nn.Linear(512, vocab * seq_len)
    sd = dict(p)
    sd.update(common.bn_buffers())
    G.load_state_dict(sd)
    return G


def g1_sampler(ref):
    g = common.gen(101)
    logits = torch.randn((4, 8, 175), generator=g) * 2
    U = torch.rand((4, 8, 175), generator=g)
    dy = torch.randn((4, 8, 175), generator=g)
    out = {}
    sm = ref.CustomSoftmax()
    for tau in (0.5, 1.0, 5.0):
        for hard in (False, True):
            lg = logits.clone().requires_grad_(True)
            q = Queue(); q.push(U)
            with injected(ref, q, Queue(), Queue()):
                y = sm(lg, tau, hard)
            (dl,) = torch.autograd.grad(y, lg, dy)
            out[f"y_tau{tau}_hard{int(hard)}"] = y.detach()
            out[f"dlogits_tau{tau}_hard{int(hard)}"] = dl
    out["relaxed_tau0.7"] = ref.CustomSoftmax("relaxed_bernoulli")(logits, 0.7, False)
    out["plain"] = ref.CustomSoftmax("softmax")(logits, 0.7, False)
    out["anneal"] = torch.tensor([ref.CustomSoftmax.anneal_tau_(5.0, 0.5, e, 100, 10) for e in (0, 7, 10, 50)])
    save("g1_sampler.pt", out)


def g2_generator(ref):
    out = {}
    for seq_len in (8, 120):
        p = common.generator_params(202, 175, seq_len)
        G = load_generator(ref, p, 175, seq_len)
        G.train()
        nz = common.noise(203, 6, seq_len)
        labels = torch.tensor([0, 1, 1, 0, 1, 0])
        q = Queue(); q.push(nz["U"])
        with injected(ref, q, Queue(), Queue()):
            y = G(nz["z"], labels, 0.5, False)
        dy = torch.randn(y.shape, generator=common.gen(204))
        names = [n for n, _ in G.named_parameters()]
        grads = torch.autograd.grad(y, list(G.parameters()), dy)
        tag = f"T{seq_len}"
        out[f"{tag}.y"] = y.detach() if seq_len == 8 else y.detach()[:, :4].clone()
        for n, gr in zip(names, grads):
            if n == "model.8.weight" and seq_len == 120:
                out[f"{tag}.grad_norm.{n}"] = gr.norm()
                out[f"{tag}.grad_head.{n}"] = gr[:64].clone()
            else:
                out[f"{tag}.grad.{n}"] = gr
        for b in ("model.3.running_mean", "model.3.running_var", "model.6.running_mean", "model.6.running_var"):
            out[f"{tag}.{b}"] = G.state_dict()[b].clone()
        # eval-mode forward (generate_fake_data.py:61-66) after the one training call above
        G.eval()
        q = Queue(); q.push(nz["U2"])
        with injected(ref, q, Queue(), Queue()), torch.no_grad():
            tok = G(nz["z"], labels, 0.5, True).argmax(-1)
        out[f"{tag}.eval_tokens"] = tok
    save("g2_generator.pt", out)


def g3_critic(ref):
    out = {}
    p = common.critic_params(301, scale=2.0)
    D = load_critic(ref, p)
    D.train()
    g = common.gen(302)
    x = torch.softmax(torch.randn((4, 16, 175), generator=g) * 3, -1)
    dout = torch.randn((4, 3), generator=g)
    lengths = torch.tensor([16, 11, 7, 3])
    from torch.nn.utils.rnn import pack_padded_sequence
    for tag, use_len in (("dense", False), ("packed", True)):
        xin = x.clone().requires_grad_(True)
        inp = pack_padded_sequence(xin, lengths, batch_first=True, enforce_sorted=False) if use_len else xin
        o = D(inp)
        names = [n for n, _ in D.named_parameters()]
        grads = torch.autograd.grad(o, [xin] + list(D.parameters()), dout)
        out[f"{tag}.out"] = o.detach()
        out[f"{tag}.dx"] = grads[0]
        for n, gr in zip(names, grads[1:]):
            out[f"{tag}.grad.{n}"] = gr
    save("g3_critic.pt", out)


def g4_gp(ref):
    out = {}
    for scale, tag in ((1.0, "s1"), (3.0, "s3")):
        p = common.critic_params(401, scale=scale)
        D = load_critic(ref, p)
        D.train()
        tokens, labels, lengths = common.batch(402, 5, 12, lengths=[12, 12, 9, 4, 1])
        real = torch.nn.functional.one_hot(tokens, 175).float()
        g = common.gen(403)
        for Tf in (7, 12, 15):
            fake = torch.softmax(torch.randn((5, Tf, 175), generator=g) * 2, -1)
            alpha = torch.rand((5, 1, 1), generator=g)
            q = Queue(); q.push(alpha)
            with injected(ref, Queue(), q, Queue()):
                gp = ref.compute_gradient_penalty(D, real, fake, labels, lengths)
            names = [n for n, _ in D.named_parameters()]
            grads = torch.autograd.grad(10.0 * gp, list(D.parameters()), allow_unused=True)
            k = f"{tag}.Tf{Tf}"
            out[f"{k}.fake"] = fake
            out[f"{k}.alpha"] = alpha
            out[f"{k}.gp"] = gp.detach()
            for n, gr in zip(names, grads):
                if gr is not None:
                    out[f"{k}.grad.{n}"] = gr
    save("g4_gp.pt", out)


def g5_ce(ref):
    g = common.gen(501)
    logits = torch.randn((9, 2), generator=g).requires_grad_(True)
    labels = torch.tensor([0, 1, 1, 0, 0, 1, 1, 1, 0])
    crit = torch.nn.CrossEntropyLoss(torch.tensor([1.0, 2.0]))
    loss = crit(torch.softmax(logits, -1), labels)   # wgan_gp.py:268,299,310
    (dl,) = torch.autograd.grad(loss, logits)
    save("g5_ce.pt", {"logits": logits.detach(), "labels": labels, "loss": loss.detach(), "dlogits": dl})


def g6_adam(ref):
    g = common.gen(601)
    w = torch.randn((7, 5), generator=g).requires_grad_(True)
    w2 = torch.randn((3,), generator=g).requires_grad_(True)
    opt = torch.optim.Adam([w, w2], lr=2e-4, betas=(0.5, 0.999))
    out = {"w0": w.detach().clone(), "w20": w2.detach().clone()}
    for s in range(3):
        opt.zero_grad()
        gw = torch.randn((7, 5), generator=g) * (10.0 ** (s - 1))
        w.grad = gw.clone()
        out[f"gw{s}"] = gw
        if s != 1:                       # step 1: w2 has no grad -> Adam skips it
            gw2 = torch.randn((3,), generator=g)
            w2.grad = gw2.clone()
            out[f"gw2{s}"] = gw2
        else:
            w2.grad = None
        opt.step()
        out[f"w{s + 1}"] = w.detach().clone()
        out[f"w2{s + 1}"] = w2.detach().clone()
    save("g6_adam.pt", out)


class ListLoader(list):
    """A list that train_wgan can enumerate and len().  At B=64, L=512 one iteration's double-backward
    graph takes ~30 GiB of host memory and the loop's local names (gp, d_loss, fake_data, ...) keep the
    previous iteration's graph alive until they are rebound, i.e. while the next one is being built: two
    generations do not fit in this container.  Before a batch is handed out, the harness therefore
    unbinds those dead names in the caller's frame and runs the cycle collector.  No arithmetic of the
    reference is touched: every one of these names is assigned again before it is read."""
    DEAD = ("gp", "d_loss", "g_loss", "real_validity", "fake_validity", "fake_data", "out", "real_class_output",
            "fake_class_output", "d_class_loss", "g_class_loss", "packed_input", "real_data")

    def __iter__(self):
        import ctypes
        import gc
        for k, item in enumerate(list.__iter__(self)):
            f = sys._getframe(1)
            while f is not None and f.f_code.co_name != "train_wgan":
                f = f.f_back
            if f is not None and k > 0:
                for name in self.DEAD:
                    if name in f.f_locals:
                        f.f_locals[name] = None
                ctypes.pythonapi.PyFrame_LocalsToFast(ctypes.py_object(f), ctypes.c_int(0))
            gc.collect()
            if os.environ.get("GOLDEN_TRACE"):
                import resource
                print(f"[batch {k}] peak RSS so far {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.1f} GiB",
                      file=sys.__stderr__, flush=True)
            yield item


def run_reference_steps(ref, pg, pd, batches, noises, vocab, gen_seq_len, opt_ns):
    """Drive the reference's own train_wgan (wgan_gp.py:261-348) over the given batches."""
    G = load_generator(ref, pg, vocab, gen_seq_len)
    D = load_critic(ref, pd, vocab, drop=0.0)
    oG = torch.optim.Adam(G.parameters(), lr=opt_ns.lr, betas=(opt_ns.b1, opt_ns.b2))
    oD = torch.optim.Adam(D.parameters(), lr=opt_ns.lr, betas=(opt_ns.b1, opt_ns.b2))
    snaps = {"D": [], "G": []}

    def snap(tag, mod):
        def hook(opt, args, kwargs):
            snaps[tag].append({n: (None if q.grad is None else q.grad.detach().clone())
                               for n, q in mod.named_parameters()})
        return hook
    oD.register_step_pre_hook(snap("D", D))
    oG.register_step_pre_hook(snap("G", G))
    qU, qA, qZ = Queue(), Queue(), Queue()
    for i, nz in enumerate(noises):
        qZ.push(nz["z"]); qU.push(nz["U"]); qA.push(nz["alpha"])
        if i % opt_ns.n_critic == 0:
            qU.push(nz["U2"])
    import io
    buf = io.StringIO()
    with injected(ref, qU, qA, qZ), contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        ref.train_wgan(G, D, oG, oD, ListLoader(batches), ListLoader([]), opt_ns, torch.device("cpu"))
    return G, D, buf.getvalue(), snaps


def parse_losses(text):
    import re
    return [(float(a), float(b)) for a, b in re.findall(r"\[D loss: ([-\d.eE+naninf]+)\] \[G loss: ([-\d.eE+naninf]+)\]", text)]


def g7_full_step(ref):
    ns = argparse.Namespace(vocab_size=175, epochs=1, batch_size=8, lr=2e-4, b1=0.5, b2=0.999, latent_dim=100,
                            seq_len=32, n_critic=5, clip_value=0.01, lambda_gp=10.0, hard=False, tau=0.5)
    out = {}
    for Tf in (4, 32):
        pg = common.generator_params(701, 175, Tf)
        pd = common.critic_params(702, scale=1.5)
        batches = [common.batch(710 + i, 8, 32, min_len=3) for i in range(6)]
        noises = [common.noise(720 + i, 8, Tf) for i in range(6)]
        # 6 batches: generator updates at i = 0 and i = 5, so the second update sees 5 critic steps
        G, D, text, snaps = run_reference_steps(ref, pg, pd, batches, noises, 175, Tf, ns)
        k = f"Tf{Tf}"
        out[f"{k}.printed_losses"] = torch.tensor(parse_losses(text), dtype=torch.float64)
        for it in (0, 5):
            for n, v in snaps["D"][it].items():
                out[f"{k}.D.grad{it}.{n}"] = v
        for n, v in D.state_dict().items():
            out[f"{k}.D.{n}"] = v.clone()
        for n, v in G.state_dict().items():
            if n == "model.8.weight" and Tf == 32:
                out[f"{k}.G.delta_norm.{n}"] = (v - pg[n]).norm()
                out[f"{k}.G.head.{n}"] = v[:32].clone()
            else:
                out[f"{k}.G.{n}"] = v.clone()
    save("g7_full_step.pt", out)


def g8_full_size(ref):
    """One critic iteration + the generator update at the north-star shape: B=64, L=512, V=175,
    generator seq_len 512, fixed lengths and ADFA-shaped lengths.  Scalars and norms only."""
    import time
    ns = argparse.Namespace(vocab_size=175, epochs=1, batch_size=64, lr=2e-4, b1=0.5, b2=0.999, latent_dim=100,
                            seq_len=512, n_critic=5, clip_value=0.01, lambda_gp=10.0, hard=False, tau=0.5)
    out = {}
    for tag, lengths in (("fixed", [512] * 64), ("adfa", common.adfa_lengths(805, 64).tolist())):
        pg = common.generator_params(801, 175, 512)
        pd = common.critic_params(802)
        batches = [common.batch(810, 64, 512, lengths=lengths)]
        noises = [common.noise(820, 64, 512)]
        t0 = time.time()
        G, D, text, snaps = run_reference_steps(ref, pg, pd, batches, noises, 175, 512, ns)
        out[f"{tag}.seconds"] = torch.tensor(time.time() - t0)
        out[f"{tag}.printed_losses"] = torch.tensor(parse_losses(text), dtype=torch.float64)
        for n, v in snaps["D"][0].items():
            out[f"{tag}.D.grad.{n}"] = v
        for n, v in snaps["G"][0].items():
            if v.numel() > 200000:
                out[f"{tag}.G.grad_norm.{n}"] = v.norm()
                out[f"{tag}.G.grad_head.{n}"] = v.reshape(-1)[:4096].clone()
            else:
                out[f"{tag}.G.grad.{n}"] = v
        for n, v in G.state_dict().items():
            if "running" in n:
                out[f"{tag}.G.{n}"] = v.clone()
        print(tag, "reference seconds", float(out[f"{tag}.seconds"]), out[f"{tag}.printed_losses"])
    save("g8_full_size.pt", out)


def g9_full_step_full_size(ref):
    """One FULL step and the start of the next at the north-star shape through the reference's own
    train_wgan: 6 batches at B=64, L=512, V=175, generator seq_len 512, ADFA-shaped lengths that differ
    from batch to batch (5 critic updates + the generator update after the first, then the 6th batch with
    its critic and generator update; peaks at ~62 GiB of host memory).  Records the printed losses, every critic gradient of updates 0 and
    5, the final critic parameters, and heads / update norms / BatchNorm buffers of the generator."""
    import time
    ns = argparse.Namespace(vocab_size=175, epochs=1, batch_size=64, lr=2e-4, b1=0.5, b2=0.999, latent_dim=100,
                            seq_len=512, n_critic=5, clip_value=0.01, lambda_gp=10.0, hard=False, tau=0.5)
    pg = common.generator_params(901, 175, 512)
    pd = common.critic_params(902)
    batches = [common.batch(910 + i, 64, 512, lengths=common.adfa_lengths(930 + i, 64).tolist()) for i in range(6)]
    noises = [common.noise(920 + i, 64, 512) for i in range(6)]
    t0 = time.time()
    G, D, text, snaps = run_reference_steps(ref, pg, pd, batches, noises, 175, 512, ns)
    out = {"seconds": torch.tensor(time.time() - t0),
           "printed_losses": torch.tensor(parse_losses(text), dtype=torch.float64)}
    for it in range(6):
        for n, v in snaps["D"][it].items():
            if it in (0, 5):
                out[f"D.grad{it}.{n}"] = v
            out[f"D.grad_norm{it}.{n}"] = v.double().norm()
    for n, v in D.state_dict().items():
        out[f"D.{n}"] = v.clone()
    for n, v in G.state_dict().items():
        if v.numel() > 200000:
            out[f"G.delta_norm.{n}"] = (v - pg[n]).double().norm()
            # the LAST rows (time step 511): the critic's gradient reaches the early time steps of a
            # 512-step sequence vanishingly small, Adam leaves the first rows exactly where they were
            out[f"G.tail.{n}"] = v.reshape(-1)[-65536:].clone()
        elif "num_batches" not in n:
            out[f"G.{n}"] = v.clone()
    for it in range(2):
        for n, v in snaps["G"][it].items():
            out[f"G.grad_norm{it}.{n}"] = v.double().norm()
    print("g9: reference seconds", float(out["seconds"]), out["printed_losses"])
    save("g9_full_step_full_size.pt", out)


def g10_generate_fake_data(ref):
    """The consumer of the training step's output, generate_fake_data.py:24-30, 60-66, 74-79: a checkpoint
    written as torch.save(generator.state_dict(), ...) (wgan_gp.py:350-352) is loaded by the reference's
    own load_model into the reference's Generator(vocab_size=175) (seq_len 120), put in eval mode and
    sampled as generator(noise, labels, tau, hard).argmax(-1).tolist().  Records the token lists for hard
    False / True and two temperatures.  fetch_graphs itself needs torch_geometric (absent here); what IS
    recorded of it is the vocabulary quirk its first statement has (graph_encoder.py:30-37, 64): the
    generated tokens are ints while the loaded vocabulary's keys are strings, so build_vocabulary mints a
    new id (>= 175) for every distinct generated token instead of finding it."""
    import tempfile
    # seaborn (a plotting import of the reference's utils.py, pulled in through wgan.py) is absent here: an
    # empty stand-in module, like torch_geometric's; no reference code is changed
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    import generate_fake_data as gfd            # the reference's script (imports wgan / graph_encoder as well)
    pg = common.generator_params(1101, 175, 120)
    sd = dict(pg)
    bn = common.bn_buffers()
    g = common.gen(1102)
    # running statistics as a trained checkpoint has them (not the initial 0 / 1)
    bn["model.3.running_mean"] = torch.randn(256, generator=g) * 0.05
    bn["model.3.running_var"] = 1.0 + 0.2 * torch.rand(256, generator=g)
    bn["model.6.running_mean"] = torch.randn(512, generator=g) * 0.05
    bn["model.6.running_var"] = 1.0 + 0.2 * torch.rand(512, generator=g)
    bn["model.3.num_batches_tracked"] = torch.tensor(37)
    bn["model.6.num_batches_tracked"] = torch.tensor(37)
    sd.update(bn)
    n = 70                                        # more than one 64-sample launch of the fused kernel
    noise = torch.randn((n, 100), generator=g)
    labels = torch.zeros((n,), dtype=torch.long)  # generate_fake_data.py:115: class 0 = malware
    out = {"noise": noise, "labels": labels}
    for k, v in bn.items():
        out[f"bn.{k}"] = v
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "saved_models"))
        torch.save(sd, os.path.join(tmp, "saved_models", "wgan_generator_5.pth"))
        os.chdir(tmp)
        try:
            args = argparse.Namespace(vocab_size=175, num_classes=2, num_layers=3, softmax="gumbel_softmax")
            model = gfd.load_model("wgan_generator_5.pth", args)
        finally:
            os.chdir(cwd)
    model.eval()
    for tau in (0.5, 1.0):
        for hard in (False, True):
            U = torch.rand((n, 120, 175), generator=common.gen(1103 + int(hard) + int(tau * 10)))
            q = Queue(); q.push(U)
            with injected(ref, q, Queue(), Queue()), torch.no_grad():
                gen_data = model(noise, labels, tau, hard).argmax(-1)
            out[f"U.tau{tau}.hard{int(hard)}.seed"] = torch.tensor(1103 + int(hard) + int(tau * 10))
            out[f"tokens.tau{tau}.hard{int(hard)}"] = gen_data
    # the vocabulary quirk (recorded, not fixed)
    enc = gfd.GraphEncoder()
    enc.syscall_vocabs = {str(i): i for i in range(175)}        # what load_vocabulary returns: string keys
    seq = out["tokens.tau0.5.hard1"][0].tolist()
    enc.build_vocabulary(seq)
    minted = sorted(v for k, v in enc.syscall_vocabs.items() if isinstance(k, int))
    out["quirk.sequence0_distinct_tokens"] = torch.tensor(len(set(seq)))
    out["quirk.minted_ids"] = torch.tensor(minted)
    print("g10: distinct tokens in sequence 0:", len(set(seq)), "-> ids minted by build_vocabulary:",
          minted[:3], "...", minted[-1])
    save("g10_generate_fake_data.pt", out)


def g11_attention(ref):
    """MaskedMultiheadAttention (transformer.py:196-282) of the imported reference: outputs and the gradients of
    inputs and parameters for a given upstream gradient, without masks, with the reference's own
    create_causal_mask / create_padding_mask, and with batch_first=True."""
    import transformer as T_ref                   # /root/reference/transformer.py (first on sys.path)
    assert T_ref.__file__.startswith("/root/reference")
    out = {}
    for tag, (seed, N, T, E, H) in {"small": (1301, 3, 20, 32, 2), "selftest": (1302, 8, 50, 64, 4),
                                    "ragged": (1303, 2, 37, 64, 1)}.items():
        c = common.attention_case(seed, N, T, E, H)
        for mode in ("nomask", "masks", "batch_first"):
            m = T_ref.MaskedMultiheadAttention(E, H, batch_first=(mode == "batch_first"))
            with torch.no_grad():
                m.in_proj_weight.copy_(c["in_w"]); m.in_proj_bias.copy_(c["in_b"])
                m.fc_out.weight.copy_(c["out_w"]); m.fc_out.bias.copy_(c["out_b"])
            q, k, v = (c[n].clone().requires_grad_(True) for n in "qkv")
            am = T_ref.create_causal_mask(T) if mode != "nomask" else None
            pm = T_ref.create_padding_mask(c["tokens"]) if mode != "nomask" else None
            if mode == "batch_first":
                # the module transposes (0, 1) first: feed [T, N, E] so that the same problem results
                y = m(v.transpose(0, 1), k.transpose(0, 1), q.transpose(0, 1), attn_mask=am, src_key_padding_mask=pm)
                y = y.transpose(0, 1)
            else:
                y = m(v, k, q, attn_mask=am, src_key_padding_mask=pm)
            grads = torch.autograd.grad(y, [q, k, v, m.in_proj_weight, m.in_proj_bias, m.fc_out.weight, m.fc_out.bias],
                                        c["dy"])
            key = f"{tag}.{mode}"
            out[f"{key}.y"] = y.detach().clone()
            for n, gv in zip(("dq", "dk", "dv", "d_in_w", "d_in_b", "d_out_w", "d_out_b"), grads):
                out[f"{key}.{n}"] = gv.clone()
    save("g11_attention.pt", out)


def g12_wgan_clip(ref):
    """The critic update of the reference's weight-clipping WGAN (wgan.py:174-198) through its own classes:
    CWGAN.train_step over a ONE-batch loader (i = 0: the generator branch, which would raise AttributeError on
    the undefined self.loss, wgan.py:214, is skipped by the inverted test `if i % n_critic`, :206), three
    calls in a row on different batches.  Records d_loss per call (train_step returns it), the critic's
    gradients at every update and its parameters after the three clamped Adam steps."""
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    import wgan as W                                  # /root/reference/wgan.py
    assert W.__file__.startswith("/root/reference")
    V, B, T = 175, 6, 40
    pd, pg = common.wgan_clip_params(1401, V)
    opt = argparse.Namespace(lr=2e-4, beta1=0.5, beta2=0.999, clip_value=0.1, n_critic=5)
    G = W.Generator(V, 2, 3, softmax="gumbel_softmax")
    D = W.Discriminator(V, 2, 2, dropout=0.0, use_gpu=False)
    G.load_state_dict(pg); D.load_state_dict(pd)
    model = W.CWGAN(G, D, opt)
    snaps = []
    model.d_optimizer.register_step_pre_hook(
        lambda o, a, k: snaps.append({n: q.grad.detach().clone() for n, q in D.named_parameters()}))
    out = {}
    real_randint = torch.randint
    for it in range(3):
        tokens, labels, lengths = common.batch(1410 + it, B, T, vocab=V, min_len=4)
        g = common.gen(1420 + it)
        noise = real_randint(0, V, (B, 100), generator=g)
        U = torch.rand((B, 100, V), generator=g)
        qU = Queue(); qU.push(U)
        torch.randint = lambda *a, **k: noise.clone()             # wgan.py:183
        try:
            import contextlib, io
            with injected(ref, qU, Queue(), Queue()), contextlib.redirect_stderr(io.StringIO()):
                d_loss, _ = model.train_step(ListLoader([(tokens, labels, lengths)]), 0.5, False)
        finally:
            torch.randint = real_randint
        out[f"d_loss{it}"] = torch.tensor(d_loss, dtype=torch.float64)
        for n, v in snaps[it].items():
            out[f"grad{it}.{n}"] = v
    for n, v in D.state_dict().items():
        out[f"D.{n}"] = v.clone()
    print("g12: d_loss", [float(out[f"d_loss{i}"]) for i in range(3)])
    save("g12_wgan_clip.pt", out)


def g14_wgan_clip_full_size(ref):
    """One critic update of the reference's CWGAN.train_step (wgan.py:174-198) at the size BASELINE names for the
    weight-clipping variant: B = 128, L = 1024, V = 175, ragged lengths (the longest = 1024), no dropout (the
    masks of nn.LSTM's inter-layer dropout cannot be captured).  Records d_loss, every critic gradient of the
    update and the parameters after the clamped Adam step."""
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    import wgan as W                                  # /root/reference/wgan.py
    assert W.__file__.startswith("/root/reference")
    V, B, T = 175, 128, 1024
    pd, pg = common.wgan_clip_params(1801, V)
    opt = argparse.Namespace(lr=2e-4, beta1=0.5, beta2=0.999, clip_value=0.1, n_critic=5)
    G = W.Generator(V, 2, 3, softmax="gumbel_softmax")
    D = W.Discriminator(V, 2, 2, dropout=0.0, use_gpu=False)
    G.load_state_dict(pg); D.load_state_dict(pd)
    model = W.CWGAN(G, D, opt)
    snaps = []
    model.d_optimizer.register_step_pre_hook(
        lambda o, a, k: snaps.append({n: q.grad.detach().clone() for n, q in D.named_parameters()}))
    tokens, labels, lengths = common.batch(1810, B, T, vocab=V, min_len=64)
    g = common.gen(1820)
    real_randint = torch.randint
    noise = real_randint(0, V, (B, 100), generator=g)
    U = torch.rand((B, 100, V), generator=g)
    qU = Queue(); qU.push(U)
    torch.randint = lambda *a, **k: noise.clone()             # wgan.py:183
    import time
    t0 = time.perf_counter()
    try:
        import io
        with injected(ref, qU, Queue(), Queue()), contextlib.redirect_stderr(io.StringIO()):
            d_loss, _ = model.train_step(ListLoader([(tokens, labels, lengths)]), 0.5, False)
    finally:
        torch.randint = real_randint
    out = {"d_loss": torch.tensor(d_loss, dtype=torch.float64), "seconds": torch.tensor(time.perf_counter() - t0)}
    for n, v in snaps[0].items():
        out[f"grad.{n}"] = v
    for n, v in D.state_dict().items():
        out[f"D.{n}"] = v.clone()
    print("g14: d_loss", float(out["d_loss"]), "seconds", float(out["seconds"]))
    save("g14_wgan_clip_full_size.pt", out)


def g15_wgan_modules(ref):
    """The module surface of the reference's wgan.py beyond the critic update: RelaxedEmbedding.forward (:36-45,
    both branches, with gradients), Attention.forward (:82-89: pooled vector and weights, with gradients for an
    upstream on both), Discriminator.forward (:106-122) with gradients with respect to every parameter and to
    a relaxed input, and CWGAN.train (:230-246) with tau annealing over one-batch loaders (i = 0 only, so the
    broken generator branch is never entered)."""
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    import wgan as W
    assert W.__file__.startswith("/root/reference")
    V, B, T = 175, 5, 24
    pd, pg = common.wgan_clip_params(1901, V)
    g = common.gen(1902)
    out = {}
    # RelaxedEmbedding
    emb = W.RelaxedEmbedding(V, 32)
    emb.device = torch.device("cpu")
    emb.embed_layer = torch.nn.Embedding(V, 32)
    with torch.no_grad():
        emb.embed_layer.weight.copy_(pd["sequence_embedding.embed_layer.weight"])
    tok = torch.randint(0, V, (B, T), generator=g)
    soft = torch.softmax(torch.randn((B, T, V), generator=g), -1).requires_grad_(True)
    dy = torch.randn((B, T, 32), generator=g)
    y_tok = emb(tok)
    (gw_tok,) = torch.autograd.grad(y_tok, [emb.embed_layer.weight], dy)
    y_soft = emb(soft)
    gw_soft, gx_soft = torch.autograd.grad(y_soft, [emb.embed_layer.weight, soft], dy)
    out.update({"emb.tok": tok, "emb.soft": soft.detach().clone(), "emb.dy": dy, "emb.y_tok": y_tok.detach().clone(),
                "emb.y_soft": y_soft.detach().clone(), "emb.dW_tok": gw_tok, "emb.dW_soft": gw_soft, "emb.dx_soft": gx_soft})
    # Attention
    att = W.Attention(64)
    with torch.no_grad():
        att.attn.weight.copy_(pd["attention.attn.weight"]); att.attn.bias.copy_(pd["attention.attn.bias"])
        att.context_vector.weight.copy_(pd["attention.context_vector.weight"])
    hidden = torch.randn((B, T, 64), generator=g).requires_grad_(True)
    lo = torch.randn((B, T, 64), generator=g).requires_grad_(True)
    d_pool, d_w = torch.randn((B, 64), generator=g), torch.randn((B, T), generator=g)
    pooled, weights = att(hidden, lo)
    grads = torch.autograd.grad([pooled, weights], [hidden, lo, att.attn.weight, att.attn.bias, att.context_vector.weight],
                                [d_pool, d_w])
    out.update({"att.hidden": hidden.detach().clone(), "att.lstm_outputs": lo.detach().clone(), "att.d_pooled": d_pool,
                "att.d_weights": d_w, "att.pooled": pooled.detach().clone(), "att.weights": weights.detach().clone()})
    for n, gv in zip(("d_hidden", "d_lstm_outputs", "d_attn_w", "d_attn_b", "d_context"), grads):
        out[f"att.{n}"] = gv.clone()
    # Discriminator.forward, differentiable: tokens and a relaxed input
    D = W.Discriminator(V, 2, 2, dropout=0.0, use_gpu=False)
    D.load_state_dict(pd)
    tokens, labels, lengths = common.batch(1910, B, T, vocab=V, min_len=3)
    dval = torch.randn((B, 1), generator=g)
    names = [n for n, _ in D.named_parameters()]
    val = D(tokens, labels, lengths)
    gt = torch.autograd.grad(val, [q for _, q in D.named_parameters()], dval, allow_unused=True)
    out.update({"D.tokens": tokens, "D.labels": labels, "D.lengths": lengths, "D.dval": dval, "D.val_tok": val.detach().clone()})
    for n, gv in zip(names, gt):
        out[f"D.grad_tok.{n}"] = torch.zeros_like(dict(D.named_parameters())[n]) if gv is None else gv.clone()
    x = torch.softmax(torch.randn((B, T, V), generator=g) * 3.0, -1).requires_grad_(True)
    full = torch.full((B,), T, dtype=torch.long)
    val2 = D(x, labels, full)
    gs = torch.autograd.grad(val2, [x] + [q for _, q in D.named_parameters()], dval, allow_unused=True)
    out.update({"D.soft": x.detach().clone(), "D.val_soft": val2.detach().clone(), "D.dx_soft": gs[0].clone()})
    for n, gv in zip(names, gs[1:]):
        out[f"D.grad_soft.{n}"] = torch.zeros_like(dict(D.named_parameters())[n]) if gv is None else gv.clone()
    # CWGAN.train: three epochs over a one-batch loader with tau annealing every epoch
    opt = argparse.Namespace(lr=2e-4, beta1=0.5, beta2=0.999, clip_value=0.1, n_critic=5)
    G2 = W.Generator(V, 2, 3, softmax="gumbel_softmax")
    D2 = W.Discriminator(V, 2, 2, dropout=0.0, use_gpu=False)
    G2.load_state_dict(pg); D2.load_state_dict(pd)
    model = W.CWGAN(G2, D2, opt)
    noises = [torch.randint(0, V, (B, 100), generator=g) for _ in range(3)]
    Us = [torch.rand((B, 100, V), generator=g) for _ in range(3)]
    qU, qN = Queue(), Queue()
    qU.push(*Us); qN.push(*noises)
    real_randint = torch.randint
    torch.randint = qN.pop
    import io
    buf = io.StringIO()
    try:
        with injected(ref, qU, Queue(), Queue()), contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(buf):
            model.train(ListLoader([(tokens, labels, lengths)]), epochs=3, tau=0.5, hard=False, anneal_tau=True, anneal_interval=1)
    finally:
        torch.randint = real_randint
    out["train.noise"] = torch.stack(noises); out["train.U"] = torch.stack(Us)
    out["train.d_losses"] = torch.tensor(model.losses["d_loss"], dtype=torch.float64)
    out["train.g_losses"] = torch.tensor(model.losses["g_loss"], dtype=torch.float64)
    out["train.printed"] = buf.getvalue()
    for n, v in D2.state_dict().items():
        out[f"train.D.{n}"] = v.clone()
    print("g15: train printed:\n" + buf.getvalue())
    save("g15_wgan_modules.pt", out)


def g16_encoder_layer(ref):
    """CustomTransformerEncoderLayer (transformer.py:285-317) of the imported reference with dropout 0: the layer
    alone (batch_first False and True) with the reference's causal and padding masks -- output and the gradients
    of the input and of every parameter the forward uses -- and nn.TransformerEncoder stacking two of them in
    train mode, as the reference's own self-test does (:340-348)."""
    import transformer as T_ref
    assert T_ref.__file__.startswith("/root/reference")
    out = {}
    N, T, E, H, DFF = 4, 24, 64, 4, 128
    g = common.gen(1601)
    used = ("attn.in_proj_weight", "attn.in_proj_bias", "attn.fc_out.weight", "attn.fc_out.bias", "linear1.weight",
            "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")
    for mode in ("plain", "batch_first"):
        torch.manual_seed(1602)
        layer = T_ref.CustomTransformerEncoderLayer(E, H, DFF, dropout=0.0, batch_first=(mode == "batch_first"))
        with torch.no_grad():      # biases and LayerNorm parameters away from their 0 / 1 initial values
            for n, q in layer.named_parameters():
                if n.endswith("bias") or n.startswith("norm"):
                    q.add_(0.2 * torch.randn(q.shape, generator=g))
        layer.train()
        x = torch.randn((N, T, E), generator=g)
        tokens = torch.randint(0, 4, (N, T), generator=g)
        dy = torch.randn((N, T, E), generator=g)
        am, pm = T_ref.create_causal_mask(T), T_ref.create_padding_mask(tokens)
        xin = x.clone().requires_grad_(True)
        # in both modes dim 0 of the input is what the attention module treats as the batch: with batch_first the
        # layer's transpose (:299-300) and the attention module's own (:223-226) cancel
        y = layer(xin, src_mask=am, src_key_padding_mask=pm)
        params = dict(layer.named_parameters())
        grads = torch.autograd.grad(y, [xin] + [params[n] for n in used], dy)
        out[f"{mode}.state"] = {k: v.clone() for k, v in layer.state_dict().items()}
        out[f"{mode}.x"], out[f"{mode}.tokens"], out[f"{mode}.dy"] = x, tokens, dy
        out[f"{mode}.y"] = y.detach().clone()
        out[f"{mode}.dx"] = grads[0].clone()
        for n, gv in zip(used, grads[1:]):
            out[f"{mode}.grad.{n}"] = gv.clone()
    # the reference's self-test arrangement: nn.TransformerEncoder over two layers, train mode
    torch.manual_seed(1603)
    layer = T_ref.CustomTransformerEncoderLayer(E, H, DFF, dropout=0.0, batch_first=False)
    enc = torch.nn.TransformerEncoder(layer, num_layers=2)
    enc.train()
    x = torch.randn((N, T, E), generator=g)
    tokens = torch.randint(0, 4, (N, T), generator=g)
    dy = torch.randn((N, T, E), generator=g)
    xin = x.clone().requires_grad_(True)
    y = enc(xin, mask=T_ref.create_causal_mask(T), src_key_padding_mask=T_ref.create_padding_mask(tokens))
    (dx,) = torch.autograd.grad(y, [xin], dy)
    out["stack.state"] = {k: v.clone() for k, v in enc.state_dict().items()}
    out["stack.x"], out["stack.tokens"], out["stack.dy"] = x, tokens, dy
    out["stack.y"], out["stack.dx"] = y.detach().clone(), dx.clone()
    print("g16: |y|", float(out["plain.y"].norm()), float(out["batch_first.y"].norm()), float(out["stack.y"].norm()))
    save("g16_encoder_layer.pt", out)


def g13_graph_encoder(ref):
    """GraphEncoder.encode (graph_encoder.py:57-133) on the sequences of common.graph_sequences with one encoder
    (its vocabulary grows across calls), plus the float64 centralities from the same networkx calls the reference
    makes (graph_encoder.py:86-91) on the same graph, and the exception the reference raises on a sequence whose
    Katz system is singular (ten self-transitions: 1 - 0.1 * 10 = 0 on an otherwise empty row)."""
    class Data:      # stand-in for torch_geometric.data.Data (absent here): keeps the fields, validates nothing
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def validate(self, raise_on_error=True):
            return True

    sys.modules["torch_geometric.data"].Data = Data
    import graph_encoder as ge
    import networkx as nx
    ge.Data = Data           # the module may already have been imported (by wgan_gp) with the empty stand-in
    enc = ge.GraphEncoder()
    out = {"networkx": nx.__version__}
    for name, seq in common.graph_sequences():
        graph, mapping = enc.encode(seq)
        out[f"{name}.x"] = graph.x.clone()
        out[f"{name}.edge_index"] = graph.edge_index.clone()
        out[f"{name}.edge_features"] = graph.edge_features.clone()
        out[f"{name}.num_nodes"] = graph.num_nodes
        out[f"{name}.mapping_keys"] = list(mapping.keys())
        out[f"{name}.mapping_values"] = list(mapping.values())
        n = graph.num_nodes
        f64 = torch.zeros(n, 4, dtype=torch.float64)
        if n > 1:
            G = nx.DiGraph()
            for i in range(len(seq) - 1):
                u, v = mapping[seq[i]], mapping[seq[i + 1]]
                if G.has_edge(u, v):
                    G[u][v]["weight"] += 1
                else:
                    G.add_edge(u, v, weight=1)
            b, c, p = nx.betweenness_centrality(G), nx.closeness_centrality(G), nx.pagerank(G, weight="weight")
            k = nx.katz_centrality_numpy(G, weight="weight") if n > 2 else {i: 0 for i in G}
            for i in range(n):
                f64[i] = torch.tensor([k[i], b[i], c[i], p[i]], dtype=torch.float64)
        out[f"{name}.f64"] = f64
    out["vocab_keys"] = list(enc.syscall_vocabs.keys())
    out["vocab_values"] = list(enc.syscall_vocabs.values())
    singular = [1] * 11 + [2, 3]
    try:
        ge.GraphEncoder().encode(singular)
        out["singular.raises"] = ""
    except Exception as e:      # noqa: BLE001
        out["singular.raises"] = type(e).__name__
    try:
        ge.GraphEncoder().encode([])
        out["empty.raises"] = ""
    except Exception as e:      # noqa: BLE001
        out["empty.raises"] = f"{type(e).__name__}: {e}"
    print("g13:", {k: v for k, v in out.items() if k.endswith("raises") or k.endswith("num_nodes")})
    save("g13_graph_encoder.pt", out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    ref = import_reference()
    todo = [g1_sampler, g2_generator, g3_critic, g4_gp, g5_ce, g6_adam, g7_full_step, g10_generate_fake_data, g11_attention, g12_wgan_clip, g13_graph_encoder, g15_wgan_modules, g16_encoder_layer]
    if a.full:
        todo += [g8_full_size, g9_full_step_full_size, g14_wgan_clip_full_size]
    for fn in todo:
        if a.only and a.only not in fn.__name__:
            continue
        fn(ref)


if __name__ == "__main__":
    main()
