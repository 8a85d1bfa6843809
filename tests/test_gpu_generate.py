"""The consumer of the training step's output, generate_fake_data.py: a checkpoint written the way
train_wgan writes it (wgan_gp.py:350-352) is reloaded the way load_model does (generate_fake_data.py:24-30),
put in eval mode and sampled as generator(noise, labels, tau, hard).argmax(-1) (:60-66).  The token lists
must be those the reference's own load_model + Generator produced from the same checkpoint, noise and
uniforms (g10), through the module's forward and through the fused kernel (Generator.sample_tokens /
s2g_generate_tokens).

Fixture note (recorded, not fixed): fetch_graphs (generate_fake_data.py:77-79) looks the generated INT
tokens up in a vocabulary whose keys are STRINGS, so GraphEncoder.build_vocabulary (graph_encoder.py:30-37)
mints a new id >= 175 for every distinct generated token; g10 holds the ids minted for sequence 0."""
import pytest
import torch

import common

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _checkpoint(G):
    sd = dict(common.generator_params(1101, 175, 120))
    for k, v in G.items():
        if k.startswith("bn."):
            sd[k[3:]] = v
    return sd


def test_generate_fake_data_flow_reproduces_reference_tokens(golden, tmp_path):
    import wgan_gp as W
    G = golden("g10_generate_fake_data.pt")
    # a generator as train_wgan leaves it, saved as wgan_gp.py:350-352 saves it
    trained = W.Generator(175, 2).to(DEV)
    trained.load_state_dict(_checkpoint(G))
    path = tmp_path / "wgan_generator_5.pth"
    torch.save(trained.state_dict(), path)
    assert set(torch.load(path, weights_only=True)) == set(_checkpoint(G))        # exactly the reference's keys
    # generate_fake_data.py:24-30, 58-66
    model = W.Generator(vocab_size=175)
    model.load_state_dict(torch.load(path))
    model = model.to(DEV)
    model.eval()
    noise, labels = G["noise"].to(DEV), G["labels"].to(DEV)
    n = noise.shape[0]
    for tau in (0.5, 1.0):
        for hard in (False, True):
            want = G[f"tokens.tau{tau}.hard{int(hard)}"]
            U = torch.rand((n, 120, 175), generator=common.gen(int(G[f"U.tau{tau}.hard{int(hard)}.seed"])))
            real_rand = torch.rand
            torch.rand = lambda *a, **k: U.clone()             # the draw of wgan_gp.py:58
            try:
                with torch.no_grad():
                    gen_data = model(noise, labels, tau, hard).argmax(-1)
            finally:
                torch.rand = real_rand
            fake_data = gen_data.cpu().detach().tolist()         # generate_fake_data.py:74
            assert fake_data == want.tolist(), f"module path, tau {tau}, hard {hard}"
            fused = model.sample_tokens(noise, labels, uniforms=U.to(DEV))
            assert fused.dtype == torch.int64 and tuple(fused.shape) == (n, 120)
            assert fused.cpu().tolist() == want.tolist(), f"fused kernel, tau {tau}, hard {hard}"
    # eval mode: the BatchNorm buffers were used, not updated
    assert torch.equal(model.state_dict()["model.3.running_mean"].cpu(), G["bn.model.3.running_mean"])
    assert int(model.state_dict()["model.3.num_batches_tracked"]) == 37
    # the recorded vocabulary quirk: ids minted for ints start at the vocabulary size
    assert int(G["quirk.minted_ids"].min()) == 175
    assert G["quirk.minted_ids"].numel() == int(G["quirk.sequence0_distinct_tokens"])


@pytest.mark.parametrize("B,L", [(64, 512), (5, 7)])
def test_fused_tokens_with_in_kernel_draws_match_the_sampler(B, L):
    """s2g_generate_tokens with its own Philox draws == argmax of the unfused path (output layer, in-kernel
    Gumbel-softmax with the same seed / offset, argmax) at the north-star shape and at a ragged one."""
    from s2g import ops
    from s2g.generator import GeneratorEngine
    eng = GeneratorEngine(175, L, torch.device(DEV))
    eng.load(common.generator_params(1201, 175, L))
    g = common.gen(1202)
    z = torch.randn(B, 100, generator=g).to(DEV)
    labels = torch.randint(0, 2, (B,), generator=g).to(DEV)

    class Fixed:
        def take(self):
            return 99, 5
    eng.rng = Fixed()
    tok = eng.generate_tokens(z, labels)
    for hard in (False, True):
        y = eng.forward(z, labels, None, 0.5, hard, training=False, save=False)
        want = ops.argmax_rows(y)
        torch.cuda.synchronize()
        assert torch.equal(tok, want), f"hard {hard}: {(tok != want).sum().item()} of {tok.numel()} tokens differ"
    assert int(tok.min()) >= 0 and int(tok.max()) < 175
