"""The CPU restatement of the weight-clipping WGAN's critic update (oracle/wgan_clip_cpu.py) against what the
reference's own CWGAN.train_step produced (g12): loss, gradients of every update, clamped parameters."""
import torch

import common
from conftest import assert_close
from oracle import wgan_clip_cpu as O


def test_oracle_critic_updates_match_reference_golden(golden):
    G = golden("g12_wgan_clip.pt")
    V, B, T = 175, 6, 40
    pd, pg = common.wgan_clip_params(1401, V)
    opt = O.Adam(pd, 2e-4, 0.5, 0.999)
    for it in range(3):
        tokens, labels, lengths = common.batch(1410 + it, B, T, vocab=V, min_len=4)
        g = common.gen(1420 + it)
        noise = torch.randint(0, V, (B, 100), generator=g)
        U = torch.rand((B, 100, V), generator=g)
        d_loss, grads = O.critic_step(pd, opt, pg, tokens, labels, lengths, noise, U, 0.5, 0.1)
        assert abs(float(d_loss) - float(G[f"d_loss{it}"])) < 1e-6
        for n, gv in grads.items():
            assert_close(gv, G[f"grad{it}.{n}"], 1e-4, f"grad{it} {n}", atol=1e-9)
    for n, v in pd.items():
        assert_close(v, G[f"D.{n}"], 1e-5, n)
        assert float(v.abs().max()) <= 0.1 + 1e-7
