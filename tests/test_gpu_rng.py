"""Device-side draws (csrc/rng.hip) through the C ABI against the numpy Philox restatement, itself pinned
by the Random123 known answers (tests/test_philox_cpu.py): the uniform fill bit for bit, the in-kernel
Gumbel-softmax against the injected-U sampler fed the same uniforms, the dropout mask, the normal fill."""
import numpy as np
import pytest
import torch

import common
from oracle import philox

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SEED = 0x1234_5678_9ABC_DEF1


class FixedRng:
    """a DeviceRng whose next draw uses a chosen offset"""

    def __init__(self, seed, offset):
        self.seed, self.offset = seed, offset

    def take(self):
        return self.seed, self.offset


@pytest.mark.parametrize("n,offset", [(4096, 0), (1001, 5), (3, 2 ** 40 + 17)])
def test_uniform_fill_is_the_philox_stream(n, offset):
    from s2g import ops
    out = torch.full((n,), -1.0, device=DEV)
    ops.rng_fill(out, 0, FixedRng(SEED, offset))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), philox.fill_uniform(n, offset, SEED))


def test_normal_fill_is_box_muller_of_the_stream():
    from s2g import ops
    n = 1 << 18
    out = torch.empty(n, device=DEV)
    ops.rng_fill(out, 1, FixedRng(SEED, 3))
    torch.cuda.synchronize()
    u = philox.fill_uniform(n, 3, SEED).astype(np.float64).reshape(-1, 4)
    r0, r1 = np.sqrt(-2 * np.log(1 - u[:, 0])), np.sqrt(-2 * np.log(1 - u[:, 2]))
    a0, a1 = 2 * np.pi * u[:, 1], 2 * np.pi * u[:, 3]
    want = np.stack([r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)], 1).reshape(-1)
    got = out.double().cpu().numpy()
    assert np.abs(got - want).max() < 1e-4          # fp32 log / sin / cos against float64
    assert abs(got.mean()) < 0.01 and abs(got.std() - 1.0) < 0.01
    assert abs(((got - got.mean()) ** 4).mean() / got.var() ** 2 - 3.0) < 0.05      # kurtosis of a normal


@pytest.mark.parametrize("hard", [False, True])
def test_in_kernel_gumbel_softmax_equals_injected_uniforms(hard):
    """s2g_gumbel_softmax_fwd_rng == s2g_gumbel_softmax_fwd fed the uniforms the kernel draws itself."""
    from s2g import ops
    rows, V = 4 * 37 + 3, 175
    logits = (torch.randn(rows, V, generator=common.gen(5)) * 2).to(DEV)
    y_rng, soft_rng = ops.gumbel_softmax_fwd(logits, None, 0.5, 0, hard, rng=FixedRng(SEED, 9))
    U = torch.from_numpy(philox.sampler_uniforms(rows, V, 9, SEED)).to(DEV)
    y_inj, soft_inj = ops.gumbel_softmax_fwd(logits, U, 0.5, 0, hard)
    torch.cuda.synchronize()
    assert torch.equal(y_rng, y_inj) and torch.equal(soft_rng, soft_inj)
    assert torch.allclose(soft_rng.sum(-1), torch.ones(rows, device=DEV), atol=1e-5)
    # a second launch with another offset gives another sample
    y2, _ = ops.gumbel_softmax_fwd(logits, None, 0.5, 0, hard, rng=FixedRng(SEED, 10))
    assert not torch.equal(y2, y_rng)


def test_dropout_mask_is_thresholded_philox_stream():
    from s2g import ops
    rows, p = 1000, 0.3
    m = ops.dropout_mask_rng(rows, p, DEV, FixedRng(SEED, 21))
    torch.cuda.synchronize()
    u = philox.fill_uniform(rows * 52, 21, SEED).reshape(rows, 52)
    want = np.where(u >= np.float32(p), np.float32(1.0 / (1.0 - p)), np.float32(0.0)).astype(np.float32)
    want[:, 50:] = 0.0
    got = m.cpu().numpy()
    assert np.array_equal(got != 0, want != 0)
    assert np.allclose(got, want, rtol=1e-6)
    keep = (got[:, :50] != 0).mean()
    assert abs(keep - 0.7) < 0.01


def test_trainer_draws_are_reproducible_from_the_seed():
    """Two trainers with the same seed draw the same z / uniforms / alpha / masks: same parameters after a
    full step without any injected noise; another seed gives another trajectory."""
    from s2g.trainer import WGANGPTrainer, class_weight_sum
    outs = []
    for seed in (7, 7, 8):
        tr = WGANGPTrainer(175, 12, DEV, drop_rate=0.3, seed=seed)
        tr.G.load(common.generator_params(31, 175, 12)); tr.D.load(common.critic_params(32, scale=1.5))
        batches = []
        for i in range(5):
            tok, lab, ln = common.batch(40 + i, 4, 16, min_len=2)
            tok = torch.nn.functional.pad(tok, (0, 16 - tok.shape[1]))
            batches.append((tok.to(torch.int32).to(DEV).contiguous(), lab.to(DEV), ln.to(torch.int32).to(DEV),
                            class_weight_sum(lab)))
        tr.D.overlap = False
        tr.full_step(batches)
        torch.cuda.synchronize()
        outs.append((tr.D.params.flat.clone(), tr.G.params.flat.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][0], outs[2][0])
