import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "sequence-to-graph-gan_amd"), GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=True)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def assert_update_close(new, old, ref_new, tol, what=""):
    """Compares what an optimizer step MOVED (new - old against ref_new - old): one Adam step changes every
    element by about lr, ~1e-3 of a parameter's norm, so comparing the parameters themselves at 1e-3 would
    accept an update of the wrong sign."""
    new, old, ref_new = (torch.as_tensor(t).detach().double().cpu() for t in (new, old, ref_new))
    du, du_ref = new - old, ref_new - old
    assert float(du_ref.norm()) > 0, f"{what}: the reference did not move this tensor"
    err = float((du - du_ref).norm()) / float(du_ref.norm())
    assert err <= tol, f"{what}: update differs by {err:.3e} relative (tol {tol:.1e})"


# Generator biases that feed a BatchNorm layer: their gradient is mathematically zero (the batch mean is
# subtracted right after), what arrives is rounding noise and Adam turns it into an implementation-dependent
# drift; update comparisons skip them.
NOISE_ONLY = ("model.2.bias", "model.5.bias")


def assert_close(a, b, tol, what="", atol=0.0, elem=True):
    """Two bounds.  (1) Relative L2: |a - b|_2 <= tol |b|_2 (+ `atol` absolute L2 slack per sqrt(numel), for tensors
    that are mathematically zero, e.g. the gradient of a bias that feeds a batch-norm).  (2) Element by element:
    |a_i - b_i| <= tol |b_i| + tol rms(b) (+ 5 atol): the norm alone would accept a tensor whose small entries are
    wrong by 100 % as long as its large entries are right; here an entry may be off by `tol` of its own size plus
    `tol` of the tensor's root-mean-square entry (sums of thousands of fp32 terms taken in another order differ by
    rounding that scales with the tensor, not with the entry)."""
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    err = float((a - b).norm())
    lim = tol * float(b.norm()) + atol * (b.numel() ** 0.5)
    assert err <= lim, f"{what}: L2 error {err:.3e} > {lim:.3e} (rel {err / (float(b.norm()) + 1e-30):.3e}, tol {tol:.1e})"
    if elem and b.numel():
        rms = float(b.norm()) / (b.numel() ** 0.5)
        bound = tol * b.abs() + (tol * rms + 5.0 * atol)       # atol is an rms-level slack: 5 sigma per entry
        excess = ((a - b).abs() - bound).reshape(-1)
        worst = int(excess.argmax())
        assert float(excess[worst]) <= 0.0, (
            f"{what}: element {worst}: {float(a.reshape(-1)[worst]):.6e} vs {float(b.reshape(-1)[worst]):.6e} "
            f"(|diff| {float((a - b).abs().reshape(-1)[worst]):.3e} > {float(bound.reshape(-1)[worst]):.3e}; "
            f"tensor rms {rms:.3e}, tol {tol:.1e})")
