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


def assert_close(a, b, tol, what="", atol=0.0):
    """Relative L2 check; `atol` (absolute L2 slack per sqrt(numel)) covers tensors that are
    mathematically zero (e.g. the gradient of a bias that feeds a batch-norm)."""
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    err = float((a - b).norm())
    lim = tol * float(b.norm()) + atol * (b.numel() ** 0.5)
    assert err <= lim, f"{what}: L2 error {err:.3e} > {lim:.3e} (rel {err / (float(b.norm()) + 1e-30):.3e}, tol {tol:.1e})"
