#!/usr/bin/env python3
"""Times the generator's output-layer update at the north-star shape (batch 64, 512 -> 89 600): the
weight-gradient GEMM followed by the flat Adam, against the fused s2g_linear_wgrad_adam.  Development aid."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequence-to-graph-gan_amd"))
import torch  # noqa: E402

from s2g import ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    B, N, K = 64, 89600, 512
    dy = torch.randn(B, N, device=dev) * 1e-3
    a = torch.randn(B, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    m, v, g = torch.zeros_like(W), torch.zeros_like(W), torch.zeros_like(W)
    sc = ops.Scratch(dev)

    def unfused():
        ops.linear_bwd_dw(dy, a, g, sc)
        ops.adam(W.view(-1), g.view(-1), m.view(-1), v.view(-1), 2e-4, 0.5, 0.999, 1e-8, 3)

    nbytes = W.numel() * 4
    for name, fn, passes in (("gemm + adam", unfused, 10),
                             ("fused", lambda: ops.linear_wgrad_adam(dy, a, W, m, v, 2e-4, 0.5, 0.999, 1e-8, 3), 6),
                             ("fused, 192 CUs", lambda: ops.linear_wgrad_adam(dy, a, W, m, v, 2e-4, 0.5, 0.999, 1e-8, 3, workgroups=192), 6),
                             ("fused + grad out", lambda: ops.linear_wgrad_adam(dy, a, W, m, v, 2e-4, 0.5, 0.999, 1e-8, 3, grad=g), 7)):
        us = timeit(fn)
        print(f"{name:18s} {us:7.1f} us  {passes * nbytes / us / 1e6:6.2f} TB/s over {passes} passes of {nbytes / 1e6:.0f} MB")


if __name__ == "__main__":
    main()
