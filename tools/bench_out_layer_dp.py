#!/usr/bin/env python3
"""Times the output Linear's weight-gradient product dl2^T a3 at the global batch sizes of 1, 2, 4 and 8 ranks
(64 rows per rank) and the flat Adam step that follows it: the compute side of exchanging the gradient's factors
instead of the gradient.  Development aid."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequence-to-graph-gan_amd"))

import torch  # noqa: E402

from s2g import ops  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
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
    N, K = 175 * 512, 512
    sc = ops.Scratch(dev)
    dW = torch.empty(N, K, device=dev)
    W, m, v = (torch.randn(N * K, device=dev) * 0.01 for _ in range(3))
    v.abs_()
    for M in (64, 128, 256, 512):
        dl2 = torch.randn(M, N, device=dev) * 0.01
        a3 = torch.randn(M, K, device=dev)
        t = timeit(lambda: ops.linear_bwd_dw(dl2, a3, dW, sc))
        print(f"rows {M}: product {t:.0f} us ({2 * M * N * K / t / 1e6:.1f} TFLOP/s)")
    t = timeit(lambda: ops.adam(W, dW.view(-1), m, v, 2e-4, 0.5, 0.999, 1e-8, 3))
    print(f"flat Adam {t:.0f} us")


if __name__ == "__main__":
    main()
