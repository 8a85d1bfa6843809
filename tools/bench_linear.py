#!/usr/bin/env python3
"""Times the generator's output Linear (64 x 512 -> 89600) through the generic GEMM and through the
weight-streaming small-batch kernel.  Development aid."""
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


dev = "cuda:0"
x = torch.randn(64, 512, device=dev)
W = torch.randn(89600, 512, device=dev) * 0.05
b = torch.randn(89600, device=dev)
y = torch.empty(64, 89600, device=dev)
print(f"generic GEMM {timeit(lambda: ops.linear_fwd(x, W, b, y)):.1f} us, "
      f"weight-streaming kernel {timeit(lambda: ops.linear_fwd_small_batch(x, W, b, y)):.1f} us "
      f"(183.5 MB of weights / 6 TB/s = 31 us)")
o = torch.empty(1, device=dev)
print(f"reference points: W.sum() {timeit(lambda: torch.sum(W, dim=None, out=o) if False else W.sum()):.1f} us, "
      f"W.clone() {timeit(lambda: W.clone()):.1f} us (read + write 183.5 MB each)")
