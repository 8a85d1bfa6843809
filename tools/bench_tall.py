#!/usr/bin/env python3
"""Development aid: the tall products of the critic's input layer (s2g_gemm_tall) against the generic
GEMM and torch, at the north-star shape (32768 rows)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequence-to-graph-gan_amd"))

import torch  # noqa: E402

from s2g import ops  # noqa: E402
from s2g.ops import ptr  # noqa: E402


def timeit(fn, reps=30):
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
    M = 32768
    for name, K, N, ldb in (("x @ W_ih0^T", 175, 200, 200), ("delta @ M2", 200, 200, 200), ("delta @ W_ih0", 200, 175, 176)):
        A = torch.randn(M, K, device=dev)
        B = torch.zeros(K, ldb, device=dev)
        B[:, :N] = torch.randn(K, N, device=dev)
        C1 = torch.empty(M, N, device=dev)
        C2 = torch.empty(M, N, device=dev)
        t_tall = timeit(lambda: ops.gemm_tall(ptr(A), K, ptr(B), ldb, ptr(C1), N, M, N, K))
        t_gen = timeit(lambda: ops.gemm_raw(ptr(A), K, 1, ptr(B), ldb, 1, ptr(C2), N, M, N, K))
        ref = A.double() @ B[:, :N].double()
        e1 = ((C1.double() - ref).abs().max() / ref.abs().max()).item()
        e2 = ((C2.double() - ref).abs().max() / ref.abs().max()).item()
        fl = 2.0 * M * N * K
        print(f"{name:16s} tall {t_tall:6.1f} us ({fl / t_tall * 1e-6:5.1f} TF, err {e1:.1e})   generic {t_gen:6.1f} us (err {e2:.1e})")
    # ragged M
    M2, K, N = 1000, 175, 200
    A = torch.randn(M2, K, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M2, N, device=dev)
    ops.gemm_tall(ptr(A), K, ptr(B), N, ptr(C), N, M2, N, K)
    print("ragged M err", ((C - A @ B).abs().max() / (A @ B).abs().max()).item())


if __name__ == "__main__":
    main()
