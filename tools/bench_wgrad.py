#!/usr/bin/env python3
"""Times the grouped weight-gradient GEMM of one critic iteration in isolation (HIP events) at
the north-star shape (B=64, T=512, V=175).  Development aid for kernel tuning."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequence-to-graph-gan_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402

import common  # noqa: E402
from s2g import critic as C  # noqa: E402
from s2g.ops import ptr  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, T, V = 64, 512, 175
    G4, H = 200, 50
    eng = C.CriticEngine(V, dev)
    eng.load(common.critic_params(802))
    ws = eng.workspace(3 * B, T, B)
    for k in ("delta0", "delta1", "cell0", "cell1", "abar0", "abar1", "celld0", "celld1", "G"):
        ws[k].normal_()
    xreal = torch.randn(B * T, V, device=dev)
    xhat = torch.randn(B * T, V, device=dev)
    fake = torch.randn(B, T, V, device=dev)
    g = eng.grads.views
    d0, d1, c0, c1 = ws["delta0"], ws["delta1"], ws["cell0"], ws["cell1"]
    a0, a1, cd0, cd1 = ws["abar0"], ws["abar1"], ws["celld0"], ws["celld1"]
    Tw = Tf = T
    oi = 2 * B * Tw * G4
    n12, n3 = 2 * B * Tw, B * Tw
    jobs = [(g["lstm.weight_ih_l0"], V, [
        (ptr(d0), B * T, ptr(xreal), V, 1, T, Tw, T, False),
        (ptr(d0, B * Tw * G4), B * Tf, ptr(fake), V, 1, Tf, Tw, Tf, False),
        (ptr(a0), B * T, ptr(xhat), V, 1, T, Tw, T, False),
        (ptr(d0, oi), n3, ptr(ws["G"]), V, 1, 0, 0, 0, False)]),
            (g["lstm.weight_hh_l0"], H, [
        (ptr(d0), n12, ptr(c0, 2), G4, 4, Tw, Tw, Tw, True),
        (ptr(a0), n3, ptr(c0, oi + 2), G4, 4, Tw, Tw, Tw, True),
        (ptr(d0, oi), n3, ptr(cd0, 1), G4, 4, Tw, Tw, Tw, True)]),
            (g["lstm.weight_ih_l1"], H, [
        (ptr(d1), n12, ptr(c0, 3), G4, 4, 0, 0, 0, False),
        (ptr(a1), n3, ptr(c0, oi + 3), G4, 4, 0, 0, 0, False),
        (ptr(d1, oi), n3, ptr(cd0, 2), G4, 4, 0, 0, 0, False)]),
            (g["lstm.weight_hh_l1"], H, [
        (ptr(d1), n12, ptr(c1, 2), G4, 4, Tw, Tw, Tw, True),
        (ptr(a1), n3, ptr(c1, oi + 2), G4, 4, Tw, Tw, Tw, True),
        (ptr(d1, oi), n3, ptr(cd1, 1), G4, 4, Tw, Tw, Tw, True)])]

    def timeit(sub, key):
        def run():
            C._wgrad_grouped(eng, sub, (key, B, T, Tf))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    print(f"W_ih0 only (N=175): {timeit(jobs[:1], 'b0'):.1f} us;  three N=50 jobs: {timeit(jobs[1:], 'b1'):.1f} us")
    if "--side" in sys.argv:       # the same launch on the CU-masked second stream (3/4 of the CUs)
        side = eng.side()
        if side is None:
            print("CU-masked stream refused by the runtime")
        else:
            torch.cuda.synchronize()
            with torch.cuda.stream(side[0]):
                print(f"all jobs on the masked stream: {timeit(jobs, 'bench'):.1f} us")
            torch.cuda.synchronize()
    us = timeit(jobs, "bench")
    flops = 2.0 * G4 * (4 * V * n3 + 3 * H * (n12 + 2 * n3))
    print(f"wgrad grouped + reduces: {us:.1f} us  ({flops / us * 1e-6:.1f} TFLOP/s algorithmic)")
    # reference check of one product against torch
    ref = (d0[:n3].view(B, Tw, G4)[:, :T].reshape(-1, G4).T @ xreal
           + d0[n3:2 * n3].T @ fake.view(-1, V) + a0.T @ xhat + d0[2 * n3:].T @ ws["G"].view(-1, V))
    perm = torch.tensor([(j & 3) * H + (j >> 2) for j in range(G4)], device=dev)
    out = torch.empty_like(ref)
    out[perm] = ref
    err = (g["lstm.weight_ih_l0"] - out).abs().max().item() / out.abs().max().item()
    print(f"max rel err vs torch: {err:.2e}")


if __name__ == "__main__":
    main()
