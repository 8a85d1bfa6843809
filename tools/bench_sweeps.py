#!/usr/bin/env python3
"""Times the four LSTM sweeps in isolation (HIP events) at the north-star shape:
A/B on 3*64 sequences, C/D on 64 sequences, 512 steps.  Development aid for kernel tuning."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequence-to-graph-gan_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402

import common  # noqa: E402
from s2g.critic import CriticEngine, make_desc  # noqa: E402
from s2g.ops import ptr  # noqa: E402


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


_FLUSH = None


def timeit_cold(fn, reps=10):
    """Like timeit, but a 1 GiB fill between launches evicts the operands from L2 and the memory-side
    cache first (inside a training step a sweep's inputs were written long before it reads them)."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(1 << 28, dtype=torch.float32, device="cuda:0")
    tot = 0.0
    for _ in range(reps):
        _FLUSH.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


def main():
    global timeit
    if "--cold" in sys.argv:
        timeit = timeit_cold
    dev = torch.device("cuda:0")
    B, T, V = 64, 512, 175
    for a in sys.argv[1:]:
        if a.startswith("--T="):
            T = int(a[4:])
    use_mask = "--mask" in sys.argv
    S = T                                   # workspace rows per sequence (--stride=N: padded, N >= T)
    for a in sys.argv[1:]:
        if a.startswith("--stride="):
            S = int(a[9:])
    eng = CriticEngine(V, dev)
    eng.load(common.critic_params(802))
    ws = eng.workspace(3 * B, S, B)
    P = torch.randn(B * T, 200, device=dev) * 0.3
    tok = torch.randint(0, V, (B, T), device=dev, dtype=torch.int32)
    mask = None
    if use_mask:
        mask = (torch.rand(3 * B * S, 52, device=dev) > 0.3).float() / 0.7
    ent = []
    for b in range(B):
        ent.append(dict(len=T, tok_off=b * T, ca=1.0, cp=0.0, ws_row=b * S, mask_off=b * S if use_mask else -1))
    for b in range(B):
        ent.append(dict(len=T, p_off=b * T, p_len=T, ca=0.0, cp=1.0, ws_row=(B + b) * S,
                        mask_off=(B + b) * S if use_mask else -1))
    for b in range(B):
        ent.append(dict(len=T, tok_off=b * T, p_off=b * T, p_len=T, ca=0.3, cp=0.7, ws_row=(2 * B + b) * S,
                        mask_off=(2 * B + b) * S if use_mask else -1))
    for b in range(B):
        ent.append(dict(len=T, tok_off=b * T, p_off=b * T, p_len=T, ca=0.3, cp=0.7, ws_row=b * S,
                        mask_off=b * S if use_mask else -1))
    desc = make_desc(ent, dev)
    ws["dhT"].normal_()
    ws["Pv"].normal_()
    r0 = 2 * B * S
    mrel = ptr(mask, r0 * 52) if mask is not None else None
    res = {}
    res["fwd_192"] = timeit(lambda: eng.sweep_fwd(ws, ptr(desc), 3 * B, T, tok, P, mask))
    res["fwd_64"] = timeit(lambda: eng.sweep_fwd(ws, ptr(desc), B, T, tok, P, mask))
    res["bwd_192"] = timeit(lambda: eng.sweep_bwd(ws, ptr(desc), 3 * B, S, mask, adj_seq0=2 * B))
    res["bwd_64"] = timeit(lambda: eng.sweep_bwd(ws, ptr(desc), B, S, mask))
    res["tangent_64"] = timeit(lambda: eng.sweep_tangent(ws, ptr(desc, 3 * B * 8), B, mrel, r0))
    res["bwd2_64"] = timeit(lambda: eng.sweep_bwd2(ws, ptr(desc, 3 * B * 8), B, S, mrel, r0))
    if "--tall" in sys.argv:
        from s2g import ops
        fake = torch.rand(B * T, V, device=dev)
        Pf = torch.empty(B * T, 200, device=dev)
        G = torch.empty(B * T, V, device=dev)
        tall = {}
        tall["project"] = timeit(lambda: eng.project(fake, Pf))
        tall["delta_m2"] = timeit(lambda: ops.gemm_tall(ptr(ws["delta0"], r0 * 200), 200, ptr(eng.m2()), 200,
                                                        ptr(ws["Pv"]), 200, B * T, 200, 200))
        tall["delta_w0"] = timeit(lambda: ops.gemm_tall(ptr(ws["delta0"], r0 * 200), 200, ptr(eng.W0p), eng.Vp,
                                                        ptr(G), V, B * T, V, 200))
        print("tall products:", " ".join(f"{k}={v:.1f}us" for k, v in tall.items()))
    if "--wgrad" in sys.argv:
        from s2g import critic as C
        G4 = 200
        d0, d1, c0, c1 = ws["delta0"].normal_(), ws["delta1"].normal_(), ws["cell0"].normal_(), ws["cell1"].normal_()
        a0, a1 = ws["abar0"].normal_(), ws["abar1"].normal_()
        x = torch.rand(B * T, V, device=dev)
        n12, n3, oi = 2 * B * T, B * T, r0 * 200
        gw = [torch.empty(200, n, device=dev) for n in (V, 50, 50, 50)]
        for phases in ((2,), (1,), (0,), (3,)):
            def tm(A, K, Bp, sbk, sbn, segT, shift, ph):
                return (A, K, Bp, sbk, sbn, segT, segT, segT, shift, ph)
            jobs = [(gw[0], V, []), (gw[1], 50, []), (gw[2], 50, []), (gw[3], 50, [])]
            if 2 in phases:
                jobs[0][2].append(tm(ptr(a0), n3, ptr(x), V, 1, T, False, 2))
                jobs[1][2].append(tm(ptr(a0), n3, ptr(c0, oi + 2), G4, 4, T, True, 2))
                jobs[2][2].append(tm(ptr(a1), n3, ptr(c0, oi + 3), G4, 4, 0, False, 2))
                jobs[3][2].append(tm(ptr(a1), n3, ptr(c1, oi + 2), G4, 4, T, True, 2))
            if 3 in phases:                     # the wide product alone
                jobs[0][2].append(tm(ptr(a0), n3, ptr(x), V, 1, T, False, 3))
            if 1 in phases:
                jobs[1][2].append(tm(ptr(d0, oi), n3, ptr(c0, 1), G4, 4, T, True, 1))
                jobs[2][2].append(tm(ptr(d1, oi), n3, ptr(c0, 2), G4, 4, 0, False, 1))
                jobs[3][2].append(tm(ptr(d1, oi), n3, ptr(c1, 1), G4, 4, T, True, 1))
            if 0 in phases:
                for k in range(3):
                    jobs[0][2].append(tm(ptr(d0, k * n3 * 200), n3, ptr(x), V, 1, T, False, 0))
                jobs[1][2].append(tm(ptr(d0), n12, ptr(c0, 2), G4, 4, T, True, 0))
                jobs[2][2].append(tm(ptr(d1), n12, ptr(c0, 3), G4, 4, 0, False, 0))
                jobs[3][2].append(tm(ptr(d1), n12, ptr(c1, 2), G4, 4, T, True, 0))
            built = C._wgrad_build(eng, jobs, ("bench", phases))
            t = timeit(lambda: C._wgrad_launch(eng, built, phases[0]))
            tr = timeit(lambda: C._wgrad_reduce(eng, built))
            print(f"wgrad phase {phases[0]}: gemm {t:.1f}us blocks={built[0][phases[0]][3]} reduce {tr:.1f}us")
    print(f"T={T} per-step ns:", " ".join(f"{k}={v * 1e3 / (T + 2):.0f}" for k, v in res.items()))
    print(" ".join(f"{k}={v:.1f}us" for k, v in res.items()), f"sum4={res['fwd_192'] + res['bwd_192'] + res['tangent_64'] + res['bwd2_64']:.1f}us")


if __name__ == "__main__":
    main()
