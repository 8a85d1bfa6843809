#!/usr/bin/env python3
"""Times the HBM-bound kernels of the step ALONE (HIP events, 20 launches each, operands far larger than the
caches) at the north-star shape and prints their rate against 8 TB/s using ALGORITHMIC bytes (what the kernel
must read and write once).  Inside the step these kernels run beside the LSTM sweeps on CU-masked streams and
share HBM with them; profiles/pmc_traffic.json has those in-step figures.

    python tools/bench_hbm_kernels.py [out.json]
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/bench_hbm_kernels.py
    python tools/bench_hbm_kernels.py --from-stats DIR/*/*_kernel_stats.csv out.json

The event timings of the first form include ~10-15 us of launch and event latency per call, which is most of
a 10 us kernel; the second and third forms take the kernels' own durations from the profiler's trace.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequence-to-graph-gan_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402

from s2g import ops  # noqa: E402
from s2g.lib import call  # noqa: E402
from s2g.ops import ptr, stream_ptr  # noqa: E402

DEV = torch.device("cuda:0")
B, T, V = 64, 512, 175
PEAK = 8000.0      # GB/s


def timeit(fn, reps=20):
    flush = torch.empty(1 << 28, dtype=torch.float32, device=DEV)      # 1 GiB: evicts L2 / the memory-side cache
    for _ in range(2):
        fn()
    tot = 0.0
    for _ in range(reps):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3      # us


ALGO = {   # kernel name in the profiler's trace -> (label, algorithmic bytes, note)
    "s2g_k_gumbel_softmax_fwd_v4<true>": ("sampler forward, in-kernel draws", 8 * B * T * V, "read logits, write y"),
    "s2g_k_gumbel_softmax_fwd_v4<false>": ("sampler forward, injected uniforms", 12 * B * T * V, "read logits + U, write y"),
    "s2g_k_gumbel_softmax_bwd_v4": ("sampler backward", 12 * B * T * V, "read y, dy; write dlogits"),
    "s2g_k_one_hot": ("one-hot operand", 4 * B * T * V, "write x"),
    "s2g_k_gp_interp": ("interpolation operand", 8 * B * T * V, "read fake, write x_hat"),
    "s2g_k_gp_norm_penalty": ("penalty norm + scale in place", 12 * B * T * V,
                              "read g, read g again + write v; one workgroup per sample: 64 of 256 CUs"),
    "s2g_k_gp_dots": ("penalty reduction", 8 * B * T * 200, "read delta, M2 delta"),
    "s2g_k_adam": ("Adam, 46 M parameters", 28 * 46_000_000, "read p, g, m, v; write p, m, v"),
    "s2g_k_dropout_mask_rng": ("dropout mask", 4 * 3 * B * T * 52, "write mask"),
    "s2g_k_linear_wgrad_adam": ("output layer: weight gradient + Adam, 45.9 M parameters", 24 * 89600 * 512,
                                "read W, m, v; write W, m, v (the gradient stays in registers / LDS)"),
}


def from_stats(path, out):
    import csv
    res = {}
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0].replace("void ", "").strip()
        if name in ALGO:
            label, nbytes, note = ALGO[name]
            ns = float(r["AverageNs"])
            res[name] = {"what": label, "avg_us": ns / 1e3, "min_us": float(r["MinNs"]) / 1e3, "launches": int(r["Calls"]),
                         "algorithmic_mb": nbytes / 1e6, "gb_per_s": nbytes / ns, "frac_of_hbm_peak": nbytes / ns / PEAK,
                         "note": note}
    for k, r in res.items():
        print(f"{k:40s} {r['avg_us']:8.1f} us  {r['gb_per_s']:7.0f} GB/s  {r['frac_of_hbm_peak']:.2f} of 8 TB/s")
    with open(out, "w") as f:
        json.dump({"source": "rocprofv3 --kernel-trace --stats of tools/bench_hbm_kernels.py: each kernel alone, cold caches "
                             "(a 1 GiB fill between launches), algorithmic bytes / average duration, peak 8000 GB/s", **res},
                  f, indent=1)


def main():
    if len(sys.argv) > 3 and sys.argv[1] == "--from-stats":
        return from_stats(sys.argv[2], sys.argv[3])
    n = B * T * V
    logits = torch.randn(B, T, V, device=DEV)
    U = torch.rand(B, T, V, device=DEV)
    y = torch.empty_like(logits)
    dy = torch.randn_like(logits)
    tokens = torch.randint(0, V, (B, T), device=DEV, dtype=torch.int32)
    lengths = torch.full((B,), T, device=DEV, dtype=torch.int32)
    alpha = torch.rand(B, device=DEV)
    xhat = torch.empty(B * T, V, device=DEV)
    G = torch.randn(B * T, V, device=DEV)
    pen, norms = torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    d0, pv = torch.randn(B * T, 200, device=DEV), torch.randn(B * T, 200, device=DEV)
    part = torch.empty(B, 4, device=DEV)
    npar = 46_000_000
    p, g, m, v = (torch.randn(npar, device=DEV) for _ in range(4))
    v.abs_()
    rng = ops.DeviceRng(1)
    mask = torch.empty(3 * B * T, 52, device=DEV)
    res = {}

    def rec(name, us, nbytes, note=""):
        res[name] = {"us": us, "algorithmic_mb": nbytes / 1e6, "gb_per_s": nbytes / us / 1e3,
                     "frac_of_hbm_peak": nbytes / us / 1e3 / PEAK, "note": note}

    rec("s2g_k_gumbel_softmax_fwd_v4<true> (in-kernel draws)",
        timeit(lambda: ops.gumbel_softmax_fwd(logits, None, 0.5, 0, False, out=y, rng=rng)), 8 * n, "read logits, write y")
    rec("s2g_k_gumbel_softmax_fwd_v4<false> (injected uniforms)",
        timeit(lambda: ops.gumbel_softmax_fwd(logits, U, 0.5, 0, False, out=y)), 12 * n, "read logits + U, write y")
    rec("s2g_k_gumbel_softmax_bwd_v4", timeit(lambda: ops.gumbel_softmax_bwd(y, dy, 2.0)), 12 * n, "read y, dy; write dlogits")
    rec("s2g_k_one_hot", timeit(lambda: call("s2g_one_hot", ptr(tokens), ptr(lengths), T, ptr(xhat), B * T, V, stream_ptr())),
        4 * n, "write x")
    rec("s2g_k_gp_interp", timeit(lambda: call("s2g_gp_interp", ptr(tokens), ptr(lengths), ptr(logits), ptr(alpha), B, T, T, V,
                                               ptr(xhat), stream_ptr())), 8 * n, "read fake, write x_hat")
    rec("s2g_k_gp_norm_penalty (scale in place)",
        timeit(lambda: call("s2g_gp_norm_penalty", ptr(G), B, T * V, 10.0, float(B), ptr(pen), ptr(norms), 1, stream_ptr())),
        12 * n, "read g (reduction), read g again + write v: one workgroup per sample, 64 of 256 CUs")
    rec("s2g_k_gp_dots", timeit(lambda: call("s2g_gp_dots", ptr(d0), ptr(pv), B, T * 200, ptr(part), stream_ptr())),
        8 * B * T * 200, "read delta, M2 delta")
    rec("s2g_k_adam (46 M parameters)", timeit(lambda: ops.adam(p, g, m, v, 2e-4, 0.5, 0.999, 1e-8, 3), reps=8), 28 * npar,
        "read p, g, m, v; write p, m, v")
    Wl, ml, vl = p[:89600 * 512], m[:89600 * 512], v[:89600 * 512]
    dyl, al = torch.randn(B, 89600, device=DEV) * 1e-3, torch.randn(B, 512, device=DEV)
    rec("s2g_k_linear_wgrad_adam (output layer, 45.9 M parameters)",
        timeit(lambda: ops.linear_wgrad_adam(dyl, al, Wl, ml, vl, 2e-4, 0.5, 0.999, 1e-8, 3), reps=8), 24 * 89600 * 512,
        "read W, m, v; write W, m, v")
    rec("s2g_k_dropout_mask_rng", timeit(lambda: ops.dropout_mask_rng(3 * B * T, 0.3, DEV, rng)), 4 * mask.numel(),
        "write mask (includes torch's allocation of the output)")
    for k, r in res.items():
        print(f"{k:58s} {r['us']:8.1f} us  {r['gb_per_s']:7.0f} GB/s  {r['frac_of_hbm_peak']:.2f} of 8 TB/s")
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
