#!/usr/bin/env python3
"""Shader-clock stamps of the phases of s2g_k_graph_features for graph 0 of a 64-graph launch (development aid).

Needs a build of the library with the stamps compiled in:

    cd sequence-to-graph-gan_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
        -mllvm -amdgpu-mfma-vgpr-form=1 -DGR_STAMPS s2g_lib.hip -o ../../tools/_ubench/libs2g_stamps.so
    S2G_LIB=$PWD/tools/_ubench/libs2g_stamps.so python tools/graph_phases.py [out.json]

Lane 0 of every wave writes clock64() at the phase boundaries into the (otherwise unused) Katz workspace and adds
up, per elimination step, its time in the pivot owners' section, in the trailing update and at the barrier.  Lane 0
sits in the first half-wave, so the owners' section is recorded on every other step (printed doubled)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequence-to-graph-gan_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import common  # noqa: E402
from s2g import lib  # noqa: E402
from s2g.ops import ptr, stream_ptr  # noqa: E402

V, N, L = 175, 64, 512
out = {}
for name, skew in (("uniform", False), ("skewed", True)):
    tok, _ = common.graph_token_batch(77, N, L, vocab=V, skew=skew)
    tokd = tok.cuda()
    feat = torch.empty(N, V, 4, dtype=torch.float64, device="cuda")
    info = torch.empty(N, 4, dtype=torch.int32, device="cuda")
    work = torch.zeros(N * V * V, dtype=torch.float64, device="cuda")
    for _ in range(2):
        work.zero_()
        lib.call("s2g_graph_features", ptr(tokd), None, N, L, V, ptr(feat), ptr(info), ptr(work), work.numel() * 8, stream_ptr())
        torch.cuda.synchronize()
    w = work[:256].cpu().view(16, 16)
    n = int(info[0, 0])
    if float(w[0, 6]) == 0.0:
        raise SystemExit("no stamps found: build the library with -DGR_STAMPS and point S2G_LIB at it")
    r0, rl = w[0], w[15]
    res = {"nodes": n,
           "build": float(r0[1] - r0[0]), "pagerank_wave0": float(r0[2] - r0[1]),
           "brandes_slowest_wave": float(max(w[k, 2] - w[k, 1] for k in range(1, 16))),
           "solver_init": float(rl[4] - rl[3]), "solve": float(r0[5] - r0[4]), "tail": float(r0[6] - r0[5]),
           "per_step_owner_section": 2 * float(w[:, 8].sum()) / n, "per_step_trailing_update": float(w[:, 9].mean()) / n,
           "per_step_barrier_wait": float(w[:, 10].mean()) / n, "unit": "shader-clock ticks"}
    out[name] = res
    print(name, json.dumps(res))
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
