#!/usr/bin/env python3
"""Development aid: builds an instrumented copy of the library (shader-clock stamps inside the layer-0
cell wave of the forward sweep and the layer-1 cell wave of the second reverse sweep), runs one
64-sequence sweep of each and prints where a step's cycles go.

    python tools/stamp_fwd.py build            # on the build host (hipcc): writes tools/_stamp/libs2g_stamp.so
    python tools/stamp_fwd.py run [out.json]   # on the GPU box
"""
import ctypes
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sequence-to-graph-gan_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_stamp")

STAMP_DEF = '''
__device__ unsigned long long s2g_stamps[128 * 8];
#define S2G_STAMP(k, dep)                                                                   \\
  {                                                                                         \\
    asm volatile("s_nop 0" ::"v"(dep));                                                     \\
    const unsigned long long c_ = clock64();                                                \\
    if (blockIdx.x == 0 && threadIdx.x == 1 && t >= 200 && t < 328) s2g_stamps[(t - 200) * 8 + (k)] = c_; \\
  }
'''


def build():
    shutil.rmtree(OUT, ignore_errors=True)
    shutil.copytree(SRC, OUT)
    p = os.path.join(OUT, "lstm_fwd.hip")
    s = open(p).read()
    s = s.replace('#define PF 6\n', '#define PF 6\n' + STAMP_DEF, 1)
    def ins(anchor, text, after=False):
        nonlocal s
        assert anchor in s, anchor
        s = s.replace(anchor, anchor + text if after else text + anchor, 1)
    ins("          float hv[28];\n          load_slice28(h0buf", "          S2G_STAMP(0, c)\n")
    ins("          const float cpt = (t < sq.p_len) ? cp : 0.0f;\n", "          S2G_STAMP(1, hv[27])\n")
    ins("          const float v0 = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(kact * a.x)), actA, actB);\n"
        "          const float v1 = fast_sigmoid(a.y);\n          const float gi = from_even(v0), gf = from_even(v1);      // lane 1",
        "          S2G_STAMP(2, a.x)\n")
    ins("          if (owner) {\n            h0buf[(t & 1) * HV + hp] = h;\n", "          S2G_STAMP(3, y)\n")
    ins("          sc = make_float4(c, tc, h, y);\n          __syncthreads();\n", "          S2G_STAMP(4, c)\n", after=True)
    open(p, "w").write(s)
    # sweep D, layer-1 cell wave (wave 0, lane 1 = an owner lane): slots 8..12 of the same table
    p = os.path.join(OUT, "lstm_bwd.hip")
    s = open(p).read()
    s = s.replace('#define PF 6\n', '#define PF 6\n' + STAMP_DEF.replace("s2g_stamps", "s2g_stamps_d")
                  .replace("S2G_STAMP", "S2G_STAMP_D"), 1)
    ins("            float hb = matvecT_oct(w01, w23, dbuf + ((t + 1) & 1) * DVEC + 28 * s);\n            if (!top) hb +=",
        "            if (top) S2G_STAMP_D(0, cb)\n")
    ins("            // ---- the part behind the mat-vec\n", "            if (top) S2G_STAMP_D(1, hb)\n")
    ins("            const float4 d = make_float4(d01.x, d01.y, d23.x, d23.y);\n", "            if (top) S2G_STAMP_D(2, d.x)\n", after=True)
    ins("              db[p0] = d.x; db[p1] = d.y; db[p2] = d.z; db[p3] = d.w;\n            }\n          }\n          __syncthreads();\n        }\n        ISSUE_D(j, i + PFD)",
        "")
    s = s.replace("              db[p0] = d.x; db[p1] = d.y; db[p2] = d.z; db[p3] = d.w;\n            }\n          }\n          __syncthreads();\n        }\n        ISSUE_D(j, i + PFD)",
                  "              db[p0] = d.x; db[p1] = d.y; db[p2] = d.z; db[p3] = d.w;\n            }\n            if (top) S2G_STAMP_D(3, cb)\n          }\n          __syncthreads();\n          if (top && t >= 0 && t < len) S2G_STAMP_D(4, cb)\n        }\n        ISSUE_D(j, i + PFD)", 1)
    open(p, "w").write(s)
    lib = os.path.join(OUT, "s2g_lib.hip")
    s = open(lib).read()
    s += ('\nextern "C" int s2g_read_stamps(unsigned long long* host) {\n'
          '  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(s2g_stamps), sizeof(unsigned long long) * 128 * 8);\n}\n'
          'extern "C" int s2g_read_stamps_d(unsigned long long* host) {\n'
          '  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(s2g_stamps_d), sizeof(unsigned long long) * 128 * 8);\n}\n')
    open(lib, "w").write(s)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
                    "-amdgpu-mfma-vgpr-form=1", "s2g_lib.hip", "-o", "libs2g_stamp.so"], cwd=OUT, check=True)
    for f in os.listdir(OUT):
        if not f.endswith(".so"):
            os.remove(os.path.join(OUT, f))


def run(out_json=None):
    sys.path.insert(0, os.path.join(ROOT, "sequence-to-graph-gan_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import json
    import numpy as np
    import torch
    from s2g import lib
    lib.LIB_PATH = os.path.join(OUT, "libs2g_stamp.so")
    lib._LIB = lib.load(lib.LIB_PATH)
    import common
    from s2g.critic import CriticEngine, make_desc
    from s2g.ops import ptr
    dev = torch.device("cuda:0")
    B, T, V = 64, 512, 175
    eng = CriticEngine(V, dev)
    eng.load(common.critic_params(802))
    ws = eng.workspace(B, T, B)
    tok = torch.randint(0, V, (B, T), device=dev, dtype=torch.int32)
    mask = (torch.rand(B * T, 52, device=dev) > 0.3).float() / 0.7
    P = torch.randn(B * T, 200, device=dev) * 0.3
    ent = [dict(len=T, tok_off=b * T, p_off=b * T, p_len=T, ca=0.3, cp=0.7, ws_row=b * T, mask_off=b * T)
           for b in range(B)]
    desc = make_desc(ent, dev)
    ws["dhT"].normal_()
    ws["Pv"].normal_()

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3

    def table(getter, names):
        buf = (ctypes.c_ulonglong * (128 * 8))()
        rc = getattr(lib._LIB, getter)(buf)
        assert rc == 0, rc
        a = np.array(buf[:], dtype=np.int64).reshape(128, 8)[:, :5]
        a = a[np.argsort(a[:, 0])]                  # the reverse sweep fills the table back to front
        d = np.diff(a, axis=1)
        nxt = a[1:, 0] - a[:-1, 4]                   # after the barrier -> next step's top
        step = a[1:, 0] - a[:-1, 0]
        vals = list(d[1:-1].mean(axis=0)) + [nxt[1:].mean()]
        return float(step[1:].mean()), {n: float(v) for n, v in zip(names, vals)}

    report = {"note": "shader-clock ticks (clock64(), s_memtime) read by one lane of one cell wave "
                      "of workgroup 0 at five points of every step, steps 200..327 of a 512-step sweep over 64 sequences "
                      "with dropout masks; the instrumented build is slower than the product (extra scalar waits)"}
    us = timed(lambda: eng.sweep_fwd(ws, ptr(desc), B, T, tok, P, mask))
    step, parts = table("s2g_read_stamps", ["lds read of h(t-1) + deferred stores + ring refill", "x + mat-vec (50 v_pk_fma) + 2-lane reduce",
                                             "activations + cell update", "lds write of h, y + barrier", "loop back"])
    report["s2g_k_lstm2_fwd / layer-0 cell wave"] = {"instrumented_sweep_us": us, "ns_per_step": us * 1e3 / (T + 2),
                                                     "ticks_per_step": step, "ticks_by_phase": parts}
    eng.sweep_bwd(ws, ptr(desc), B, T, mask, adj_seq0=0)
    eng.sweep_tangent(ws, ptr(desc), B, ptr(mask), 0)
    us = timed(lambda: eng.sweep_bwd2(ws, ptr(desc), B, T, ptr(mask), 0))
    step, parts = table("s2g_read_stamps_d", ["operand prep before the mat-vec result is needed + transposed mat-vec (50 v_pk_fma) + quad reduce",
                                               "second-order cell backward behind the mat-vec", "lds write of the gate gradients",
                                               "barrier", "ring refill + loop back"])
    report["s2g_k_lstm2_bwd2 / layer-1 cell wave"] = {"instrumented_sweep_us": us, "ns_per_step": us * 1e3 / (T + 2),
                                                      "ticks_per_step": step, "ticks_by_phase": parts}
    print(json.dumps(report, indent=1))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run(sys.argv[2] if len(sys.argv) > 2 else None)
