#!/usr/bin/env python3
"""Development aid: builds an instrumented copy of the library (shader-clock stamps inside the layer-0
cell wave of the forward sweep), runs one 64-sequence sweep and prints where a step's cycles go.

    python tools/stamp_fwd.py build     # on the build host (hipcc): writes tools/_stamp/libs2g_stamp.so
    python tools/stamp_fwd.py run       # on the GPU box
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
    lib = os.path.join(OUT, "s2g_lib.hip")
    s = open(lib).read()
    s += ('\nextern "C" int s2g_read_stamps(unsigned long long* host) {\n'
          '  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(s2g_stamps), sizeof(unsigned long long) * 128 * 8);\n}\n')
    open(lib, "w").write(s)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
                    "-amdgpu-mfma-vgpr-form=1", "s2g_lib.hip", "-o", "libs2g_stamp.so"], cwd=OUT, check=True)
    for f in os.listdir(OUT):
        if not f.endswith(".so"):
            os.remove(os.path.join(OUT, f))


def run():
    sys.path.insert(0, os.path.join(ROOT, "sequence-to-graph-gan_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
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
    ws = eng.workspace(3 * B, T, B)
    tok = torch.randint(0, V, (B, T), device=dev, dtype=torch.int32)
    mask = (torch.rand(3 * B * T, 52, device=dev) > 0.3).float() / 0.7
    P = torch.randn(B * T, 200, device=dev) * 0.3
    ent = [dict(len=T, tok_off=b * T, p_off=b * T, p_len=T, ca=0.3, cp=0.7, ws_row=b * T, mask_off=b * T)
           for b in range(B)]
    desc = make_desc(ent, dev)
    for _ in range(3):
        eng.sweep_fwd(ws, ptr(desc), B, T, tok, P, mask)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.sweep_fwd(ws, ptr(desc), B, T, tok, P, mask)
    e1.record()
    torch.cuda.synchronize()
    print(f"instrumented sweep: {e0.elapsed_time(e1) * 1e3:.1f} us = {e0.elapsed_time(e1) * 1e6 / (T + 2):.0f} ns per step")
    buf = (ctypes.c_ulonglong * (128 * 8))()
    rc = lib._LIB.s2g_read_stamps(buf)
    assert rc == 0, rc
    import numpy as np
    a = np.array(buf[:], dtype=np.int64).reshape(128, 8)[:, :5]
    d = np.diff(a, axis=1)
    nxt = a[1:, 0] - a[:-1, 4]                   # after the barrier -> next step's top
    step = a[1:, 0] - a[:-1, 0]
    names = ["lds read + deferred stores + refill (until the slice landed)", "x + mat-vec + reduce", "activations + cell",
             "lds write + barrier", "loop"]
    vals = list(d[1:-1].mean(axis=0)) + [nxt[1:].mean()]
    print("cycles per step:", step[1:].mean())
    for n, v in zip(names, vals):
        print(f"  {n:18s} {v:8.1f}")


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
