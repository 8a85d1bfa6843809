"""The PRODUCT data-parallel path: two fresh processes (ranks 0 and 1, both on GPU 0, gloo between them),
each running WGANGPTrainer(comm=DistComm()) on its 8-sequence shard of global batches of 16, against ONE
process on the whole batches: critic and generator gradients, loss values, parameters after a full step
through the overlapped schedule, BatchNorm buffers and Adam moments.  Differences come from the order of
fp32 additions only (per-rank partial sums, all-reduced), hence the tight tolerances."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(rank, world, port, out, **extra):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), OUT=out, PYTHONFAULTHANDLER="1", **extra)
    return subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "dp_worker.py")], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def _finish(procs, limit):
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=limit)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("data-parallel worker did not finish:\n" + "\n".join(logs))
    for p in procs:
        assert p.returncode == 0, "\n---- next rank ----\n".join(logs)      # every rank's output: a stall needs both stacks
    return logs


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_two_ranks_match_one_rank_at_the_global_batch(tmp_path):
    # the parent never touches the GPU: the children are fresh processes
    port = _free_port()
    outs = [str(tmp_path / f"rank{r}.pt") for r in range(2)]
    _finish([_spawn(r, 2, port, outs[r]) for r in range(2)], 240)
    single = str(tmp_path / "single.pt")
    _finish([_spawn(0, 1, port, single)], 240)
    r0, r1, ref = (torch.load(p, weights_only=True) for p in (*outs, single))
    assert int(r0["overlapped"]) == int(ref["overlapped"])
    worst = {}
    for k, want in ref.items():
        if k == "overlapped":
            continue
        got = r0[k]
        # both ranks hold the same replicated state after every update (same all-reduced gradients)
        if k.startswith(("D.", "G.", "G_head.", "buf.", "mD", "vD")):
            assert torch.equal(got, r1[k]), f"ranks disagree on {k}"
        tol = 1e-5          # measured: <= 4e-6 everywhere (fp32 summation order only)
        err = _rel(got, want)
        # absolute slack for tensors that are mathematically zero (the bias of a Linear that feeds BatchNorm)
        slack = 1e-7 * want.numel() ** 0.5
        ok = float((got.double() - want.double()).norm()) <= tol * float(want.double().norm()) + slack
        worst[k] = err
        assert ok, f"{k}: relative L2 error {err:.3e} > {tol:.0e}"
    assert torch.allclose(r0["losses"], torch.stack((r0["d_loss"], r0["g_loss"])).reshape(-1), rtol=1e-6, atol=1e-7)
    # what each update moved must agree too (parameters alone would hide a wrong update behind their norm)
    import common
    p0 = common.critic_params(1002, scale=1.5)
    for name, init in p0.items():
        du_got, du_ref = r0["D." + name] - init, ref["D." + name] - init
        assert _rel(du_got, du_ref) <= 1e-3, f"update of {name}: {_rel(du_got, du_ref):.3e}"
    print("worst relative errors:", sorted(worst.items(), key=lambda kv: -kv[1])[:8])
