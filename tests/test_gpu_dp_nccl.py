"""The data-parallel path over RCCL (torch.distributed backend "nccl"): fresh processes, one per GPU, each running
WGANGPTrainer(comm=DistComm()) on its shard of global batches of 16, against ONE process on the whole batches --
the same worker and the same comparisons as tests/test_gpu_dp2.py, with RCCL instead of gloo moving the data.

  * two ranks on two DISTINCT GPUs: needs two visible GPUs, skipped on a one-GPU box;
  * one rank with the multi-rank code path forced (replicated small layers with the all-gather of d(a3),
    all-reduced buckets, three-launch critic update): every RCCL call of the step on whatever GPU there is.

The parent process never touches the GPU (counting devices does not initialise it)."""
import os

import pytest
import torch

from test_gpu_dp2 import _finish, _free_port, _rel, _spawn

pytestmark = pytest.mark.gpu


def _compare(got, other, ref, tol=1e-5):
    worst = {}
    for k, want in ref.items():
        if k == "overlapped":
            continue
        if other is not None and k.startswith(("D.", "G.", "G_head.", "buf.", "mD", "vD")):
            assert torch.equal(got[k], other[k]), f"ranks disagree on {k}"
        err = _rel(got[k], want)
        slack = 1e-7 * want.numel() ** 0.5          # tensors that are mathematically zero
        ok = float((got[k].double() - want.double()).norm()) <= tol * float(want.double().norm()) + slack
        worst[k] = err
        assert ok, f"{k}: relative L2 error {err:.3e} > {tol:.0e}"
    return worst


def test_one_rank_over_rccl_matches_the_single_process_schedule(tmp_path):
    port = _free_port()
    out, single = str(tmp_path / "rccl1.pt"), str(tmp_path / "single.pt")
    _finish([_spawn(0, 1, port, out, S2G_DP_BACKEND="nccl", S2G_DP_FORCE_SPLIT="1")], 300)
    _finish([_spawn(0, 1, port, single)], 300)
    got, ref = (torch.load(p, weights_only=True) for p in (out, single))
    worst = _compare(got, None, ref)
    print("worst relative errors:", sorted(worst.items(), key=lambda kv: -kv[1])[:5])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one rank per GPU over RCCL)")
def test_two_ranks_on_two_gpus_over_rccl_match_one_rank(tmp_path):
    world = min(2, torch.cuda.device_count())
    port = _free_port()
    outs = [str(tmp_path / f"rank{r}.pt") for r in range(world)]
    _finish([_spawn(r, world, port, outs[r], S2G_DP_BACKEND="nccl") for r in range(world)], 300)
    single = str(tmp_path / "single.pt")
    _finish([_spawn(0, 1, port, single)], 300)
    r0, r1, ref = (torch.load(p, weights_only=True) for p in (*outs, single))
    worst = _compare(r0, r1, ref)
    print("worst relative errors:", sorted(worst.items(), key=lambda kv: -kv[1])[:5])
