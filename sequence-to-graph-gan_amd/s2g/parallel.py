"""Data-parallel layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on ROCm; "gloo" for CPU rehearsals).  The reference is single-process (helper_tools.py:10-24);
the batch dimension shards exactly because the critic is per-sample independent, and the three
cross-sample couplings are handled explicitly:

  * batch means (wgan_gp.py:313) and the penalty mean (:256): every rank normalises by the
    GLOBAL batch size, so gradients combine with a plain SUM all-reduce;
  * the weighted cross-entropy mean (wgan_gp.py:267-268) divides by the GLOBAL sum of class
    weights (global_wsum);
  * the generator's BatchNorm batch statistics (wgan_gp.py:150): the small layers in front of the
    output Linear (100 -> 128 -> 256 -> 512, 0.18 GFLOP at a global batch of 512) run REPLICATED on the
    global batch on every rank, so the statistics are exact and local and no collective sits inside a
    generator forward.  Every rank knows the whole batch's inputs: z is a pure function of (seed, draw
    offset, position) of the shared counter-based generator, and the labels are gathered once per batch.
    Only the output Linear (99.6 % of the parameters and of the work) is sharded by sample.  Backward:
    one all-gather of d(a3) [B_global, 512] replaces the BatchNorm-sum exchanges, after which the small
    layers' gradients are computed on the global batch, identical on every rank, and need no all-reduce.

What travels per update: the critic's gradients as ONE flat fp32 bucket (66 k floats: latency-bound, one
all-reduce), and for the generator's output Linear (46 M parameters at L*V = 89 600) not the gradient but its
factors: dW = dl2^T a3 has rank <= batch, a3 of the global batch is on every rank already, so the ranks
all-gather their rows of dl2 [B, L*V] (world * 23 MB received at B = 64) and each forms the global gradient
itself -- against 2 (world - 1) / world x 184 MB per rank for an all-reduce of the product."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class DistComm:
    def __init__(self, group=None, force_split: bool = False):
        """force_split: take the multi-rank code paths (split BatchNorm kernels, three-launch critic update,
        every all-reduce issued) even in a group of one rank: a rehearsal of the RCCL calls and of their
        stream on a one-GPU machine."""
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.split = self.world > 1 or force_split
        self.calls = 0          # collectives issued so far (all-reduces and all-gathers)

    def allreduce(self, t: torch.Tensor) -> None:
        """Sum over the ranks, in place, ordered behind the work enqueued on the current stream."""
        if self.split:
            self.calls += 1
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                # rehearsal of several ranks on one GPU: gloo's own staging of device tensors (its copies and
                # stream waits) stalls now and then with several processes on one card, also on a drained GPU,
                # so the tensor crosses the host here and gloo only ever sees host memory
                host = t.cpu()              # (synchronises with the current stream)
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(host)
                return
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def allgather_rows(self, t: torch.Tensor) -> torch.Tensor:
        """[rows, ...] of every rank stacked in rank order -> [world * rows, ...] (a copy for one rank)."""
        out, wait = self.allgather_rows_start(t)
        wait()
        return out

    def allgather_rows_start(self, t: torch.Tensor):
        """Starts allgather_rows(t) and returns (out, wait): `out` holds the result once wait() has been called
        (it orders the current stream behind the collective; the input is read in the order of the stream that was
        current at the start).  Several gathers started back to back run one after the other on the communicator's
        stream while the caller works on the results that have arrived."""
        t = t.contiguous()
        if not self.split:
            return t.clone(), (lambda: None)
        self.calls += 1
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if dist.get_backend(self.group) == "gloo":
            if t.is_cuda:                   # through the host, as in allreduce
                host = t.cpu()
                parts = [torch.empty_like(host) for _ in range(self.world)]
                dist.all_gather(parts, host, group=self.group)
                out.copy_(torch.cat(parts, 0))
            else:
                dist.all_gather(list(out.chunk(self.world, 0)), t, group=self.group)
            return out, (lambda: None)
        work = dist.all_gather_into_tensor(out, t, group=self.group, async_op=True)
        return out, (lambda: (work.wait(), None)[1])

    def broadcast_int(self, value: int, device=None) -> int:
        """Rank 0's `value` on every rank (set-up time: e.g. the seed of the draws all ranks must share)."""
        if self.world == 1:
            return int(value)
        gloo = dist.get_backend(self.group) == "gloo"
        t = torch.tensor([int(value) & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device="cpu" if gloo else device)
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        return int(t[0])

    def barrier(self) -> None:
        if self.world > 1:
            dist.barrier(group=self.group)


def init_from_env(backend: str = "nccl", force_split: bool = False):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run).
    Returns (comm or None, rank, local_rank, world).  force_split: build a process group and a
    DistComm(force_split=True) even for one rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1 and not force_split:
        return None, 0, local, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return DistComm(force_split=force_split), rank, local, world


def shard(n_global: int, rank: int, world: int):
    """Contiguous batch shard [lo, hi) of this rank (equal shards; n_global % world == 0)."""
    if n_global % world:
        raise ValueError("global batch must divide evenly over the ranks")
    per = n_global // world
    return rank * per, (rank + 1) * per


def global_wsum(labels_cpu: torch.Tensor, comm, device=None) -> float:
    """Global sum of the class weights [1,2] over all ranks' labels (wgan_gp.py:267)."""
    n1 = int((labels_cpu == 1).sum())
    w = 1.0 * (labels_cpu.numel() - n1) + 2.0 * n1
    if comm is None or not comm.split:
        return w
    t = torch.tensor([w], dtype=torch.float64, device=device or "cpu")
    comm.allreduce(t)
    return float(t[0])
