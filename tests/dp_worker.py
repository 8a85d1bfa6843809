"""One rank of the data-parallel parity test (tests/test_gpu_dp2.py starts WORLD_SIZE of these as fresh
processes; WORLD_SIZE=1 gives the single-process reference at the global batch).  Every rank runs the
PRODUCT trainer, WGANGPTrainer(comm=DistComm()), on its contiguous shard of the same seeded global
batches with injected noise and dropout masks, and writes what it ends with to OUT."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sequence-to-graph-gan_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import common  # noqa: E402

B, T, TF, V = 16, 32, 24, 175
# S2G_DP_BACKEND=gloo (default): every rank on GPU 0, gloo moves the data, the kernels are the product's;
# nccl: rank r on GPU LOCAL_RANK, RCCL moves the data.  S2G_DP_FORCE_SPLIT=1: the multi-rank code path even for one rank.
BACKEND = os.environ.get("S2G_DP_BACKEND", "gloo")
DEV = "cuda:0" if BACKEND == "gloo" else "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))


def main():
    # a rank that is still alive after this many seconds prints every thread's stack and exits: the parent then
    # fails with the stacks in its message instead of sitting out its own time limit
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("S2G_DP_WATCHDOG", "100")), exit=True)
    from s2g.parallel import global_wsum, init_from_env, shard
    from s2g.trainer import WGANGPTrainer
    comm, rank, _, world = init_from_env(BACKEND, force_split=os.environ.get("S2G_DP_FORCE_SPLIT") == "1")
    lo, hi = shard(B, rank, world)
    n = hi - lo
    torch.cuda.set_device(torch.device(DEV))
    tr = WGANGPTrainer(V, TF, DEV, drop_rate=0.3, comm=comm)
    tr.G.load(common.generator_params(1001, V, TF))
    tr.D.load(common.critic_params(1002, scale=1.5))

    def batch(i):
        tokens, labels, lengths = common.batch(1010 + i, B, T, min_len=3)
        tokens = torch.nn.functional.pad(tokens, (0, T - tokens.shape[1]))
        wsum = global_wsum(labels[lo:hi], comm, DEV)
        nz = common.noise(1020 + i, B, TF)
        g = common.gen(1030 + i)
        mask = (torch.rand((3, B, T, 52), generator=g) > 0.3).float() / 0.7
        mask_g = (torch.rand((B, TF, 52), generator=g) > 0.3).float() / 0.7
        noise = dict(z=nz["z"][lo:hi].contiguous().to(DEV), U=nz["U"][lo:hi].contiguous().to(DEV),
                     alpha=nz["alpha"].view(-1)[lo:hi].contiguous().to(DEV), U2=nz["U2"][lo:hi].contiguous().to(DEV),
                     mask=mask[:, lo:hi].reshape(3 * n * T, 52).contiguous().to(DEV),
                     mask_g=mask_g[lo:hi].reshape(n * TF, 52).contiguous().to(DEV))
        dev = (tokens[lo:hi].to(torch.int32).contiguous().to(DEV), labels[lo:hi].to(DEV),
               lengths[lo:hi].to(torch.int32).to(DEV), wsum)
        return dev, noise

    out = {}
    # (1) one critic update and one generator update, launch after launch: gradients and loss values
    (tok, lab, ln, wsum), nz = batch(0)
    tr.critic_step(tok, lab, ln, wsum, nz["z"], nz["U"], nz["alpha"], nz["mask"])
    tr.generator_step(U=nz["U2"], mask=nz["mask_g"])
    both = tr.losses()
    out["d_loss"], out["g_loss"] = tr.d_loss().cpu(), tr.g_loss().cpu()
    out["losses"] = both.cpu()
    torch.cuda.synchronize()
    for name, v in tr.D.grads.views.items():
        out["gradD." + name] = v.cpu().clone()          # all-reduced: the global gradient on every rank
    for name, v in tr.G.grads.views.items():
        if v.numel() <= 1 << 20:
            out["gradG." + name] = v.cpu().clone()
        else:
            out["gradG_head." + name] = v.reshape(-1)[:1 << 16].cpu().clone()
            out["gradG_norm." + name] = v.double().norm().cpu()
    # (2) one full step through the overlapped schedule (streams, prefetched forwards)
    pairs = [batch(i) for i in range(1, 6)]
    tr.full_step([p[0] for p in pairs], [p[1] for p in pairs])
    torch.cuda.synchronize()
    out["overlapped"] = torch.tensor(int(bool(tr.D._side)))
    for name, v in tr.D.params.views.items():
        out["D." + name] = v.cpu().clone()
    for name, v in tr.G.params.views.items():
        if v.numel() <= 1 << 20:
            out["G." + name] = v.cpu().clone()
        else:
            out["G_head." + name] = v.reshape(-1)[:1 << 16].cpu().clone()
            out["G_norm." + name] = v.double().norm().cpu()
    for name, v in tr.G.buffers.items():
        out["buf." + name] = v.cpu().clone()
    out["mD"], out["vD"] = tr.mD.cpu().clone(), tr.vD.cpu().clone()
    out["d_loss_last"] = tr.d_loss().cpu()
    torch.save(out, os.environ["OUT"])
    if comm is not None:
        comm.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
