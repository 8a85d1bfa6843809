"""Data-parallel conventions of s2g/parallel.py rehearsed with 2 CPU processes over gloo: each rank
takes a contiguous shard of the batch, normalises its loss by the GLOBAL batch size and the GLOBAL
class-weight sum and SUM-all-reduces gradients; the generator's BatchNorm layers run replicated on the
global batch (every rank knows all inputs), their backward on the all-gathered upstream gradient; the
result must equal the single-process step on the whole batch (oracle arithmetic, float64).  The output
Linear's gradient is formed from gathered FACTORS (rows of d(pre-activation), in column pieces started back to
back) and the global input rows, not by summing the ranks' products."""
import os
import socket

import torch
import torch.multiprocessing as mp
import torch.nn.functional as F

import common

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _replicated_bn(x_global, gamma, beta, dy_local, comm, eps=0.8):
    """what GeneratorEngine does with several ranks: BatchNorm on the GLOBAL batch on every rank (exact
    statistics, no exchange), this rank's rows go on to the sharded output layer; backward: the ranks'
    upstream gradient rows are gathered in rank order and the layer's backward runs on the global batch.
    Returns (y of the global batch, dx of the global batch, dgamma, dbeta)."""
    x = x_global.detach().clone().requires_grad_(True)
    g, b = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    mean, var = x.mean(0), x.var(0, unbiased=False)
    y = (x - mean) / torch.sqrt(var + eps) * g + b
    dy = comm.allgather_rows(dy_local)
    dx, dg, db = torch.autograd.grad(y, [x, g, b], dy)
    return y.detach(), dx, dg, db


def _dp_critic_loss(O, p, real, lengths, fake, labels, alpha, lam, n_global, wsum):
    """local share of d_loss with global normalisers (what critic_iteration is given)."""
    out_r = O.critic_forward(p, real, lengths)
    out_f = O.critic_forward(p, fake, None)
    fake_p = O.pad_or_crop_time(fake, real.shape[1])
    xh = (alpha * real + (1 - alpha) * fake_p).detach().requires_grad_(True)
    d = O.critic_forward(p, xh, lengths)[:, 0]
    (g,) = torch.autograd.grad(d, xh, torch.ones_like(d), create_graph=True)
    pen = ((g.reshape(g.shape[0], -1).norm(2, dim=1) - 1) ** 2).sum() / n_global
    w = torch.tensor([1.0, 2.0], dtype=real.dtype)

    def ce_sum(out):
        logp = F.log_softmax(F.softmax(out[:, 1:], -1), -1)
        return -(w[labels] * logp[torch.arange(len(labels)), labels]).sum() / wsum
    return (-out_r[:, 0].sum() + out_f[:, 0].sum()) / n_global + lam * pen + 0.5 * (ce_sum(out_r) + ce_sum(out_f))


def _worker(rank, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(WORLD), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle import wgan_gp_cpu as O
    from s2g.parallel import global_wsum, init_from_env, shard
    comm, r, _, world = init_from_env("gloo")
    assert (r, world) == (rank, WORLD) and comm.world == WORLD
    B, T, Tf, V = 8, 10, 6, 23
    dt = torch.float64
    p = {k: v.requires_grad_(True) for k, v in common.critic_params(91, vocab=V, scale=2.0, dtype=dt).items()}
    tokens, labels, lengths = common.batch(92, B, T, vocab=V, min_len=2)
    g = common.gen(93)
    fake = torch.softmax(torch.randn((B, Tf, V), generator=g, dtype=dt), -1)
    alpha = torch.rand((B, 1, 1), generator=g, dtype=dt)
    real = F.one_hot(tokens, V).to(dt)
    lo, hi = shard(B, rank, world)
    wsum = global_wsum(labels[lo:hi], comm)
    assert wsum == float(1.0 * (labels == 0).sum() + 2.0 * (labels == 1).sum())
    loss = _dp_critic_loss(O, p, real[lo:hi], lengths[lo:hi], fake[lo:hi], labels[lo:hi], alpha[lo:hi], 10.0, B, wsum)
    names = list(p)
    grads = torch.autograd.grad(loss, [p[n] for n in names])
    flat = torch.cat([x.reshape(-1) for x in grads])            # one flat bucket
    comm.allreduce(flat)
    lt = loss.detach().reshape(1).clone()
    comm.allreduce(lt)

    # BatchNorm replicated on the global batch; every rank contributes its rows of the upstream gradient
    x = torch.randn((B, 5), generator=common.gen(94), dtype=dt)
    gamma, beta = torch.rand(5, generator=g, dtype=dt) + 0.5, torch.rand(5, generator=g, dtype=dt)
    dy = torch.randn((B, 5), generator=common.gen(95), dtype=dt)
    y_rep, dx_rep, dg_rep, db_rep = _replicated_bn(x, gamma, beta, dy[lo:hi], comm)
    assert comm.calls == 4          # world-size gather at set-up aside: wsum, gradient bucket, loss, d(y) gather

    # output Linear: dW = dl2^T a3 over the GLOBAL batch from the gathered rows of dl2 (three column pieces,
    # all started before the first is waited for) and a3 of the global batch, which every rank holds
    NV = 11
    a3 = torch.randn((B, 5), generator=common.gen(96), dtype=dt)
    dl2 = torch.randn((B, NV), generator=common.gen(97), dtype=dt)
    cuts = [(0, 4), (4, 8), (8, NV)]
    started = [comm.allgather_rows_start(dl2[lo:hi, c0:c1]) for c0, c1 in cuts]
    dW_fac = torch.empty(NV, 5, dtype=dt)
    for (c0, c1), (rows, wait) in zip(cuts, started):
        wait()
        assert rows.shape == (B, c1 - c0)
        dW_fac[c0:c1] = rows.t() @ a3
    assert comm.calls == 7
    dW_sum = dl2[lo:hi].t() @ a3[lo:hi]         # what an all-reduce of the ranks' products would deliver
    comm.allreduce(dW_sum)

    if rank == 0:
        p1 = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
        full, _ = O.critic_loss(p1, real, lengths, fake, labels, alpha, 10.0)
        gfull = torch.autograd.grad(full, [p1[n] for n in names])
        ref = torch.cat([x.reshape(-1) for x in gfull])
        xr = x.clone().requires_grad_(True)
        gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        mean, var = xr.mean(0), xr.var(0, unbiased=False)
        y_ref = (xr - mean) / torch.sqrt(var + 0.8) * gr + br
        dx_ref, dg_ref, db_ref = torch.autograd.grad(y_ref, [xr, gr, br], dy)
        ret["grad_err"] = float((flat - ref).norm() / ref.norm())
        ret["loss_err"] = abs(float(lt[0]) - float(full))
        ret["fac_err"] = max(float((dW_fac - dl2.t() @ a3).abs().max()), float((dW_fac - dW_sum).abs().max()))
        ret["bn_err"] = max(float((y_rep - y_ref.detach()).abs().max()), float((dx_rep - dx_ref).abs().max()),
                            float((dg_rep - dg_ref).abs().max()), float((db_rep - db_ref).abs().max()))
    comm.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_step_equals_single_rank():
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(port, ret), nprocs=WORLD, join=True)
        assert ret["grad_err"] < 1e-10, dict(ret)
        assert ret["loss_err"] < 1e-10, dict(ret)
        assert ret["bn_err"] < 1e-10, dict(ret)
        assert ret["fac_err"] < 1e-12, dict(ret)
