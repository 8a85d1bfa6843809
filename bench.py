#!/usr/bin/env python3
"""Benchmark of the WGAN-GP training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full WGAN-GP step of the reference's loop (wgan_gp.py:274-343): n_critic = 5
critic updates on 5 distinct real batches + 1 generator update (after the first batch's critic
update), at V = 175, L = 512, B = 64 per GPU, generator seq_len 512, dropout 0.3, synthetic
tokens already resident in HBM.  value = n_critic * B_global / t_step (sequences/s, whole job).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "sequence-to-graph-gan_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

V, L, B_PER_GPU, N_CRITIC, GEN_SEQ_LEN = 175, 512, 64, 5, 512
# Algorithmic work of the dominant kernels (fp32 FLOP per sequence-step, both layers):
# forward/tangent sweeps: W_hh0 h + W_ih1 y + W_hh1 h = 3 * 2 * 200 * 50; reverse sweeps likewise.
SWEEP_FLOP_PER_STEP = 3 * 2 * 200 * 50
FP32_PEAK_TFLOPS = 157.3       # MI355X fp32 vector = fp32 matrix peak


def make_batches(seed: int, n: int, B: int, device, profile: str):
    import common
    from s2g.trainer import class_weight_sum
    out = []
    for i in range(n):
        lengths = [L] * B if profile == "fixed" else common.adfa_lengths(seed + 100 + i, B, L).tolist()
        tokens, labels, lengths = common.batch(seed + i, B, L, vocab=V, lengths=lengths)
        out.append((tokens.to(torch.int32).to(device).contiguous(), labels.to(device),
                    lengths.to(torch.int32).to(device), class_weight_sum(labels), labels))
    return out


def cpu_baseline(seconds_budget: float = 60.0):
    """The CPU restatement of the reference step (oracle/, pinned to the reference's outputs by
    tests/test_oracle_golden.py) timed on this host: a few critic iterations (wgan_gp.py:277-318, about
    12 s of CPU work) at B = 64, L = 512, V = 175, generator seq_len 512 -- a bounded sample of the same
    workload (a full step is 5 of these + a generator update that costs <1% of one on the CPU)."""
    import common
    from oracle import wgan_gp_cpu as O
    threads = torch.get_num_threads()
    pg = common.generator_params(801, V, GEN_SEQ_LEN)
    pd = common.critic_params(802, vocab=V)
    bn = common.bn_buffers()
    tokens, labels, lengths = common.batch(810, B_PER_GPU, L, vocab=V, lengths=[L] * B_PER_GPU)
    nz = common.noise(820, B_PER_GPU, GEN_SEQ_LEN, vocab=V)
    cfg = dict(vocab=V, tau=0.5, hard=False, lambda_gp=10.0, n_critic=N_CRITIC, gen_seq_len=GEN_SEQ_LEN)
    opt = O.Adam(pd, 2e-4, 0.5, 0.999)
    # critic iterations until ~12 s of CPU work are through (at least 2, at most 8; the first warms the allocator)
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < min(seconds_budget, 12.0) and n < 8):
        O.critic_iteration(pg, bn, pd, opt, tokens, labels, lengths, nz["z"], nz["U"], nz["alpha"], cfg)
        n += 1
    dt = time.perf_counter() - t0
    # what the imported reference itself needed in the build container (recorded with the golden vectors)
    ref_rec = None
    try:
        g8 = torch.load(os.path.join(ROOT, "tests", "golden", "g8_full_size.pt"), weights_only=True)
        secs = float(g8["fixed.seconds"])
        ref_rec = {"sequences_per_s": B_PER_GPU / secs, "seconds": secs,
                   "what": "the reference's own train_wgan (imported unmodified, CPU, 8 threads of the build container): one "
                           "critic iteration + the generator update at B=64, L=512, generator seq_len 512, no dropout; "
                           "recorded with tests/golden/g8_full_size.pt"}
    except Exception:
        pass
    return {"value": n * B_PER_GPU / dt, "unit": "sequences/s", "cores": threads, "kind": "port",
            "reference_recorded": ref_rec,
            "sample": f"{n} critic iterations (B=64, L=512, V=175, gen seq_len 512, no dropout) in {dt:.1f} s; "
                      f"host has {os.cpu_count()} logical CPUs, torch used {threads} threads"}


class TimedLoader:
    """A DataLoader stand-in for train_wgan (CPU batches shaped like collate_fn's output, custom_dataset.py:30-47):
    drains the GPU and takes a time stamp when it hands out the first batch after the warm-up steps."""

    def __init__(self, batches, warm_batches):
        self.batches, self.warm, self.t0 = batches, warm_batches, None

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for k, b in enumerate(self.batches):
            if k == self.warm:
                torch.cuda.synchronize()
                self.t0 = time.perf_counter()
            yield b


def bench_train_wgan(dev, steps, warmup, device_rng, drop_rate):
    """The drop-in path: the module surface of wgan_gp.py (Generator, Discriminator, torch.optim.Adam) driven by
    train_wgan over a synthetic loader, exactly as main() does for real data (wgan_gp.py:393-442 of the
    reference).  Returns (sequences/s, ms per full step) over `steps` full steps after `warmup`."""
    import argparse as ap
    import contextlib
    import io
    import common
    import wgan_gp as W
    opt = ap.Namespace(vocab_size=V, epochs=1, batch_size=B_PER_GPU, lr=2e-4, b1=0.5, b2=0.999, latent_dim=100, seq_len=L,
                       n_critic=N_CRITIC, clip_value=0.01, lambda_gp=10.0, hard=False, tau=0.5, device_rng=device_rng)
    D = W.Discriminator(V, 50, 2, 2, drop_rate=drop_rate).to(dev)
    G = W.Generator(V, 2, seq_len=GEN_SEQ_LEN).to(dev)
    oG = torch.optim.Adam(G.parameters(), lr=opt.lr, betas=(opt.b1, opt.b2))
    oD = torch.optim.Adam(D.parameters(), lr=opt.lr, betas=(opt.b1, opt.b2))
    pool = [common.batch(3000 + i, B_PER_GPU, L, vocab=V, lengths=[L] * B_PER_GPU) for i in range(2 * N_CRITIC)]
    n = (steps + warmup) * N_CRITIC
    loader = TimedLoader([pool[k % len(pool)] for k in range(n)], warmup * N_CRITIC)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        W.train_wgan(G, D, oG, oD, loader, [], opt, dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - loader.t0
    return steps * N_CRITIC * B_PER_GPU / dt, dt / steps * 1e3


def bench_attention(steps, warmup):
    """BASELINE config 5's kernel: MaskedMultiheadAttention (transformer.py:196-282) forward + backward at the
    TenGAN shape (batch 16, L = 512, embed 64, 4 heads; tengan_config.py:7-12) with the causal and padding
    masks, through the module surface.  One JSON line of its own; not the headline metric."""
    import common
    import transformer as T
    dev = torch.device("cuda", 0)
    N, L_, E, H = 16, 512, 64, 4
    c = common.attention_case(1310, N, L_, E, H)
    m = T.MaskedMultiheadAttention(E, H).to(dev)
    q, k, v = (c[n].to(dev).requires_grad_(True) for n in "qkv")
    am, pm, dy = T.create_causal_mask(L_).to(dev), T.create_padding_mask(c["tokens"]).to(dev), c["dy"].to(dev)
    params = [q, k, v, m.in_proj_weight, m.in_proj_bias, m.fc_out.weight, m.fc_out.bias]

    def step():
        y = m(v, k, q, attn_mask=am, src_key_padding_mask=pm)
        torch.autograd.grad(y, params, dy)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    D = E // H
    core = 2.0 * N * H * L_ * L_ * D * (2 + 4)          # forward: QK^T, PV; backward: dP, dQ, dK, dV
    proj = 2.0 * N * L_ * (3 * E * 3 * E + E * E) * 3   # in / out projections, forward + two backward products
    print(json.dumps({"metric": "MaskedMultiheadAttention forward+backward sequences/sec (TenGAN shape)",
                      "value": N / dt, "unit": "sequences/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
                      "ms_per_step": dt * 1e3, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "attention fwd+bwd, batch 16, L=512, embed 64, 4 heads, causal + padding masks"},
                      "achieved_tflops": (core + proj) / dt / 1e12, "peak_tflops": FP32_PEAK_TFLOPS}), flush=True)


def bench_wgan_clip(steps, warmup):
    """BASELINE config 4: the weight-clipping WGAN variant (reference wgan.py) at L = 1024, B = 128: one step =
    one critic update of CWGAN.train_step (:174-198: D(real tokens), G(noise).detach(), D(fake, 100 steps),
    Wasserstein loss, Adam, clamp), dropout 0.3, device-resident batches.  The reference's generator branch
    cannot run (undefined self.loss, :214), so the critic update is what there is to measure.  One JSON line
    of its own; not the headline metric."""
    import common
    from s2g.wgan_clip import ClipCriticEngine, ClipGeneratorEngine
    dev = torch.device("cuda", 0)
    Vc, Bc, Lc = 175, 128, 1024
    pd, pg = common.wgan_clip_params(1701, Vc)
    D, G = ClipCriticEngine(Vc, 2, dev, dropout=0.3), ClipGeneratorEngine(Vc, 3, dev)
    D.load(pd); G.load(pg)
    pool = []
    for i in range(4):
        tok, lab, ln = common.batch(1710 + i, Bc, Lc, vocab=Vc, lengths=[Lc] * Bc)
        pool.append((tok.to(torch.int32).to(dev).contiguous(), lab.to(dev), ln.to(torch.int32).to(dev),
                     torch.randint(0, Vc, (Bc, 100), generator=common.gen(1720 + i)).to(torch.int32).to(dev)))

    def step(k):
        tok, lab, ln, noise = pool[k % len(pool)]
        # the generator's forward is handed over as a callable: it runs beside the real pass, like the fake pass
        D.critic_step(tok, lab, ln, Lc, lambda: G.forward(noise, lab, 0.5), 2e-4, 0.5, 0.999, 0.1, fake_len=noise.shape[1])
    for k in range(warmup):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(warmup + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"metric": "weight-clipping WGAN critic-update sequences/sec at V=175 L=1024 B=128",
                      "value": Bc / dt, "unit": "sequences/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
                      "ms_per_step": dt * 1e3, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "wgan.py critic update: D(real L=1024) + G forward + D(fake L=100) + backward + "
                                             "Adam + clamp, B=128, dropout 0.3"}}), flush=True)


def bench_graph(steps, warmup, cpu_seconds=10.0):
    """The step after the hot path (reference graph_encoder.py:57-133): node features of the transition graphs of
    a batch of generated-length sequences (L = 512, V = 175), device-resident tokens in, [N, V, 4] float64
    features out.  One step = one s2g_graph_features launch over N = 1024 sequences.  Two symbol distributions:
    uniform draws (what an untrained generator emits: ~166 distinct symbols, Katz matrix in the global workspace)
    and skewed draws (trace-like: a few dozen distinct symbols, Katz matrix in LDS).  The CPU figure is the
    numpy restatement of the networkx algorithms on one core.  One JSON line of its own."""
    import common
    from oracle import graph_cpu
    from s2g import graph
    dev = torch.device("cuda", 0)
    N, Lg, Vg = 1024, 512, 175
    res = {}
    for name, skew in (("uniform", False), ("skewed", True)):
        tok, _ = common.graph_token_batch(77, N, Lg, vocab=Vg, skew=skew)
        tokd = tok.to(dev)
        for _ in range(warmup):
            graph.graph_features(tokd, None, vocab=Vg)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        t0 = time.perf_counter()
        ev[0].record()
        for _ in range(steps):
            feat, info = graph.graph_features(tokd, None, vocab=Vg)
        ev[1].record()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        inf = info.cpu()
        t1, cnt = time.perf_counter(), 0
        while time.perf_counter() - t1 < cpu_seconds / 2 and cnt < N:
            try:
                graph_cpu.features(tok[cnt].tolist())
            except Exception:      # noqa: BLE001  (a singular Katz system: still a timed graph)
                pass
            cnt += 1
        cpu = cnt / (time.perf_counter() - t1)
        res[name] = {"graphs_per_s": N / dt, "ms_per_launch": dt * 1e3, "kernel_ms": ev[0].elapsed_time(ev[1]) / steps,
                     "mean_nodes": float(inf[:, 0].float().mean()), "mean_edges": float(inf[:, 3].float().mean()),
                     "mean_pagerank_iterations": float(inf[:, 2].float().mean()),
                     "flagged": int((inf[:, 1] != 0).sum()), "cpu_graphs_per_s": cpu, "cpu_sample": f"{cnt} graphs, 1 core"}
    u = res["uniform"]
    print(json.dumps({"metric": "sequence-to-graph node-feature graphs/sec at V=175 L=512", "value": u["graphs_per_s"],
                      "unit": "graphs/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": u["ms_per_launch"],
                      "higher_is_better": True, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": "GraphEncoder centralities (betweenness, closeness, pagerank, katz), 1024 "
                                             "sequences of 512 uniform symbols over 175 per launch"},
                      "cpu_baseline": {"value": u["cpu_graphs_per_s"], "unit": "graphs/s", "cores": 1, "kind": "port",
                                       "sample": u["cpu_sample"]},
                      "by_distribution": res}), flush=True)


def main():
    global GEN_SEQ_LEN
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lengths", default="fixed", choices=["fixed", "adfa"])
    ap.add_argument("--gen-seq-len", type=int, default=GEN_SEQ_LEN,
                    help="the generator's seq_len: 512 = the headline shape; 120 = the reference's hard-coded default "
                         "(wgan_gp.py:145), i.e. fakes shorter than the real sequences")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-wgan", action="store_true", help="skip the train_wgan() path measurement (profiling runs)")
    ap.add_argument("--drop-rate", type=float, default=0.3)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 (default, the headline): exact-fp32 MFMA everywhere; bf16: bf16 operands / fp32 accumulation "
                         "in the critic's input-layer products (the 2e-2 path)")
    ap.add_argument("--variant", default="wgan_gp", choices=["wgan_gp", "attention", "wgan_clip", "graph"],
                    help="wgan_gp (default): the WGAN-GP step; attention: the transformer.py attention module alone; "
                         "wgan_clip: the critic update of the weight-clipping variant (wgan.py) at L=1024, B=128; graph: sequence -> graph node features (graph_encoder.py) for 1024 sequences")
    ap.add_argument("--path", default="trainer", choices=["trainer", "train_wgan"],
                    help="which path `value` is measured on: WGANGPTrainer.full_step on device-resident batches "
                         "(default, the headline) or the drop-in train_wgan() loop of wgan_gp.py on a CPU loader; "
                         "the other one is reported beside it on one GPU")
    a = ap.parse_args()
    GEN_SEQ_LEN = a.gen_seq_len

    if a.variant == "attention":
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
        bench_attention(a.steps, a.warmup)
        return
    if a.variant == "wgan_clip":
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
        bench_wgan_clip(a.steps, a.warmup)
        return
    if a.variant == "graph":
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
        bench_graph(a.steps, a.warmup)
        return
    from s2g import lib
    from s2g.parallel import init_from_env
    from s2g.trainer import WGANGPTrainer
    lib.get()                                    # fail loudly if libs2g.so is missing
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    # S2G_REHEARSE=1: every rank on GPU 0 with the gloo backend -- exercises the multi-rank code path
    # on a one-GPU machine (its number is meaningless; RCCL needs one GPU per rank)
    # S2G_REHEARSE=rccl1: ONE rank, but with an RCCL process group and the multi-rank code path (split
    # BatchNorm kernels, three-launch critic update, every all-reduce issued on RCCL's stream)
    rehearse = os.environ.get("S2G_REHEARSE") == "1"
    rccl1 = os.environ.get("S2G_REHEARSE") == "rccl1"
    comm, rank, local, world = init_from_env("gloo" if rehearse else "nccl", force_split=rccl1)
    if world != max(a.gpus, 1):
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run")
    if rehearse:
        local = 0
        if os.environ.get("S2G_TRACE_COMM") == "1" and comm is not None:
            inner, count = comm.allreduce, [0]

            def traced(t):
                count[0] += 1
                print(f"[rank {rank}] allreduce #{count[0]} numel={t.numel()} stream={torch.cuda.current_stream().cuda_stream:#x}: "
                      "waiting for the GPU", flush=True)
                torch.cuda.synchronize()
                print(f"[rank {rank}] allreduce #{count[0]}: GPU idle, reducing", flush=True)
                inner(t)
                print(f"[rank {rank}] allreduce #{count[0]} done", flush=True)
            comm.allreduce = traced
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import common
    tr = WGANGPTrainer(V, GEN_SEQ_LEN, dev, lr=2e-4, b1=0.5, b2=0.999, lambda_gp=10.0, tau=0.5, hard=False,
                       n_critic=N_CRITIC, drop_rate=a.drop_rate, comm=comm, seed=42,   # the trainer derives per-rank noise streams
                       dtype=a.dtype)
    if os.environ.get("S2G_NO_OVERLAP") == "1":
        tr.D.overlap = False                                      # one stream, launch after launch (debugging aid)
    tr.G.load(common.generator_params(801, V, GEN_SEQ_LEN))      # same initial weights on every rank
    tr.D.load(common.critic_params(802, vocab=V))
    pool = make_batches(1000 + 7919 * rank, 2 * N_CRITIC, B_PER_GPU, dev, a.lengths)
    if comm is not None:
        from s2g.parallel import global_wsum
        pool = [(t, lab, ln, global_wsum(lab_cpu, comm, dev), lab_cpu) for (t, lab, ln, _, lab_cpu) in pool]

    # per-kernel timing of the five heavy kernels of a critic iteration (the four LSTM sweeps and the
    # grouped weight-gradient GEMM): HIP events on the stream the kernels are launched on, recorded
    # inside the timed region.  Every 6th critic iteration is bracketed (14 event records each), so the
    # sampled position walks through the five iterations of a step: bracketing all of them costs 3-8 % of
    # the step itself (7.39 against 7.14 ms at 10 steps, 7.84 against 7.17 ms at 20).
    ev = {}
    tr.D.timed_events = ev
    tr.D.timed_every = 6

    def one_step(k):
        batches = [pool[(k * N_CRITIC + i) % len(pool)][:4] for i in range(N_CRITIC)]
        # the loader peeks one batch ahead: the next step's first fake batch does not depend on this step's
        # remaining updates of the critic, and the generator stays fixed until that batch's critic update
        tr.full_step(batches, lookahead=pool[((k + 1) * N_CRITIC) % len(pool)][:4])

    # The step runs on a non-blocking stream of its own: the CU-masked second stream of the critic is a
    # blocking stream, i.e. it would synchronise implicitly with every launch on the NULL stream.
    torch.cuda.synchronize()
    prewarm = 0
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        if world > 1:
            # several ranks: the same FIXED number of untimed steps on every rank (the ranks issue collectives in
            # lockstep, so none may decide by its own clock when it is warm)
            n_pre = 3 if rehearse else 100      # (the gloo rehearsal stages every collective through the host)
            for k in range(n_pre):
                one_step(k)
            torch.cuda.synchronize()
            prewarm = n_pre
        else:
            # A freshly leased box enqueues slowly for its first tens of seconds (host side: 7 ms per step against
            # 4 ms a minute later, same binary), which makes the step launch-bound.  Untimed blocks of steps run
            # until the host's enqueue time per step stops improving (at least 3 blocks, at most ~12 s); the W
            # warmup steps and the K timed steps follow unchanged.
            best, stale, t_pre = float("inf"), 0, time.perf_counter()
            while time.perf_counter() - t_pre < 12.0:
                t_blk = time.perf_counter()
                for k in range(25):
                    one_step(prewarm + k)
                enq = (time.perf_counter() - t_blk) / 25
                torch.cuda.synchronize()
                prewarm += 25
                stale = 0 if enq < 0.95 * best else stale + 1
                best = min(best, enq)
                if prewarm >= 75 and stale >= 2:
                    break
        for k in range(a.warmup):
            one_step(k)
        ev.clear()
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()
        calls0 = comm.calls if comm is not None else 0
        t0 = time.perf_counter()
        for k in range(a.steps):
            one_step(a.warmup + k)
        t_host = time.perf_counter() - t0          # host time to enqueue the steps (launch-bound if close to dt)
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if comm is not None:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax[0])

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        print(f"[bench] {ms_per_step:.3f} ms/step, host enqueue {t_host / a.steps * 1e3:.3f} ms/step", file=sys.stderr)
        seqs =N_CRITIC * B_PER_GPU * world * a.steps
        # algorithmic FLOP per launch (averaged over the batches of the pool)
        real = sum(int(p[2].sum()) for p in pool[:N_CRITIC]) / N_CRITIC      # live real steps per batch
        fake_steps = B_PER_GPU * GEN_SEQ_LEN
        rows3 = 2 * real + fake_steps                                         # real + interp + fake sequence-steps
        work = {
            "s2g_k_lstm2_fwd": rows3 * SWEEP_FLOP_PER_STEP,
            "s2g_k_lstm2_bwd": rows3 * SWEEP_FLOP_PER_STEP,
            "s2g_k_lstm2_tangent": real * SWEEP_FLOP_PER_STEP,
            "s2g_k_lstm2_bwd2": real * SWEEP_FLOP_PER_STEP,
            # dW_ih0: the fake third's and the x_hat product [200 x rows] x [rows x V] and the 200-wide Gram product of
            # the penalty's first-order factor (the real third is a bucketed sum, not a product);
            # dW_hh0, dW_ih1, dW_hh1: 4 row-blocks x [200 x 50] each
            "s2g_k_gemm_grouped": 2.0 * 200 * (V * (real + fake_steps) + 200 * real + 3 * 50 * (3 * real + fake_steps)),
        }
        # "name#phase" entries are the launches of one kernel inside one critic iteration (the grouped
        # GEMM runs as three launches, one per phase): their average durations add up to the kernel's
        # time per critic iteration
        per_iter, counts, totals, launches = {}, {}, {}, {}
        for tag, pairs in ev.items():
            name = tag.split("#")[0]
            durs = [s.elapsed_time(e) for s, e in pairs]
            launches[tag] = sum(durs) / max(len(durs), 1)
            per_iter[name] = per_iter.get(name, 0.0) + sum(durs) / max(len(durs), 1)
            counts[name] = counts.get(name, 0) + len(durs)
            totals[name] = totals.get(name, 0.0) + sum(durs)
        kernels = {}
        for name, per_it in per_iter.items():
            # a kernel launched n times per critic iteration: work per launch = work per iteration / n and
            # average launch duration = (sum of all its launches) / (number of launches); the quotient
            # equals work per iteration / time per iteration
            n_tags = sum(1 for tag in ev if tag.split("#")[0] == name)
            avg = totals[name] / max(counts[name], 1)
            tf = (work[name] / n_tags) / (avg * 1e-3) / 1e12 if avg > 0 else 0.0
            kernels[name] = {"avg_launch_ms": avg, "launches_timed": counts[name],
                             "launches_per_critic_iteration": n_tags, "ms_per_critic_iteration": per_it,
                             "flop_per_launch": work[name] / n_tags,
                             "ms_per_step": totals[name] / a.steps, "achieved_tflops": tf,
                             "frac": tf / FP32_PEAK_TFLOPS}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        kern_ms, achieved = kernels[dom]["avg_launch_ms"], kernels[dom]["achieved_tflops"]
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")       # per-launch HBM bytes from rocprofv3 PMC passes
        if os.path.exists(pmc):
            with open(pmc) as f:
                t = json.load(f).get(dom)
            if t:
                traffic, traffic_src = t["bytes_per_launch"], t["source"]
        # the HBM-bound kernels of the step (sampler, penalty pieces, one-hot / interpolation operands, Adam, draws):
        # bytes per launch from the committed PMC passes, durations from the committed kernel-trace run of this
        # same command (profiles/), not from this run
        hbm_kernels = {}
        if os.path.exists(pmc):
            with open(pmc) as f:
                table = json.load(f)
            for name in ("s2g_k_gumbel_softmax_fwd_v4<true>", "s2g_k_gumbel_softmax_fwd_v4<false>", "s2g_k_gumbel_softmax_bwd_v4",
                         "s2g_k_gp_interp", "s2g_k_embed_grad_tokens_part", "s2g_k_gp_dots", "s2g_k_adam",
                         "s2g_k_dropout_mask_rng", "s2g_k_linear_wstream", "s2g_k_linear_wgrad_adam"):
                t = table.get(name)
                if t and "avg_ns" in t:
                    hbm_kernels[name] = {"bytes_per_launch": t["bytes_per_launch"], "avg_us": t["avg_ns"] / 1e3,
                                         "gb_per_s": t["gb_per_s"], "frac_of_hbm_peak": t["frac_of_8_tb_s"]}
        line = {
            "metric": "WGAN-GP train sequences/sec at V=175 L=512", "value": seqs / dt, "unit": "sequences/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.dtype == "f32" else "bf16 operands / f32 accumulate in the critic's input-layer and "
                                                     "weight-gradient products, f32 elsewhere",
            "data": "synthetic",
            "config": {"workload": "full WGAN-GP step = 5 critic updates (real+fake+gradient penalty) + 1 generator "
                                   f"update; V=175, L=512, B=64 per GPU, generator seq_len {GEN_SEQ_LEN}, tau 0.5, lambda_gp 10, "
                                   f"critic dropout {a.drop_rate}, lengths {a.lengths}",
                       "global_batch": B_PER_GPU * world, "seq_len": L, "parallelism": f"dp{world}"},
            "ms_per_critic_iteration_est": ms_per_step / (N_CRITIC + 0.3),
            "host_enqueue_ms_per_step": t_host / a.steps * 1e3,
            "prewarm_steps": prewarm,
            "collectives_per_step": ((comm.calls - calls0) / a.steps if comm is not None else 0),
            "roofline": {"bound": "mfma", "kernel": dom,
                         "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": kern_ms, "launches_timed": kernels[dom]["launches_timed"],
                         "note": "dominant kernel = largest summed time among the five timed kernels; exact-fp32 "
                                 "arithmetic, peak = fp32 vector = fp32 MFMA peak; the LSTM sweeps are chains of 512 "
                                 "dependent steps (one workgroup per sequence) and are bounded by per-step latency, "
                                 "not by the FLOP or HBM roof"},
            "kernels": kernels, "launch_avg_ms": launches,
            "hbm_bound_kernels": {"source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes and a "
                                            "--kernel-trace --stats run of this command; peak 8000 GB/s)", **hbm_kernels},
            "second_stream": ("cu-masked, 3/4 of the CUs" if tr.D._side else None),
        }
        if world == 1 and not a.no_train_wgan:
            # the drop-in path (python3 wgan_gp.py ..., README.md:83): same step through train_wgan() and the nn.Module
            # surface, batches arriving from a CPU loader; with device-side draws, and with the reference-compatible
            # draws on the CPU generator (z, Gumbel uniforms, alpha: 2 x 23 MB of torch.rand + copies per batch)
            v_dev, ms_dev = bench_train_wgan(dev, a.steps, a.warmup, True, a.drop_rate)
            v_cpu, ms_cpu = bench_train_wgan(dev, 2, 1, False, a.drop_rate)
            line["train_wgan_path"] = {"value": v_dev, "unit": "sequences/s", "ms_per_step": ms_dev, "rng": "device (--device_rng)",
                                       "reference_compatible_cpu_rng": {"value": v_cpu, "ms_per_step": ms_cpu, "steps": 2}}
            if a.path == "train_wgan":
                line["trainer_path"] = {"value": line["value"], "ms_per_step": line["ms_per_step"]}
                line["value"], line["ms_per_step"] = v_dev, ms_dev
                line["config"]["workload"] += "; measured through train_wgan() on a CPU loader with --device_rng"
        if world == 1 and not a.no_train_wgan and a.dtype == "f32":
            # the reduced-precision path beside the headline (BASELINE's second configuration names a bf16 MFMA critic):
            # bf16 operands / fp32 accumulation in the critic's input-layer and weight-gradient products, fp32
            # recurrence, reductions and generator; within 2e-2 of the reference's fp32 gradients (measured 2e-3)
            tb = WGANGPTrainer(V, GEN_SEQ_LEN, dev, lr=2e-4, b1=0.5, b2=0.999, lambda_gp=10.0, tau=0.5, hard=False,
                               n_critic=N_CRITIC, drop_rate=a.drop_rate, seed=42, dtype="bf16")
            tb.G.load(common.generator_params(801, V, GEN_SEQ_LEN))
            tb.D.load(common.critic_params(802, vocab=V))

            def bstep(k):
                tb.full_step([pool[(k * N_CRITIC + i) % len(pool)][:4] for i in range(N_CRITIC)],
                             lookahead=pool[((k + 1) * N_CRITIC) % len(pool)][:4])
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                for k in range(a.warmup + 5):
                    bstep(k)
                torch.cuda.synchronize()
                tb0 = time.perf_counter()
                for k in range(a.steps):
                    bstep(a.warmup + 5 + k)
                torch.cuda.synchronize()
                tb1 = time.perf_counter() - tb0
            line["bf16_path"] = {"value": seqs / tb1, "unit": "sequences/s", "ms_per_step": tb1 / a.steps * 1e3,
                                 "dtype": "bf16 operands / f32 accumulate in the critic's input-layer and weight-gradient "
                                          "products, f32 elsewhere", "tolerance": "2e-2 (north star, bf16); measured 2e-3"}
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if comm is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
