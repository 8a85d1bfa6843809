"""Does a gloo all-reduce of a device tensor work with two ranks on one GPU?  (rehearsal sanity probe)"""
import os
import sys
import time

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
dev = torch.device("cuda:0")
t = torch.full((1000,), float(rank + 1), device=dev)
t0 = time.time()
dist.all_reduce(t)
torch.cuda.synchronize()
print(rank, "default stream ok", float(t[0]), f"{time.time() - t0:.2f}s", flush=True)
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    u = torch.full((1 << 20,), float(rank + 1), device=dev)
    for _ in range(3):
        dist.all_reduce(u)
    torch.cuda.synchronize()
print(rank, "side stream ok", float(u[0]), flush=True)
v = torch.zeros(66000, device=dev)
w = v[100:60000]
with torch.cuda.stream(s):
    w.fill_(1.0)
    dist.all_reduce(w)
torch.cuda.synchronize()
print(rank, "slice ok", float(w[0]), flush=True)
dist.destroy_process_group()
