#!/usr/bin/env python3
"""Probe: can two ranks (torchrun, gloo, both on device 0 here; one per GPU on a real node) map each other's device
buffers through torch's CUDA-IPC storage sharing, and write into them?  usage: torchrun --nproc-per-node 2 tools/proto/ipc_probe.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ndev = torch.cuda.device_count()
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % ndev)
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)
mine = torch.full((world, 1024), -1.0, dtype=torch.float64, device=dev)
h = mine.untyped_storage()._share_cuda_()
handles = [None] * world
dist.all_gather_object(handles, (h, mine.storage_offset(), tuple(mine.shape)))
peers = []
for r, (hr, off, shape) in enumerate(handles):
    if r == rank:
        peers.append(mine)
        continue
    st = torch.UntypedStorage._new_shared_cuda(*hr)
    t = torch.empty(0, dtype=torch.float64, device=torch.device("cuda", hr[0])).set_(st, off, shape)
    peers.append(t)
dist.barrier()
row = torch.full((1024,), float(rank + 1), dtype=torch.float64, device=dev)
t0 = time.perf_counter()
for p in peers:
    p[rank].copy_(row)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
dist.barrier()
torch.cuda.synchronize()
ok = all(bool((mine[r] == r + 1).all()) for r in range(world))
print(f"rank {rank}: device {dev}, peers' devices {[int(h_[0][0]) for h_ in handles]}, every row present: {ok}, push {dt * 1e6:.0f} us", flush=True)
dist.barrier()
del peers
dist.destroy_process_group()
sys.exit(0 if ok else 1)
