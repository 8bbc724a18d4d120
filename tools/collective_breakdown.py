#!/usr/bin/env python3
"""Where a compressed all-gather step's time goes at world 1 (one-rank RCCL group): every phase of
CompressedExchangePlan.all_gather timed with a device synchronise around it, next to the unsynchronised step."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dietgpu_amd import distributed as D  # noqa: E402

os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
D.init(backend="nccl", device=dev)
g = torch.Generator(device=dev).manual_seed(1)
shard = torch.randn((256, 512 * 1024), generator=g, device=dev).to(torch.bfloat16)
for chunks in (1, 2, 4):
    plan = D.CompressedAllGatherPlan(shard, chunks=chunks)
    for _ in range(20):
        plan.run(shard)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        plan.run(shard)
    torch.cuda.synchronize()
    print(f"chunks {chunks}: step {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms")

plan = D.CompressedAllGatherPlan(shard, chunks=1)
plan.run(shard)
W = plan.width
codec = plan.codec
snd = plan.send[: 256 * W].view(256, W)
rcv = plan.recv[: 256 * W]
cs = torch.cuda.current_stream().cuda_stream


def timed(name, fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"  {name:42s} {(time.perf_counter() - t0) / reps * 1e6:8.1f} us per call (back to back, one sync at the end)")


timed("compress_into (256 rows, capped)", lambda: codec.compress_into(shard, snd, W, plan.sizes, cs))
timed("all_gather_into_tensor (184 MB, 1 rank)", lambda: dist.all_gather_into_tensor(rcv, snd.view(-1)))
timed("decompress_from (256 rows, bounded)", lambda: codec.decompress_from(rcv.view(256, W), W, plan.out[0], plan.status[0], cs))
timed("stats ops (max + async all_reduce, count)", lambda: (plan._post_compress(), torch.sum(plan.status.view(-1), dim=0, keepdim=True, dtype=torch.int32, out=plan.stat[1:2])))
                                                      
timed("stats.tolist() (device-to-host + sync)", lambda: plan.stat.tolist())
timed("plain all_gather_into_tensor of the raw shard", lambda: dist.all_gather_into_tensor(plan.out.view(-1), shard.view(-1)))
dist.destroy_process_group()
