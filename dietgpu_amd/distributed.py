"""Multi-GPU plumbing for the batched API: one process per GPU (torchrun),
independent batch elements sharded contiguously across ranks, no data-path
collective.  The codec has no exchange step (every element carries its own
statistics), so RCCL is used only to (a) all-gather the 4-byte compressed sizes
so every rank knows the whole batch's layout and (b) reduce timings.

Backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None, device=None):
    """Initialises torch.distributed from the torchrun environment (RANK, WORLD_SIZE, MASTER_*)."""
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl" and device is not None:
        kwargs["device_id"] = device
    dist.init_process_group(backend=backend, **kwargs)


def shard_range(num_elements, rank, world):
    """Element-contiguous partition: rank g of G owns [g*B/G, (g+1)*B/G) with the
    remainder spread over the first ranks."""
    base, rem = divmod(num_elements, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_sizes(local_sizes, num_elements):
    """All-gathers per-element compressed sizes (int32 tensor of this rank's shard,
    on the backend's device) into the full [num_elements] vector on every rank."""
    world = dist.get_world_size()
    counts = [shard_range(num_elements, r, world) for r in range(world)]
    width = max(e - s for s, e in counts)
    padded = torch.zeros((width,), dtype=torch.int32, device=local_sizes.device)
    padded[: local_sizes.numel()] = local_sizes
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return torch.cat([o[: e - s] for o, (s, e) in zip(out, counts)])


def max_over_ranks(seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
