"""Compressed all-gather on the GPU: one-rank RCCL group (the GPU box has a
single device), real HIP float codec.  The two-rank exchange logic is covered on CPU by
tests/test_sharding_gloo.py::test_compressed_all_gather_world2."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_compressed_all_gather_single_rank_rccl():
    import torch.distributed as dist

    import dietgpu_amd
    from dietgpu_amd import distributed as D

    dietgpu_amd.lib()  # fails loudly if the HIP extension is missing
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    D.init(backend="nccl", device=dev)  # "nccl" is RCCL on ROCm
    try:
        g = torch.Generator(device="cpu").manual_seed(7)
        mine = [torch.randn(100_000 + 4096 * i, generator=g).to(torch.bfloat16).to(dev) for i in range(4)]
        gathered, stats = D.compressed_all_gather(mine)
        assert len(gathered) == 1
        for a, b in zip(gathered[0], mine):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        # bf16 N(0,1) compresses to ~0.68 of its size; the wire matrix is trimmed to the widest row
        assert stats["payload_bytes"] < 0.75 * stats["raw_bytes"]
        assert stats["wire_bytes"] < 0.80 * stats["raw_bytes"]
    finally:
        dist.destroy_process_group()


def test_pipelined_compressed_all_gather_single_rank_rccl():
    # fixed-width rows, no host synchronisation between phases, compress / exchange / decompress on
    # separate streams, through torch.ops.dietgpu.* (one-rank RCCL group: the box has one GPU)
    import torch.distributed as dist

    import dietgpu_amd
    from dietgpu_amd import distributed as D

    dietgpu_amd.lib()
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    D.init(backend="nccl", device=dev)
    try:
        g = torch.Generator(device="cpu").manual_seed(11)
        mine = [torch.randn(262144, generator=g).to(torch.bfloat16).to(dev) for _ in range(13)]
        gathered, stats = D.compressed_all_gather_pipelined(mine, chunks=4)
        assert stats["overflow_chunks"] == 0 and stats["wire_bytes"] <= 0.76 * stats["raw_bytes"]
        for a, b in zip(gathered[0], mine):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        # the reusable plan (what bench.py --collective times): C ABI calls on prebuilt pointer arrays
        shard = torch.stack(mine)
        plan = D.CompressedAllGatherPlan(shard, chunks=3)
        for _ in range(3):
            out, redo = plan.run(shard)
            torch.cuda.synchronize()
            assert redo == 0 and torch.equal(out[0].view(torch.int16), shard.view(torch.int16))
        assert bool(plan.status.all())
        # incompressible rows do not fit the fixed width: detected on the device, gathered again uncompressed
        noise = [torch.randint(-32768, 32767, (65536,), generator=g, dtype=torch.int16).to(dev).view(torch.bfloat16)
                 for _ in range(5)]
        gathered, stats = D.compressed_all_gather_pipelined(noise, chunks=2)
        assert stats["overflow_chunks"] == 2
        for a, b in zip(gathered[0], noise):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        nshard = torch.stack(noise)
        out, redo = D.CompressedAllGatherPlan(nshard, chunks=2).run(nshard)
        torch.cuda.synchronize()
        assert redo == 2 and torch.equal(out[0].view(torch.int16), nshard.view(torch.int16))
    finally:
        dist.destroy_process_group()


def test_bench_collective_mode_two_ranks_one_device():
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DGPU_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--collective", "--steps", "2", "--warmup", "1", "--batch", "16"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["bit_exact"] and d["config"]["overflow_chunks"] == 0
    assert d["config"]["wire_bytes_per_rank"] < d["config"]["per_rank_bytes"]
