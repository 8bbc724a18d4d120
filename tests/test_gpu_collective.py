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
        # the reusable plan (what bench.py --collective times): the C ABI's capped / bounded stride entry points,
        # rows compressed straight into the send matrix at a width taken from the data
        shard = torch.stack(mine)
        plan = D.CompressedAllGatherPlan(shard, chunks=3)
        for step in range(3):
            out, redo = plan.run(shard)
            torch.cuda.synchronize()
            assert redo == 0 and torch.equal(out[0].view(torch.int16), shard.view(torch.int16))
            assert plan.last["width"] < 0.72 * shard.shape[1] * 2      # ~0.68 of the raw row + 1/64 headroom
            assert plan.last["largest_archive"] <= plan.last["width"]
        assert bool(plan.status.all())
        # incompressible rows do not fit the width: exactly THOSE rows are gathered again uncompressed
        noise = [torch.randint(-32768, 32767, (262144,), generator=g, dtype=torch.int16).to(dev).view(torch.bfloat16)
                 for _ in range(2)]
        mixed = torch.stack(mine[:5] + noise[:1] + mine[5:] + noise[1:])
        plan2 = D.CompressedAllGatherPlan(mixed, chunks=2)
        plan2.width = plan.width                                       # as if the previous steps had been compressible
        out, redo = plan2.run(mixed)
        torch.cuda.synchronize()
        assert redo == 2 and torch.equal(out[0].view(torch.int16), mixed.view(torch.int16))
        assert plan2.status.view(-1).tolist().count(0) == 2
        out, redo = plan2.run(mixed)                                   # the width followed the data: nothing falls back
        torch.cuda.synchronize()
        assert redo == 0 and torch.equal(out[0].view(torch.int16), mixed.view(torch.int16))
        gathered, stats = D.compressed_all_gather_pipelined(noise, chunks=2)
        assert stats["overflow_chunks"] == 2
        for a, b in zip(gathered[0], noise):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        # BASELINE config 4 data (fp16, 50 % zeros, ratio 0.7498) must NOT fall back (round 2's fixed 0.75 did)
        import refgen

        sp = torch.from_numpy(refgen.sparse_fp16(8, 512 * 1024).view("int16")).to(dev).view(torch.float16)
        plan4 = D.CompressedAllGatherPlan(sp, chunks=2, prob_bits=11)
        for _ in range(2):
            out, redo = plan4.run(sp)
            torch.cuda.synchronize()
            assert redo == 0 and torch.equal(out[0].view(torch.int16), sp.view(torch.int16))
        # a plan of depth 2, pipelined: nothing is read back when a step is enqueued; the wait of step k comes behind the
        # launch of step k + 1 (different data per step: the two buffer sets must not be mixed up); the step with
        # incompressible rows falls back at ITS wait
        plan6 = D.CompressedAllGatherPlan(shard, chunks=2, depth=2)
        noisy = shard.clone()
        noisy[3], noisy[9] = noise[0], noise[1]
        inputs = [shard, noisy, shard.flip(0).contiguous(), shard]
        handles = []
        for k, x in enumerate(inputs):
            handles.append(plan6.run_async(x))
            if k:
                out, redo = handles[k - 1].wait()
                torch.cuda.synchronize()
                assert redo == (2 if k - 1 == 1 else 0) and torch.equal(out[0].view(torch.int16), inputs[k - 1].view(torch.int16))
        out, redo = handles[-1].wait()
        torch.cuda.synchronize()
        assert redo == 0 and torch.equal(out[0].view(torch.int16), shard.view(torch.int16))
        # all-to-all on the same machinery (world 1: block 0 comes back)
        plan5 = D.CompressedExchangePlan(torch.bfloat16, shard.shape[1], shard.shape[0], chunks=2, device=dev)
        got, redo = plan5.all_to_all(shard.view(1, shard.shape[0], shard.shape[1]))
        torch.cuda.synchronize()
        assert redo == 0 and torch.equal(got.view(torch.int16), shard.view(1, *shard.shape).view(torch.int16))
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
    assert d["n_gpus"] == 2 and d["bit_exact"] and d["config"]["rows_sent_uncompressed"] == 0
    assert d["config"]["wire_bytes_per_rank"] < d["config"]["per_rank_bytes"]


def test_bench_helpers_over_rccl_at_world_one():
    # bench.py's three collectives on the codec path -- D.max_over_ranks (the step time), D.gather_scalars (every rank's
    # step time) and D.gather_sizes (the one all-gather of compressed sizes after the timed region) -- through backend
    # "nccl" (= RCCL) with device tensors: the branch the 2- and 8-rank gloo runs never take.  One rank is what a
    # one-GPU box offers; the calls, dtypes, devices and shapes are those of N ranks.
    import torch.distributed as dist

    from dietgpu_amd import distributed as D

    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    D.init(backend="nccl", device=dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        assert D.max_over_ranks(0.125, dev) == 0.125
        assert D.gather_scalars(0.25, dev) == [0.25]
        sizes = torch.arange(1, 257, dtype=torch.int32, device=dev)
        got = D.gather_sizes(sizes, 256)
        assert got.device.type == "cuda" and torch.equal(got, sizes)
        assert D.shard_range(256, 0, 1) == (0, 256)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("chunks", [1, 3])
def test_exchange_plan_stream_forms_agree(chunks):
    # a plan of ONE chunk runs its step on the caller's stream alone (no hand-offs between a compress and a decompress
    # stream); more chunks pipeline over three streams.  Same payload either way, and a kernel that follows on the
    # caller's stream sees the complete output.
    import torch.distributed as dist

    from dietgpu_amd import distributed as D

    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    D.init(backend="nccl", device=dev)
    try:
        g = torch.Generator(device=dev).manual_seed(3)
        shard = torch.randn((24, 40_000), generator=g, device=dev).to(torch.bfloat16)
        plan = D.CompressedAllGatherPlan(shard, chunks=chunks)
        for _ in range(3):
            out, redo = plan.all_gather(shard)
            copy = out.clone()  # (on the caller's stream, right behind the step)
            torch.cuda.synchronize()
            assert redo == 0 and torch.equal(copy[0].view(torch.int16), shard.view(torch.int16))
        h = plan.all_gather_async(shard)
        out, redo = h.wait(clone=True)
        assert redo == 0 and torch.equal(out[0].view(torch.int16), shard.view(torch.int16))
    finally:
        dist.destroy_process_group()
