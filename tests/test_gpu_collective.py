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
