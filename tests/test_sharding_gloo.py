"""N > 1 path on CPU: world_size-2 gloo processes exercise the batch sharding,
the compressed-size all-gather and the max-over-ranks timing reduction that
bench.py uses on RCCL.  The per-rank codec here is the CPU oracle (test
infrastructure): what is under test is the distributed plumbing."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_elements, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    import oracle as O
    import refgen
    from dietgpu_amd import distributed as D

    D.init(backend="gloo")
    start, end = D.shard_range(num_elements, rank, world)
    sizes = []
    for e in range(start, end):
        w = refgen.generate_floats(O.BFLOAT16, 1000 + 37 * e)
        sizes.append(O.float_compress(O.BFLOAT16, w, 10).size)
    local = torch.tensor(sizes, dtype=torch.int32)
    full = D.gather_sizes(local, num_elements)
    t = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    per_rank = D.gather_scalars(10.0 + rank, torch.device("cpu"))
    assert per_rank == [10.0 + r for r in range(world)]
    dist.barrier()
    q.put((rank, start, end, full.tolist(), t))
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    sys.path.insert(0, ROOT)
    from dietgpu_amd.distributed import shard_range

    for n in (0, 1, 7, 8, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                s, e = shard_range(n, r, world)
                assert 0 <= s <= e <= n
                covered.extend(range(s, e))
            assert covered == list(range(n))
    assert shard_range(2048, 3, 8) == (768, 1024)  # BASELINE config 5: 256 elements per GPU


def test_two_rank_gloo_size_allgather():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    import refgen

    world, n = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [O.float_compress(O.BFLOAT16, refgen.generate_floats(O.BFLOAT16, 1000 + 37 * e), 10).size for e in range(n)]
    seen = set()
    for rank, start, end, full, t in results:
        assert full == want          # every rank sees the whole batch's sizes
        assert t == 2.0              # max over ranks of (1 + rank)
        seen.update(range(start, end))
    assert seen == set(range(n))


# ---------------------------------------------------------------- compressed all-gather
class _OracleCodec:
    """CPU stand-in for the GPU float codec (test infrastructure): same interface, oracle inside."""

    def __init__(self, O):
        self.O = O

    def compress(self, tensors):
        O = self.O
        arch = [O.float_compress(O.BFLOAT16, t.view(torch.int16).numpy().view(np.uint16), 10) for t in tensors]
        cap = max(O.float_max_compressed_size(O.BFLOAT16, t.numel()) for t in tensors)
        comp = torch.zeros((len(arch), cap), dtype=torch.uint8)
        for i, a in enumerate(arch):
            comp[i, : a.size] = torch.from_numpy(a.copy())
        return comp, torch.tensor([a.size for a in arch], dtype=torch.int32)

    def decompress(self, rows, outs):
        O = self.O
        status = torch.ones((len(rows),), dtype=torch.uint8)
        for i, (r, o) in enumerate(zip(rows, outs)):
            a = r.numpy()
            # a row cut off at the exchange width (pipelined variant, overflow case) is not decodable
            ans_header_end = 16 + O.float_uncomp_data_size(O.BFLOAT16, o.numel()) + 32
            if a.size < ans_header_end or O.float_info(a)["compressed"] > a.size:
                status[i] = 0
                continue
            rc, w, _ = O.float_decompress(O.BFLOAT16, a, 10, o.numel())
            assert rc == 0 and w.size == o.numel()
            o.view(torch.int16).copy_(torch.from_numpy(w.view(np.int16).copy()))
        return status


def _cag_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    import oracle as O
    from dietgpu_amd import distributed as D

    D.init(backend="gloo")
    g = torch.Generator().manual_seed(100 + rank)
    mine = [torch.randn(5000 + 16 * i, generator=g).to(torch.bfloat16) for i in range(3)]
    gathered, stats = D.compressed_all_gather(mine, codec=_OracleCodec(O))
    # the pipelined variant (fixed-width rows, no host sync between phases) must return the same tensors;
    # with width_fraction 0.5 every bf16 N(0,1) row overflows and the uncompressed fallback runs
    same = [torch.randn(6000, generator=g).to(torch.bfloat16) for _ in range(5)]
    g2, stats2 = D.compressed_all_gather_pipelined(same, chunks=2, width_fraction=0.8, codec=_OracleCodec(O))
    g3, stats3 = D.compressed_all_gather_pipelined(same, chunks=3, width_fraction=0.5, codec=_OracleCodec(O))
    ok = stats2["overflow_chunks"] == 0 and stats3["overflow_chunks"] == 3 and stats2["wire_bytes"] < stats2["raw_bytes"]
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        _ = [torch.randn(5000 + 16 * i, generator=gr) for i in range(3)]
        want = [torch.randn(6000, generator=gr).to(torch.bfloat16) for _ in range(5)]
        for w_, b, c in zip(want, g2[r], g3[r]):
            ok = ok and torch.equal(w_.view(torch.int16), b.view(torch.int16)) and torch.equal(w_.view(torch.int16), c.view(torch.int16))
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        want = [torch.randn(5000 + 16 * i, generator=gr).to(torch.bfloat16) for i in range(3)]
        for a, b in zip(gathered[r], want):
            ok = ok and torch.equal(a.view(torch.int16), b.view(torch.int16))
    dist.barrier()
    q.put((rank, ok, stats))
    dist.destroy_process_group()


def test_compressed_all_gather_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cag_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, stats in res:
        assert ok, f"rank {rank}: gathered tensors differ"
        # bf16 N(0,1): the wire carries fewer bytes than the raw tensors
        assert stats["payload_bytes"] < stats["raw_bytes"]
        assert stats["wire_bytes"] < stats["raw_bytes"]


# ---------------------------------------------------------------- the exchange plan (all-gather + all-to-all)
class _OracleStrideCodec:
    """CPU stand-in for dietgpu_amd.distributed.CAbiFloatCodec (test infrastructure): rows are compressed by the
    oracle into a strided matrix, nothing is stored beyond the capacity; a row whose archive claims more bytes than
    the receiver has is rejected -- the contract of dgpu_float_compress_stride_capped / _decompress_stride_bounded."""

    def __init__(self, O, elems):
        self.O, self.elems = O, elems
        self.cap = O.float_max_compressed_size(O.BFLOAT16, elems)
        nb = (elems + 4095) // 4096
        self.min_width = (16 + O.float_uncomp_data_size(O.BFLOAT16, elems) + O.ans_compressed_overhead(nb) + 31) // 16 * 16

    def compress_into(self, rows, out, width, sizes, stream):
        O = self.O
        for i in range(rows.shape[0]):
            a = O.float_compress(O.BFLOAT16, rows[i].view(torch.int16).numpy().view(np.uint16), 10)
            k = min(a.size, width)
            out[i, :k] = torch.from_numpy(a[:k].copy())
            sizes[i] = a.size

    def decompress_from(self, rows, width, out, status, stream):
        O = self.O
        for i in range(rows.shape[0]):
            a = rows[i].numpy()
            info = O.float_info(a)
            if info["compressed"] > width:
                status[i] = 0
                continue
            rc, w, _ = O.float_decompress(O.BFLOAT16, a[: info["compressed"]], 10, self.elems)
            assert rc == 0
            out[i].view(torch.int16).copy_(torch.from_numpy(w.view(np.int16).copy()))
            status[i] = 1


def _plan_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    import oracle as O
    from dietgpu_amd import distributed as D

    D.init(backend="gloo")
    elems, rows = 3 * 4096 + 40, 6
    ok = True

    def shard_of(r, step, noisy=()):
        g = torch.Generator().manual_seed(1000 * step + r)
        x = torch.randn(rows, elems, generator=g).to(torch.bfloat16)
        for i in noisy:
            x[i] = torch.randint(-32768, 32767, (elems,), generator=g, dtype=torch.int16).view(torch.bfloat16)
        return x

    plan = D.CompressedExchangePlan(torch.bfloat16, elems, rows, chunks=2, device="cpu", codec=_OracleStrideCodec(O, elems))
    log = []
    # step 0: the width is probed from the data; step 1: rank 1 has two incompressible rows -> exactly those go
    # uncompressed; step 2: clean data again (the width has grown to the noisy rows' size: still no fall-back)
    for step, noisy in ((0, {}), (1, {1: (2, 5)}), (2, {})):
        mine = shard_of(rank, step, noisy.get(rank, ()))
        out, redo = plan.all_gather(mine)
        for r in range(world):
            want = shard_of(r, step, noisy.get(r, ()))
            ok = ok and torch.equal(out[r].view(torch.int16), want.view(torch.int16))
        log.append((redo, plan.last["width"], plan.last["largest_archive"], plan.last["rows_sent_uncompressed"]))
    ok = ok and log[0][0] == 0 and log[0][1] < int(0.72 * elems * 2)      # probed: ~0.68 of the raw row + headroom
    ok = ok and log[1][0] == (2 if rank == 1 else 0)                       # per ROW, on the rank that owns them
    ok = ok and log[2][0] == 0
    # all-to-all on a second plan: block j of `send` goes to rank j
    m = rows // world
    plan2 = D.CompressedExchangePlan(torch.bfloat16, elems, rows, chunks=2, device="cpu", codec=_OracleStrideCodec(O, elems))
    for step, noisy in ((0, {}), (1, {0: (4,)})):   # rank 0's row 4 = block 1, row 1: only rank 1 cannot decode it
        mine = shard_of(rank, 10 + step, noisy.get(rank, ())).view(world, m, elems).contiguous()
        got, redo = plan2.all_to_all(mine)
        for src in range(world):
            want = shard_of(src, 10 + step, noisy.get(src, ())).view(world, m, elems)[rank]
            ok = ok and torch.equal(got[src].view(torch.int16), want.view(torch.int16))
        log.append((redo, plan2.last["width"]))
    ok = ok and log[3][0] == 0 and log[4][0] == (1 if rank == 1 else 0)
    # a plan of depth 2, pipelined: step k is waited for BEHIND the call that enqueues step k + 1 (two buffer sets); the
    # step with incompressible rows falls back per row at its wait, the steps around it are untouched
    plan3 = D.CompressedExchangePlan(torch.bfloat16, elems, rows, chunks=2, device="cpu", codec=_OracleStrideCodec(O, elems), depth=2)
    steps = ((20, {}), (21, {0: (3,)}), (22, {}), (23, {}))
    shards = [shard_of(rank, st, nz.get(rank, ())) for st, nz in steps]
    handles, redos = [], []
    for k, mine in enumerate(shards):
        handles.append(plan3.all_gather_async(mine))
        if k >= 1:
            out, redo = handles[k - 1].wait()
            redos.append(redo)
            st, nz = steps[k - 1]
            for r in range(world):
                ok = ok and torch.equal(out[r].view(torch.int16), shard_of(r, st, nz.get(r, ())).view(torch.int16))
    out, redo = handles[-1].wait()
    redos.append(redo)
    for r in range(world):
        ok = ok and torch.equal(out[r].view(torch.int16), shard_of(r, steps[-1][0], {}).view(torch.int16))
    ok = ok and redos == [0, (1 if rank == 0 else 0), 0, 0] and handles[1].wait()[1] == redos[1]  # (wait is idempotent)
    # wait(clone=True): an output of the caller's own, which outlives the buffer set's reuse by step k + depth
    kept, _ = handles[-1].wait(clone=True)
    ok = ok and kept.data_ptr() != out.data_ptr() and torch.equal(kept.view(torch.int16), out.view(torch.int16))
    for extra in range(2):
        plan3.all_gather_async(shards[extra]).wait()
    for r in range(world):
        ok = ok and torch.equal(kept[r].view(torch.int16), shard_of(r, steps[-1][0], {}).view(torch.int16))
    log.append(tuple(redos))
    dist.barrier()
    q.put((rank, ok, log))
    dist.destroy_process_group()


def test_exchange_plan_world2_all_gather_and_all_to_all():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, log in res:
        assert ok, (rank, log)
