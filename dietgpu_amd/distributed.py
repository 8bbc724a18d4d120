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


def gather_scalars(value, device):
    """Every rank's float, in rank order, on every rank."""
    if dist.get_backend() != "nccl":
        device = "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def max_over_ranks(seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---------------------------------------------------------------------------
# Compressed all-gather (the use the reference's README
# motivates, README.md:68-72,104): every rank compresses its tensors with the
# float codec, the ranks exchange sizes and then the compressed rows over
# RCCL/xGMI, and every rank decompresses everything it received.  Two phases
# because rows are variable-length: (1) all-gather of the int32 sizes, (2) one
# all-gather of a [rows, width] byte matrix whose width is the largest
# compressed row across ranks (rounded to 16 bytes) -- not the worst-case
# capacity the codec wrote into.  xGMI is point-to-point and a ring all-gather
# is bound by one link per hop, so the bytes saved by the codec (x0.67 for bf16
# gradients/activations) translate directly into collective time.
class GpuFloatCodec:
    """Default codec of the compressed collectives: the HIP float codec through `torch.ops.dietgpu.*`
    (csrc/torch_ops.cpp over the C ABI: no per-call Python marshalling of pointer arrays).  `temp_mem`
    (optional uint8 tensor) is handed to every call, as the reference's ops take it."""

    def __init__(self, temp_mem=None):
        from . import load_torch_ops

        self.ops = load_torch_ops()
        self.temp_mem = temp_mem

    def compress(self, tensors):
        comp, sizes, _ = self.ops.compress_data(True, tensors, False, self.temp_mem)
        return comp, sizes

    def decompress(self, rows, outs):
        status = torch.zeros((len(rows),), dtype=torch.uint8, device=outs[0].device)
        self.ops.decompress_data(True, rows, outs, False, self.temp_mem, status, None)
        return status


def compressed_all_gather(tensors, codec=None):
    """All-gathers a list of equally-shaped float tensors per rank, moving compressed bytes.

    `tensors`: this rank's list (same count, shapes and dtype on every rank).
    Returns (gathered, stats): gathered[r][i] is rank r's i-th tensor, bit-exact;
    stats = {"raw_bytes", "wire_bytes", "payload_bytes"} per rank-to-rank copy.
    `codec` needs compress(list) -> (uint8 [n, cap], int32 [n]) and
    decompress(list of uint8 rows, list of outputs) -> uint8 status [n]."""
    codec = codec or GpuFloatCodec()
    world = dist.get_world_size()
    n = len(tensors)
    dev = tensors[0].device
    comp, sizes = codec.compress(tensors)
    sizes = sizes.to(torch.int32)

    # phase 1: sizes
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = torch.stack(all_sizes)  # [world, n]
    width = int(all_sizes.max().item())
    width = (width + 15) // 16 * 16

    # phase 2: payload, trimmed to the widest compressed row
    payload = comp[:, :width].contiguous()
    gathered_payload = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered_payload, payload)

    gathered = []
    for r in range(world):
        rows = [gathered_payload[r][i, : int(all_sizes[r, i])] for i in range(n)]
        outs = [torch.empty_like(t) for t in tensors]
        status = codec.decompress(rows, outs)
        if not bool(status.all().item()):
            raise RuntimeError(f"decompression of rank {r}'s rows failed")
        gathered.append(outs)
    raw = sum(t.numel() * t.element_size() for t in tensors)
    stats = {
        "raw_bytes": raw,
        "wire_bytes": n * width,
        "payload_bytes": int(all_sizes[dist.get_rank()].sum().item()),
    }
    return gathered, stats


# ---------------------------------------------------------------------------
# Pipelined compressed all-gather: no host synchronisation between the phases, compression of chunk
# k + 1 overlapped with the exchange of chunk k (and with the decompression of chunk k - 1).
#
#   * rows are exchanged at a FIXED width W = width_fraction x the raw row bytes (rounded up to 16), so the
#     payload collective's shape is known on the host without reading the compressed sizes back.  bf16 / fp16
#     activations and gradients compress to ~0.67 (README.md:45-60 of the reference); the default 0.75
#     leaves headroom.  A row that does not fit is detected from the all-gathered sizes ON THE DEVICE;
#   * the only host synchronisation is ONE read of the "some row overflowed" flag at the very end; if it
#     is set (incompressible data) the affected chunks are gathered again uncompressed;
#   * streams: compress on `comp_stream`; every collective is asynchronous (RCCL runs it on its own
#     stream, ordered after the compress stream at the call); decompress on `dec_stream` after the
#     work handle.  On CPU tensors (gloo, the unit tests) everything degenerates to in-order execution.
def compressed_all_gather_pipelined(tensors, chunks=4, width_fraction=0.75, codec=None):
    """All-gathers a list of equally-shaped float tensors per rank, moving compressed bytes.

    Returns (gathered, stats): gathered[r][i] is rank r's i-th tensor, bit-exact;
    stats = {"raw_bytes", "wire_bytes", "overflow_chunks"} per rank-to-rank copy."""
    codec = codec or GpuFloatCodec()
    world = dist.get_world_size()
    n = len(tensors)
    dev = tensors[0].device
    on_gpu = dev.type == "cuda"
    row_bytes = tensors[0].numel() * tensors[0].element_size()
    assert all(t.shape == tensors[0].shape and t.dtype == tensors[0].dtype for t in tensors)
    # archive = 16-byte float header + non-compressed plane + ANS archive; W must at least hold the overhead
    width = (int(row_bytes * width_fraction) + 15) // 16 * 16
    bounds = [shard_range(n, k, min(chunks, n)) for k in range(min(chunks, n))]

    cur = torch.cuda.current_stream(dev) if on_gpu else None
    comp_stream = torch.cuda.Stream(dev) if on_gpu else None
    dec_stream = torch.cuda.Stream(dev) if on_gpu else None
    if on_gpu:
        comp_stream.wait_stream(cur)  # the inputs were produced on the caller's stream
        dec_stream.wait_stream(cur)

    def on(stream):
        return torch.cuda.stream(stream) if on_gpu else _NullContext()

    pending = []   # (chunk index, payload work, sizes work, gathered payloads, gathered sizes)
    keep = []      # buffers RCCL may still be reading
    for k, (lo, hi) in enumerate(bounds):
        with on(comp_stream):
            comp, sizes = codec.compress(tensors[lo:hi])
            sizes = sizes.to(torch.int32)
            w = min(width, comp.shape[1])
            payload = comp[:, :w].contiguous()
            gp = [torch.empty_like(payload) for _ in range(world)]
            gs = [torch.empty_like(sizes) for _ in range(world)]
            wp = dist.all_gather(gp, payload, async_op=True)
            ws = dist.all_gather(gs, sizes, async_op=True)
            keep.append((comp, payload, sizes))
        pending.append((k, w, wp, ws, gp, gs))

    gathered = [[None] * n for _ in range(world)]
    overflow_flags = []
    for k, w, wp, ws, gp, gs in pending:
        lo, hi = bounds[k]
        with on(dec_stream):
            wp.wait()
            ws.wait()
            all_sizes = torch.stack(gs)                      # [world, rows] on the device
            over = all_sizes > w
            overflow_flags.append(over.any())                # stays on the device
            for r in range(world):
                # rows cut off at the exchange width must not be decoded: blank their header word (device-side)
                gp[r][:, :4].masked_fill_(over[r][:, None], 0)
                rows = [gp[r][i] for i in range(hi - lo)]    # fixed-width rows; the archives say how long they are
                outs = [torch.empty_like(t) for t in tensors[lo:hi]]
                codec.decompress(rows, outs)
                gathered[r][lo:hi] = outs
    if on_gpu:
        cur.wait_stream(dec_stream)
        cur.wait_stream(comp_stream)

    # the one host synchronisation: did any row not fit into the fixed width?
    flags = torch.stack(overflow_flags).to("cpu")
    redo = [k for k in range(len(bounds)) if bool(flags[k])]
    for k in redo:
        lo, hi = bounds[k]
        raw = torch.stack([t.reshape(-1) for t in tensors[lo:hi]])
        got = [torch.empty_like(raw) for _ in range(world)]
        dist.all_gather(got, raw)
        for r in range(world):
            gathered[r][lo:hi] = [got[r][i].reshape(tensors[lo + i].shape) for i in range(hi - lo)]
    stats = {
        "raw_bytes": n * row_bytes,
        "wire_bytes": n * min(width, keep[0][0].shape[1]) + sum((hi - lo) * row_bytes for lo, hi in (bounds[k] for k in redo)),
        "overflow_chunks": len(redo),
        "chunks": len(bounds),
    }
    return gathered, stats


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ---------------------------------------------------------------------------
# The same exchange as a reusable PLAN for a fixed shard shape (a training loop gathers the same
# buffers every step): every buffer, pointer array and stream is created once, a step is K compress
# calls, 2 K asynchronous collectives and K x world decompress calls straight on the C ABI -- no Python
# lists of tensors, no per-step allocation (the generic function above spends milliseconds of host time
# on 256-tensor lists, far more than the codec's 0.25 ms).
class CompressedAllGatherPlan:
    def __init__(self, shard, chunks=4, width_fraction=0.75, prob_bits=10):
        import ctypes as C

        from . import lib
        from .ops import _DTYPE_TO_FT

        assert shard.is_cuda and shard.dim() == 2 and shard.is_contiguous()
        self.C, self.L = C, lib()
        self.world = dist.get_world_size()
        self.dev = shard.device
        self.n, self.elems = shard.shape
        self.dtype = shard.dtype
        self.ft = _DTYPE_TO_FT[shard.dtype]
        self.P = prob_bits
        self.row_bytes = self.elems * shard.element_size()
        self.cap = int(self.L.dgpu_float_max_compressed_size(self.ft, self.elems))
        self.width = min((int(self.row_bytes * width_fraction) + 15) // 16 * 16, self.cap)
        self.bounds = [shard_range(self.n, k, min(chunks, self.n)) for k in range(min(chunks, self.n))]
        u8 = dict(dtype=torch.uint8, device=self.dev)
        self.comp = torch.empty((self.n, self.cap), **u8)
        self.payload = torch.empty((self.n, self.width), **u8)
        self.sizes = torch.zeros((self.n,), dtype=torch.int32, device=self.dev)
        self.out = torch.empty((self.world, self.n, self.elems), dtype=self.dtype, device=self.dev)
        self.status = torch.zeros((self.world, self.n), **u8)
        self.gp = [torch.empty((self.world, hi - lo, self.width), **u8) for lo, hi in self.bounds]
        self.gs = [torch.empty((self.world, hi - lo), dtype=torch.int32, device=self.dev) for lo, hi in self.bounds]
        self.overflow = torch.zeros((len(self.bounds),), dtype=torch.bool, device=self.dev)
        tb = max(int(self.L.dgpu_float_compress_temp_bytes(self.ft, self.n, self.elems)),
                 int(self.L.dgpu_float_decompress_temp_bytes(self.ft, self.n, self.elems, prob_bits)))
        # one temp region per stream: calls on different streams run concurrently
        self.temp_c = torch.empty((tb,), **u8)
        self.temp_d = torch.empty((tb,), **u8)
        self.comp_stream = torch.cuda.Stream(self.dev)
        self.dec_stream = torch.cuda.Stream(self.dev)
        self.err = C.c_int32(-1)
        self._shard_ptr = None

    def _arrays(self, shard):
        C = self.C
        if self._shard_ptr == shard.data_ptr():
            return
        self._shard_ptr = shard.data_ptr()
        self.c_in, self.c_out, self.c_sz, self.d_in, self.d_out = [], [], [], [], []
        for k, (lo, hi) in enumerate(self.bounds):
            m = hi - lo
            self.c_in.append((C.c_void_p * m)(*[shard.data_ptr() + (lo + i) * self.row_bytes for i in range(m)]))
            self.c_out.append((C.c_void_p * m)(*[self.comp.data_ptr() + (lo + i) * self.cap for i in range(m)]))
            self.c_sz.append((C.c_uint32 * m)(*([self.elems] * m)))
            self.d_in.append([(C.c_void_p * m)(*[self.gp[k].data_ptr() + (r * m + i) * self.width for i in range(m)])
                              for r in range(self.world)])
            self.d_out.append([(C.c_void_p * m)(*[self.out.data_ptr() + ((r * self.n) + lo + i) * self.row_bytes
                                                   for i in range(m)]) for r in range(self.world)])

    def run(self, shard):
        """Returns (gathered [world, n, elems], number of chunks that had to be gathered uncompressed)."""
        C, L = self.C, self.L
        assert shard.shape == (self.n, self.elems) and shard.dtype == self.dtype and shard.is_contiguous()
        self._arrays(shard)
        cur = torch.cuda.current_stream(self.dev)
        self.comp_stream.wait_stream(cur)
        self.dec_stream.wait_stream(cur)
        works = []
        with torch.cuda.stream(self.comp_stream):
            cs = C.c_void_p(self.comp_stream.cuda_stream)
            for k, (lo, hi) in enumerate(self.bounds):
                m = hi - lo
                rc = L.dgpu_float_compress(C.c_void_p(self.temp_c.data_ptr()), self.temp_c.numel(), None, self.ft, self.P, 0,
                                           m, self.c_in[k], self.c_sz[k], self.c_out[k],
                                           C.c_void_p(self.sizes.data_ptr() + 4 * lo), cs)
                if rc:
                    raise RuntimeError(L.dgpu_last_error().decode())
                self.payload[lo:hi].copy_(self.comp[lo:hi, : self.width])  # fixed-width rows for the exchange
                wp = dist.all_gather_into_tensor(self.gp[k].view(self.world * m, self.width), self.payload[lo:hi], async_op=True)
                ws = dist.all_gather_into_tensor(self.gs[k].view(-1), self.sizes[lo:hi], async_op=True)
                works.append((wp, ws))
        with torch.cuda.stream(self.dec_stream):
            ds = C.c_void_p(self.dec_stream.cuda_stream)
            for k, (lo, hi) in enumerate(self.bounds):
                m = hi - lo
                works[k][0].wait()
                works[k][1].wait()
                over = self.gs[k] > self.width                    # [world, m], on the device
                self.overflow[k] = over.any()
                # a row cut off at the exchange width must not be decoded (its block table points past the
                # row): blank its header word so that the decoder rejects it -- still no host involvement
                self.gp[k][:, :, :4].masked_fill_(over[:, :, None], 0)
                for r in range(self.world):
                    rc = L.dgpu_float_decompress(C.c_void_p(self.temp_d.data_ptr()), self.temp_d.numel(), None, self.ft, self.P, 0,
                                                 m, self.d_in[k][r], self.d_out[k][r], self.c_sz[k],
                                                 C.c_void_p(self.status.data_ptr() + r * self.n + lo), None, ds,
                                                 C.byref(self.err))
                    if rc:
                        raise RuntimeError(L.dgpu_last_error().decode())
        cur.wait_stream(self.comp_stream)
        cur.wait_stream(self.dec_stream)
        # the one host synchronisation: rows that did not fit the fixed width (incompressible data)
        redo = [k for k, f in enumerate(self.overflow.tolist()) if f]
        for k in redo:
            lo, hi = self.bounds[k]
            got = torch.empty((self.world, hi - lo, self.elems), dtype=self.dtype, device=self.dev)
            dist.all_gather_into_tensor(got.view(self.world * (hi - lo), self.elems), shard[lo:hi])
            self.out[:, lo:hi] = got
        return self.out, len(redo)

    @property
    def wire_bytes(self):
        return self.n * self.width
