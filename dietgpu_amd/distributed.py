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
                st = codec.decompress(rows, outs)
                # any row that did not decode (cut off at the exchange width, or rejected for another reason) sends
                # its chunk through the uncompressed fall-back: a failed decode never passes for a result
                overflow_flags[-1] = overflow_flags[-1] | (st == 0).any().to(overflow_flags[-1].device)
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
# The same exchanges as a reusable PLAN for a fixed shard shape (a training loop moves the same buffers every
# step): every buffer and stream is created once; a step is K compress calls, K asynchronous collectives and
# K x world decompress calls straight on the C ABI -- no Python lists of tensors, no per-step allocation, no copy
# between the codec's output and the send buffer:
#
#   * rows travel at a fixed WIDTH (bytes).  The encoder writes every row straight into the send matrix at that
#     stride and stores nothing beyond it (dgpu_float_compress_stride_capped); the decoder is told how many bytes
#     a received row has and rejects a row whose archive claims more (dgpu_float_decompress_stride_bounded).  So a
#     row that did not fit shows up as status 0 on every receiver -- no size exchange on the data path at all;
#   * the width comes from the DATA: the first call probes (compresses its first chunk, one extra host sync, all
#     ranks agree through an all-reduce MAX); every call folds the largest compressed row of all ranks into the ONE
#     device-to-host read it ends with, and the next call uses that + 1/64 headroom.  Steady-state traffic has the
#     same statistics step after step (activations, gradients), so fall-backs are rare -- and cheap:
#   * fall-back is per ROW: rows with status 0 are gathered again uncompressed in one padded collective, the others
#     are not touched;
#   * all_gather(shard [n, elems]) -> [world, n, elems] and all_to_all(send [world, m, elems]) -> [world, m, elems]
#     share all of this.
# The codec is pluggable (compress rows into a strided matrix with a capacity, decompress rows with a bound), so the
# world-2 logic runs on CPU under gloo with the oracle as codec (tests/test_sharding_gloo.py).
class CAbiFloatCodec:
    """The HIP float codec through the C ABI's capped / bounded stride entry points."""

    def __init__(self, dtype, elems, max_rows, prob_bits=10, device=None):
        import ctypes as C

        from . import lib
        from .ops import _DTYPE_TO_FT

        self.C, self.L = C, lib()
        self.ft = _DTYPE_TO_FT[dtype]
        self.P = prob_bits
        self.elems = elems
        if elems <= 4096:
            # dgpu_float_compress_stride_capped takes rows of more than one 4096-word block (single-block rows go to
            # the two-elements-per-wavefront encoder, which does not clamp its copy-out); say so HERE, not at the
            # first exchange.  Rows that small gain nothing from the codec anyway (~700 bytes of tables per row).
            raise ValueError(f"compressed collectives need rows of more than 4096 elements (got {elems}): "
                             "exchange such shards uncompressed, or fold several rows into one")
        self.cap = int(self.L.dgpu_float_max_compressed_size(self.ft, elems))
        self.elem_bytes = torch.empty((), dtype=dtype).element_size()
        self.min_width = (16 + (elems + 15) // 16 * 16 * (3 if self.ft == 3 else 1) + 32 + 512 + 136 * ((elems + 4095) // 4096 + 1) + 15) // 16 * 16
        tb = max(int(self.L.dgpu_float_compress_temp_bytes(self.ft, max_rows, elems)),
                 int(self.L.dgpu_float_decompress_temp_bytes(self.ft, max_rows, elems, prob_bits)))
        # one temp region per stream: calls on different streams run concurrently
        self.temp_c = torch.empty((tb,), dtype=torch.uint8, device=device)
        self.temp_d = torch.empty((tb,), dtype=torch.uint8, device=device)
        self.err = C.c_int32(-1)

    def compress_into(self, rows, out, width, sizes, stream):
        """rows [m, elems] (contiguous) -> out [m, width] (contiguous uint8), sizes [m] int32 = full archive sizes."""
        C, L = self.C, self.L
        m = rows.shape[0]
        rc = L.dgpu_float_compress_stride_capped(
            C.c_void_p(self.temp_c.data_ptr()), self.temp_c.numel(), None, self.ft, self.P, 0, m,
            C.c_void_p(rows.data_ptr()), self.elems, self.elems * self.elem_bytes,
            C.c_void_p(out.data_ptr()), width, width, C.c_void_p(sizes.data_ptr()), C.c_void_p(stream))
        if rc:
            raise RuntimeError(L.dgpu_last_error().decode())

    def decompress_from(self, rows, width, out, status, stream):
        """rows [m, width] -> out [m, elems]; status [m] uint8 = 0 for rows that are incomplete / not decodable."""
        C, L = self.C, self.L
        m = rows.shape[0]
        rc = L.dgpu_float_decompress_stride_bounded(
            C.c_void_p(self.temp_d.data_ptr()), self.temp_d.numel(), None, self.ft, self.P, 0, m,
            C.c_void_p(rows.data_ptr()), width, width, C.c_void_p(out.data_ptr()), self.elems * self.elem_bytes,
            self.elems, C.c_void_p(status.data_ptr()), None, C.c_void_p(stream), C.byref(self.err))
        if rc:
            raise RuntimeError(L.dgpu_last_error().decode())


class CompressedExchangePlan:
    """all_gather / all_to_all of float rows that moves compressed bytes.  See the comment block above."""

    HEADROOM = 1.0 / 64.0

    def __init__(self, dtype, elems, rows, chunks=4, prob_bits=10, device=None, codec=None, initial_width=None, depth=1):
        """`depth`: buffer sets of the plan.  1: every call ends with its one host read (all_gather / all_to_all).  2: a
        step can be ENQUEUED (all_gather_async) while the previous one is still unchecked, so that the host read of step
        k (ExchangeHandle.wait) happens behind the launch of step k + 1 and a steady-state step is launch-only."""
        self.world = dist.get_world_size()
        self.dev = torch.device(device) if device is not None else torch.device("cpu")
        self.on_gpu = self.dev.type == "cuda"
        self.dtype, self.elems, self.rows = dtype, elems, rows
        self.row_bytes = elems * torch.empty((), dtype=dtype).element_size()
        self.codec = codec or CAbiFloatCodec(dtype, elems, rows, prob_bits, self.dev)
        self.cap = self.codec.cap
        self.chunks = max(1, min(chunks, rows))
        self.bounds = [shard_range(rows, k, self.chunks) for k in range(self.chunks)]
        u8 = dict(dtype=torch.uint8, device=self.dev)
        # send / receive matrices are flat and sized for the worst case; a step uses the first rows x width bytes.
        # `depth` sets of them; the attributes below (send, recv, sizes, status, stat, out) always name the set of the step
        # that is being enqueued or finished (_bind)
        self._slots = [{
            "send": torch.empty((rows * self.cap,), **u8),
            "recv": torch.empty((self.world * rows * self.cap,), **u8),
            "sizes": torch.zeros((rows,), dtype=torch.int32, device=self.dev),
            "status": torch.zeros((self.world, rows), **u8),
            "stat": torch.zeros((2,), dtype=torch.int32, device=self.dev),  # {largest archive of all ranks, rows decoded}
            "max_work": None, "out": None, "pending": None,
        } for _ in range(max(1, int(depth)))]
        self._cur = 0
        self._bind(self._slots[0])
        self.width = None if initial_width is None else self._round_width(initial_width)
        self._width_agreed = initial_width is None  # widths derived from all-reduced sizes agree by construction
        self.comp_stream = torch.cuda.Stream(self.dev) if self.on_gpu else None
        self.dec_stream = torch.cuda.Stream(self.dev) if self.on_gpu else None
        self.last = {}

    # ---- helpers
    def _bind(self, slot):
        self._slot = slot
        self.send, self.recv, self.sizes, self.status, self.stat = slot["send"], slot["recv"], slot["sizes"], slot["status"], slot["stat"]
        self._max_work, self.out = slot["max_work"], slot["out"]

    def _unbind(self):
        self._slot["max_work"], self._slot["out"] = self._max_work, self.out

    def _next_slot(self):
        """The buffer set of the step about to be enqueued; its previous step must have been waited for."""
        self._unbind()
        self._cur = (self._cur + 1) % len(self._slots)
        slot = self._slots[self._cur]
        if slot["pending"] is not None:
            slot["pending"].wait()  # (a caller that never waits: the step that used these buffers is finished first)
        self._bind(slot)
        return slot

    def _streams_of_step(self):
        """(caller's stream, compress stream, decompress stream) of the step being enqueued.  A step of ONE chunk has
        nothing to overlap between its compress and its decompress side: everything stays on the caller's stream, and
        the four stream hand-offs of the pipelined form -- an event record, a wait and ~15 us of idle GPU each at world
        1 -- are not paid (profiles/r06_collective_world1.txt)."""
        cur = torch.cuda.current_stream(self.dev) if self.on_gpu else None
        if not self.on_gpu or self.chunks == 1:
            return cur, cur, cur
        self.comp_stream.wait_stream(cur)  # the inputs were produced on the caller's stream
        self.dec_stream.wait_stream(cur)
        return cur, self.comp_stream, self.dec_stream

    def _join_streams_of_step(self, cur, cs, ds):
        if self.on_gpu and cs is not cur:
            cur.wait_stream(cs)
            cur.wait_stream(ds)

    def _round_width(self, nbytes):
        w = (int(nbytes) + 15) // 16 * 16
        return max(min(w, self.cap), min(self.codec.min_width, self.cap))

    def _stream_ptr(self, s):
        return s.cuda_stream if s is not None else 0

    def _on(self, s):
        return torch.cuda.stream(s) if s is not None else _NullContext()

    def _agree_on_width(self):
        """A caller-given initial width must be the SAME on every rank (it is the row stride of the collectives):
        the first call makes it so (MAX over ranks, one tiny all-reduce, once)."""
        if self._width_agreed or self.world == 1:
            self._width_agreed = True
            return
        w = torch.tensor([self.width], dtype=torch.int32, device=self.dev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        self.width = self._round_width(int(w.item()))
        self._width_agreed = True

    def _probe_width(self, rows2d):
        """First call: compress the first chunk once, all ranks agree on max(size) -> width (one host sync)."""
        lo, hi = self.bounds[0]
        m = hi - lo
        view = self.send[: m * self.cap].view(m, self.cap)
        self.codec.compress_into(rows2d[lo:hi], view, self.cap, self.sizes[lo:hi], self._stream_ptr(torch.cuda.current_stream(self.dev) if self.on_gpu else None))
        mx = self.sizes[lo:hi].max().to(torch.int32).reshape(1)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        self.width = self._round_width(int(mx.item()) * (1.0 + self.HEADROOM) + 64)
        self._width_agreed = True

    def _post_compress(self):
        """Right after the last compress call (on the compress stream): the largest archive of all ranks, as an
        asynchronous all-reduce that overlaps with the exchange and the decompression."""
        torch.amax(self.sizes, dim=0, keepdim=True, out=self.stat[0:1])
        self._max_work = dist.all_reduce(self.stat[0:1], op=dist.ReduceOp.MAX, async_op=True) if self.world > 1 else None

    def _post_decode(self):
        """The device side of a step's check, enqueued with the step (on the caller's stream, behind the decode stream):
        rows that decoded, agreed over the ranks -- the fall-back is a collective, every rank must agree on whether it
        runs (in the all-to-all only sender and receiver see a row's status; in the all-gather a local decode rejection
        must not desynchronise the ranks) -- and both statistics on their way to pinned host memory."""
        torch.sum(self.status.view(-1), dim=0, keepdim=True, dtype=torch.int32, out=self.stat[1:2])
        if self.world > 1:
            dist.all_reduce(self.stat[1:2], op=dist.ReduceOp.MIN)
        if self._max_work is not None:
            self._max_work.wait()
            self._max_work = None
        slot = self._slot
        slot["width"] = self.width
        if self.on_gpu:
            if slot.get("host") is None:
                slot["host"] = torch.empty((2,), dtype=torch.int32).pin_memory()
                slot["event"] = torch.cuda.Event()
            slot["host"].copy_(self.stat, non_blocking=True)
            slot["event"].record()

    def _finish(self, kind, fallback):
        """The ONE device-to-host read of a step: {largest archive over all ranks, rows that decoded}."""
        slot = self._slot
        if self.on_gpu:
            slot["event"].synchronize()
            largest, decoded = (int(v) for v in slot["host"].tolist())
        else:
            largest, decoded = (int(v) for v in self.stat.tolist())
        failed = self.status.numel() - decoded
        used_width = slot["width"]
        redo = fallback() if failed else 0
        self.last = {"kind": kind, "width": used_width, "largest_archive": largest, "rows_sent_uncompressed": redo,
                     "wire_bytes": self.rows * used_width + redo * self.row_bytes, "raw_bytes": self.rows * self.row_bytes}
        # next step's width: what the data needed + headroom (it only moves when the data moves)
        self.width = self._round_width(largest * (1.0 + self.HEADROOM) + 64)
        return redo

    def _handle(self, kind, fallback):
        self._post_decode()
        self._unbind()
        h = ExchangeHandle(self, self._slot, kind, fallback)
        self._slot["pending"] = h
        return h

    # ---- all-gather
    def all_gather(self, shard):
        """shard [rows, elems] -> (gathered [world, rows, elems], rows that had to be sent uncompressed)."""
        return self.all_gather_async(shard).wait()

    def all_gather_async(self, shard):
        """Enqueues the step and returns an ExchangeHandle WITHOUT reading anything back: `handle.wait()` -> (gathered,
        rows sent uncompressed) does the step's one host read and, if rows did not fit the exchange width, their
        uncompressed fall-back.  The output is complete only after wait(); `shard` must stay unchanged until then.  With
        a plan of depth 2 the wait of step k belongs behind the call that enqueues step k + 1."""
        assert shard.shape == (self.rows, self.elems) and shard.dtype == self.dtype and shard.is_contiguous()
        self._next_slot()
        if self.out is None or self.out.shape[0] != self.world or self.out.dim() != 3:
            self.out = torch.empty((self.world, self.rows, self.elems), dtype=self.dtype, device=self.dev)
        if self.width is None:
            self._probe_width(shard)
        self._agree_on_width()
        W, world = self.width, self.world
        cur, cs, ds = self._streams_of_step()
        works = []
        with self._on(cs):
            for k, (lo, hi) in enumerate(self.bounds):
                m = hi - lo
                snd = self.send[lo * W : hi * W].view(m, W)
                self.codec.compress_into(shard[lo:hi], snd, W, self.sizes[lo:hi], self._stream_ptr(cs))
                rcv = self.recv[world * lo * W : world * hi * W]
                works.append(_all_gather_flat(rcv, snd.view(-1), world))
            self._post_compress()
        with self._on(ds):
            for k, (lo, hi) in enumerate(self.bounds):
                m = hi - lo
                if works[k] is not None:
                    works[k].wait()
                rcv = self.recv[world * lo * W : world * hi * W].view(world, m, W)
                for r in range(world):
                    self.codec.decompress_from(rcv[r], W, self.out[r, lo:hi], self.status[r, lo:hi], self._stream_ptr(ds))
        self._join_streams_of_step(cur, cs, ds)

        status_of_step, out_of_step = self.status, self.out

        def fallback():
            # per ROW: the rows that failed are gathered again uncompressed, padded to the largest count of any rank.
            # Every rank decoded the same archives, so the status matrices SHOULD be equal -- but the fall-back is a
            # collective whose shapes derive from them, so they are made equal (a row is bad if ANY rank could not
            # decode it: MIN over ranks of a world x rows byte matrix) rather than assumed to be.
            agreed = status_of_step.to(torch.int32)
            if world > 1:
                dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            bad = (agreed == 0)                                       # [world, rows]
            counts = bad.sum(dim=1).tolist()
            width = max(counts)
            mine = bad[dist.get_rank()].nonzero().flatten()
            pack = torch.zeros((width, self.elems), dtype=self.dtype, device=self.dev)
            pack[: mine.numel()] = shard[mine]
            got = torch.empty((world, width, self.elems), dtype=self.dtype, device=self.dev)
            w = _all_gather_flat(got.view(-1), pack.view(-1), world)
            if w is not None:
                w.wait()
            for r in range(world):
                idx = bad[r].nonzero().flatten()
                out_of_step[r, idx] = got[r, : idx.numel()]
            return counts[dist.get_rank()]

        # (the count of decoded rows is all-reduced too: every rank must agree on WHETHER the fall-back collective runs)
        return self._handle("all_gather", fallback)

    # ---- all-to-all
    def all_to_all(self, send):
        """send [world, m, elems] (row block j goes to rank j) -> (received [world, m, elems] (block i came from rank i),
        rows of this rank's receive side that had to be sent uncompressed)."""
        return self.all_to_all_async(send).wait()

    def all_to_all_async(self, send):
        """all_to_all without the host read: see all_gather_async."""
        world = self.world
        assert send.dim() == 3 and send.shape[0] == world and send.shape[2] == self.elems and send.is_contiguous()
        m = send.shape[1]
        assert world * m == self.rows and send.dtype == self.dtype
        self._next_slot()
        if self.out is None or self.out.shape != send.shape:
            self.out = torch.empty_like(send)
        flat = send.view(world * m, self.elems)
        if self.width is None:
            self._probe_width(flat)
        self._agree_on_width()
        W = self.width
        cur, cs, ds = self._streams_of_step()
        # chunk k = rows [lo_k, hi_k) of EVERY destination block, so that each chunk is a complete all-to-all
        cb = [shard_range(m, k, min(self.chunks, m)) for k in range(min(self.chunks, m))]
        works = []
        base = 0
        with self._on(cs):
            for lo, hi in cb:
                c = hi - lo
                snd = self.send[base * W : (base + world * c) * W].view(world, c, W)
                for j in range(world):
                    self.codec.compress_into(send[j, lo:hi], snd[j], W, self.sizes[j * m + lo : j * m + hi], self._stream_ptr(cs))
                rcv = self.recv[base * W : (base + world * c) * W]
                works.append((_all_to_all_flat(rcv, snd.view(-1), world), base, lo, hi))
                base += world * c
            self._post_compress()
        st = self.status.view(-1)[: world * m].view(world, m)
        with self._on(ds):
            for wk, b0, lo, hi in works:
                c = hi - lo
                if wk is not None:
                    wk.wait()
                rcv = self.recv[b0 * W : (b0 + world * c) * W].view(world, c, W)
                for i in range(world):
                    self.codec.decompress_from(rcv[i], W, self.out[i, lo:hi], st[i, lo:hi], self._stream_ptr(ds))
        self._join_streams_of_step(cur, cs, ds)
        if world * m < self.status.numel():
            self.status.view(-1)[world * m :] = 1
        out_of_step = self.out

        def fallback():
            # per ROW.  Only sender and receiver know which rows failed: the receiver tells every sender (a tiny
            # all-gather of the status matrix), then the rows travel uncompressed in an all-to-all with uneven splits
            mine = (st == 0).to(torch.uint8)                              # [src, m] as seen by me, the receiver
            allst = torch.empty((world, world, m), dtype=torch.uint8, device=self.dev)
            w0 = _all_gather_flat(allst.view(-1), mine.reshape(-1).contiguous(), world)
            if w0 is not None:
                w0.wait()
            me = dist.get_rank()
            bad_send = allst[:, me, :].bool()                             # [dst, m]: my rows that dst could not decode
            bad_recv = mine.bool()                                        # [src, m]
            in_split = bad_send.sum(dim=1).tolist()
            out_split = bad_recv.sum(dim=1).tolist()
            pack = send[bad_send]                                         # rows ordered by destination
            got = torch.empty((sum(out_split), self.elems), dtype=self.dtype, device=self.dev)
            dist.all_to_all_single(got, pack.contiguous(), out_split, in_split)
            out_of_step[bad_recv] = got
            return sum(out_split)

        return self._handle("all_to_all", fallback)


class ExchangeHandle:
    """A step of a CompressedExchangePlan that has been enqueued but not checked.  wait() -> (output, rows that had to be
    sent uncompressed): the step's ONE device-to-host read (largest archive of all ranks, rows that decoded), the per-row
    uncompressed fall-back if rows did not fit, and the width of the next step.

    Lifetimes and ordering (the plan owns `depth` buffer sets and hands them out round robin):
      * the tensor wait() returns IS the buffer set's output: it is valid until that buffer set is used again, i.e.
        until the enqueue of step k + depth -- copy it (or pass clone=True) to keep it longer;
      * the caller's input (`shard` / `send`) must stay unchanged until wait(): the per-row fall-back reads it then;
      * wait() may run a collective (the fall-back), so every rank must wait for its handles in the SAME order relative
        to its other collectives -- in particular before enqueueing step k + depth: an enqueue that finds its buffer
        set still pending finishes that step first, and does so on every rank only if every rank left it pending."""

    def __init__(self, plan, slot, kind, fallback):
        self.plan, self.slot, self.kind, self.fallback, self.result = plan, slot, kind, fallback, None

    def wait(self, clone=False):
        if self.result is None:
            p = self.plan
            cur = p._slot
            p._unbind()
            p._bind(self.slot)
            try:
                redo = p._finish(self.kind, self.fallback)
                self.result = (p.out, redo)
            finally:
                p._unbind()
                p._bind(cur)
            self.slot["pending"] = None
        if clone:  # an output of the caller's own instead of the buffer set's (valid beyond step k + depth)
            return (self.result[0].clone(), self.result[1])
        return self.result


def _all_gather_flat(out_flat, in_flat, world):
    """all-gather of flat byte / word tensors, asynchronous where the backend can (returns the work handle or None)."""
    if dist.get_backend() == "gloo" or not in_flat.is_cuda:
        parts = list(out_flat.view(world, -1).unbind(0))
        dist.all_gather(parts, in_flat)
        return None
    return dist.all_gather_into_tensor(out_flat, in_flat, async_op=True)


def _all_to_all_flat(out_flat, in_flat, world):
    """all-to-all of equal flat slices (slice j of in_flat goes to rank j), asynchronous on RCCL."""
    if not in_flat.is_cuda or dist.get_backend() == "gloo":
        dist.all_to_all_single(out_flat, in_flat)
        return None
    return dist.all_to_all_single(out_flat, in_flat, async_op=True)


class CompressedAllGatherPlan(CompressedExchangePlan):
    """The all-gather of one [n, elems] shard (kept under its round-2 name; bench.py --collective)."""

    def __init__(self, shard, chunks=4, prob_bits=10, initial_width=None, codec=None, depth=1):
        assert shard.dim() == 2 and shard.is_contiguous()
        super().__init__(shard.dtype, shard.shape[1], shard.shape[0], chunks=chunks, prob_bits=prob_bits,
                         device=shard.device, codec=codec, initial_width=initial_width, depth=depth)

    def run(self, shard):
        return self.all_gather(shard)

    def run_async(self, shard):
        return self.all_gather_async(shard)

    @property
    def wire_bytes(self):
        return self.last.get("wire_bytes", self.rows * (self.width or self.cap))
