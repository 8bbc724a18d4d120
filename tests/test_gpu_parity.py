"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bit-exact everywhere (integer / byte work): compressed archives must equal the
oracle's byte for byte, decoded data must equal the original.  Shapes follow the
reference's own tests (ANSTest.cu:243-282, ANSStatisticsTest.cu:44-207,
FloatTest.cu:270-311, ans_test.py, float_test.py) plus BASELINE.md's configs.
Nothing here reads /root/reference.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle as O
import refgen

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
FT_DTYPE = {O.FLOAT16: torch.float16, O.BFLOAT16: torch.bfloat16, O.FLOAT32: torch.float32}


class TorchOpsSurface:
    """The same calls through torch.ops.dietgpu.* (csrc/torch_ops.cpp: the reference's op names and schemas,
    DietGpu.cpp:915-937) -- the fast tensor surface.  The ops fix probBits at 10 (DietGpu.cpp:114): tests that ask
    for another precision are skipped on this surface.  Everything else (lib(), size queries) is the package's."""

    name = "torch_ops"

    def __init__(self, pkg):
        self._pkg = pkg
        self.ops = pkg.load_torch_ops()

    def __getattr__(self, attr):
        return getattr(self._pkg, attr)

    @staticmethod
    def _p10(prob_bits):
        if prob_bits != 10:
            pytest.skip("torch.ops.dietgpu fixes probBits at 10 (DietGpu.cpp:114)")

    def compress_data(self, as_float, ts, checksum=False, temp_mem=None, out_compressed=None, out_compressed_bytes=None, prob_bits=10):
        self._p10(prob_bits)
        return self.ops.compress_data(as_float, ts, checksum, temp_mem, out_compressed, out_compressed_bytes)

    def decompress_data(self, as_float, ts_in, ts_out, checksum=False, temp_mem=None, out_status=None,
                        out_decompressed_words=None, prob_bits=10):
        self._p10(prob_bits)
        return self.ops.decompress_data(as_float, ts_in, ts_out, checksum, temp_mem, out_status, out_decompressed_words)

    def compress_data_split_size(self, as_float, t_in, splits, checksum=False, temp_mem=None, out_compressed=None,
                                 out_compressed_bytes=None, prob_bits=10):
        self._p10(prob_bits)
        return self.ops.compress_data_split_size(as_float, t_in, splits, checksum, temp_mem, out_compressed, out_compressed_bytes)

    def decompress_data_split_size(self, as_float, ts_in, t_out, splits, checksum=False, temp_mem=None, out_status=None,
                                   out_decompressed_words=None, prob_bits=10):
        self._p10(prob_bits)
        return self.ops.decompress_data_split_size(as_float, ts_in, t_out, splits, checksum, temp_mem, out_status,
                                                   out_decompressed_words)

    def compress_data_simple(self, as_float, ts, checksum=False, prob_bits=10):
        self._p10(prob_bits)
        return self.ops.compress_data_simple(as_float, ts, checksum)

    def decompress_data_simple(self, as_float, ts, checksum=False, prob_bits=10):
        self._p10(prob_bits)
        return self.ops.decompress_data_simple(as_float, ts, checksum)


# Every test of this module (and of the modules that import the fixture) runs on BOTH tensor surfaces: the ctypes
# mirror dietgpu_amd.ops and torch.ops.dietgpu.* -- the parity evidence is carried by the fast surface too.
@pytest.fixture(scope="module", params=["ctypes", "torch_ops"])
def dg(request):
    import dietgpu_amd

    dietgpu_amd.lib()  # fails loudly if the HIP extension is missing
    if request.param == "torch_ops":
        return TorchOpsSurface(dietgpu_amd)
    # the ctypes route of dietgpu_amd.ops at EVERY precision (by default it hands prob_bits 10 to torch.ops.dietgpu)
    dietgpu_amd.prefer_torch_ops(False)
    request.addfinalizer(lambda: dietgpu_amd.prefer_torch_ops(True))
    return dietgpu_amd


def to_dev_bytes(x):
    return torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).copy()).to(DEV)


def words_to_tensor(ft, w):
    t = torch.from_numpy(np.ascontiguousarray(w).view(np.int32 if ft == O.FLOAT32 else np.int16).copy()).to(DEV)
    return t.view(FT_DTYPE[ft])


def tensor_to_words(ft, t):
    v = t.view(torch.int32 if ft == O.FLOAT32 else torch.int16).cpu().numpy()
    return v.view(np.uint32 if ft == O.FLOAT32 else np.uint16)


def roundup(n, m):
    return (n + m - 1) // m * m


def gpu_ans_encode(dg, xs, prob_bits=10, checksum=False, temp=None):
    ts = [to_dev_bytes(x) for x in xs]
    comp, sizes, _ = dg.compress_data(False, ts, checksum, temp, prob_bits=prob_bits)
    sizes = sizes.cpu().numpy()
    comp = comp.cpu().numpy()
    return [comp[i, : sizes[i]].copy() for i in range(len(xs))]


def gpu_ans_decode(dg, archives, sizes, prob_bits=10, checksum=False):
    ins = [to_dev_bytes(a) for a in archives]
    outs = [torch.empty((n,), dtype=torch.uint8, device=DEV) for n in sizes]
    status = torch.zeros((len(ins),), dtype=torch.uint8, device=DEV)
    osz = torch.zeros((len(ins),), dtype=torch.int32, device=DEV)
    dg.decompress_data(False, ins, outs, checksum, None, status, osz, prob_bits=prob_bits)
    return [o.cpu().numpy() for o in outs], status.cpu().numpy(), osz.cpu().numpy()


# ----------------------------------------------------------------- statistics
@pytest.mark.parametrize("size", [1, 2, 11, 32, 55, 64, 99, 1000, 12345, 1234567])
def test_histogram_unaligned(dg, size):
    # ANSStatisticsTest.cu:44-95: batch of 3, stride = size + 11 => unaligned starts
    B, stride = 3, size + 11
    rng = np.random.default_rng(size)
    buf = rng.integers(0, 256, B * stride + 64, dtype=np.uint8)
    t = torch.from_numpy(buf).to(DEV)
    hist = torch.zeros((B, 256), dtype=torch.int32, device=DEV)
    for off in (0, 1, 5):
        rc = dg.lib().dgpu_ans_histogram_batch_stride(
            B, C.c_void_p(t.data_ptr() + off), size, stride, C.c_void_p(hist.data_ptr()),
            C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        h = hist.cpu().numpy()
        for b in range(B):
            ref = np.bincount(buf[off + b * stride : off + b * stride + size], minlength=256)
            assert (h[b] == ref).all()


def gpu_normalize(dg, counts, totals, prob_bits):
    B = counts.shape[0]
    hist = torch.from_numpy(counts.astype(np.int64).astype(np.int32)).to(DEV).contiguous()
    sizes = torch.from_numpy(np.asarray(totals, np.int64).astype(np.int32)).to(DEV)
    table = torch.zeros((B, 256, 4), dtype=torch.int32, device=DEV)
    rc = dg.lib().dgpu_ans_calc_weights(
        B, prob_bits, C.c_void_p(sizes.data_ptr()), 0, C.c_void_p(hist.data_ptr()),
        C.c_void_p(table.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    return table.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_normalize_matches_oracle(dg, prob_bits):
    rng = np.random.default_rng(prob_bits)
    rows = []
    # reference known answers (ANSStatisticsTest.cu:127-167)
    d = np.ones(10000, np.uint8)
    d[:256] = np.arange(256)
    rows.append(np.bincount(d, minlength=256))
    rows.append(np.full(256, 64))
    # adversarial fp32 rounding: totals near powers of two, tiny and huge counts
    for _ in range(200):
        k = int(rng.integers(1, 257))
        c = np.zeros(256, np.int64)
        idx = rng.choice(256, k, replace=False)
        mode = rng.integers(0, 4)
        if mode == 0:
            c[idx] = rng.integers(1, 1 << 20, k)
        elif mode == 1:
            c[idx] = rng.integers(1, 4, k)
            c[idx[0]] = int(rng.integers(1 << 20, 1 << 31))
        elif mode == 2:
            c[idx] = (rng.pareto(0.7, k) * 10 + 1).astype(np.int64) % (1 << 24) + 1
        else:
            c[idx] = 1 << int(rng.integers(0, 20))
        rows.append(c)
    rows.append(np.zeros(256, np.int64))  # empty element
    counts = np.stack(rows).astype(np.uint32)
    totals = counts.astype(np.int64).sum(1)
    got = gpu_normalize(dg, counts, totals, prob_bits)
    for b in range(counts.shape[0]):
        want = O.normalize(counts[b], int(totals[b]), prob_bits)
        live = want[:, 0] > 0
        assert (got[b][:, 0] == want[:, 0]).all(), f"pdf row {b}"
        assert (got[b][:, 1] == want[:, 1]).all(), f"cdf row {b}"
        assert (got[b][live] == want[live]).all(), f"magic/shift row {b}"


# ------------------------------------------------------------------ ANS codec
SIZE_SETS = [[1], [1, 1], [4096, 4095, 4096], [1234, 2345, 3456], [10000, 10013, 10000]]


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
@pytest.mark.parametrize("lam", [1.0, 10.0, 100.0, 1000.0])
def test_ans_batch_pointer(dg, prob_bits, lam):
    # ANSTest.cu:248-260, checksum on
    for sizes in SIZE_SETS:
        xs = [refgen.generate_symbols(n, lam) for n in sizes]
        got = gpu_ans_encode(dg, xs, prob_bits, checksum=True)
        for x, g in zip(xs, got):
            want = O.ans_encode(x, prob_bits, use_checksum=True)
            assert g.size % 16 == 0
            assert g.size == want.size and (g == want).all()
        # decode from buffers truncated to the reported size (ans_test.py:21-26)
        outs, status, osz = gpu_ans_decode(dg, got, sizes, prob_bits, checksum=True)
        for x, o, s, z in zip(xs, outs, status, osz):
            assert s == 1 and z == x.size and (o == x).all()


def test_ans_batch_pointer_large(dg):
    # ANSTest.cu:262-275: 100 elements, sizes U[100, 10000]
    rng = np.random.default_rng(10)
    sizes = rng.integers(100, 10000, 100).tolist()
    xs = [refgen.generate_symbols(n, 20.0) for n in sizes]
    got = gpu_ans_encode(dg, xs, 10)
    for x, g in zip(xs, got):
        want = O.ans_encode(x, 10)
        assert g.size == want.size and (g == want).all()
    outs, status, _ = gpu_ans_decode(dg, got, sizes, 10)
    assert status.all()
    for x, o in zip(xs, outs):
        assert (o == x).all()


def test_ans_decodes_oracle_archives(dg):
    # archives produced by the CPU oracle decode on the GPU (and vice versa above)
    xs = [refgen.generate_symbols(n, 30.0) for n in (1, 33, 4096, 8191, 100000)]
    for p in (9, 10, 11):
        arch = [O.ans_encode(x, p) for x in xs]
        outs, status, osz = gpu_ans_decode(dg, arch, [x.size for x in xs], p)
        assert status.all()
        for x, o in zip(xs, outs):
            assert (o == x).all()


def test_ans_batch_stride(dg):
    # ANSTest.cu:277-282: 13 x 8208 bytes, through the stride entry points
    B, n = 13, 8208
    x = refgen.generate_symbols(B * n, 20.0).reshape(B, n)
    t = torch.from_numpy(x.copy()).to(DEV)
    lib = dg.lib()
    stride = int(lib.dgpu_ans_max_compressed_size(n))
    comp = torch.zeros((B, stride), dtype=torch.uint8, device=DEV)
    sizes = torch.zeros((B,), dtype=torch.int32, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.dgpu_ans_encode_batch_stride(None, 0, None, 10, 1, B, C.c_void_p(t.data_ptr()), n, n, None,
                                          C.c_void_p(comp.data_ptr()), stride, C.c_void_p(sizes.data_ptr()), st)
    assert rc == 0
    hs = sizes.cpu().numpy()
    hc = comp.cpu().numpy()
    for b in range(B):
        want = O.ans_encode(x[b], 10, use_checksum=True)
        assert hs[b] == want.size and (hc[b, : hs[b]] == want).all()
    out = torch.zeros((B, n), dtype=torch.uint8, device=DEV)
    status = torch.zeros((B,), dtype=torch.uint8, device=DEV)
    osz = torch.zeros((B,), dtype=torch.int32, device=DEV)
    err = C.c_int32(-1)
    rc = lib.dgpu_ans_decode_batch_stride(None, 0, None, 10, 1, B, C.c_void_p(comp.data_ptr()), stride,
                                          C.c_void_p(out.data_ptr()), n, n, C.c_void_p(status.data_ptr()),
                                          C.c_void_p(osz.data_ptr()), st, C.byref(err))
    assert rc == 0 and err.value == -1
    assert status.cpu().numpy().all() and (osz.cpu().numpy() == n).all()
    assert (out.cpu().numpy() == x).all()


def test_ans_empty(dg):
    # ans_test.py:68-77
    ts = [torch.empty((0,), dtype=torch.uint8, device=DEV)]
    comp = dg.compress_data_simple(False, ts, True)
    assert comp[0].numel() == 544
    assert (comp[0].cpu().numpy() == O.ans_encode(np.zeros(0, np.uint8), 10, use_checksum=True)).all()
    dec = dg.decompress_data_simple(False, comp, True)
    assert dec[0].numel() == 0


def test_ans_capacity_too_small(dg):
    x = refgen.generate_symbols(5000, 20.0)
    arch = gpu_ans_encode(dg, [x, x])
    ins = [to_dev_bytes(a) for a in arch]
    outs = [torch.zeros((4999,), dtype=torch.uint8, device=DEV), torch.zeros((5000,), dtype=torch.uint8, device=DEV)]
    status = torch.zeros((2,), dtype=torch.uint8, device=DEV)
    osz = torch.zeros((2,), dtype=torch.int32, device=DEV)
    dg.decompress_data(False, ins, outs, False, None, status, osz)
    assert status.cpu().tolist() == [0, 1]
    assert osz.cpu().tolist() == [5000, 5000]  # required size is reported on failure
    assert (outs[0].cpu().numpy() == 0).all()  # nothing written
    assert (outs[1].cpu().numpy() == x).all()


def test_ans_checksum_mismatch_detected(dg):
    x = refgen.generate_symbols(20000, 20.0)
    arch = gpu_ans_encode(dg, [x], checksum=True)[0]
    bad = arch.copy()
    bad[20] ^= 0x5A  # stored checksum
    out = torch.zeros((x.size,), dtype=torch.uint8, device=DEV)
    with pytest.raises(RuntimeError, match="checksum mismatch"):
        dg.decompress_data(False, [to_dev_bytes(bad)], [out], True)
    dg.decompress_data(False, [to_dev_bytes(arch)], [out], True)
    assert (out.cpu().numpy() == x).all()


def test_ans_temp_mem_and_usage(dg):
    # with and without caller temp memory (ans_test.py:51-66); float32 tensors bytewise
    temp = torch.empty((64 * 1024 * 1024,), dtype=torch.uint8, device=DEV)
    g = torch.Generator(device="cpu").manual_seed(1)
    ts = [torch.randn((n,), generator=g).to(DEV) for n in (10000, 100000, 1000000)]
    for tm in (None, temp):
        for checksum in (False, True):
            comp, sizes, used = dg.compress_data(False, ts, checksum, tm)
            assert used > 0
            trunc = [comp[i, : int(sizes[i])].clone() for i in range(len(ts))]
            outs = [torch.empty_like(t) for t in ts]
            status = torch.zeros((len(ts),), dtype=torch.uint8, device=DEV)
            osz = torch.zeros((len(ts),), dtype=torch.int32, device=DEV)
            dg.decompress_data(False, trunc, outs, checksum, tm, status, osz)
            for t, o, s, z in zip(ts, outs, status.tolist(), osz.tolist()):
                assert s == 1 and z == t.numel() * 4 and torch.equal(t, o)


def test_ans_split_size(dg):
    # ans_test.py:79-139
    import random

    random.seed(3)
    for _ in range(3):
        sizes = []
        for _ in range(random.randrange(1, 15)):
            s = random.randrange(1, 10000)
            sizes.append(s + 4 - (s % 4))
        t = torch.randint(0, 65, (sum(sizes),), dtype=torch.uint8, device=DEV)
        sizes_t = torch.IntTensor(sizes)
        splits = torch.split(t, sizes)
        comp_ts, _, _ = dg.compress_data_split_size(False, t, sizes_t, True)
        host = t.cpu().numpy()
        off = 0
        for c, s in zip(comp_ts, sizes):
            want = O.ans_encode(host[off : off + s], 10, use_checksum=True)
            assert c.numel() == want.size and (c.cpu().numpy() == want).all()
            off += s
        dec = dg.decompress_data_simple(False, [c.clone() for c in comp_ts], True)
        for a, b in zip(splits, dec):
            assert torch.equal(a, b)
        out = torch.empty_like(t)
        simple = dg.compress_data_simple(False, list(splits), True)
        dg.decompress_data_split_size(False, simple, out, sizes_t, True)
        assert torch.equal(t, out)


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_ans_worst_case_block(dg, prob_bits):
    # a block of globally rare symbols: exercises the encoder's LDS stage bound
    rng = np.random.default_rng(3)
    x = np.zeros(1 << 20, np.uint8)
    x[:4096] = rng.integers(1, 256, 4096, dtype=np.uint8)
    x[40960:45056] = rng.integers(1, 256, 4096, dtype=np.uint8)
    got = gpu_ans_encode(dg, [x], prob_bits)[0]
    want = O.ans_encode(x, prob_bits)
    assert got.size == want.size and (got == want).all()
    outs, status, _ = gpu_ans_decode(dg, [got], [x.size], prob_bits)
    assert status.all() and (outs[0] == x).all()


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_decoder_staging_modes_raw(dg, prob_bits):
    # The decoder stages a wave's two blocks WHOLE when both have <= 1024 compressed words and goes through
    # the 2 KiB word ring otherwise (wave-uniform choice): one element holds waves of both kinds and
    # waves that mix a compressible with an incompressible block; 16-block and small tiles, a partial tail.
    rng = np.random.default_rng(4242 + prob_bits)

    def blocks(kinds, tail=0):
        parts = []
        for k in kinds:
            if k == "c":  # a few symbols: ~1.5 bits each
                parts.append(rng.choice(np.array([0, 1, 2, 3], np.uint8), 4096, p=[0.7, 0.2, 0.07, 0.03]))
            else:         # uniform bytes: far above the table's average code length
                parts.append(rng.integers(0, 256, 4096, dtype=np.uint8))
        if tail:
            parts.append(rng.integers(0, 4, tail, dtype=np.uint8))
        return np.concatenate(parts)

    xs = [blocks("ccrrcrrc" * 5, 777), blocks("ccrcrr"), blocks("cc"), blocks("rr"), blocks("c" * 33, 4095)]
    got = gpu_ans_encode(dg, xs, prob_bits)
    for x, g in zip(xs, got):
        want = O.ans_encode(x, prob_bits)
        assert g.size == want.size and (g == want).all()
    outs, status, osz = gpu_ans_decode(dg, got, [x.size for x in xs], prob_bits)
    assert status.all() and osz.tolist() == [x.size for x in xs]
    for x, o in zip(xs, outs):
        assert (o == x).all()


def test_pointer_batches_in_arithmetic_progression(dg):
    # A pointer batch whose addresses form an arithmetic progression with equal sizes is handled as a stride batch
    # (no parameter block): ascending rows, DESCENDING rows (the stride wraps around 2^64), the same row four times
    # (stride 0), a batch of one -- and the same rows in scrambled order (a genuine pointer list) for comparison.
    rng = np.random.default_rng(99)
    n = 4096 * 5 + 123
    words = (rng.standard_normal((6, n)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    t = torch.from_numpy(words.view(np.int16)).to(DEV).view(torch.bfloat16)
    want = [O.float_compress(O.BFLOAT16, words[i], 10) for i in range(6)]
    for order in ([0, 1, 2, 3, 4, 5], [5, 4, 3, 2, 1, 0], [2, 2, 2, 2], [3], [0, 2, 4], [4, 0, 5, 1]):
        ts = [t[i] for i in order]
        comp, sizes, _ = dg.compress_data(True, ts, False, prob_bits=10)
        hs = sizes.cpu().numpy()
        hc = comp.cpu().numpy()
        arch = []
        for k, i in enumerate(order):
            assert hs[k] == want[i].size and (hc[k, : hs[k]] == want[i]).all(), (order, k)
            arch.append(comp[k, : hs[k]])
        # decode into the rows of one output tensor, in the same (possibly descending) order
        out = torch.zeros((6, n), dtype=torch.bfloat16, device=DEV)
        outs = [out[i] for i in order] if len(set(order)) == len(order) else [torch.empty_like(t[0]) for _ in order]
        status = torch.zeros((len(order),), dtype=torch.uint8, device=DEV)
        dg.decompress_data(True, arch, outs, False, None, status, None, prob_bits=10)
        assert status.cpu().numpy().all()
        for k, i in enumerate(order):
            assert (tensor_to_words(O.BFLOAT16, outs[k]) == words[i]).all(), (order, k)


@pytest.mark.parametrize("workload", ["bf16", "u8"])
def test_stride_detected_batches_through_the_c_abi(dg, workload):
    # bench.Codec hands the C ABI the rows of one tensor as pointer lists (unbounded decode entry points): both
    # directions take the stride path; then the same with two elements swapped (parameter block).  Archives of
    # every row against the oracle, round trip exact.
    import bench

    data, ft, _, P, _ = bench.make_workload(workload, 12, 4321, DEV, 4096 * 3 + 640)
    codec = bench.Codec(dg, data, ft, P)
    for scattered in (False, True):
        if scattered:
            codec.scatter_pointers()
        codec.sizes.zero_()
        codec.out.zero_()
        codec.step()
        codec.verify()
        hs = codec.sizes.cpu().numpy()
        hc = codec.comp.cpu().numpy()
        rows = data.view(torch.uint8).cpu().numpy().reshape(12, -1)
        for k in range(12):
            r = k if not scattered or k > 1 else 1 - k  # element k of the call is row r
            if ft:
                want = O.float_compress(ft, rows[r].view(np.uint16), P)
            else:
                want = O.ans_encode(rows[r], P)
            assert hs[k] == want.size and (hc[r, : hs[k]] == want).all(), (workload, scattered, k)


def _block_words(ans_archive):
    """compressed u16 words of every block of an ANS archive (ANSCoalescedHeader, GpuANSUtils.cuh:67-229)"""
    hdr = ans_archive[:32].view(np.uint32)
    nb = int(hdr[1])
    off = 32 + 512 + 128 * nb
    bw = ans_archive[off : off + 8 * nb].view(np.uint32).reshape(nb, 2)
    return bw[:, 0] & 0xFFFF


def test_decoder_staging_boundary(dg):
    # 17 nearly equiprobable symbols = a little over 4 bits per symbol: every block has 1024 .. 1027 words, i.e.
    # sits on or just past the whole-block staging limit (1024 words), so waves of both kinds alternate
    rng = np.random.default_rng(777)
    p = np.array([1.3, 1.2, 1.1] + [1.0] * 11 + [0.9, 0.8, 0.7])
    x = rng.choice(17, 4096 * 96, p=p / p.sum()).astype(np.uint8) * 13 + 3
    w = _block_words(O.ans_encode(x, 10))
    assert (w <= 1024).any() and (w > 1024).any() and (np.abs(w.astype(int) - 1024) <= 2).any(), (w.min(), w.max())
    got = gpu_ans_encode(dg, [x], 10)
    want = O.ans_encode(x, 10)
    assert got[0].size == want.size and (got[0] == want).all()
    outs, status, _ = gpu_ans_decode(dg, got, [x.size], 10)
    assert status.all() and (outs[0] == x).all()
    # the same exponent bytes inside bf16 words (float join at the flush on both staging paths)
    words = (x.astype(np.uint16) << 7) | rng.integers(0, 128, x.size, dtype=np.uint16) | (rng.integers(0, 2, x.size, dtype=np.uint16) << 15)
    t = words_to_tensor(O.BFLOAT16, words)
    comp, sizes, _ = dg.compress_data(True, [t], False, prob_bits=10)
    n = int(sizes[0].item())
    wantf = O.float_compress(O.BFLOAT16, words, 10)
    assert n == wantf.size and (comp[0, :n].cpu().numpy() == wantf).all()
    out = torch.empty_like(t)
    status = torch.zeros((1,), dtype=torch.uint8, device=DEV)
    dg.decompress_data(True, [comp[0, :n]], [out], False, None, status, None, prob_bits=10)
    assert status.cpu().numpy().all() and (tensor_to_words(O.BFLOAT16, out) == words).all()


# ---------------------------------------------------------------- float codec
@pytest.mark.parametrize("ft", [O.FLOAT16, O.BFLOAT16, O.FLOAT32])
@pytest.mark.parametrize("prob_bits", [9, 10])
def test_float_batch(dg, ft, prob_bits):
    # FloatTest.cu:270-284 (+ BatchSize1 :300-311)
    rng = np.random.default_rng(ft * 100 + prob_bits)
    for batch in (1, 3, 16, 23):
        for mult16 in (False, True):
            ns = []
            for _ in range(batch):
                n = int(rng.integers(1, 8192))
                ns.append((n + 15) // 16 * 16 if mult16 else n)
            ws = [refgen.generate_floats(ft, n) for n in ns]
            ts = [words_to_tensor(ft, w) for w in ws]
            comp, sizes, _ = dg.compress_data(True, ts, True, prob_bits=prob_bits)
            hs = sizes.cpu().numpy()
            hc = comp.cpu().numpy()
            arch = []
            for i, w in enumerate(ws):
                want = O.float_compress(ft, w, prob_bits, use_checksum=True)
                assert hs[i] == want.size and (hc[i, : hs[i]] == want).all(), (ft, prob_bits, batch, i)
                arch.append(comp[i, : hs[i]].clone())
            outs = [torch.empty_like(t) for t in ts]
            status = torch.zeros((batch,), dtype=torch.uint8, device=DEV)
            osz = torch.zeros((batch,), dtype=torch.int32, device=DEV)
            dg.decompress_data(True, arch, outs, True, None, status, osz, prob_bits=prob_bits)
            assert status.cpu().numpy().all() and osz.cpu().tolist() == ns
            for w, o in zip(ws, outs):
                assert (tensor_to_words(ft, o) == w).all()


@pytest.mark.parametrize("ft", [O.FLOAT16, O.BFLOAT16, O.FLOAT32])
@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_float_incompressible_exponents(dg, ft, prob_bits):
    # random bit patterns: ~8 bits of entropy per exponent byte, so every block overflows the
    # float encoder's small LDS stage and goes through its spill path (full and partial
    # blocks, several tiles, ragged batch); one element mixes both regimes
    rng = np.random.default_rng(1000 + ft * 10 + prob_bits)
    dt = np.uint32 if ft == O.FLOAT32 else np.uint16
    hi = 1 << (32 if ft == O.FLOAT32 else 16)
    ns = [4096 * 40, 4096 * 9 + 1234, 777, 4096 * 16, 4096 * 33 + 5]
    ws = [rng.integers(0, hi, n, dtype=np.uint64).astype(dt) for n in ns]
    mixed = refgen.generate_floats(ft, 4096 * 24)
    mixed[4096 * 5 : 4096 * 9] = rng.integers(0, hi, 4096 * 4, dtype=np.uint64).astype(dt)
    ws.append(mixed)
    ts = [words_to_tensor(ft, w) for w in ws]
    comp, sizes, _ = dg.compress_data(True, ts, False, prob_bits=prob_bits)
    hs = sizes.cpu().numpy()
    hc = comp.cpu().numpy()
    arch = []
    for i, w in enumerate(ws):
        want = O.float_compress(ft, w, prob_bits)
        assert hs[i] == want.size and (hc[i, : hs[i]] == want).all(), (ft, prob_bits, i)
        arch.append(comp[i, : hs[i]].clone())
    outs = [torch.empty_like(t) for t in ts]
    status = torch.zeros((len(ts),), dtype=torch.uint8, device=DEV)
    osz = torch.zeros((len(ts),), dtype=torch.int32, device=DEV)
    dg.decompress_data(True, arch, outs, False, None, status, osz, prob_bits=prob_bits)
    assert status.cpu().numpy().all()
    for w, o in zip(ws, outs):
        assert (tensor_to_words(ft, o) == w).all()


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_ragged_batches(dg, seed):
    # random ragged batches: element sizes from empty to ~60 blocks (so that elements span 0..8
    # encoder tiles and both decoder workgroup shapes), random type / probBits / checksum,
    # compressible and incompressible elements mixed; archives byte-identical to the oracle,
    # decode bit-exact, reported sizes exact
    rng = np.random.default_rng(9000 + seed)
    for trial in range(5):
        ft = int(rng.choice([0, O.FLOAT16, O.BFLOAT16, O.FLOAT32]))
        P = int(rng.choice([9, 10, 11]))
        cksum = bool(rng.integers(0, 2))
        B = int(rng.integers(1, 40))
        ns = []
        for _ in range(B):
            kind = rng.integers(0, 10)
            if kind == 0:
                ns.append(0)
            elif kind < 4:
                ns.append(int(rng.integers(1, 5000)))
            else:
                ns.append(int(rng.integers(1, 60)) * 4096 + int(rng.integers(0, 2)) * int(rng.integers(0, 4096)))
        if ft == 0:
            xs = []
            for n in ns:
                lam = float(rng.choice([2.0, 30.0, 300.0]))
                x = refgen.generate_symbols(max(n, 1), lam)[:n] if rng.integers(0, 4) else rng.integers(0, 256, n, dtype=np.uint8)
                xs.append(np.ascontiguousarray(x, np.uint8))
            got = gpu_ans_encode(dg, [x if x.size else np.zeros(0, np.uint8) for x in xs], P, cksum)
            for i, x in enumerate(xs):
                want = O.ans_encode(x, P, use_checksum=cksum)
                assert got[i].size == want.size and (got[i] == want).all(), (seed, trial, "raw", i, x.size)
            outs, status, osz = gpu_ans_decode(dg, got, [x.size for x in xs], P, cksum)
            assert status.all() and osz.tolist() == [x.size for x in xs]
            for x, o in zip(xs, outs):
                assert (o == x).all()
        else:
            dt = np.uint32 if ft == O.FLOAT32 else np.uint16
            hi = 1 << (32 if ft == O.FLOAT32 else 16)
            ws = []
            for n in ns:
                w = refgen.generate_floats(ft, max(n, 1))[:n] if rng.integers(0, 4) else rng.integers(0, hi, n, dtype=np.uint64).astype(dt)
                ws.append(np.ascontiguousarray(w, dt))
            ts = [words_to_tensor(ft, w) for w in ws]
            comp, sizes, _ = dg.compress_data(True, ts, cksum, prob_bits=P)
            hs = sizes.cpu().numpy()
            hc = comp.cpu().numpy()
            arch = []
            for i, w in enumerate(ws):
                want = O.float_compress(ft, w, P, use_checksum=cksum)
                assert hs[i] == want.size and (hc[i, : hs[i]] == want).all(), (seed, trial, ft, P, i, w.size)
                arch.append(comp[i, : hs[i]].clone())
            outs = [torch.empty_like(t) for t in ts]
            status = torch.zeros((B,), dtype=torch.uint8, device=DEV)
            osz = torch.zeros((B,), dtype=torch.int32, device=DEV)
            dg.decompress_data(True, arch, outs, cksum, None, status, osz, prob_bits=P)
            assert status.cpu().numpy().all() and osz.cpu().tolist() == ns
            for w, o in zip(ws, outs):
                assert (tensor_to_words(ft, o) == w).all()


def test_maximum_batch_size(dg):
    # 65535 elements in one call (the API's maximum; grid.y limit), a few hundred bytes each
    B = 65535
    rng = np.random.default_rng(123)
    base = rng.integers(0, 48, 1 << 20, dtype=np.uint8)
    sizes = rng.integers(1, 400, B)
    offs = rng.integers(0, (1 << 20) - 400, B)
    buf = torch.from_numpy(base).to(DEV)
    ts = [buf[int(o) : int(o) + int(n)] for o, n in zip(offs, sizes)]
    # inputs must be 4-byte aligned for the raw codec: use aligned offsets
    offs = (offs // 4) * 4
    ts = [buf[int(o) : int(o) + int(n)] for o, n in zip(offs, sizes)]
    comp, csz, _ = dg.compress_data(False, ts, False, prob_bits=10)
    hs = csz.cpu().numpy()
    hc = comp.cpu().numpy()
    for i in list(range(0, B, 997)) + [B - 1]:
        x = base[int(offs[i]) : int(offs[i]) + int(sizes[i])]
        want = O.ans_encode(x, 10)
        assert hs[i] == want.size and (hc[i, : hs[i]] == want).all(), i
    outs = [torch.empty((int(n),), dtype=torch.uint8, device=DEV) for n in sizes]
    status = torch.zeros((B,), dtype=torch.uint8, device=DEV)
    dg.decompress_data(False, [comp[i] for i in range(B)], outs, False, None, status, None, prob_bits=10)
    assert bool(status.all().item())
    for i in list(range(0, B, 1499)) + [B - 1]:
        assert torch.equal(outs[i], ts[i]), i


def test_parameter_cache_eviction_and_no_temp(dg):
    # 40 different pointer sets (the parameter cache holds 16) interleaved with repeats, no temp
    # memory passed at all (library-owned overflow slab): every call must still be exact
    rng = np.random.default_rng(5)
    sets = []
    for k in range(40):
        ws = [refgen.generate_floats(O.BFLOAT16, int(rng.integers(1, 3)) * 4096 * 8 + 17 * k + i) for i in range(3)]
        sets.append((ws, [words_to_tensor(O.BFLOAT16, w) for w in ws]))
    order = list(range(40)) + [0, 1, 2, 39, 38, 0, 17, 17, 3]
    for k in order:
        ws, ts = sets[k]
        comp, sizes, _ = dg.compress_data(True, ts, False, None, prob_bits=10)  # temp_mem=None
        hs = sizes.cpu().numpy()
        hc = comp.cpu().numpy()
        outs = [torch.empty_like(t) for t in ts]
        status = torch.zeros((len(ts),), dtype=torch.uint8, device=DEV)
        dg.decompress_data(True, [comp[i, : hs[i]].clone() for i in range(len(ts))], outs, False, None, status, None,
                           prob_bits=10)
        assert status.cpu().numpy().all()
        for i, w in enumerate(ws):
            want = O.float_compress(O.BFLOAT16, w, 10)
            assert hs[i] == want.size and (hc[i, : hs[i]] == want).all(), (k, i)
            assert (tensor_to_words(O.BFLOAT16, outs[i]) == w).all()


def test_concurrent_streams(dg):
    # Two streams compress and decompress different batches at the same time, repeatedly, without
    # host synchronisation in between: per-call state (tickets, claim words, descriptors, spill
    # slots) lives in each call's temp memory, the arrival counters are per stream, the parameter
    # cache is shared.  Each encoder grid is sized for the whole chip, so the two kernels also
    # run with only part of their workgroups resident.
    rng = np.random.default_rng(77)
    batches = []
    for k in range(2):
        ws = [refgen.generate_floats(O.BFLOAT16, 4096 * 8 * (12 + 5 * k) + 100 * i) for i in range(24)]
        batches.append(ws)
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    results = [[], []]
    for rep in range(4):
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                ts = [words_to_tensor(O.BFLOAT16, w) for w in batches[k]]
                comp, sizes, _ = dg.compress_data(True, ts, False, prob_bits=10)
                outs = [torch.empty_like(t) for t in ts]
                status = torch.zeros((len(ts),), dtype=torch.uint8, device=DEV)
                arch = [comp[i] for i in range(len(ts))]
                dg.decompress_data(True, arch, outs, False, None, status, None, prob_bits=10)
                results[k].append((comp, sizes, outs, status))
    torch.cuda.synchronize()
    for k in range(2):
        want = [O.float_compress(O.BFLOAT16, w, 10) for w in batches[k]]
        for comp, sizes, outs, status in results[k]:
            hs = sizes.cpu().numpy()
            hc = comp.cpu().numpy()
            assert status.cpu().numpy().all()
            for i, w in enumerate(batches[k]):
                assert hs[i] == want[i].size and (hc[i, : hs[i]] == want[i]).all(), (k, i)
                assert (tensor_to_words(O.BFLOAT16, outs[i]) == w).all()


@pytest.mark.parametrize("modulo", [2, 3, 7])
def test_encoder_with_absent_workgroups(dg, modulo):
    # The encoder's static tile map must not depend on the whole grid being resident: with a
    # part of the workgroups starting late (test hook), the running ones take over the tiles
    # they depend on, the late ones pick up what is left -- and the archives stay
    # byte-identical.  Many tiles per element so that take-over chains form.
    L = dg.lib()
    rng = np.random.default_rng(50 + modulo)
    ns = [4096 * 8 * 40 + 77, 4096 * 8 * 23, 4096 * 8 * 64, 5, 4096 * 8 * 31 + 4095, 4096 * 8 * 9]
    ws = [refgen.generate_floats(O.BFLOAT16, n) for n in ns]
    xs = [refgen.generate_symbols(n, 20.0) if hasattr(refgen, "generate_symbols")
          else rng.integers(0, 64, n, dtype=np.uint8) for n in (4096 * 8 * 50 + 1, 4096 * 8 * 17, 300)]
    L.dgpu_debug_set_absent_workgroups(modulo)
    try:
        ts = [words_to_tensor(O.BFLOAT16, w) for w in ws]
        comp, sizes, _ = dg.compress_data(True, ts, False, prob_bits=10)
        hs = sizes.cpu().numpy()
        hc = comp.cpu().numpy()
        got_raw = gpu_ans_encode(dg, xs, 10)
    finally:
        L.dgpu_debug_set_absent_workgroups(0)
    for i, w in enumerate(ws):
        want = O.float_compress(O.BFLOAT16, w, 10)
        assert hs[i] == want.size and (hc[i, : hs[i]] == want).all(), (modulo, i)
    for x, g in zip(xs, got_raw):
        want = O.ans_encode(x, 10)
        assert g.size == want.size and (g == want).all()


@pytest.mark.parametrize("ft", [O.FLOAT16, O.BFLOAT16, O.FLOAT32])
def test_float_unaligned_io(dg, ft):
    # inputs / outputs that are only float-word aligned
    n = 20000
    w = refgen.generate_floats(ft, n + 3)
    base = words_to_tensor(ft, w)
    t = base[3:]
    assert t.data_ptr() % 16 != 0
    comp, sizes, _ = dg.compress_data(True, [t])
    want = O.float_compress(ft, w[3:], 10)
    assert int(sizes[0]) == want.size and (comp[0, : want.size].cpu().numpy() == want).all()
    outbuf = torch.zeros_like(base)
    out = outbuf[1 : 1 + n]
    dg.decompress_data(True, [comp[0, : want.size].clone()], [out])
    assert (tensor_to_words(ft, out) == w[3:]).all()


@pytest.mark.parametrize("small", [True, False])
@pytest.mark.parametrize("ft", [0, O.FLOAT16, O.BFLOAT16, O.FLOAT32])
def test_elements_at_every_word_alignment(dg, ft, small):
    # elements of a split tensor / rows of a matrix start anywhere: the vector paths of histogram, encoder and decoder
    # take every word-aligned address (raw bytes: every 4-byte aligned one, the C ABI's requirement).  Element i starts
    # i words (raw: 4 i bytes) past a 16-byte boundary, its output (i + 3) words past one; sizes put the element's end on
    # both sides of every 16-byte part of a lane's slice (the encoder loads the part that straddles the end as the 16
    # bytes that END there), below 16 bytes (scalar path), on whole blocks and tiles; `small`: every element fits one
    # block (the kernels that take a pair of elements per wavefront).
    rng = np.random.default_rng(515 + 7 * ft + int(small))
    wb = 1 if ft == 0 else (4 if ft == O.FLOAT32 else 2)
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[wb]
    step = 4 if ft == 0 else 1  # words between successive misalignments
    per16 = 16 // wb
    ns = [1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31, 32, 33, 100, 511, 513, 2047, 4000, 4001, 4002, 4003, 4005, 4009, 4095, 4096]
    ns += [int(n) for n in rng.integers(1, 4097, 12)]
    if not small:
        ns += [4097, 8191, 8192, 8193, 4096 * 3 + 5, 4096 * 8, 4096 * 8 + 123, 4096 * 17 + 1, 4096 * 40 - 3]
    ws = []
    for i, n in enumerate(ns):
        if ft == 0:
            w = rng.integers(0, 256, n, dtype=np.uint8) if i % 3 == 0 else refgen.generate_symbols(n, 20.0 + 30 * i)[:n]
        else:
            w = rng.integers(0, 1 << (8 * wb), n, dtype=np.uint64).astype(dt) if i % 3 == 0 else refgen.generate_floats(ft, n)[:n]
        ws.append(np.ascontiguousarray(w, dt))
    room = sum(roundup(n, per16) + 2 * per16 for n in ns) + 64
    host_in = np.zeros(room, dt)
    offs_in, offs_out, pos = [], [], 0
    for i, n in enumerate(ns):
        mis = (i * step) % per16
        offs_in.append(pos + mis)
        offs_out.append(pos + (mis + 3 * step) % per16)
        host_in[pos + mis : pos + mis + n] = ws[i]
        pos += roundup(n, per16) + 2 * per16
    dev_in = torch.from_numpy(host_in).to(DEV) if ft == 0 else words_to_tensor(ft, host_in)
    dev_out = torch.zeros_like(dev_in)
    assert dev_in.data_ptr() % 16 == 0 and dev_out.data_ptr() % 16 == 0
    ins = [dev_in[o : o + n] for o, n in zip(offs_in, ns)]
    outs = [dev_out[o : o + n] for o, n in zip(offs_out, ns)]
    comp, sizes, _ = dg.compress_data(ft != 0, ins, True)
    hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
    arch = []
    for i, w in enumerate(ws):
        want = O.float_compress(ft, w, 10, use_checksum=True) if ft else O.ans_encode(w, 10, use_checksum=True)
        assert hs[i] == want.size and not (hc[i, : hs[i]] != want).any(), (ft, w.size, offs_in[i] % per16)
        arch.append(comp[i, : hs[i]].clone())
    status = torch.zeros((len(ns),), dtype=torch.uint8, device=DEV)
    osz = torch.zeros((len(ns),), dtype=torch.int32, device=DEV)
    dg.decompress_data(ft != 0, arch, outs, True, None, status, osz)
    assert status.cpu().numpy().all() and osz.cpu().tolist() == ns
    got = dev_out.cpu().numpy() if ft == 0 else tensor_to_words(ft, dev_out)
    want_out = np.zeros(room, dt)
    for o, w in zip(offs_out, ws):
        want_out[o : o + w.size] = w
    assert (got == want_out).all()  # every element restored, and not a word outside the elements written


def test_float_simple_and_empty(dg):
    # float_test.py:78-108
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        ts = [torch.randn((10000,), device=DEV).to(dt), torch.randn((100000,), device=DEV).to(dt)]
        comp = dg.compress_data_simple(True, ts, True)
        for c, t in zip(comp, ts):
            assert c.numel() < t.numel() * t.element_size()
        dec = dg.decompress_data_simple(True, comp, True)
        for a, b in zip(ts, dec):
            assert a.dtype == b.dtype and torch.equal(a, b)
        e = [torch.empty((0,), dtype=dt, device=DEV)]
        ce = dg.compress_data_simple(True, e, True)
        assert ce[0].numel() == 16 + 544
        de = dg.decompress_data_simple(True, ce, True)
        assert de[0].numel() == 0 and de[0].dtype == dt


def test_float_split_size(dg):
    # float_test.py:110-178
    import random

    random.seed(5)
    for dt, ft in ((torch.bfloat16, O.BFLOAT16), (torch.float16, O.FLOAT16), (torch.float32, O.FLOAT32)):
        for align16 in (True, False):
            sizes = []
            for _ in range(random.randrange(1, 12)):
                s = random.randrange(1, 10000)
                if align16:
                    s = (s + 15) // 16 * 16
                sizes.append(s)
            t = torch.randn((sum(sizes),), device=DEV).to(dt)
            sizes_t = torch.IntTensor(sizes)
            comp_ts, _, _ = dg.compress_data_split_size(True, t, sizes_t, True)
            host = tensor_to_words(ft, t)
            off = 0
            for c, s in zip(comp_ts, sizes):
                want = O.float_compress(ft, host[off : off + s], 10, use_checksum=True)
                assert c.numel() == want.size and (c.cpu().numpy() == want).all()
                off += s
            out = torch.empty_like(t)
            dg.decompress_data_split_size(True, [c.clone() for c in comp_ts], out, sizes_t, True)
            assert torch.equal(t.view(torch.int16 if ft != O.FLOAT32 else torch.int32),
                               out.view(torch.int16 if ft != O.FLOAT32 else torch.int32))


def test_float_capacity_and_checksum(dg):
    w = refgen.generate_floats(O.FLOAT16, 3000)
    t = words_to_tensor(O.FLOAT16, w)
    comp, sizes, _ = dg.compress_data(True, [t], True)
    arch = comp[0, : int(sizes[0])].clone()
    small = torch.zeros((2999,), dtype=torch.float16, device=DEV)
    status = torch.ones((1,), dtype=torch.uint8, device=DEV)
    osz = torch.zeros((1,), dtype=torch.int32, device=DEV)
    dg.decompress_data(True, [arch], [small], False, None, status, osz)
    assert status.item() == 0 and osz.item() == 3000
    bad = arch.clone()
    bad[12] ^= 0x3C  # stored float checksum
    out = torch.empty_like(t)
    with pytest.raises(RuntimeError, match="checksum mismatch"):
        dg.decompress_data(True, [bad], [out], True)


# ------------------------------------------------- BASELINE configs, full size
import os

_THREADS = os.cpu_count() or 8


def _check_all_rows(comp, sizes, want_rows, want_sizes):
    """Every row of the batch, byte for byte (FloatTest.cu:286-298 LargeBatch compares every element)."""
    hs = sizes.cpu().numpy().astype(np.int64)
    assert (hs == want_sizes.astype(np.int64)).all(), np.nonzero(hs != want_sizes)[0][:8]
    width = int(hs.max())
    got = comp[:, :width].cpu().numpy()
    mask = np.arange(width)[None, :] < hs[:, None]
    diff = (got != want_rows[:, :width]) & mask
    assert not diff.any(), f"rows {np.nonzero(diff.any(axis=1))[0][:8]} differ from the oracle"


def test_baseline_config2_zipf_bytes(dg):
    # 256 independent rows, default_rng(1234 + b)
    B, n = 256, 1 << 20
    xs = refgen.zipf_bytes(B, n)
    t = torch.from_numpy(xs).to(DEV)
    ts = list(t.unbind(0))
    comp, sizes, _ = dg.compress_data(False, ts)
    want, wsz = O.ans_encode_batch(xs, 10, threads=_THREADS)
    _check_all_rows(comp, sizes, want, wsz)
    ratio = sizes.cpu().numpy().sum() / xs.size
    assert abs(ratio - 0.695) < 0.02
    outs = [torch.empty((n,), dtype=torch.uint8, device=DEV) for _ in range(B)]
    dg.decompress_data(False, list(comp.unbind(0)), outs)
    assert torch.equal(torch.stack(outs), t)


def test_baseline_config3_bf16(dg):
    B, n = 256, 512 * 1024
    w = refgen.normal_bf16(B, n)
    t = torch.from_numpy(w.view(np.int16)).to(DEV).view(torch.bfloat16)
    ts = list(t.unbind(0))
    comp, sizes, _ = dg.compress_data(True, ts)
    want, wsz = O.float_compress_batch(O.BFLOAT16, w, 10, threads=_THREADS)
    _check_all_rows(comp, sizes, want, wsz)
    ratio = sizes.cpu().numpy().sum() / (w.size * 2)
    assert abs(ratio - 0.673) < 0.01
    outs = [torch.empty((n,), dtype=torch.bfloat16, device=DEV) for _ in range(B)]
    status = torch.zeros((B,), dtype=torch.uint8, device=DEV)
    dg.decompress_data(True, list(comp.unbind(0)), outs, False, None, status)
    assert status.cpu().numpy().all()
    assert torch.equal(torch.stack(outs).view(torch.int16), t.view(torch.int16))


def test_baseline_config4_sparse_fp16_p11(dg):
    B, n = 256, 512 * 1024
    w = refgen.sparse_fp16(B, n)
    t = torch.from_numpy(w.view(np.int16)).to(DEV).view(torch.float16)
    ts = list(t.unbind(0))
    comp, sizes, _ = dg.compress_data(True, ts, prob_bits=11)
    want, wsz = O.float_compress_batch(O.FLOAT16, w, 11, threads=_THREADS)
    _check_all_rows(comp, sizes, want, wsz)
    outs = [torch.empty((n,), dtype=torch.float16, device=DEV) for _ in range(B)]
    dg.decompress_data(True, list(comp.unbind(0)), outs, prob_bits=11)
    assert torch.equal(torch.stack(outs).view(torch.int16), t.view(torch.int16))


def test_baseline_config5_shard_of_one_rank(dg):
    # BASELINE config 5 = 2048 x 1 MiB bf16 over 8 ranks: rank g codes elements [256 g, 256 g + 256)
    # generated with seed 1234 + g.  One GPU here: rank 3's shard, all rows.
    from dietgpu_amd.distributed import shard_range

    assert shard_range(2048, 3, 8) == (768, 1024)
    B, n = 256, 512 * 1024
    w = refgen.normal_bf16(B, n, seed=1234 + 3)
    t = torch.from_numpy(w.view(np.int16)).to(DEV).view(torch.bfloat16)
    comp, sizes, _ = dg.compress_data(True, list(t.unbind(0)))
    want, wsz = O.float_compress_batch(O.BFLOAT16, w, 10, threads=_THREADS)
    _check_all_rows(comp, sizes, want, wsz)


# ------------------------------------------ multi-window look-back (> 64 tiles per element)
# The ordered compaction replaces BatchPrefixSum.cuh:33-194, whose own test goes to 512 x 512
# entries (BatchPrefixSumTest.cu:85-127).  One look-back step covers 64 predecessor tiles; elements
# of 65, 129 and >= 1025 tiles need 2, 3 and 17 steps.  Archives are compared with the oracle byte
# for byte (a round trip alone would not notice a gap or a wrong start offset), also with part of
# the workgroups starting late.
_TILE = 8 * 4096


@pytest.mark.parametrize("tiles,absent", [(65, 0), (129, 0), (129, 3), (1025, 0), (1027, 7)])
def test_lookback_windows_raw(dg, tiles, absent):
    rng = np.random.default_rng(tiles)
    n = tiles * _TILE - 1234  # last tile partial
    # skewed bytes whose statistics drift along the element: block sizes vary from tile to tile
    x = (rng.exponential(12.0 + 30.0 * np.linspace(0, 1, n)) % 256).astype(np.uint8)
    small = rng.integers(0, 50, 3 * _TILE + 5, dtype=np.uint8)
    L = dg.lib()
    L.dgpu_debug_set_absent_workgroups(absent)
    try:
        got = gpu_ans_encode(dg, [x, small], 10)
    finally:
        L.dgpu_debug_set_absent_workgroups(0)
    for data, g in zip((x, small), got):
        want = O.ans_encode(data, 10)
        assert g.size == want.size
        bad = np.nonzero(g != want)[0]
        assert bad.size == 0, f"first differing byte {bad[:4]} of {want.size}"
    outs, status, osz = gpu_ans_decode(dg, got, [n, small.size], 10)
    assert status.all() and (outs[0] == x).all() and (outs[1] == small).all()


@pytest.mark.parametrize("absent", [0, 2])
def test_lookback_windows_bf16_40m(dg, absent):
    n = 40 * 1000 * 1000 + 7  # 1221 tiles
    rng = np.random.default_rng(40)
    f = rng.standard_normal(n, dtype=np.float32) * np.exp(rng.standard_normal(n, dtype=np.float32))
    w = refgen.f32_to_bf16_rne(f)
    t = words_to_tensor(O.BFLOAT16, w)
    L = dg.lib()
    L.dgpu_debug_set_absent_workgroups(absent)
    try:
        comp, sizes, _ = dg.compress_data(True, [t])
    finally:
        L.dgpu_debug_set_absent_workgroups(0)
    want = O.float_compress(O.BFLOAT16, w, 10)
    assert int(sizes[0]) == want.size
    got = comp[0, : want.size].cpu().numpy()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"first differing byte {bad[:4]} of {want.size}"
    out = torch.empty_like(t)
    dg.decompress_data(True, [comp[0, : want.size]], [out])
    assert torch.equal(out.view(torch.int16), t.view(torch.int16))


@pytest.mark.parametrize("absent", [0, 3])
@pytest.mark.parametrize("ft", [O.BFLOAT16, O.FLOAT32])
def test_two_level_lookback_of_several_elements(dg, ft, absent):
    # Float elements of more than 64 tiles walk the look-back in two levels (groups of 64 tiles: an arrival word and a
    # descriptor per group, kernels_encode.h lookBackTwoLevel).  Three elements of 130 tiles each -- two whole groups and a
    # group of two, the last tile of a group not the last of its element and the other way round -- and, beside them in
    # the same rectangle, one of 64 tiles (one level) and one of 65 (a group of one); with late workgroups the group's
    # last tile can publish the inclusive prefix before the group's arrivals are complete.
    tile = 8 * 4096
    ns = [130 * tile, 130 * tile - 5000, 64 * tile, 130 * tile, 65 * tile - 17]
    rng = np.random.default_rng(640 + ft + absent)
    ws = []
    for i, n in enumerate(ns):
        if i == 3:  # incompressible: every block spills, aggregates near their maximum
            dt = np.uint32 if ft == O.FLOAT32 else np.uint16
            ws.append(rng.integers(0, 1 << (8 * dt().itemsize), n, dtype=np.uint64).astype(dt))
        else:
            ws.append(np.ascontiguousarray(np.roll(refgen.generate_floats(ft, n), 31 * i)))
    ts = [words_to_tensor(ft, w) for w in ws]
    L = dg.lib()
    L.dgpu_debug_set_work_lists(0)  # (the rectangle: the two levels exist there)
    L.dgpu_debug_set_absent_workgroups(absent)
    try:
        comp, sizes, _ = dg.compress_data(True, ts, True)
    finally:
        L.dgpu_debug_set_absent_workgroups(0)
        L.dgpu_debug_set_work_lists(-1)
    hs = sizes.cpu().numpy()
    rows = []
    for i, w in enumerate(ws):
        want = O.float_compress(ft, w, 10, use_checksum=True)
        assert hs[i] == want.size, (i, hs[i], want.size)
        got = comp[i, : want.size].cpu().numpy()
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (i, bad[:4], want.size)
        rows.append(comp[i, : want.size].clone())
    outs = [torch.empty_like(t) for t in ts]
    status = torch.zeros((len(ts),), dtype=torch.uint8, device=DEV)
    dg.decompress_data(True, rows, outs, True, None, status)
    assert status.cpu().numpy().all() and all((tensor_to_words(ft, o) == w).all() for o, w in zip(outs, ws))


# ------------------------------------------------------------ malformed archives
def _decode_status(dg, as_float, arch_np, out_like, prob_bits=10):
    arch = torch.from_numpy(arch_np.copy()).to(DEV)
    out = torch.full_like(out_like, 0)
    status = torch.full((1,), 7, dtype=torch.uint8, device=DEV)
    osz = torch.zeros((1,), dtype=torch.int32, device=DEV)
    dg.decompress_data(as_float, [arch], [out], False, None, status, osz, prob_bits=prob_bits)
    torch.cuda.synchronize()
    return int(status.item()), out


def test_decoder_rejects_malformed_ans_archives(dg):
    # The reference guards the format with device asserts (GpuANSDecode.cuh:323,448,
    # GpuANSUtils.cuh:109-112), compiled out in release builds.  Here a corrupt header, block
    # table or pdf table is reported through outSuccess and nothing is decoded from it.
    x = refgen.generate_symbols(5 * 4096 + 100, 20.0)
    good = O.ans_encode(x, 10)
    ref = torch.from_numpy(x).to(DEV)
    st, out = _decode_status(dg, False, good, ref)
    assert st == 1 and torch.equal(out, ref)
    nb = 6
    bw0 = 32 + 512 + 128 * nb  # blockWords table

    def u32(a, off):
        return a[off : off + 4].view(np.uint32)

    cases = []
    for name, off, val in [
        ("magic", 0, 0xd00d0002), ("numBlocks+1", 4, nb + 1), ("numBlocks-1", 4, nb - 1), ("numBlocks huge", 4, 1 << 30),
        ("total-1", 8, x.size - 1), ("probBits", 16, 11),
        ("totalCompressedWords small", 12, 8),
        ("block0 uncompressed size", bw0, (4095 << 16) | int(u32(good, bw0)[0] & 0xffff)),
        ("block2 start past the end", bw0 + 2 * 8 + 4, int(u32(good, 12)[0])),
        ("block3 start unaligned", bw0 + 3 * 8 + 4, int(u32(good, bw0 + 3 * 8 + 4)[0]) + 3),
        ("last block size", bw0 + 5 * 8, (101 << 16) | int(u32(good, bw0 + 5 * 8)[0] & 0xffff)),
    ]:
        bad = good.copy()
        u32(bad, off)[0] = val
        cases.append((name, bad))
    bad = good.copy()
    bad[32:34].view(np.uint16)[0] += 1  # pdf no longer sums to 2^probBits
    cases.append(("pdf sum", bad))
    for name, bad in cases:
        st, out = _decode_status(dg, False, bad, ref)
        assert st == 0, name
    # a corrupt element does not disturb its neighbours in the batch
    archs = [torch.from_numpy(cases[3][1].copy()).to(DEV), torch.from_numpy(good.copy()).to(DEV)]
    outs = [torch.zeros_like(ref), torch.zeros_like(ref)]
    status = torch.zeros((2,), dtype=torch.uint8, device=DEV)
    dg.decompress_data(False, archs, outs, False, None, status, None)
    assert status.cpu().tolist() == [0, 1] and torch.equal(outs[1], ref)


def test_decoder_rejects_malformed_float_archives(dg):
    w = refgen.generate_floats(O.BFLOAT16, 3 * 4096 + 77)
    good = O.float_compress(O.BFLOAT16, w, 10)
    ref = words_to_tensor(O.BFLOAT16, w)
    st, out = _decode_status(dg, True, good, ref)
    assert st == 1 and torch.equal(out.view(torch.int16), ref.view(torch.int16))
    for name, off, val in [("float magic", 0, 0xf00f0003), ("float size larger", 4, w.size + 16),
                           ("float size huge", 4, 0xfffffff0), ("float type", 8, O.FLOAT16),
                           ("float size smaller", 4, w.size - 16)]:
        bad = good.copy()
        bad[off : off + 4].view(np.uint32)[0] = val
        st, _ = _decode_status(dg, True, bad, ref)
        assert st == 0, name


def test_decoder_rejects_truncated_archives(dg):
    # the tensor API knows every input tensor's size (the *_bounded C entry points): an archive cut short --
    # at any length -- is reported through the status instead of being read past its tensor
    x = refgen.generate_symbols(6 * 4096 + 5, 20.0)
    good = O.ans_encode(x, 10)
    ref = torch.from_numpy(x).to(DEV)
    for cut in (0, 8, 31, 32, 100, 544, 1300, good.size - 16, good.size - 1):
        st, _ = _decode_status(dg, False, good[:cut] if cut else good[:1], ref)
        assert st == 0, cut
    w = refgen.generate_floats(O.FLOAT32, 2 * 4096 + 9)
    fgood = O.float_compress(O.FLOAT32, w, 10)
    fref = words_to_tensor(O.FLOAT32, w)
    for cut in (4, 15, 16, 40, 16 + 8192 * 3, fgood.size - 2):
        st, _ = _decode_status(dg, True, fgood[:cut], fref)
        assert st == 0, cut
    st, out = _decode_status(dg, True, np.concatenate([fgood, np.zeros(64, np.uint8)]), fref)  # slack after the archive is fine
    assert st == 1 and torch.equal(out.view(torch.int32), fref.view(torch.int32))


@pytest.mark.parametrize("seed", range(4))
def test_decoder_fuzzed_headers_do_not_fault(dg, seed):
    # random bit flips anywhere in the header / pdf / state / block tables: the decoder either
    # decodes something (flips in states or pdfs that keep the invariants) or reports failure, and
    # the process survives; capacity-bounded output is the only memory written
    rng = np.random.default_rng(1000 + seed)
    x = refgen.generate_symbols(9 * 4096 + 11, 50.0)
    good = O.ans_encode(x, 10)
    overhead = 32 + 512 + 128 * 10 + 8 * 10
    ref = torch.from_numpy(x).to(DEV)
    guard = torch.full((x.size + 4096,), 0xAB, dtype=torch.uint8, device=DEV)
    for _ in range(60):
        bad = good.copy()
        for _k in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, overhead))
            bad[pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
        arch = torch.from_numpy(bad).to(DEV)
        out = guard[: x.size]
        status = torch.zeros((1,), dtype=torch.uint8, device=DEV)
        dg.decompress_data(False, [arch], [out], False, None, status, None)
        torch.cuda.synchronize()
        assert (guard[x.size :] == 0xAB).all()  # nothing written past the capacity


# ------------------------------------------------- library-owned state (overflow slab, streams)
def test_overflow_slab_growth_inside_one_call(dg):
    # No temp memory, checksum on: a small call warms the slab, the next one is large enough to
    # outgrow it twice inside ONE call.  Allocations handed out before the growth (checksums,
    # tables) must stay valid until the kernels that read them have run.
    L = dg.lib()
    L.dgpu_release_all_stream_state()
    small = [to_dev_bytes(refgen.generate_symbols(3000 + i, 20.0)) for i in range(2)]
    dg.compress_data(False, small, True, None)
    rng = np.random.default_rng(9)
    for B, n in ((3000, 700), (20000, 300), (64, 3 << 20)):
        base = rng.integers(0, 60, n * 4 + 4096, dtype=np.uint8)
        buf = torch.from_numpy(base).to(DEV)
        offs = (rng.integers(0, n * 3, B) // 4) * 4
        lens = rng.integers(max(n // 2, 1), n, B)
        ts = [buf[int(o) : int(o) + int(m)] for o, m in zip(offs, lens)]
        comp, sizes, _ = dg.compress_data(False, ts, True, None)
        hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
        step = max(B // 40, 1)
        for i in list(range(0, B, step)) + [B - 1]:
            want = O.ans_encode(base[int(offs[i]) : int(offs[i]) + int(lens[i])], 10, use_checksum=True)
            assert hs[i] == want.size and (hc[i, : hs[i]] == want).all(), (B, i)
        outs = [torch.empty((int(m),), dtype=torch.uint8, device=DEV) for m in lens]
        dg.decompress_data(False, [comp[i] for i in range(B)], outs, True, None)
        for i in range(0, B, step):
            assert torch.equal(outs[i], ts[i])


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_whole_block_elements_in_every_tile_variant(dg, prob_bits):
    # elements of exactly 1, 2, 3, 4, 5, 9 and 17 full blocks: waves with ONE full block take the straight-line row
    # paths with an idle / shadow upper half; batches whose largest element has <= 2 / <= 4 / more blocks use the
    # one-wavefront, 4-block and 8-block (16-block decode) tile variants
    rng = np.random.default_rng(prob_bits)
    for max_blocks in (1, 2, 3, 4, 5, 9, 17):
        counts = [k for k in (1, 2, 3, 4, 5, 9, 17) if k <= max_blocks] + [max_blocks, 1]
        xs = [(rng.exponential(6.0 + 9 * i, k * 4096) % 256).astype(np.uint8) for i, k in enumerate(counts)]
        got = gpu_ans_encode(dg, xs, prob_bits, True)
        for x, g in zip(xs, got):
            want = O.ans_encode(x, prob_bits, use_checksum=True)
            assert g.size == want.size and not (g != want).any(), (max_blocks, x.size)
        outs, status, _ = gpu_ans_decode(dg, got, [x.size for x in xs], prob_bits, True)
        assert status.all() and all((o == x).all() for o, x in zip(outs, xs))
        for ft in (O.FLOAT16, O.BFLOAT16, O.FLOAT32):
            ws = [refgen.generate_floats(ft, k * 4096) for k in counts]
            ts = [words_to_tensor(ft, w) for w in ws]
            comp, sizes, _ = dg.compress_data(True, ts, False, prob_bits=prob_bits)
            hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
            for i, w in enumerate(ws):
                want = O.float_compress(ft, w, prob_bits)
                assert hs[i] == want.size and not (hc[i, : hs[i]] != want).any(), (max_blocks, ft, w.size)
            outs = [torch.empty_like(t) for t in ts]
            dg.decompress_data(True, [comp[i, : hs[i]] for i in range(len(ts))], outs, False, prob_bits=prob_bits)
            assert all((tensor_to_words(ft, o) == w).all() for o, w in zip(outs, ws))


@pytest.mark.parametrize("blocks", [1, 2, 4])
@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_ragged_batches_of_small_elements(dg, prob_bits, blocks):
    # batches whose LARGEST element has 1 / 2 / 4 blocks use the single-block (one stage / ring / store buffer per
    # wavefront, the upper half shadowing or idle), one-wavefront and 4-block tile variants: empty, partial and
    # whole blocks, compressible and incompressible (the float encoder spills the latter), with checksums
    rng = np.random.default_rng(4400 + prob_bits + 10 * blocks)
    ns = [0, 1, 31, 32, 33, 100, 2048, 4095, 4096, 4096, 3000, 4096, 0, 4064]
    ns += [blocks * 4096, blocks * 4096 - 5, (blocks - 1) * 4096 + 1, blocks * 4096]
    ns += [4096] * 6  # pairs of whole blocks with an incompressible member (the straight-line path, spilling)
    xs = []
    for i, n in enumerate(ns):
        x = rng.integers(0, 256, n, dtype=np.uint8) if i % 3 == 0 else refgen.generate_symbols(max(n, 1), 20.0 + 40 * i)[:n]
        xs.append(np.ascontiguousarray(x, np.uint8))
    got = gpu_ans_encode(dg, xs, prob_bits, True)
    for x, g in zip(xs, got):
        want = O.ans_encode(x, prob_bits, use_checksum=True)
        assert g.size == want.size and not (g != want).any(), ("raw", x.size)
    outs, status, osz = gpu_ans_decode(dg, got, [x.size for x in xs], prob_bits, True)
    assert status.all() and osz.tolist() == ns and all((o == x).all() for o, x in zip(outs, xs))
    for ft in (O.FLOAT16, O.BFLOAT16, O.FLOAT32):
        dt = np.uint32 if ft == O.FLOAT32 else np.uint16
        hi = 1 << (32 if ft == O.FLOAT32 else 16)
        ws = []
        for i, n in enumerate(ns):
            w = rng.integers(0, hi, n, dtype=np.uint64).astype(dt) if i % 3 == 1 else refgen.generate_floats(ft, max(n, 1))[:n]
            ws.append(np.ascontiguousarray(w, dt))
        ts = [words_to_tensor(ft, w) for w in ws]
        comp, sizes, _ = dg.compress_data(True, ts, True, prob_bits=prob_bits)
        hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
        arch = []
        for i, w in enumerate(ws):
            want = O.float_compress(ft, w, prob_bits, use_checksum=True)
            assert hs[i] == want.size and not (hc[i, : hs[i]] != want).any(), (ft, w.size)
            arch.append(comp[i, : hs[i]].clone())
        outs = [torch.empty_like(t) for t in ts]
        status = torch.zeros((len(ns),), dtype=torch.uint8, device=DEV)
        osz = torch.zeros((len(ns),), dtype=torch.int32, device=DEV)
        dg.decompress_data(True, arch, outs, True, None, status, osz, prob_bits=prob_bits)
        assert status.cpu().numpy().all() and osz.cpu().tolist() == ns
        assert all((tensor_to_words(ft, o) == w).all() for o, w in zip(outs, ws))


@pytest.mark.parametrize("lead", [0, 1, 3, 8])
def test_partial_last_blocks_at_every_row_count(dg, lead):
    # elements of `lead` whole blocks + a last block of 1 .. 4095 symbols on both sides of every 8-row group boundary:
    # both coders run the whole groups of such a block on the straight-line path and the rows above them predicated
    # (encodeRows / decodeBlock, kTail) -- next to a whole block in the same wavefront, alone in it, or (lead 0) next
    # to another element's block of a different length.  Compressible blocks are staged whole in LDS, incompressible
    # ones go through the word ring (decoder) and the spill slots (float encoder).
    rng = np.random.default_rng(9100 + lead)
    lasts = [1, 31, 32, 33, 255, 256, 257, 288, 1023, 1024, 2049, 3585, 3840, 3841, 4064, 4065, 4095]
    ns = [lead * 4096 + n for n in lasts] + [lead * 4096 + int(n) for n in rng.integers(1, 4096, 15)]
    xs = []
    for i, n in enumerate(ns):
        x = rng.integers(0, 256, n, dtype=np.uint8) if i % 2 == 0 else refgen.generate_symbols(n, 20.0 + 30 * i)[:n]
        xs.append(np.ascontiguousarray(x, np.uint8))
    got = gpu_ans_encode(dg, xs, 10, True)
    for x, g in zip(xs, got):
        want = O.ans_encode(x, 10, use_checksum=True)
        assert g.size == want.size and not (g != want).any(), ("raw", x.size)
    outs, status, osz = gpu_ans_decode(dg, got, ns, 10, True)
    assert status.all() and osz.tolist() == ns and all((o == x).all() for o, x in zip(outs, xs))
    for ft in (O.BFLOAT16, O.FLOAT32):
        dt = np.uint32 if ft == O.FLOAT32 else np.uint16
        hi = 1 << (32 if ft == O.FLOAT32 else 16)
        ws = []
        for i, n in enumerate(ns):
            w = rng.integers(0, hi, n, dtype=np.uint64).astype(dt) if i % 2 == 1 else refgen.generate_floats(ft, n)[:n]
            ws.append(np.ascontiguousarray(w, dt))
        ts = [words_to_tensor(ft, w) for w in ws]
        comp, sizes, _ = dg.compress_data(True, ts, True, prob_bits=10)
        hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
        arch = []
        for i, w in enumerate(ws):
            want = O.float_compress(ft, w, 10, use_checksum=True)
            assert hs[i] == want.size and not (hc[i, : hs[i]] != want).any(), (ft, w.size)
            arch.append(comp[i, : hs[i]].clone())
        outs = [torch.empty_like(t) for t in ts]
        status = torch.zeros((len(ns),), dtype=torch.uint8, device=DEV)
        osz = torch.zeros((len(ns),), dtype=torch.int32, device=DEV)
        dg.decompress_data(True, arch, outs, True, None, status, osz, prob_bits=10)
        assert status.cpu().numpy().all() and osz.cpu().tolist() == ns
        assert all((tensor_to_words(ft, o) == w).all() for o, w in zip(outs, ws))


def _single_block_count_vectors(rng, trials):
    """Byte rows of <= 4096 symbols whose histograms drive every branch of the normalisation: a few symbols, all 256,
    singletons next to one giant (the deficit branch: every count-1 symbol is lifted to probability 1), powers of two,
    heavy tails; plus an empty row."""
    rows = []
    for t in range(trials):
        k = int(rng.integers(1, 257))
        syms = rng.choice(256, k, replace=False)
        mode = t % 6
        if mode == 0:
            c = rng.integers(1, 33, k)
        elif mode == 1:
            c = np.ones(k, np.int64)
            c[0] = 4096 - (k - 1)
        elif mode == 2:
            c = (rng.pareto(0.7, k) * 3 + 1).astype(np.int64)
        elif mode == 3:
            c = 1 << rng.integers(0, 8, k)
        elif mode == 4:
            c = np.ones(k, np.int64)
        else:
            c = rng.integers(1, 4, k)
            c[: max(1, k // 16)] = rng.integers(64, 512, max(1, k // 16))
        c = np.asarray(c, np.int64)
        while c.sum() > 4096:
            c = np.maximum(c // 2, 1) if c.max() > 1 else c[:-1]
        x = np.repeat(syms.astype(np.uint8), c)
        rng.shuffle(x)
        rows.append(np.ascontiguousarray(x))
    rows.append(np.zeros(0, np.uint8))
    return rows


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_statistics_of_single_block_elements(dg, prob_bits):
    # batches of single-block elements count and normalise with ONE WAVEFRONT per element (k_stats_single) instead of
    # a workgroup: whole archives (pdf table, coded words, checksum) against the oracle for histograms that take the
    # surplus and the deficit branch, as raw bytes and as the compressed byte of the three float types
    rng = np.random.default_rng(7700 + prob_bits)
    rows = _single_block_count_vectors(rng, 96)
    got = gpu_ans_encode(dg, rows, prob_bits, True)
    for x, g in zip(rows, got):
        want = O.ans_encode(x, prob_bits, use_checksum=True)
        assert g.size == want.size and not (g != want).any(), ("raw", x.size, np.unique(x).size)
    outs, status, osz = gpu_ans_decode(dg, got, [x.size for x in rows], prob_bits, True)
    assert status.all() and osz.tolist() == [x.size for x in rows] and all((o == x).all() for o, x in zip(outs, rows))
    for ft in (O.FLOAT16, O.BFLOAT16, O.FLOAT32):
        ws = []
        for x in rows:
            e = x.astype(np.uint32)
            r = rng.integers(0, 1 << 32, x.size, dtype=np.uint64).astype(np.uint32)
            if ft == O.FLOAT16:
                w = ((e << 8) | (r & 0xff)).astype(np.uint16)
            elif ft == O.BFLOAT16:
                w = ((e << 7) | (r & 0x7f) | ((r >> 8 & 1) << 15)).astype(np.uint16)
            else:
                w = ((e << 23) | (r & 0x7fffff) | ((r >> 24 & 1) << 31)).astype(np.uint32)
            ws.append(np.ascontiguousarray(w))
        ts = [words_to_tensor(ft, w) for w in ws]
        comp, sizes, _ = dg.compress_data(True, ts, True, prob_bits=prob_bits)
        hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
        for i, w in enumerate(ws):
            want = O.float_compress(ft, w, prob_bits, use_checksum=True)
            assert hs[i] == want.size and not (hc[i, : hs[i]] != want).any(), (ft, w.size)


@pytest.mark.parametrize("prob_bits", [10, 11])
def test_statistics_of_single_block_elements_at_every_word_alignment(dg, prob_bits):
    # split-size batches put single-block byte elements at 4-byte (not 16-byte) aligned addresses: the wavefront's
    # head / vector / tail split of the element (k_stats_single) at all four alignments, ragged sizes
    rng = np.random.default_rng(7800 + prob_bits)
    sizes = [4, 1004, 4092, 36, 2052, 4096, 12, 3000, 8, 4088, 20, 100, 4096, 1]
    xs = [np.ascontiguousarray(refgen.generate_symbols(max(n, 1), 15.0 + 30 * i)[:n], np.uint8) for i, n in enumerate(sizes)]
    t = to_dev_bytes(np.concatenate(xs))
    sizes_t = torch.tensor(sizes, dtype=torch.int32)
    comp_ts, _, _ = dg.compress_data_split_size(False, t, sizes_t, True, prob_bits=prob_bits)
    for x, c in zip(xs, comp_ts):
        want = O.ans_encode(x, prob_bits, use_checksum=True)
        g = c.cpu().numpy()
        assert g.size == want.size and not (g != want).any(), x.size


def test_two_host_threads_on_one_stream(dg):
    # ctypes releases the GIL during a call: two threads enqueue on the SAME stream with no temp memory, i.e. both
    # carve the stream's overflow slab.  Calls serialise on the per-stream lock; every archive must be exact.
    import threading

    rng = np.random.default_rng(21)
    jobs = []
    for k in range(2):
        ws = [refgen.generate_floats(O.BFLOAT16, 4096 * (5 + 3 * k) + 100 * i + k) for i in range(12)]
        jobs.append((ws, [words_to_tensor(O.BFLOAT16, w) for w in ws], [O.float_compress(O.BFLOAT16, w, 10) for w in ws]))
    errors = []
    stream = torch.cuda.current_stream()

    def work(k):
        ws, ts, want = jobs[k]
        try:
            with torch.cuda.stream(stream):
                for rep in range(25):
                    comp, sizes, _ = dg.compress_data(True, ts, False, None)
                    hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
                    for i in range(len(ws)):
                        if hs[i] != want[i].size or (hc[i, : hs[i]] != want[i]).any():
                            errors.append((k, rep, i))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:5]


def test_stream_state_is_bounded_and_releasable(dg):
    L = dg.lib()
    L.dgpu_release_all_stream_state()
    assert L.dgpu_debug_stream_state_count() == 0
    w = refgen.generate_floats(O.BFLOAT16, 4096 * 8 + 3)
    want = O.float_compress(O.BFLOAT16, w, 10)
    streams = [torch.cuda.Stream(device=DEV) for _ in range(40)]
    for st in streams:
        with torch.cuda.stream(st):
            t = words_to_tensor(O.BFLOAT16, w)
            comp, sizes, _ = dg.compress_data(True, [t], False, None)
        st.synchronize()
        n = int(sizes[0])
        assert n == want.size and (comp[0, :n].cpu().numpy() == want).all()
    assert 0 < L.dgpu_debug_stream_state_count() <= 32
    torch.cuda.synchronize()
    assert L.dgpu_release_all_stream_state() > 0
    assert L.dgpu_debug_stream_state_count() == 0
    # and the library keeps working afterwards
    t = words_to_tensor(O.BFLOAT16, w)
    comp, sizes, _ = dg.compress_data(True, [t], False, None)
    assert int(sizes[0]) == want.size


# ------------------------------------------------- against the reference itself (oracle/_ref)
def _ref():
    from oracle import ref as R

    if not R.available():
        pytest.skip("oracle/_ref not built (needs /root/reference: make -C oracle ref)")
    return R


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_hip_archives_equal_the_reference_and_interoperate(dg, prob_bits):
    # The reference's own sources, executed on the CPU SIMT emulation (oracle/ref_shim/): the HIP encoder's
    # archives equal its archives (indeterminate header bytes blanked), the REFERENCE decoder reads HIP
    # archives and the HIP decoder reads the reference's.
    from refmask import mask_ans, mask_float

    R = _ref()
    xs = [refgen.generate_symbols(n, lam) for n, lam in ((1, 1.0), (4096, 10.0), (4097, 100.0), (70001, 20.0), (0, 1.0),
                                                        (3 * 8 * 4096 + 11, 1000.0))]
    for ck in (False, True):
        got = gpu_ans_encode(dg, xs, prob_bits, ck)
        want = R.ans_encode_batch(xs, prob_bits, ck)
        for x, g, w in zip(xs, got, want):
            assert g.size == w.size and not (mask_ans(g) != mask_ans(w)).any(), (x.size, ck)
        outs, ok, osz, rc = R.ans_decode_batch(got, [x.size for x in xs], prob_bits, ck)  # reference reads HIP archives
        assert rc == 0 and ok.all() and all((o == x).all() for o, x in zip(outs, xs))
        outs, status, _ = gpu_ans_decode(dg, want, [x.size for x in xs], prob_bits, ck)   # HIP reads the reference's
        assert status.all() and all((o == x).all() for o, x in zip(outs, xs))
    for ft in (O.FLOAT16, O.BFLOAT16, O.FLOAT32):
        ws = [refgen.generate_floats(ft, n) for n in (1, 17, 4096, 8 * 4096 + 3, 50003)]
        ts = [words_to_tensor(ft, w) for w in ws]
        comp, sizes, _ = dg.compress_data(True, ts, True, prob_bits=prob_bits)
        hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
        got = [hc[i, : hs[i]].copy() for i in range(len(ws))]
        want = R.float_compress_batch(ft, ws, prob_bits, True)
        for w_, g, r in zip(ws, got, want):
            assert g.size == r.size and not (mask_float(g) != mask_float(r)).any(), (ft, w_.size)
        outs, ok, osz, rc = R.float_decompress_batch(ft, got, [w.size for w in ws], prob_bits, True)
        assert rc == 0 and ok.all() and all((o == w).all() for o, w in zip(outs, ws))
        outs = [torch.empty_like(t) for t in ts]
        dg.decompress_data(True, [to_dev_bytes(r) for r in want], outs, True, prob_bits=prob_bits)
        assert all((tensor_to_words(ft, o) == w).all() for o, w in zip(outs, ws))


def test_hip_statistics_equal_the_reference(dg):
    R = _ref()
    rng = np.random.default_rng(77)
    x = (rng.exponential(9.0, 100000) % 256).astype(np.uint8)
    t = to_dev_bytes(x)
    hist = torch.zeros((1, 256), dtype=torch.int32, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert dg.lib().dgpu_ans_histogram_batch_stride(1, C.c_void_p(t.data_ptr()), x.size, x.size, C.c_void_p(hist.data_ptr()), st) == 0
    assert (hist.cpu().numpy()[0].astype(np.uint32) == R.histogram(x)).all()
    counts = np.stack([np.bincount(x[: 1000 * (k + 1)], minlength=256) for k in range(40)]).astype(np.uint32)
    totals = counts.sum(axis=1).astype(np.uint32)
    for P in (9, 10, 11):
        got = gpu_normalize(dg, counts, totals, P).reshape(40, 256, 4)
        want = R.normalize_batch(counts, totals, P)
        live = want[:, :, 0] > 0
        assert (got[:, :, :2] == want[:, :, :2]).all() and (got[live] == want[live]).all()


def test_large_single_element(dg):
    # float_test.py:66-76 shape class: one big tensor (many tiles -> multi-window look-back)
    n = 40 * 1000 * 1000 + 7
    t = torch.randn((n,), device=DEV).to(torch.bfloat16)
    comp = dg.compress_data_simple(True, [t])
    dec = dg.decompress_data_simple(True, comp)
    assert torch.equal(dec[0].view(torch.int16), t.view(torch.int16))
    assert comp[0].numel() < n * 2


# ---------------------------------------------------------------------------
# The decoder's word ring on raw-byte elements of more than 8 blocks: every block-kind mix inside a wavefront pair,
# word counts on both sides of the ring's chunk boundaries, element tails, ragged batches, corrupt tables and
# states.  (Written for the round-3 four-blocks-per-wavefront experiment, tools/experiments/round3_tree/; they
# are the densest coverage the shipped k_ans_decode's ring / whole-block staging protocol has.)
def _ring_block(rng, kind):
    if kind == "z":   # one symbol only: pdf = 2^P, the block emits no words at all
        return np.full(4096, 7, np.uint8)
    if kind == "c":   # ~1.2 bits per symbol: ~310 words, within the first three chunks
        return rng.choice(np.array([7, 1, 2, 3], np.uint8), 4096, p=[0.7, 0.2, 0.07, 0.03])
    if kind == "m":   # ~5 bits per symbol (Zipf-like): ~1300 words, ring in steady state
        return (rng.zipf(1.3, 4096) % 256).astype(np.uint8)
    return rng.integers(0, 256, 4096, dtype=np.uint8)  # "r": incompressible, up to 32 words per row


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_decode_ring_block_mixes(dg, prob_bits):
    # every combination of block kinds inside a quad, quads that mix full and partial blocks, element tails of
    # 1 / 2 / 3 blocks + a partial block, an element of exactly 9 blocks (two full quads + one block)
    rng = np.random.default_rng(777 + prob_bits)
    pats = ["zcmr" * 8 + "rrrr" + "zzzz" + "cccc", "rzrz" * 5 + "mmc", "m" * 9, "crmz" * 4 + "rr", "r" * 37, "zm" * 6 + "z"]
    tails = [0, 1, 4095, 33, 2049, 0]
    xs = []
    for p, t in zip(pats, tails):
        parts = [_ring_block(rng, k) for k in p]
        if t:
            parts.append(rng.integers(0, 256, t, dtype=np.uint8))
        xs.append(np.concatenate(parts))
    got = gpu_ans_encode(dg, xs, prob_bits)
    for x, g in zip(xs, got):
        want = O.ans_encode(x, prob_bits)
        assert g.size == want.size and (g == want).all()
    outs, status, osz = gpu_ans_decode(dg, got, [x.size for x in xs], prob_bits)
    assert status.all() and osz.tolist() == [x.size for x in xs]
    for i, (x, o) in enumerate(zip(xs, outs)):
        assert (o == x).all(), (i, int(np.flatnonzero(o != x)[0]))


def test_decode_ring_chunk_boundaries(dg):
    # blocks whose compressed word counts sit on both sides of every chunk boundary of the 1 KiB ring protocol
    # (128-word chunks: 127 .. 129, 255 .. 257, 383 .. 385, 511 .. 513, ...), found by searching symbol mixes
    rng = np.random.default_rng(31337)
    targets = set()
    for k in range(1, 12):
        targets.update({128 * k - 1, 128 * k, 128 * k + 1})
    found = {}
    base = rng.integers(0, 256, 4096, dtype=np.uint8)
    # the table comes from the whole element, so word counts are steered by how many positions of a block hold
    # the (frequent) filler symbol: sweep that count and keep blocks that land on a target
    blocks = []
    for fill in range(0, 4097, 8):
        blk = base.copy()
        blk[rng.permutation(4096)[:fill]] = 0
        blocks.append(blk)
    filler = np.zeros(4096 * 24, np.uint8)
    x = np.concatenate(blocks + [filler])
    arch = O.ans_encode(x, 10)
    words = _block_words(arch)[: len(blocks)]
    hit = sorted(set(int(w) for w in words) & targets)
    assert len(hit) >= 6, hit  # the sweep is dense enough to land on several boundaries
    got = gpu_ans_encode(dg, [x], 10)[0]
    assert got.size == arch.size and (got == arch).all()
    outs, status, _ = gpu_ans_decode(dg, [got], [x.size], 10)
    assert status.all() and (outs[0] == x).all()


def test_decode_ring_slices_and_batches(dg):
    # ragged batch: elements of 9 .. 300 blocks next to tiny ones (the grid is laid out for the largest capacity;
    # slices beyond an element's end and whole absent slices must do nothing), capacity larger than the size
    rng = np.random.default_rng(2024)
    sizes = [4096 * 300 + 17, 4096 * 9, 5, 4096 * 64, 0, 4096 * 131 + 4000, 4096 * 33 - 1]
    xs = [(rng.zipf(1.25, n) % 256).astype(np.uint8) for n in sizes]
    got = gpu_ans_encode(dg, xs, 10)
    for x, g in zip(xs, got):
        want = O.ans_encode(x, 10)
        assert g.size == want.size and (g == want).all()
    outs, status, osz = gpu_ans_decode(dg, got, [n + 100 for n in sizes], 10)
    assert status.all() and osz.tolist() == sizes
    for x, o in zip(xs, outs):
        assert (o[: x.size] == x).all()


def test_decode_ring_rejects_malformed_archives(dg):
    # the checks of test_decoder_rejects_malformed_ans_archives on an element of several 16-block tiles
    # (40 blocks), corruptions in the middle of the block table included
    x = refgen.generate_symbols(39 * 4096 + 100, 20.0)
    good = O.ans_encode(x, 10)
    ref = torch.from_numpy(x).to(DEV)
    st, out = _decode_status(dg, False, good, ref)
    assert st == 1 and torch.equal(out, ref)
    nb = 40
    bw0 = 32 + 512 + 128 * nb

    def u32(a, off):
        return a[off : off + 4].view(np.uint32)

    cases = []
    for name, off, val in [
        ("numBlocks+1", 4, nb + 1), ("total-1", 8, x.size - 1), ("probBits", 16, 9), ("totalCompressedWords small", 12, 64),
        ("block17 uncompressed size", bw0 + 17 * 8, (4095 << 16) | int(u32(good, bw0 + 17 * 8)[0] & 0xffff)),
        ("block22 start past the end", bw0 + 22 * 8 + 4, int(u32(good, 12)[0])),
        ("block35 start unaligned", bw0 + 35 * 8 + 4, int(u32(good, bw0 + 35 * 8 + 4)[0]) + 5),
        ("last block size", bw0 + 39 * 8, (101 << 16) | int(u32(good, bw0 + 39 * 8)[0] & 0xffff)),
    ]:
        bad = good.copy()
        u32(bad, off)[0] = val
        cases.append((name, bad))
    bad = good.copy()
    bad[32:34].view(np.uint16)[0] += 1
    cases.append(("pdf sum", bad))
    for name, bad in cases:
        st, _ = _decode_status(dg, False, bad, ref)
        assert st == 0, name
    # garbage lane states keep every invariant the decoder checks: it decodes garbage without faulting and writes
    # nothing past the capacity
    guard = torch.full((x.size + 8192,), 0xAB, dtype=torch.uint8, device=DEV)
    rng = np.random.default_rng(5)
    bad = good.copy()
    st_off = 32 + 512
    bad[st_off : st_off + 128 * nb] = rng.integers(0, 256, 128 * nb, dtype=np.uint8)
    status = torch.zeros((1,), dtype=torch.uint8, device=DEV)
    dg.decompress_data(False, [torch.from_numpy(bad).to(DEV)], [guard[: x.size]], False, None, status, None)
    torch.cuda.synchronize()
    assert (guard[x.size :] == 0xAB).all()
    # truncated (the tensor API bounds the archive)
    for cut in (32, 544, bw0 + 8, good.size - 16):
        st, _ = _decode_status(dg, False, good[:cut], ref)
        assert st == 0, cut


@pytest.mark.parametrize("as_float", [False, True])
def test_checksum_mismatch_reports_every_member(dg, as_float):
    # upstream pushes EVERY mismatching batch member into errorInfo (GpuANSDecode.cuh:581-590,
    # GpuFloatDecompress.cuh:720-733): three corrupted members of a batch of ten, through the tensor API message
    # and through the C ABI's dgpu_last_checksum_mismatches
    rng = np.random.default_rng(8)
    if as_float:
        words = [refgen.generate_floats(O.BFLOAT16, 5000 + 37 * i) for i in range(10)]
        ts = [words_to_tensor(O.BFLOAT16, w) for w in words]
        ck_off = 12  # GpuFloatHeader::checksum
    else:
        ts = [to_dev_bytes(rng.integers(0, 50, 9000 + 11 * i, dtype=np.uint8)) for i in range(10)]
        ck_off = 20  # ANSCoalescedHeader::checksum
    comp, sizes, _ = dg.compress_data(as_float, ts, True)
    sizes = sizes.cpu().numpy()
    rows = [comp[i, : sizes[i]].clone() for i in range(10)]
    for i in (1, 4, 9):
        rows[i][ck_off] ^= 0x3C
    outs = [torch.empty_like(t) for t in ts]
    with pytest.raises(RuntimeError, match="checksum mismatch") as ei:
        dg.decompress_data(as_float, rows, outs, True)
    text = str(ei.value)
    for i in (1, 4, 9):
        assert f"batch member {i}:" in text
    for i in (0, 2, 3, 5, 6, 7, 8):
        assert f"batch member {i}:" not in text
    L = dg.lib()
    idx = (C.c_int32 * 16)()
    want = (C.c_uint32 * 16)()
    got = (C.c_uint32 * 16)()
    n = L.dgpu_last_checksum_mismatches(idx, want, got, 16)
    assert n == 3 and list(idx[:3]) == [1, 4, 9]
    assert all(want[k] == (got[k] ^ 0x3C) for k in range(3))
    assert L.dgpu_last_checksum_mismatches(None, None, None, 0) == 3  # count only
    # a clean call resets the list
    rows2 = [comp[i, : sizes[i]] for i in range(10)]
    dg.decompress_data(as_float, rows2, outs, True)
    assert L.dgpu_last_checksum_mismatches(None, None, None, 0) == 0
    # the fast tensor surface raises the same text
    import dietgpu_amd

    dietgpu_amd.load_torch_ops()
    with pytest.raises(RuntimeError, match="batch member 4:"):
        torch.ops.dietgpu.decompress_data(as_float, rows, outs, True, None, None, None)


def test_one_gi_bf16_words_and_the_size_guard(dg):
    # (a) the largest point of the reference's README plots: ONE tensor of 1024 Mi bf16 words (2 GiB; exponent plane
    # 262 144 blocks = 32 768 encoder tiles), archive byte for byte against the oracle, round trip bit-exact.
    n = 1 << 30
    rng = np.random.default_rng(77)
    # N(0,1) in bf16, built from a 64 Mi-word piece (host time) with per-piece exponent offsets so that the pieces differ
    piece = (rng.standard_normal(1 << 26, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
    words = np.empty(n, np.uint16)
    for k in range(n >> 26):
        words[k << 26 : (k + 1) << 26] = piece + np.uint16((k % 5) << 7)
    t = torch.from_numpy(words.view(np.int16)).to(DEV).view(torch.bfloat16)
    comp, sizes, _ = dg.compress_data(True, [t])
    got_n = int(sizes[0].item())
    want = O.float_compress(O.BFLOAT16, words, 10)
    assert got_n == want.size
    got = comp[0, :got_n].cpu().numpy()
    assert (got == want).all(), int(np.flatnonzero(got != want)[0])
    out = torch.empty_like(t)
    dg.decompress_data(True, [comp[0, :got_n]], [out])
    assert torch.equal(out.view(torch.int16), t.view(torch.int16))
    del comp, out, t, got, want
    torch.cuda.empty_cache()

    # (b) the reference's size guard (GpuANSEncode.cu:22): 419 321 blocks is the largest raw input; one byte more is
    # rejected by the size query and by the encoder
    guard = 419321 * 4096
    L = dg.lib()
    assert L.dgpu_ans_max_compressed_size(guard) == 557600 + 5120 * 419321 and L.dgpu_ans_max_compressed_size(guard + 1) == 0
    x = np.tile((rng.zipf(1.3, 1 << 24) % 256).astype(np.uint8), guard // (1 << 24) + 1)[:guard]
    tx = torch.from_numpy(x).to(DEV)
    comp, sizes, _ = dg.compress_data(False, [tx])
    k = int(sizes[0].item())
    want = O.ans_encode(x, 10)
    assert k == want.size and (comp[0, :k].cpu().numpy() == want).all()
    outx = torch.empty_like(tx)
    dg.decompress_data(False, [comp[0, :k]], [outx])
    assert torch.equal(outx, tx)
    del comp, outx
    big = torch.zeros((guard + 4,), dtype=torch.uint8, device=DEV)
    with pytest.raises(Exception, match="INT32_MAX|1717538816"):
        dg.compress_data(False, [big])


def test_histogram_load_policy_does_not_change_archives(dg):
    # dgpu_set_histogram_load_policy only changes the cache policy of the histogram pass's loads
    L = dg.lib()
    w = refgen.generate_floats(O.BFLOAT16, 70 * 4096 + 123)
    x = refgen.generate_symbols(50 * 4096 + 7, 30.0)
    t, tx = words_to_tensor(O.BFLOAT16, w), to_dev_bytes(x)
    try:
        for mode in (1, 0, -1):
            L.dgpu_set_histogram_load_policy(mode)
            comp, sizes, _ = dg.compress_data(True, [t])
            n = int(sizes[0].item())
            want = O.float_compress(O.BFLOAT16, w, 10)
            assert n == want.size and (comp[0, :n].cpu().numpy() == want).all(), mode
            comp, sizes, _ = dg.compress_data(False, [tx])
            n = int(sizes[0].item())
            want = O.ans_encode(x, 10)
            assert n == want.size and (comp[0, :n].cpu().numpy() == want).all(), mode
    finally:
        L.dgpu_set_histogram_load_policy(-1)


@pytest.mark.parametrize("ft", [O.FLOAT16, O.BFLOAT16, O.FLOAT32])
def test_capped_stride_compress_and_bounded_stride_decompress(dg, ft):
    # dgpu_float_compress_stride_capped: rows of one matrix compressed straight into the rows of a send matrix at a
    # fixed width.  With room for everything the rows are the oracle's archives; with less, every byte below the
    # capacity is still the oracle's, nothing at or beyond it is written, outSize reports the full size, and the
    # bounded decoder rejects exactly the rows that did not fit (incompressible rows take the encoder's spill path).
    L = dg.lib()
    rng = np.random.default_rng(40 + ft)
    n, B = 9 * 4096 + 40, 6
    wdt = np.uint32 if ft == O.FLOAT32 else np.uint16
    rows = [refgen.generate_floats(ft, n) for _ in range(B)]
    bits = 32 if ft == O.FLOAT32 else 16
    rows[2] = rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(wdt)  # incompressible
    rows[5] = rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(wdt)
    mat = torch.stack([words_to_tensor(ft, r) for r in rows])
    want = [O.float_compress(ft, r, 10) for r in rows]
    cap_full = int(L.dgpu_float_max_compressed_size(ft, n))
    normal = max(a.size for i, a in enumerate(want) if i not in (2, 5))
    width = (normal + 64 + 15) // 16 * 16            # fits the compressible rows, not the noise
    assert all(want[i].size > width for i in (2, 5))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    eb = mat.element_size()
    for cap, stride in ((cap_full, cap_full), (width, width), (width, width + 4096)):
        out = torch.full((B, stride), 0xAB, dtype=torch.uint8, device=DEV)
        sizes = torch.zeros((B,), dtype=torch.int32, device=DEV)
        rc = L.dgpu_float_compress_stride_capped(None, 0, None, ft, 10, 0, B, C.c_void_p(mat.data_ptr()), n, n * eb,
                                                 C.c_void_p(out.data_ptr()), stride, cap, C.c_void_p(sizes.data_ptr()), stream)
        assert rc == 0, L.dgpu_last_error()
        torch.cuda.synchronize()
        got, hs = out.cpu().numpy(), sizes.cpu().numpy()
        for i in range(B):
            assert hs[i] == want[i].size
            k = min(want[i].size, cap)
            assert (got[i, :k] == want[i][:k]).all(), (cap, i, int(np.flatnonzero(got[i, :k] != want[i][:k])[0]))
            assert (got[i, max(k, min(cap, (want[i].size + 15) // 16 * 16)):] == 0xAB).all(), (cap, i)  # nothing beyond
        dec = torch.zeros_like(mat)
        status = torch.full((B,), 7, dtype=torch.uint8, device=DEV)
        err = C.c_int32(-1)
        rc = L.dgpu_float_decompress_stride_bounded(None, 0, None, ft, 10, 0, B, C.c_void_p(out.data_ptr()), stride, cap,
                                                    C.c_void_p(dec.data_ptr()), n * eb, n, C.c_void_p(status.data_ptr()), None,
                                                    stream, C.byref(err))
        assert rc == 0, L.dgpu_last_error()
        torch.cuda.synchronize()
        st = status.cpu().tolist()
        fits = [1 if want[i].size <= cap else 0 for i in range(B)]
        assert st == fits, (cap, st, fits)
        view = torch.int32 if ft == O.FLOAT32 else torch.int16
        for i in range(B):
            if fits[i]:
                assert torch.equal(dec[i].view(view), mat[i].view(view))
    # a capacity that cannot even hold the tables and the non-compressed plane is an argument error
    out = torch.empty((B, 4096), dtype=torch.uint8, device=DEV)
    rc = L.dgpu_float_compress_stride_capped(None, 0, None, ft, 10, 0, B, C.c_void_p(mat.data_ptr()), n, n * eb,
                                             C.c_void_p(out.data_ptr()), 4096, 4096, None, stream)
    assert rc == 1  # DGPU_ERR_INVALID_ARGUMENT


# ----------------------------------------------------------------- encoder dispatch modes
@pytest.mark.parametrize("mode", [0, 1])
def test_encoder_dispatch_modes(dg, mode):
    # The tiled encoder runs either as persistent workgroups with a static ticket map (0) or as one workgroup per tile
    # dispatched by the hardware (1) -- raw bytes and float tiles of 2 / 4 blocks; the library picks per call (more
    # tiles than resident workgroups: 1), this hook forces one.  Archives are byte-identical to the oracle either way:
    # late workgroups and the take-over of their tiles, look-back chains over 1027 tiles, ragged batches, BASELINE
    # config 2, small float elements with spilling members.  (8-block float tiles are always persistent.)
    L = dg.lib()
    L.dgpu_debug_set_encoder_dispatch(mode)
    try:
        test_encoder_with_absent_workgroups(dg, 3)
        test_lookback_windows_raw(dg, 129, 3)
        test_lookback_windows_raw(dg, 1027, 7)
        test_fuzz_ragged_batches(dg, 1)
        test_fuzz_ragged_batches(dg, 4)
        test_baseline_config2_zipf_bytes(dg)
        test_ragged_batches_of_small_elements(dg, 10, 2)  # float tiles of 2 / 4 blocks: both forms exist
        test_ragged_batches_of_small_elements(dg, 11, 4)
    finally:
        L.dgpu_debug_set_encoder_dispatch(-1)


@pytest.mark.parametrize("order", [0, 1, 2])
def test_decoder_workgroup_orders(dg, order):
    # k_ans_decode's 1-D grid maps workgroup index -> (element, tile) element-major, tile-major or per XCD (every XCD
    # walks the elements b = XCD mod 8); the library picks by batch size, this hook forces one.  Batch sizes below,
    # at and off the multiples of eight, ragged element sizes (tiles beyond an element's end), both tile shapes,
    # raw bytes and floats: decoded data identical, status and sizes as reported by the default order.
    L = dg.lib()
    rng = np.random.default_rng(300 + order)
    L.dgpu_debug_set_decoder_order(order)
    try:
        for B in (1, 3, 8, 13, 70):
            ns = [int(rng.integers(1, 40)) * 4096 * int(rng.choice([1, 9])) + int(rng.integers(0, 4096)) for _ in range(B)]
            ns[0] = 4096 * 16 * 5  # whole tiles
            ws = [refgen.generate_floats(O.BFLOAT16, n) for n in ns]
            ts = [words_to_tensor(O.BFLOAT16, w) for w in ws]
            comp, sizes, _ = dg.compress_data(True, ts, False, prob_bits=10)
            hs = sizes.cpu().numpy()
            arch = [comp[i, : hs[i]].clone() for i in range(B)]
            outs = [torch.empty_like(t) for t in ts]
            status = torch.zeros((B,), dtype=torch.uint8, device=DEV)
            osz = torch.zeros((B,), dtype=torch.int32, device=DEV)
            dg.decompress_data(True, arch, outs, False, None, status, osz, prob_bits=10)
            assert status.cpu().numpy().all() and (osz.cpu().numpy() == np.array(ns)).all()
            for w, o in zip(ws, outs):
                assert (tensor_to_words(O.BFLOAT16, o) == w).all()
        xs = [rng.integers(0, 90, int(n), dtype=np.uint8) for n in (4096 * 16 * 3 + 5, 100, 4096 * 40, 4096 * 16)]
        got = gpu_ans_encode(dg, xs, 10)
        outs, status, osz = gpu_ans_decode(dg, got, [x.size for x in xs], 10)
        assert status.all()
        for x, o in zip(xs, outs):
            assert (o == x).all()
    finally:
        L.dgpu_debug_set_decoder_order(-1)


@pytest.mark.parametrize("ft,prob_bits,blocks", [(O.BFLOAT16, 10, 1), (O.FLOAT16, 11, 1), (O.FLOAT32, 9, 1),
                                                 (O.BFLOAT16, 10, 2), (O.FLOAT16, 11, 4), (O.FLOAT32, 10, 2)])
def test_small_elements_with_more_spilling_wavefronts_than_pool_slots(dg, ft, prob_bits, blocks):
    # The float encoders of small elements -- k_ans_encode_pair (single-block elements) and k_ans_encode with tiles of 2
    # or 4 blocks -- run one workgroup per pair / tile and take their spill slots from a pool sized for the wavefronts
    # that can be resident (a few thousand pairs of slots), handed out through library-owned flags.  Thousands of
    # elements of random bit patterns make EVERY wavefront spill: many more spilling wavefronts than slots, slots
    # changing hands between XCDs.  Archives byte-identical to the oracle, round trip exact, and a second call (the
    # flags are zero at rest) gives the same archives.
    if getattr(dg, "name", "") == "torch_ops":
        dg._p10(prob_bits)
    rng = np.random.default_rng(2100 + ft + 10 * blocks)
    dt = np.uint32 if ft == O.FLOAT32 else np.uint16
    hi = 1 << (32 if ft == O.FLOAT32 else 16)
    B = 20000 // blocks
    ns = rng.integers(blocks * 4096 - 196, blocks * 4096 + 1, B)
    ns[::7] = blocks * 4096
    flat = rng.integers(0, hi, int(ns.sum()), dtype=np.uint64).astype(dt)
    splits = torch.from_numpy(ns.astype(np.int32))
    t = words_to_tensor(ft, flat)
    archives = None
    for rep in range(2):
        comp, sizes, _ = dg.compress_data_split_size(True, t, splits, False, prob_bits=prob_bits)
        hs = sizes.cpu().numpy()
        rows = [c[: hs[i]].cpu().numpy() for i, c in enumerate(comp)] if rep == 0 else None
        if rep == 0:
            archives = rows
            off = 0
            for i in range(0, B, 97):  # every 97th element against the oracle (all of them take seconds of Python)
                o = int(ns[:i].sum())
                want = O.float_compress(ft, flat[o : o + int(ns[i])], prob_bits)
                assert hs[i] == want.size and (rows[i] == want).all(), i
        else:
            for i in range(0, B, 501):
                assert (comp[i][: hs[i]].cpu().numpy() == archives[i]).all()
    out = torch.empty_like(t)
    status = torch.zeros((B,), dtype=torch.uint8, device=DEV)
    dg.decompress_data_split_size(True, [c[: hs[i]] for i, c in enumerate(comp)], out, splits, False, None, status, None,
                                  prob_bits=prob_bits)
    assert status.cpu().numpy().all()
    assert (tensor_to_words(ft, out) == flat).all()


# ----------------------------------------------------------------- batches whose elements differ widely in size
def _mixed_size_batch(rng, ft, big, count):
    """One element of `big` symbols next to `count` small ones (empty, below a block, a few blocks)."""
    ns = [big] + [int(n) for n in rng.integers(0, 3 * 4096, count)]
    ns[3], ns[7] = 0, 4096
    rng.shuffle(ns)
    if ft == 0:
        return ns, [np.ascontiguousarray(refgen.generate_symbols(max(n, 1), 30.0 + i)[:n] if i % 4 else
                                         rng.integers(0, 256, n, dtype=np.uint8)) for i, n in enumerate(ns)]
    dt = np.uint32 if ft == O.FLOAT32 else np.uint16
    return ns, [np.ascontiguousarray(refgen.generate_floats(ft, max(n, 1))[:n] if i % 4 else
                                     rng.integers(0, 1 << (8 * dt().itemsize), n, dtype=np.uint64).astype(dt), dt) for i, n in enumerate(ns)]


@pytest.mark.parametrize("mode", [-1, 0, 1])
@pytest.mark.parametrize("ft", [0, O.BFLOAT16, O.FLOAT32])
def test_batches_of_widely_different_sizes(dg, ft, mode):
    # The kernels' grids are rectangles laid out for the largest element; when at least a fifth of such a rectangle would
    # be empty the host lists the tiles and histogram parts that exist and the kernels work through the lists
    # (capi.hip, RaggedPlan).  mode -1: the library's policy (lists here: 1 element of 70 blocks next to 60 of < 3),
    # 0: the rectangles, 1: lists forced -- also on the batches of the ragged fuzz test, which the policy would leave
    # to the rectangles.  Archives byte-identical to the oracle in every mode, with checksums, decoded through the
    # same mode into capacities larger than the sizes (the decoder's list comes from the capacities).
    L = dg.lib()
    L.dgpu_debug_set_work_lists(mode)
    try:
        rng = np.random.default_rng(8800 + ft)
        ns, ws = _mixed_size_batch(rng, ft, 70 * 4096 + 1234, 60)
        if ft == 0:
            got = gpu_ans_encode(dg, ws, 10, True)
            for w, g in zip(ws, got):
                want = O.ans_encode(w, 10, use_checksum=True)
                assert g.size == want.size and not (g != want).any(), ("raw", w.size)
            outs, status, osz = gpu_ans_decode(dg, got, [n + (i % 5) * 1000 for i, n in enumerate(ns)], 10, True)
            assert status.all() and osz.tolist() == ns and all((o[: w.size] == w).all() for o, w in zip(outs, ws))
        else:
            ts = [words_to_tensor(ft, w) for w in ws]
            comp, sizes, _ = dg.compress_data(True, ts, True)
            hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
            arch = []
            for i, w in enumerate(ws):
                want = O.float_compress(ft, w, 10, use_checksum=True)
                assert hs[i] == want.size and not (hc[i, : hs[i]] != want).any(), (ft, w.size)
                arch.append(comp[i, : hs[i]].clone())
            outs = [torch.empty((n + (i % 5) * 1000,), dtype=FT_DTYPE[ft], device=DEV) for i, n in enumerate(ns)]
            status = torch.zeros((len(ns),), dtype=torch.uint8, device=DEV)
            osz = torch.zeros((len(ns),), dtype=torch.int32, device=DEV)
            dg.decompress_data(True, arch, outs, True, None, status, osz)
            assert status.cpu().numpy().all() and osz.cpu().tolist() == ns
            assert all((tensor_to_words(ft, o[:n]) == w).all() for o, n, w in zip(outs, ns, ws))
        if mode == 1 and ft == 0:
            test_fuzz_ragged_batches(dg, 2)
            test_encoder_with_absent_workgroups(dg, 3)
            test_decode_ring_slices_and_batches(dg)
        if mode == 1 and ft == O.BFLOAT16:
            test_ragged_batches_of_small_elements(dg, 10, 4)
            test_partial_last_blocks_at_every_row_count(dg, 3)
            test_decoder_rejects_truncated_archives(dg)
    finally:
        L.dgpu_debug_set_work_lists(-1)


def test_one_large_tensor_among_many_small_ones(dg):
    # the shape tools/ragged_probe.py times (5.7 ms per compress call on the rectangles): one 8 Mi-word bf16 tensor and
    # 255 tensors of 2048 words; round trip, sizes, and the same archives from lists and rectangles
    g = torch.Generator(device=DEV).manual_seed(77)
    ts = [torch.randn(8 << 20, generator=g, device=DEV).to(torch.bfloat16)]
    ts += [torch.randn(2048, generator=g, device=DEV).to(torch.bfloat16) for _ in range(255)]
    L = dg.lib()
    comps = []
    for mode in (-1, 0):
        L.dgpu_debug_set_work_lists(mode)
        try:
            comp, sizes, _ = dg.compress_data(True, ts, True)
            rows = [comp[i, : int(s)].clone() for i, s in enumerate(sizes.cpu().tolist())]
            outs = [torch.empty_like(t) for t in ts]
            status = torch.zeros((len(ts),), dtype=torch.uint8, device=DEV)
            dg.decompress_data(True, rows, outs, True, None, status)
            assert status.cpu().numpy().all()
            assert all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(ts, outs))
            comps.append(rows)
        finally:
            L.dgpu_debug_set_work_lists(-1)
    assert all(torch.equal(a, b) for a, b in zip(*comps))


@pytest.mark.parametrize("mode", [1, -1])
@pytest.mark.parametrize("ft", [0, O.BFLOAT16, O.FLOAT16, O.FLOAT32])
def test_size_classes_inside_one_batch(dg, ft, mode):
    # A batch whose members fall into several size classes -- single blocks, <= 2, <= 4 (decode: <= 8) blocks, more --
    # runs every class on the kernels of its own geometry, one class after the other (capi.hip, EncodeClass /
    # DecodeClass).  mode 1: every class that has a member, however few (one large tensor, some of 5-7 blocks, some of
    # 3-4, some of 2, many single blocks, empty ones); mode -1: the library's policy (classes of fewer than 32 members
    # join the next larger one; here the 300 single blocks next to the large tensors make it split).  Archives
    # byte-identical to the oracle, with checksums; decoded through the same mode into capacities that put some members
    # into ANOTHER class than their size did.
    L = dg.lib()
    L.dgpu_debug_set_size_classes(mode)
    try:
        rng = np.random.default_rng(4200 + ft)
        ns = [40 * 4096 + 77, 0, 4096, 1]
        ns += [int(n) for n in rng.integers(4 * 4096 + 1, 7 * 4096, 5)]
        ns += [int(n) for n in rng.integers(2 * 4096 + 1, 4 * 4096 + 1, 6)]
        ns += [int(n) for n in rng.integers(4096 + 1, 2 * 4096 + 1, 7)]
        ns += [int(n) for n in rng.integers(1, 4097, 300 if mode == -1 else 40)]
        rng.shuffle(ns)
        if ft == 0:
            ws = [np.ascontiguousarray(refgen.generate_symbols(max(n, 1), 20.0 + i % 7)[:n] if i % 5 else
                                       rng.integers(0, 256, n, dtype=np.uint8)) for i, n in enumerate(ns)]
            got = gpu_ans_encode(dg, ws, 10, True)
            for w, g in zip(ws, got):
                want = O.ans_encode(w, 10, use_checksum=True)
                assert g.size == want.size and not (g != want).any(), ("raw", w.size)
            caps = [n + (i % 4) * 3000 for i, n in enumerate(ns)]
            outs, status, osz = gpu_ans_decode(dg, got, caps, 10, True)
            assert status.all() and osz.tolist() == ns and all((o[: w.size] == w).all() for o, w in zip(outs, ws))
        else:
            dt = np.uint32 if ft == O.FLOAT32 else np.uint16
            ws = [np.ascontiguousarray(refgen.generate_floats(ft, max(n, 1))[:n] if i % 5 else
                                       rng.integers(0, 1 << (8 * dt().itemsize), n, dtype=np.uint64).astype(dt), dt) for i, n in enumerate(ns)]
            ts = [words_to_tensor(ft, w) for w in ws]
            comp, sizes, _ = dg.compress_data(True, ts, True)
            hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
            arch = []
            for i, w in enumerate(ws):
                want = O.float_compress(ft, w, 10, use_checksum=True)
                assert hs[i] == want.size and not (hc[i, : hs[i]] != want).any(), (ft, w.size)
                arch.append(comp[i, : hs[i]].clone())
            outs = [torch.empty((n + (i % 4) * 3000,), dtype=FT_DTYPE[ft], device=DEV) for i, n in enumerate(ns)]
            status = torch.zeros((len(ns),), dtype=torch.uint8, device=DEV)
            osz = torch.zeros((len(ns),), dtype=torch.int32, device=DEV)
            dg.decompress_data(True, arch, outs, True, None, status, osz)
            assert status.cpu().numpy().all() and osz.cpu().tolist() == ns
            assert all((tensor_to_words(ft, o[:n]) == w).all() for o, n, w in zip(outs, ns, ws))
    finally:
        L.dgpu_debug_set_size_classes(-1)
