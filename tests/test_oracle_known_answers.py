"""Pins the CPU oracle against everything the reference's own tests pin.

Reference tests restated here (citations into /root/reference/dietgpu/):
  ans/ANSStatisticsTest.cu:44-95    Histogram == CPU count (unaligned starts)
  ans/ANSStatisticsTest.cu:127-149  Normalization_NonZero  (known answer)
  ans/ANSStatisticsTest.cu:151-167  Normalization_EqualWeight (known answer)
  ans/ANSStatisticsTest.cu:169-207  Normalization invariants
  ans/ANSTest.cu:243-282            ZeroSized, BatchPointer, BatchPointerLarge, BatchStride
  float/FloatTest.cu:270-311        Batch / BatchSize1 round trips
  ans_test.py:21-26,68-77, float_test.py:88-92
plus struct-size / size-formula constants (GpuANSUtils.cuh:229,
GpuFloatUtils.cuh:74, GpuANSEncode.cu:13-25, GpuFloatCompress.cu:23-45).
"""
import struct

import numpy as np
import pytest

import oracle as O
from oracle import pyref
import refgen


# ------------------------------------------------------------------ constants
def test_size_formulas():
    # getCompressedOverhead(nb) = 32 + 512 + 128 nb + 8 roundUp(nb, 2)
    assert O.ans_compressed_overhead(0) == 544
    assert O.ans_compressed_overhead(1) == 544 + 128 + 16
    assert O.ans_compressed_overhead(256) == 35360
    assert O.ans_compressed_overhead(128) == 17952
    # getMaxCompressedSize: overhead(4096) [sic] + 5120 * blocks, rounded to 16
    assert O.ans_max_compressed_size(0) == 557600
    assert O.ans_max_compressed_size(1) == 557600 + 5120
    assert O.ans_max_compressed_size(1 << 20) == 1868320
    # float: 16 + ans max + non-comp plane
    assert O.float_uncomp_data_size(O.BFLOAT16, 524288) == 524288
    assert O.float_uncomp_data_size(O.FLOAT16, 17) == 32
    assert O.float_uncomp_data_size(O.FLOAT32, 17) == 2 * 24 + 32
    assert O.float_max_compressed_size(O.BFLOAT16, 524288) == 1737264


def test_header_layout():
    x = refgen.generate_symbols(10000, 20.0)
    a = O.ans_encode(x, 10, use_checksum=True)
    magic, nb, size, total_words, opts, ck, u0, u1 = struct.unpack_from("<8I", a.tobytes(), 0)
    assert magic == 0xD00D0001 and nb == 3 and size == 10000
    assert opts == (10 | 0x10) and u0 == 0 and u1 == 0
    assert ck == int(np.bitwise_xor.reduce(x))
    assert len(a) == O.ans_compressed_overhead(nb) + 2 * total_words
    # blockWords entries
    off = 32 + 512 + 128 * nb
    start = 0
    for b in range(nb):
        w0, w1 = struct.unpack_from("<2I", a.tobytes(), off + 8 * b)
        n = min(4096, size - 4096 * b)
        assert (w0 >> 16) == n and w1 == start and start % 8 == 0
        start += ((w0 & 0xFFFF) + 7) // 8 * 8
    assert start == total_words
    # odd block count => one zero pad entry
    assert struct.unpack_from("<2I", a.tobytes(), off + 8 * nb) == (0, 0)


# ----------------------------------------------------------------- statistics
@pytest.mark.parametrize("size", [1, 2, 11, 32, 55, 64, 99, 1000, 12345, 123457])
def test_histogram(size):
    rng = np.random.default_rng(size)
    buf = rng.integers(0, 256, size + 11 + 16, dtype=np.uint8)
    for start in (0, 1, 11):  # unaligned starts, ANSStatisticsTest.cu:52-57
        x = buf[start : start + size]
        assert (O.histogram(x) == np.bincount(x, minlength=256)).all()


def test_normalization_nonzero():
    data = np.ones(10000, np.uint8)
    data[:256] = np.arange(256)
    t = O.normalize(O.histogram(data), data.size, 10)
    for i in range(256):
        assert t[i, 0] == ((1 << 10) - 255 if i == 1 else 1)


def test_normalization_equal_weight():
    data = np.tile(np.arange(256, dtype=np.uint8), 64)
    t = O.normalize(O.histogram(data), data.size, 10)
    assert (t[:, 0] == 4).all()
    assert (t[:, 1] == 4 * np.arange(256)).all()


def test_normalization_invariants():
    data = refgen.generate_symbols(12345, 40.0)
    hist = np.bincount(data, minlength=256)
    prob_bits = 11
    t = O.normalize(O.histogram(data), data.size, prob_bits)
    w = 1 << prob_bits
    assert int(t[:, 0].sum()) == w
    for i in range(256):
        c, pdf = hist[i], int(t[i, 0])
        if c == 0:
            assert pdf == 0, i
        else:
            assert pdf > 0
            prob = np.float32(c) / np.float32(data.size)
            norm = np.float32(pdf) / np.float32(w)
            assert norm >= prob * np.float32(0.5)
            if prob > np.float32(1.0) / np.float32(w):
                assert norm <= prob * np.float32(2.0)


def test_normalization_surplus_goes_to_low_symbols():
    # diff > 0 branch compares the SYMBOL index (GpuANSStatistics.cuh:262-270):
    # 3 equiprobable symbols 100,150,200 at P=10 -> floor = 341 each, diff = 1
    # -> symbol 0 (count 0!) receives the surplus.
    counts = np.zeros(256, np.uint32)
    counts[[100, 150, 200]] = 1000
    t = O.normalize(counts, 3000, 10)
    assert t[0, 0] == 1 and t[100, 0] == 341 and t[150, 0] == 341 and t[200, 0] == 341
    assert int(t[:, 0].sum()) == 1024


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_division_magic_exact(prob_bits):
    # (mulhi(x, magic) + x) >> shift == x / pdf for every x the encoder can see
    rng = np.random.default_rng(7)
    for pdf in [1, 2, 3, 5, 7, 255, 256, 257, 511, (1 << prob_bits) - 1, 1 << prob_bits]:
        if pdf > (1 << prob_bits):
            continue
        counts = np.zeros(256, np.uint32)
        counts[0] = pdf
        counts[1] = (1 << prob_bits) - pdf
        t = O.normalize(counts, 1 << prob_bits, prob_bits)
        assert t[0, 0] == pdf
        magic, shift = int(t[0, 2]), int(t[0, 3])
        hi = pdf << (31 - prob_bits)
        xs = np.concatenate(
            [rng.integers(0, hi, 20000, dtype=np.int64), np.array([0, 1, pdf - 1, pdf, hi - 1])]
        )
        for x in xs.tolist():
            assert (((x * magic) >> 32) + x) >> shift == x // pdf


# ------------------------------------------------------- ANS round trips
SIZE_SETS = [[1], [1, 1], [4096, 4095, 4096], [1234, 2345, 3456], [10000, 10013, 10000]]


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
@pytest.mark.parametrize("lam", [1.0, 10.0, 100.0, 1000.0])
def test_ans_batch_pointer(prob_bits, lam):
    for sizes in SIZE_SETS:
        for n in sizes:
            x = refgen.generate_symbols(n, lam)
            a = O.ans_encode(x, prob_bits, use_checksum=True)
            assert len(a) % 16 == 0  # ANSTest.cu:131-135
            # exact-size buffer (ans_test.py:21-26): a has exactly the reported size
            assert O.ans_info(a)["compressed"] == len(a)
            rc, y, rep = O.ans_decode(a, prob_bits)
            assert rc == 0 and rep == n and (y == x).all()
            assert O.ans_info(a)["checksum"] == O.checksum(x)


def test_ans_batch_pointer_large():
    rng = np.random.default_rng(10)
    for n in rng.integers(100, 10000, 100).tolist():
        x = refgen.generate_symbols(n, 20.0)
        a = O.ans_encode(x, 10)
        rc, y, _ = O.ans_decode(a, 10)
        assert rc == 0 and (y == x).all()


def test_ans_zero_sized():
    a = O.ans_encode(np.zeros(0, np.uint8), 10, use_checksum=True)
    assert len(a) == 544  # header + pdf only
    rc, y, rep = O.ans_decode(a, 10)
    assert rc == 0 and rep == 0 and y.size == 0


def test_ans_capacity_too_small():
    x = refgen.generate_symbols(5000, 20.0)
    a = O.ans_encode(x, 10)
    rc, y, rep = O.ans_decode(a, 10, capacity=4999)
    assert rc == 1 and rep == 5000  # outSuccess = 0, outSize = required size


def test_ans_precomputed_histogram():
    x = refgen.generate_symbols(30000, 10.0)
    a = O.ans_encode(x, 10)
    b = O.ans_encode(x, 10, counts=np.bincount(x, minlength=256))
    assert (a == b).all()


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_worst_case_block_words(prob_bits):
    # A block made of globally rare symbols costs ~probBits bits/symbol; the
    # per-lane bound is 8*probBits + 1 words (encode LDS stage).
    rng = np.random.default_rng(3)
    n = 1 << 22
    x = np.zeros(n, np.uint8)
    x[:4096] = rng.integers(1, 256, 4096, dtype=np.uint8)
    a = O.ans_encode(x, prob_bits)
    w0 = struct.unpack_from("<I", a.tobytes(), 32 + 512 + 128 * 1024)[0] & 0xFFFF
    assert w0 <= 32 * (8 * prob_bits + 1)
    rc, y, _ = O.ans_decode(a, prob_bits)
    assert rc == 0 and (y == x).all()


# ------------------------------------------- two independent restatements agree
@pytest.mark.parametrize("prob_bits", [9, 10, 11])
@pytest.mark.parametrize("n", [0, 1, 31, 32, 33, 4095, 4096, 4097, 9000])
def test_c_oracle_matches_python_restatement(prob_bits, n):
    x = refgen.generate_symbols(n, 50.0)
    a = O.ans_encode(x, prob_bits, use_checksum=True)
    b = pyref.ans_encode(x.tobytes(), prob_bits, use_checksum=True)
    assert a.tobytes() == b
    assert pyref.ans_decode(a, prob_bits) == x.tobytes()


@pytest.mark.parametrize("ft", [O.FLOAT16, O.BFLOAT16, O.FLOAT32])
@pytest.mark.parametrize("n", [0, 1, 7, 16, 17, 5000])
def test_c_oracle_float_matches_python_restatement(ft, n):
    w = refgen.generate_floats(ft, n)
    a = O.float_compress(ft, w, 10, use_checksum=True)
    b = pyref.float_compress(ft, w.tolist(), 10, use_checksum=True)
    assert a.tobytes() == b


# ------------------------------------------------------------- float codec
def test_float_split_join_bit_patterns():
    # fp16: comp = high byte; bf16: comp = 8-bit exponent, nonComp = mant7<<1|sign
    c, nc = O.float_split(O.FLOAT16, np.array([0xABCD], np.uint16))
    assert c[0] == 0xAB and nc[0] == 0xCD
    w = np.array([0xBF80, 0x3F80, 0x7FFF, 0x8000, 0x0001], np.uint16)  # -1, 1, nan, -0, denorm
    c, nc = O.float_split(O.BFLOAT16, w)
    assert c.tolist() == [0x7F, 0x7F, 0xFF, 0x00, 0x00]
    assert nc[:5].tolist() == [0x01, 0x00, 0xFE, 0x01, 0x02]
    assert (O.float_join(O.BFLOAT16, c, nc, 5) == w).all()
    # exhaustive 16-bit round trip
    allw = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    for ft in (O.FLOAT16, O.BFLOAT16):
        c, nc = O.float_split(ft, allw)
        assert (O.float_join(ft, c, nc, allw.size) == allw).all()
    w32 = np.random.default_rng(0).integers(0, 1 << 32, 100001, dtype=np.uint64).astype(np.uint32)
    c, nc = O.float_split(O.FLOAT32, w32)
    assert (c == ((w32 >> 23) & 0xFF)).all()  # the fp32 exponent
    assert (O.float_join(O.FLOAT32, c, nc, w32.size) == w32).all()


@pytest.mark.parametrize("ft", [O.FLOAT16, O.BFLOAT16, O.FLOAT32])
@pytest.mark.parametrize("prob_bits", [9, 10])
def test_float_batch(ft, prob_bits):
    rng = np.random.default_rng(ft * 100 + prob_bits)
    for batch in (1, 3, 16, 23):
        for mult16 in (False, True):
            for _ in range(batch):
                n = int(rng.integers(1, 8192))
                if mult16:
                    n = (n + 15) // 16 * 16
                w = refgen.generate_floats(ft, n)
                a = O.float_compress(ft, w, prob_bits, use_checksum=True)
                assert len(a) % 16 == 0
                assert O.float_info(a)["compressed"] == len(a)
                rc, y, rep = O.float_decompress(ft, a, prob_bits)
                assert rc == 0 and rep == n and (y == w).all()


def test_float_header_and_ratio():
    n = 100000
    for ft, wsize, expect in ((O.BFLOAT16, 2, 0.68), (O.FLOAT16, 2, 0.87), (O.FLOAT32, 4, 0.85)):
        w = refgen.generate_floats(ft, n)
        a = O.float_compress(ft, w, 10)
        magic, cnt, opts, ck = struct.unpack_from("<4I", a.tobytes(), 0)
        assert magic == 0xF00F0001 and cnt == n and opts == ft and ck == 0
        ratio = len(a) / (n * wsize)
        assert ratio < 1.0  # float_test.py:88-92
        assert abs(ratio - expect) < 0.03  # README plot titles: 0.673 / 0.861


def test_float_empty_and_capacity():
    a = O.float_compress(O.BFLOAT16, np.zeros(0, np.uint16), 10)
    assert len(a) == 16 + 544
    rc, y, rep = O.float_decompress(O.BFLOAT16, a, 10)
    assert rc == 0 and rep == 0
    w = refgen.generate_floats(O.FLOAT16, 3000)
    a = O.float_compress(O.FLOAT16, w, 10)
    rc, _, rep = O.float_decompress(O.FLOAT16, a, 10, capacity=2999)
    assert rc == 1 and rep == 3000


def test_float_checksum_quirk():
    # only the first n BYTES (n = float count) are covered
    w = refgen.generate_floats(O.BFLOAT16, 1000)
    a = O.float_compress(O.BFLOAT16, w, 10, use_checksum=True)
    assert O.float_info(a)["checksum"] == O.checksum(w.view(np.uint8)[:1000])


# -------------------------------------------------------- BASELINE config 1
def test_baseline_config1_cpu_roundtrip():
    x = refgen.zipf_bytes(1, 1 << 20)[0]
    a = O.ans_encode(x, 10)
    rc, y, _ = O.ans_decode(a, 10)
    assert rc == 0 and (y == x).all()
    assert abs(len(a) / x.size - 0.69) < 0.02
