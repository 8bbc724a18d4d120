"""Pins the CPU oracle (oracle/dietgpu_oracle.c) to the REFERENCE ITSELF.

oracle/_ref/libdietgpu_ref.so is the reference's own source tree
(facebookresearch/dietgpu: dietgpu/ans/*.cuh, dietgpu/float/*.cuh, ...) compiled
with g++ against a CPU emulation of the CUDA execution model (oracle/ref_shim/)
and executed: its archives are what the reference's kernels compute.  These
tests compare the oracle with it byte for byte (indeterminate header bytes
blanked, tests/refmask.py) on the reference's own test shapes and generators
(ANSTest.cu:18-31,243-282, ANSStatisticsTest.cu:44-207, FloatTest.cu:110-311).

The library is built in the container that has /root/reference
(`__graft_entry__.build()` / `make -C oracle ref`) and travels prebuilt; where
it is absent these tests skip, and tests/test_golden.py still checks the
fixtures the reference generated (tests/golden/ref_*.bin).
"""
import numpy as np
import pytest

import oracle as O
import refgen
from oracle import ref as R
from refmask import mask_ans, mask_float

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference: make -C oracle ref)")


def same(a, b, mask):
    return a.size == b.size and not (mask(a) != mask(b)).any()


def test_layout_and_size_formulas():
    L = R.lib()
    assert L.dgref_sizeof_ans_header() == 32 and L.dgref_sizeof_float_header() == 16 and L.dgref_sizeof_warp_state() == 128
    for n in (0, 1, 4095, 4096, 4097, 1 << 20, (1 << 20) + 1, 123456789):
        assert L.dgref_ans_max_compressed_size(n) == O.ans_max_compressed_size(n)
        for ft in (1, 2, 3):
            assert L.dgref_float_max_compressed_size(ft, n) == O.float_max_compressed_size(ft, n)
            assert L.dgref_float_uncomp_data_size(ft, n) == O.float_uncomp_data_size(ft, n)
    for nb in (0, 1, 2, 3, 255, 256, 257, 4096):
        assert L.dgref_ans_compressed_overhead(nb) == O.ans_compressed_overhead(nb)
    # the reference's header accessors applied to an ORACLE archive find every section where the oracle put it
    x = refgen.generate_symbols(3 * 4096 + 17, 20.0)
    a = O.ans_encode(x, 11, use_checksum=True)
    f = R.ans_header_fields(a)
    assert f["magic_ok"] == 1 and f["num_blocks"] == 4 and f["total_uncompressed_words"] == x.size
    assert f["prob_bits"] == 11 and f["use_checksum"] == 1 and f["checksum"] == O.checksum(x)
    assert (f["pdf_offset"], f["states_offset"]) == (32, 544)
    assert f["block_words_offset"] == 544 + 128 * 4 and f["block_data_offset"] == 544 + 128 * 4 + 8 * 4
    assert f["total_compressed_size"] == a.size


@pytest.mark.parametrize("size", [1, 11, 32, 55, 1000, 12345])
def test_histogram(size):
    rng = np.random.default_rng(size)
    x = rng.integers(0, 256, size, dtype=np.uint8)
    for mis in (0, 1, 5, 11):  # ANSStatisticsTest.cu:52-57: unaligned starts
        assert (R.histogram(x, mis) == O.histogram(x)).all()


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_normalisation_including_adversarial_counts(prob_bits):
    rng = np.random.default_rng(prob_bits)
    rows = []
    d = np.ones(10000, np.uint8)
    d[:256] = np.arange(256)
    rows.append(np.bincount(d, minlength=256))  # ANSStatisticsTest.cu:127-149
    rows.append(np.full(256, 64))               # :151-167
    for _ in range(150):
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
    counts = np.stack(rows).astype(np.uint32)
    totals = counts.astype(np.int64).sum(axis=1)
    keep = totals < (1 << 32)
    counts, totals = counts[keep], totals[keep].astype(np.uint32)
    got = R.normalize_batch(counts, totals, prob_bits)
    for i in range(counts.shape[0]):
        want = np.asarray(O.normalize(counts[i], int(totals[i]), prob_bits)).reshape(256, 4)
        live = want[:, 0] > 0  # magic / shift of pdf-0 symbols are undefined upstream (division by zero) and never used
        assert (got[i][:, :2] == want[:, :2]).all(), i
        assert (got[i][live] == want[live]).all(), i


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
@pytest.mark.parametrize("lam", [1.0, 10.0, 100.0, 1000.0])
def test_ans_archives(prob_bits, lam):
    # ANSTest.cu:243-282 size sets, generator ANSTest.cu:18-31
    for n in (1, 31, 32, 33, 4095, 4096, 4097, 10000, 70001):
        x = refgen.generate_symbols(n, lam)
        for ck in (False, True):
            ref = R.ans_encode_batch([x], prob_bits, ck)[0]
            ora = O.ans_encode(x, prob_bits, use_checksum=ck)
            assert same(ref, ora, mask_ans), (n, ck)
    # cross decoding: the reference decodes the oracle's archive, the oracle the reference's
    x = refgen.generate_symbols(50000, lam)
    ref = R.ans_encode_batch([x], prob_bits, True)[0]
    ora = O.ans_encode(x, prob_bits, use_checksum=True)
    outs, ok, osz, rc = R.ans_decode_batch([ora], [x.size], prob_bits, True)
    assert ok[0] == 1 and osz[0] == x.size and rc == 0 and (outs[0] == x).all()
    rc2, dec, rep = O.ans_decode(ref, prob_bits)
    assert rc2 == 0 and rep == x.size and (dec == x).all()


def test_ans_batch_of_mixed_sizes_and_empty():
    rng = np.random.default_rng(3)
    xs = [refgen.generate_symbols(n, 20.0) for n in (0, 5, 4096 * 3, 100, 4096 * 2 + 1, 0, 12289)]
    refs = R.ans_encode_batch(xs, 10, False)
    for x, r in zip(xs, refs):
        assert same(r, O.ans_encode(x, 10), mask_ans), x.size
    assert refs[0].size == 544  # ans_test.py:68-77


def test_ans_decode_status_semantics():
    x = refgen.generate_symbols(9000, 20.0)
    a = O.ans_encode(x, 10, use_checksum=True)
    # capacity too small: outSuccess 0, outSize = required size, nothing else (GpuANSDecode.cuh:325-341)
    outs, ok, osz, rc = R.ans_decode_batch([a], [8999], 10, False)
    assert ok[0] == 0 and osz[0] == 9000 and rc == 0
    rc2, _, rep = O.ans_decode(a, 10, capacity=8999)  # the oracle reports the same: failure + required size
    assert rc2 != 0 and rep == 9000
    bad = a.copy()
    bad[20] ^= 0x5A  # stored checksum
    outs, ok, osz, rc = R.ans_decode_batch([bad], [9000], 10, True)
    assert rc == 1  # checksum mismatch reported for batch member 0


@pytest.mark.parametrize("ft", [O.FLOAT16, O.BFLOAT16, O.FLOAT32])
@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_float_archives(ft, prob_bits):
    # FloatTest.cu:270-311 shapes, generator FloatTest.cu:110-120
    for n in (1, 7, 8, 15, 16, 17, 4096, 4097, 12345, 70001):
        w = refgen.generate_floats(ft, n)
        for ck in (False, True):
            ref = R.float_compress_batch(ft, [w], prob_bits, ck)[0]
            ora = O.float_compress(ft, w, prob_bits, use_checksum=ck)
            assert same(ref, ora, mask_float), (n, ck)
    w = refgen.generate_floats(ft, 33333)
    ora = O.float_compress(ft, w, prob_bits, use_checksum=True)
    outs, ok, osz, rc = R.float_decompress_batch(ft, [ora], [w.size], prob_bits, True)
    assert ok[0] == 1 and osz[0] == w.size and rc == 0 and (outs[0] == w).all()
    ref = R.float_compress_batch(ft, [w], prob_bits, True)[0]
    rc2, dec, rep = O.float_decompress(ft, ref, prob_bits)
    assert rc2 == 0 and rep == w.size and (dec == w).all()


def test_float_batch_and_incompressible_exponents():
    rng = np.random.default_rng(8)
    for ft in (O.FLOAT16, O.BFLOAT16, O.FLOAT32):
        bits = 32 if ft == O.FLOAT32 else 16
        ws = [rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(np.uint32 if bits == 32 else np.uint16)
              for n in (4096 * 2 + 5, 0, 100)]
        refs = R.float_compress_batch(ft, ws, 10, False)
        for w, r in zip(ws, refs):
            assert same(r, O.float_compress(ft, w, 10), mask_float), (ft, w.size)


def test_baseline_config_rows():
    # one row of each BASELINE config, full size
    x = refgen.zipf_bytes(1, 1 << 20)[0]
    assert same(R.ans_encode_batch([x], 10)[0], O.ans_encode(x, 10), mask_ans)
    w = refgen.normal_bf16(1, 512 * 1024)[0]
    assert same(R.float_compress_batch(O.BFLOAT16, [w], 10)[0], O.float_compress(O.BFLOAT16, w, 10), mask_float)
    h = refgen.sparse_fp16(1, 512 * 1024)[0]
    assert same(R.float_compress_batch(O.FLOAT16, [h], 11)[0], O.float_compress(O.FLOAT16, h, 11), mask_float)


# ---------------------------------------------------------------------------------------------------------
# Round 3: the reference paths the pin did not execute before ("Reference paths the pin never executes")
@pytest.mark.parametrize("ft", [O.FLOAT16, O.BFLOAT16, O.FLOAT32])
def test_float_aligned16_paths_of_the_reference(ft):
    # is16ByteAligned = true selects SplitFloatAligned16 / JoinFloatAligned16 (GpuFloatCompress.cuh:85-278,
    # GpuFloatDecompress.cuh:25-270) -- the path DietGpu.cpp takes for 16-byte aligned tensors.  Same archives as the
    # unaligned path and as the oracle; the reference's aligned JOIN decodes oracle archives.
    for n, p in ((4096 * 3 + 8, 10), (16, 9), (8 * 1000 + 8, 11), (4096 * 2 + 5, 10), (33, 10), (0, 10)):
        w = refgen.generate_floats(ft, n)
        ora = O.float_compress(ft, w, p)
        a = R.float_compress_batch(ft, [w], p, False, aligned16=True)[0]
        u = R.float_compress_batch(ft, [w], p, False, aligned16=False)[0]
        assert same(a, ora, mask_float), (ft, n, p)
        assert same(a, u, mask_float)
        outs, ok, osz, rc = R.float_decompress_batch(ft, [ora], [n], p, False, aligned16=True)
        assert ok[0] == 1 and osz[0] == n and rc == 0 and (outs[0] == w).all()
    # with the checksum (its float quirk: the word count is used as a byte count) and in a batch
    ws = [refgen.generate_floats(ft, n) for n in (4096 + 16, 8, 4096 * 5)]
    refs = R.float_compress_batch(ft, ws, 10, True, aligned16=True)
    for w, r in zip(ws, refs):
        assert same(r, O.float_compress(ft, w, 10, use_checksum=True), mask_float)
    outs, ok, osz, rc = R.float_decompress_batch(ft, refs, [w.size for w in ws], 10, True, aligned16=True)
    assert rc == 0 and ok.all() and all((o == w).all() for o, w in zip(outs, ws))


def test_two_level_prefix_sum_of_the_reference():
    # more than 512 blocks per element: batchExclusivePrefixSum takes its two-level path
    # (BatchPrefixSum.cuh:69-110).  One raw element of 3 MiB + 5 bytes (769 blocks) and one bf16 tensor of
    # 2.6 M words (635 blocks of exponents), byte for byte.
    x = refgen.generate_symbols(3 * (1 << 20) + 5, 40.0)
    r = R.ans_encode_batch([x], 10)[0]
    assert R.ans_header_fields(r)["num_blocks"] == 769
    assert same(r, O.ans_encode(x, 10), mask_ans)
    outs, ok, osz, rc = R.ans_decode_batch([O.ans_encode(x, 10)], [x.size], 10)
    assert ok[0] == 1 and (outs[0] == x).all()
    w = refgen.normal_bf16(1, 2_600_000)[0]
    assert same(R.float_compress_batch(O.BFLOAT16, [w], 10, False, aligned16=True)[0], O.float_compress(O.BFLOAT16, w, 10), mask_float)


def test_baseline_config_rows_eight_each():
    # eight rows (not one) of each BASELINE config through the reference, full size, as ONE batch
    xs = list(refgen.zipf_bytes(8, 1 << 20))
    for x, r in zip(xs, R.ans_encode_batch(xs, 10)):
        assert same(r, O.ans_encode(x, 10), mask_ans)
    ws = list(refgen.normal_bf16(8, 512 * 1024))
    for w, r in zip(ws, R.float_compress_batch(O.BFLOAT16, ws, 10, False, aligned16=True)):
        assert same(r, O.float_compress(O.BFLOAT16, w, 10), mask_float)
    hs = list(refgen.sparse_fp16(8, 512 * 1024))
    for h, r in zip(hs, R.float_compress_batch(O.FLOAT16, hs, 11, False, aligned16=True)):
        assert same(r, O.float_compress(O.FLOAT16, h, 11), mask_float)


def test_stride_and_split_size_providers_of_the_reference():
    # ansEncodeBatchStride / ansDecodeBatchStride (GpuANSEncode.cu:27-53, GpuANSDecode.cu:20-45) and
    # ansEncodeBatchSplitSize / ansDecodeBatchSplitSize (GpuANSEncode.cu:115-179, GpuANSDecode.cu:122-193), and the
    # float split-size pair (GpuFloatCompress.cu:103-159, GpuFloatDecompress.cu:117-179): the same archives as the
    # pointer API and the oracle, and they decode
    rng = np.random.default_rng(3)
    m = np.stack([refgen.generate_symbols(4096 * 2 + 100, lam) for lam in (5.0, 20.0, 100.0)])
    for stride in (None, 4096 * 3):
        refs = R.ans_encode_batch_stride(m, 10, True, stride)
        for row, r in zip(m, refs):
            assert same(r, O.ans_encode(row, 10, use_checksum=True), mask_ans)
    out, ok, osz, rc = R.ans_decode_batch_stride([O.ans_encode(row, 10, use_checksum=True) for row in m], m.shape[1], 10, True)
    assert rc == 0 and ok.all() and (osz == m.shape[1]).all() and (out == m).all()

    rows = [rng.integers(0, 30, n, dtype=np.uint8) for n in (4096 * 2, 100, 4096 + 4, 8)]  # 4-byte aligned starts
    refs = R.ans_encode_batch_split_size(rows, 11)
    for row, r in zip(rows, refs):
        assert same(r, O.ans_encode(row, 11), mask_ans)
    outs, ok, osz, rc = R.ans_decode_batch_split_size([O.ans_encode(r, 11) for r in rows], [len(r) for r in rows], 11)
    assert rc == 0 and ok.all() and all((o == r).all() for o, r in zip(outs, rows))

    for ft in (O.FLOAT16, O.BFLOAT16, O.FLOAT32):
        ws = [refgen.generate_floats(ft, n) for n in (4096 + 8, 16, 4096 * 2)]
        refs = R.float_compress_split_size(ft, ws, 10)
        for w, r in zip(ws, refs):
            assert same(r, O.float_compress(ft, w, 10), mask_float), ft
        outs, ok, osz, rc = R.float_decompress_split_size(ft, [O.float_compress(ft, w, 10) for w in ws], [w.size for w in ws], 10)
        assert rc == 0 and ok.all() and all((o == w).all() for o, w in zip(outs, ws))


def _hist_variants(x):
    """Caller-supplied histograms that COVER the data (every present symbol has a non-zero count): the exact counts,
    the counts x 3 (the normalisation's deficit branch: the quantised sum comes out near 3 x 2^P), the counts + 1 on
    every bin (symbols the data does not hold get probability mass), a flat histogram."""
    c = np.bincount(x, minlength=256).astype(np.uint32)
    return {"exact": c, "x3": c * 3, "plus1": c + 1, "flat": np.full(256, 7, np.uint32)}


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_caller_supplied_histogram_of_the_reference(prob_bits):
    # `histogram_dev` of ansEncodeBatch{Pointer,Stride,SplitSize} (GpuANSCodec.h:65-164, GpuANSEncode.cuh:692-700): the
    # reference normalises the counts it is GIVEN against the element's size.  The oracle's `counts=` does the same:
    # byte-identical archives for every covering histogram, through all three batch providers, and they decode.
    rows = [refgen.generate_symbols(n, lam) for n, lam in ((4096 * 3 + 100, 10.0), (4096 * 3 + 100, 100.0), (4096 * 3 + 100, 3.0))]
    for name in ("exact", "x3", "plus1", "flat"):
        counts = np.stack([_hist_variants(r)[name] for r in rows])
        want = [O.ans_encode(r, prob_bits, use_checksum=True, counts=c) for r, c in zip(rows, counts)]
        for provider in ("pointer", "stride", "split_size"):
            got = R.ans_encode_batch_hist(rows, counts, prob_bits, True, provider)
            for g, w in zip(got, want):
                assert same(g, w, mask_ans), (name, provider)
        if name == "exact":
            for r, w in zip(rows, want):
                assert (w == O.ans_encode(r, prob_bits, use_checksum=True)).all()
        outs, ok, osz, rc = R.ans_decode_batch(want, [len(r) for r in rows], prob_bits, True)
        assert rc == 0 and ok.all() and all((o == r).all() for o, r in zip(outs, rows)), name
    # ragged batch through the pointer and split-size providers (interior sizes multiples of 4), one empty element
    rows = [refgen.generate_symbols(n, 20.0) for n in (4096 + 8, 0, 100, 4096 * 2)]
    counts = np.stack([np.bincount(r, minlength=256) + 1 for r in rows]).astype(np.uint32)
    want = [O.ans_encode(r, prob_bits, counts=c) for r, c in zip(rows, counts)]
    for provider in ("pointer", "split_size"):
        for g, w in zip(R.ans_encode_batch_hist(rows, counts, prob_bits, False, provider), want):
            assert same(g, w, mask_ans), provider


def test_compressed_info_entry_points_of_the_reference():
    # ansGetCompressedInfo(Device) / floatGetCompressedInfo(Device) (GpuANSCodec.h:309-341, GpuFloatCodec.h:252-292)
    # read what the oracle's info functions read, on oracle archives
    xs = [refgen.generate_symbols(n, 30.0) for n in (0, 1, 4096, 12345)]
    arch = [O.ans_encode(x, 10, use_checksum=True) for x in xs]
    for device in (False, True):
        sizes, ck = R.ans_get_compressed_info(arch, want_checksum=True, device=device)
        assert sizes.tolist() == [x.size for x in xs]
        assert ck.tolist() == [O.ans_info(a)["checksum"] for a in arch] == [O.checksum(x) for x in xs]
        sizes, ck = R.ans_get_compressed_info([O.ans_encode(x, 11) for x in xs], device=device)
        assert sizes.tolist() == [x.size for x in xs] and ck is None
    for ft in (O.FLOAT16, O.BFLOAT16, O.FLOAT32):
        ws = [refgen.generate_floats(ft, n) for n in (0, 5, 4096 + 3)]
        arch = [O.float_compress(ft, w, 10, use_checksum=True) for w in ws]
        for device in (False, True):
            sizes, types, ck = R.float_get_compressed_info(arch, want_checksum=True, device=device)
            assert sizes.tolist() == [w.size for w in ws] and types.tolist() == [ft] * 3
            assert ck.tolist() == [O.float_info(a)["checksum"] for a in arch]
