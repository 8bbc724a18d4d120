"""GPU tests that call the C ABI (include/dietgpu_amd.h) DIRECTLY, for the reference parameters and entry points the
tensor surfaces never exercise:

  * `histogram_dev` of ansEncodeBatch{Pointer,Stride,SplitSize} (GpuANSCodec.h:65-164; the encoder then normalises
    the counts it is given instead of counting, GpuANSEncode.cuh:692-700) -- archives byte for byte against the oracle
    AND against the reference itself (oracle/_ref), which tests/test_reference_pin.py pins to each other on the CPU;
  * a histogram that does NOT cover the data: the element fails cleanly (outSize 0, undecodable header), its
    neighbours in the batch are untouched;
  * ansDecodeBatchSplitSize / floatDecompressSplitSize in their un-bounded form (GpuANSCodec.h:265-304,
    GpuFloatCodec.h:220-258);
  * ansGetCompressedInfoDevice / floatGetCompressedInfoDevice and their host-array forms (GpuANSCodec.h:309-341,
    GpuFloatCodec.h:252-292).
Nothing here reads /root/reference."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle as O
import refgen
from oracle import ref as R
from refmask import mask_ans

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def L():
    import dietgpu_amd

    return dietgpu_amd.lib()  # fails loudly if the HIP extension is missing


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(DEV)


def _ptrs(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _u32s(vals):
    return (C.c_uint32 * len(vals))(*[int(v) for v in vals])


def _ok(L, rc):
    assert rc == 0, L.dgpu_last_error().decode()


def _hist_variants(x):
    c = np.bincount(x, minlength=256).astype(np.uint32)
    return {"exact": c, "x3": c * 3, "plus1": c + 1, "flat": np.full(256, 7, np.uint32)}


def _encode_with_histogram(L, rows, counts, prob_bits, checksum, provider, temp=True):
    """-> (list of archives, outSize array) through dgpu_ans_encode_batch_{pointer,stride,split_size}."""
    b = len(rows)
    sizes = [len(r) for r in rows]
    hist = torch.from_numpy(np.ascontiguousarray(counts, np.uint32).view(np.int32).reshape(b, 256).copy()).to(DEV)
    stride = int(L.dgpu_ans_max_compressed_size(max(max(sizes), 1)))
    out = torch.zeros((b, stride), dtype=torch.uint8, device=DEV)
    out_sizes = torch.full((b,), -1, dtype=torch.int32, device=DEV)
    tb = int(L.dgpu_ans_encode_temp_bytes(b, max(sizes)))
    tmp = torch.empty((tb,), dtype=torch.uint8, device=DEV) if temp else None
    targs = (C.c_void_p(tmp.data_ptr()), tb, None) if temp else (None, 0, None)
    if provider == "pointer":
        ins = [_dev(r if len(r) else np.zeros(4, np.uint8)) for r in rows]
        outs = [out[i] for i in range(b)]
        _ok(L, L.dgpu_ans_encode_batch_pointer(*targs, prob_bits, int(checksum), b, _ptrs(ins), _u32s(sizes), C.c_void_p(hist.data_ptr()),
                                               _ptrs(outs), C.c_void_p(out_sizes.data_ptr()), _stream()))
    elif provider == "stride":
        n = sizes[0]
        assert all(s == n for s in sizes)
        flat = _dev(np.concatenate(rows))
        _ok(L, L.dgpu_ans_encode_batch_stride(*targs, prob_bits, int(checksum), b, C.c_void_p(flat.data_ptr()), n, n,
                                              C.c_void_p(hist.data_ptr()), C.c_void_p(out.data_ptr()), stride,
                                              C.c_void_p(out_sizes.data_ptr()), _stream()))
    else:
        flat = _dev(np.concatenate(rows + [np.zeros(16, np.uint8)]))
        _ok(L, L.dgpu_ans_encode_batch_split_size(*targs, prob_bits, int(checksum), b, C.c_void_p(flat.data_ptr()), _u32s(sizes),
                                                  C.c_void_p(hist.data_ptr()), C.c_void_p(out.data_ptr()), stride,
                                                  C.c_void_p(out_sizes.data_ptr()), _stream()))
    torch.cuda.synchronize()
    osz = out_sizes.cpu().numpy()
    host = out.cpu().numpy()
    return [host[i, : max(int(osz[i]), 0)].copy() for i in range(b)], osz


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
@pytest.mark.parametrize("provider", ["pointer", "stride", "split_size"])
def test_caller_supplied_histogram(L, prob_bits, provider):
    rows = [refgen.generate_symbols(4096 * 3 + 100, lam) for lam in (10.0, 100.0, 3.0)]
    for name in ("exact", "x3", "plus1", "flat"):
        counts = np.stack([_hist_variants(r)[name] for r in rows])
        got, osz = _encode_with_histogram(L, rows, counts, prob_bits, True, provider)
        want = [O.ans_encode(r, prob_bits, use_checksum=True, counts=c) for r, c in zip(rows, counts)]
        for g, w in zip(got, want):
            assert g.size == w.size and (g == w).all(), (name, provider)
        if R.available():  # the reference itself, given the same histogram
            for g, r in zip(got, R.ans_encode_batch_hist(rows, counts, prob_bits, True, provider)):
                assert g.size == r.size and not (mask_ans(g) != mask_ans(r)).any(), (name, provider)
        # ... and they decode (the oracle and, below, the HIP decoder)
        for g, r in zip(got, rows):
            rc, y, _ = O.ans_decode(g, prob_bits)
            assert rc == 0 and (y == r).all()


def test_caller_supplied_histogram_large_and_ragged(L):
    # many tiles per element (look-back across tiles on the k_normalize -> k_ans_encode path), a ragged pointer
    # batch with an empty element and single-block elements (k_ans_encode_pair), library-owned temp memory
    rng = np.random.default_rng(5)
    rows = [(rng.zipf(1.3, n) % 256).astype(np.uint8) for n in (4096 * 70 + 5, 4096 * 9)]
    counts = np.stack([np.bincount(r, minlength=256) + 1 for r in rows]).astype(np.uint32)
    got, _ = _encode_with_histogram(L, rows, counts, 10, False, "pointer", temp=False)
    for g, r, c in zip(got, rows, counts):
        w = O.ans_encode(r, 10, counts=c)
        assert g.size == w.size and (g == w).all()
    rows = [refgen.generate_symbols(n, 20.0) for n in (4096, 0, 100, 4092)]
    counts = np.stack([np.bincount(r, minlength=256) * 2 + 1 for r in rows]).astype(np.uint32)
    for provider in ("pointer", "split_size"):
        got, _ = _encode_with_histogram(L, rows, counts, 11, True, provider)
        for g, r, c in zip(got, rows, counts):
            w = O.ans_encode(r, 11, use_checksum=True, counts=c)
            assert g.size == w.size and (g == w).all(), provider


@pytest.mark.parametrize("n", [4096 * 20 + 7, 4096, 3000])
def test_histogram_that_does_not_cover_the_data_fails_cleanly(L, n):
    # A present symbol with count 0 gets probability 0: every row of it emits, and a block of them overruns any
    # stage.  Upstream's behaviour is undefined there (its per-block scratch is overrun, GpuANSEncode.cuh:355-358);
    # here the ELEMENT is reported as failed -- outSize 0, a header no decoder accepts -- and its neighbours are fine.
    rng = np.random.default_rng(n)
    good = refgen.generate_symbols(n, 50.0)
    bad = rng.integers(0, 256, n, dtype=np.uint8)
    rows = [good, bad, good[::-1].copy()]
    counts = np.stack([np.bincount(r, minlength=256) for r in rows]).astype(np.uint32)
    counts[1] = 0
    counts[1][0] = n  # "all zeros": 255 of the 256 symbols of `bad` are not covered
    got, osz = _encode_with_histogram(L, rows, counts, 10, False, "pointer")
    assert osz[1] == 0
    for i in (0, 2):
        w = O.ans_encode(rows[i], 10, counts=counts[i])
        assert got[i].size == w.size and (got[i] == w).all()
    # the failed element's buffer does not decode: status 0 from the HIP decoder
    stride = int(L.dgpu_ans_max_compressed_size(n))
    arch = torch.zeros((stride,), dtype=torch.uint8, device=DEV)
    hist = torch.from_numpy(counts[1:2].view(np.int32).copy()).to(DEV)
    x = _dev(bad)
    osz1 = torch.zeros((1,), dtype=torch.int32, device=DEV)
    _ok(L, L.dgpu_ans_encode_batch_stride(None, 0, None, 10, 0, 1, C.c_void_p(x.data_ptr()), n, n, C.c_void_p(hist.data_ptr()),
                                          C.c_void_p(arch.data_ptr()), stride, C.c_void_p(osz1.data_ptr()), _stream()))
    out = torch.zeros((n,), dtype=torch.uint8, device=DEV)
    st = torch.full((1,), 7, dtype=torch.uint8, device=DEV)
    _ok(L, L.dgpu_ans_decode_batch_stride(None, 0, None, 10, 0, 1, C.c_void_p(arch.data_ptr()), stride, C.c_void_p(out.data_ptr()), n, n,
                                          C.c_void_p(st.data_ptr()), None, _stream(), None))
    torch.cuda.synchronize()
    assert int(osz1.item()) == 0 and int(st.item()) == 0


# ---------------------------------------------------------------------------------------------------------------
def test_ans_decode_batch_split_size_unbounded(L):
    # ansDecodeBatchSplitSize (GpuANSCodec.h:265-304): ONE output buffer, element i at the prefix sum of the split
    # sizes; interior sizes are multiples of 4 (kANSRequiredAlignment); with and without checksum verification
    rows = [refgen.generate_symbols(n, 15.0) for n in (4096 * 2 + 4, 8, 0, 4096, 12345)]
    sizes = [len(r) for r in rows]
    for use_ck in (0, 1):
        arch = [_dev(O.ans_encode(r, 10, use_checksum=bool(use_ck))) for r in rows]
        out = torch.full((sum(sizes) + 16,), 0xCD, dtype=torch.uint8, device=DEV)
        st = torch.zeros((len(rows),), dtype=torch.uint8, device=DEV)
        osz = torch.zeros((len(rows),), dtype=torch.int32, device=DEV)
        err = C.c_int32(-2)
        _ok(L, L.dgpu_ans_decode_batch_split_size(None, 0, None, 10, use_ck, len(rows), _ptrs(arch), C.c_void_p(out.data_ptr()),
                                                  _u32s(sizes), C.c_void_p(st.data_ptr()), C.c_void_p(osz.data_ptr()), _stream(),
                                                  C.byref(err)))
        torch.cuda.synchronize()
        assert st.cpu().numpy().all() and osz.cpu().numpy().tolist() == sizes and err.value == -1
        host = out.cpu().numpy()
        assert (host[: sum(sizes)] == np.concatenate(rows)).all() and (host[sum(sizes) :] == 0xCD).all()
    # a split that is too small for its archive: that member reports status 0 and the size it needs, the others decode
    small = list(sizes)
    small[3] = 4092
    out = torch.zeros((sum(small) + 16,), dtype=torch.uint8, device=DEV)
    st = torch.zeros((len(rows),), dtype=torch.uint8, device=DEV)
    osz = torch.zeros((len(rows),), dtype=torch.int32, device=DEV)
    arch = [_dev(O.ans_encode(r, 10)) for r in rows]
    _ok(L, L.dgpu_ans_decode_batch_split_size(None, 0, None, 10, 0, len(rows), _ptrs(arch), C.c_void_p(out.data_ptr()), _u32s(small),
                                              C.c_void_p(st.data_ptr()), C.c_void_p(osz.data_ptr()), _stream(), None))
    torch.cuda.synchronize()
    assert st.cpu().numpy().tolist() == [1, 1, 1, 0, 1] and osz.cpu().numpy().tolist() == sizes
    if R.available():  # the reference's own split-size decoder agrees on the same archives
        outs, ok, rsz, rc = R.ans_decode_batch_split_size([O.ans_encode(r, 10) for r in rows], sizes, 10)
        assert rc == 0 and ok.all() and rsz.tolist() == sizes


@pytest.mark.parametrize("ft", [O.FLOAT16, O.BFLOAT16, O.FLOAT32])
def test_float_decompress_split_size_unbounded(L, ft):
    # floatDecompressSplitSize (GpuFloatCodec.h:220-258): sizes in float words
    ws = [refgen.generate_floats(ft, n) for n in (4096 * 2 + 8, 16, 0, 4096 + 3)]
    sizes = [w.size for w in ws]
    wb = 4 if ft == O.FLOAT32 else 2
    for use_ck in (0, 1):
        arch = [_dev(O.float_compress(ft, w, 10, use_checksum=bool(use_ck))) for w in ws]
        out = torch.full(((sum(sizes) + 8) * wb,), 0xCD, dtype=torch.uint8, device=DEV)
        st = torch.zeros((len(ws),), dtype=torch.uint8, device=DEV)
        osz = torch.zeros((len(ws),), dtype=torch.int32, device=DEV)
        err = C.c_int32(-2)
        _ok(L, L.dgpu_float_decompress_split_size(None, 0, None, ft, 10, use_ck, len(ws), _ptrs(arch), C.c_void_p(out.data_ptr()),
                                                  _u32s(sizes), C.c_void_p(st.data_ptr()), C.c_void_p(osz.data_ptr()), _stream(),
                                                  C.byref(err)))
        torch.cuda.synchronize()
        assert st.cpu().numpy().all() and osz.cpu().numpy().tolist() == sizes and err.value == -1
        host = out.cpu().numpy()
        want = np.concatenate([w.view(np.uint8) for w in ws])
        assert (host[: want.size] == want).all() and (host[want.size :] == 0xCD).all()
    # a corrupted member with checksums on: its index comes back, the status is the checksum error
    bad = O.float_compress(ft, ws[0], 10, use_checksum=True).copy()
    bad[12] ^= 0x5A  # the checksum byte of the float header (GpuFloatUtils.cuh:61-72)
    arch = [_dev(bad)] + [_dev(O.float_compress(ft, w, 10, use_checksum=True)) for w in ws[1:]]
    out = torch.zeros(((sum(sizes) + 8) * wb,), dtype=torch.uint8, device=DEV)
    err = C.c_int32(-2)
    rc = L.dgpu_float_decompress_split_size(None, 0, None, ft, 10, 1, len(ws), _ptrs(arch), C.c_void_p(out.data_ptr()), _u32s(sizes),
                                            None, None, _stream(), C.byref(err))
    assert rc != 0 and err.value == 0 and b"Checksum mismatch in batch member 0" in L.dgpu_last_error()


def test_compressed_info_device_entry_points(L):
    # ans / float GetCompressedInfo and GetCompressedInfoDevice: the latter takes a DEVICE array of archive addresses
    xs = [refgen.generate_symbols(n, 30.0) for n in (0, 1, 4096, 12345)]
    arch = [_dev(O.ans_encode(x, 10, use_checksum=True)) for x in xs]
    addr = torch.tensor([a.data_ptr() for a in arch], dtype=torch.int64, device=DEV)
    for device_form in (True, False):
        sizes = torch.full((len(xs),), -1, dtype=torch.int32, device=DEV)
        ck = torch.full((len(xs),), -1, dtype=torch.int32, device=DEV)
        if device_form:
            _ok(L, L.dgpu_ans_get_compressed_info_device(C.c_void_p(addr.data_ptr()), len(xs), C.c_void_p(sizes.data_ptr()),
                                                         C.c_void_p(ck.data_ptr()), _stream()))
        else:
            _ok(L, L.dgpu_ans_get_compressed_info(None, 0, _ptrs(arch), len(xs), C.c_void_p(sizes.data_ptr()), C.c_void_p(ck.data_ptr()),
                                                  _stream()))
        torch.cuda.synchronize()
        assert sizes.cpu().numpy().tolist() == [x.size for x in xs]
        assert ck.cpu().numpy().tolist() == [O.checksum(x) for x in xs]
        # either output may be null (GpuANSInfo.cuh:25-35)
        sizes.fill_(-1)
        _ok(L, L.dgpu_ans_get_compressed_info_device(C.c_void_p(addr.data_ptr()), len(xs), C.c_void_p(sizes.data_ptr()), None, _stream()))
        _ok(L, L.dgpu_ans_get_compressed_info_device(C.c_void_p(addr.data_ptr()), len(xs), None, None, _stream()))
        torch.cuda.synchronize()
        assert sizes.cpu().numpy().tolist() == [x.size for x in xs]
    if R.available():
        rs, rck = R.ans_get_compressed_info([a.cpu().numpy() for a in arch], want_checksum=True, device=True)
        assert rs.tolist() == [x.size for x in xs] and rck.tolist() == [O.checksum(x) for x in xs]

    for ft in (O.FLOAT16, O.BFLOAT16, O.FLOAT32):
        ws = [refgen.generate_floats(ft, n) for n in (0, 5, 4096 + 3)]
        host_arch = [O.float_compress(ft, w, 10, use_checksum=True) for w in ws]
        arch = [_dev(a) for a in host_arch]
        addr = torch.tensor([a.data_ptr() for a in arch], dtype=torch.int64, device=DEV)
        for device_form in (True, False):
            sizes = torch.full((len(ws),), -1, dtype=torch.int32, device=DEV)
            types = torch.full((len(ws),), -1, dtype=torch.int32, device=DEV)
            ck = torch.full((len(ws),), -1, dtype=torch.int32, device=DEV)
            if device_form:
                _ok(L, L.dgpu_float_get_compressed_info_device(C.c_void_p(addr.data_ptr()), len(ws), C.c_void_p(sizes.data_ptr()),
                                                               C.c_void_p(types.data_ptr()), C.c_void_p(ck.data_ptr()), _stream()))
            else:
                _ok(L, L.dgpu_float_get_compressed_info(None, 0, _ptrs(arch), len(ws), C.c_void_p(sizes.data_ptr()),
                                                        C.c_void_p(types.data_ptr()), C.c_void_p(ck.data_ptr()), _stream()))
            torch.cuda.synchronize()
            assert sizes.cpu().numpy().tolist() == [w.size for w in ws]
            assert types.cpu().numpy().tolist() == [ft] * len(ws)
            assert ck.cpu().numpy().tolist() == [O.float_info(a)["checksum"] for a in host_arch]


def test_unbounded_pointer_decoders_and_temp_queries(L):
    # dgpu_ans_decode_batch_pointer / dgpu_float_decompress: the reference's signatures proper (no compressed sizes;
    # the tensor surfaces use the *_bounded forms), with temp memory sized by the library's own query
    assert b"gfx950" in L.dgpu_version()
    rows = [refgen.generate_symbols(n, 25.0) for n in (4096 * 5 + 1, 17, 4096)]
    arch = [_dev(O.ans_encode(r, 11, use_checksum=True)) for r in rows]
    outs = [torch.zeros((len(r),), dtype=torch.uint8, device=DEV) for r in rows]
    st = torch.zeros((3,), dtype=torch.uint8, device=DEV)
    osz = torch.zeros((3,), dtype=torch.int32, device=DEV)
    tb = int(L.dgpu_ans_decode_temp_bytes(3, max(len(r) for r in rows), 11))
    tmp = torch.empty((tb,), dtype=torch.uint8, device=DEV)
    used = C.c_size_t(0)
    err = C.c_int32(-2)
    _ok(L, L.dgpu_ans_decode_batch_pointer(C.c_void_p(tmp.data_ptr()), tb, C.byref(used), 11, 1, 3, _ptrs(arch), _ptrs(outs),
                                           _u32s([len(r) for r in rows]), C.c_void_p(st.data_ptr()), C.c_void_p(osz.data_ptr()), _stream(),
                                           C.byref(err)))
    torch.cuda.synchronize()
    assert st.cpu().numpy().all() and err.value == -1 and 0 < used.value <= tb
    for o, r in zip(outs, rows):
        assert (o.cpu().numpy() == r).all()
    for ft in (O.FLOAT16, O.BFLOAT16, O.FLOAT32):
        ws = [refgen.generate_floats(ft, n) for n in (4096 * 3 + 5, 7)]
        arch = [_dev(O.float_compress(ft, w, 9)) for w in ws]
        wb = 4 if ft == O.FLOAT32 else 2
        outs = [torch.zeros((w.size * wb,), dtype=torch.uint8, device=DEV) for w in ws]
        tb = int(L.dgpu_float_decompress_temp_bytes(ft, 2, max(w.size for w in ws), 9))
        tmp = torch.empty((tb,), dtype=torch.uint8, device=DEV)
        _ok(L, L.dgpu_float_decompress(C.c_void_p(tmp.data_ptr()), tb, None, ft, 9, 0, 2, _ptrs(arch), _ptrs(outs), _u32s([w.size for w in ws]),
                                       C.c_void_p(st.data_ptr()), C.c_void_p(osz.data_ptr()), _stream(), None))
        torch.cuda.synchronize()
        assert st.cpu().numpy()[:2].all() and osz.cpu().numpy()[:2].tolist() == [w.size for w in ws]
        for o, w in zip(outs, ws):
            assert (o.cpu().numpy() == w.view(np.uint8)).all()


def test_measurement_and_state_hooks(L):
    # dgpu_prof_* (per-kernel HIP-event timing, what bench.py's roofline figure reads), dgpu_debug_set_param_cache,
    # dgpu_release_stream_state: they do what include/dietgpu_amd.h says and leave the codec's results alone
    import json

    rows = [refgen.generate_symbols(4096 * 4, 40.0), refgen.generate_symbols(4096 * 2 + 3, 40.0)]  # ragged: a real pointer list
    counts = np.stack([np.bincount(r, minlength=256) for r in rows]).astype(np.uint32)
    L.dgpu_prof_reset()
    L.dgpu_prof_enable(1)
    try:
        L.dgpu_debug_set_param_cache(0)
        a0, _ = _encode_with_histogram(L, rows, counts, 10, False, "pointer")
        L.dgpu_debug_set_param_cache(1)
        a1, _ = _encode_with_histogram(L, rows, counts, 10, False, "pointer")
    finally:
        L.dgpu_debug_set_param_cache(1)
        L.dgpu_prof_enable(0)
    buf = C.create_string_buffer(4096)
    n = L.dgpu_prof_summary(buf, 4096)
    assert n > 0
    prof = json.loads(buf.value.decode())
    assert prof["k_normalize"]["launches"] == 2 and prof["k_ans_encode"]["launches"] == 2 and prof["k_ans_encode"]["total_ms"] > 0
    assert "k_histogram" not in prof  # a caller's histogram: nothing is counted
    L.dgpu_prof_reset()
    for x, y, r, c in zip(a0, a1, rows, counts):
        w = O.ans_encode(r, 10, counts=c)
        assert (x == w).all() and (y == w).all()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        _encode_with_histogram(L, rows, counts, 10, False, "pointer", temp=False)  # library-owned temp memory on stream s
    before = L.dgpu_debug_stream_state_count()
    assert L.dgpu_release_stream_state(C.c_void_p(s.cuda_stream)) == 1
    assert L.dgpu_debug_stream_state_count() == before - 1
