"""torch.ops.dietgpu.* (dietgpu_amd/csrc/torch_ops.cpp): the reference's own
Python tests (dietgpu/ans_test.py, dietgpu/float_test.py) restated against the
ROCm build -- same op names, argument order and checks."""
import random

import pytest
import torch


@pytest.fixture(scope="module")
def ops():
    import dietgpu_amd

    return dietgpu_amd.load_torch_ops()


def test_schema_is_the_reference_schema(ops):
    # DietGpu.cpp:915-937
    s = str(torch.ops.dietgpu.compress_data.default._schema)
    assert "bool compress_as_float, Tensor[] ts_in, bool checksum=False, Tensor? temp_mem=None" in s
    assert "-> (Tensor, Tensor, int)" in s
    s = str(torch.ops.dietgpu.decompress_data.default._schema)
    assert "Tensor[] ts_in, Tensor[] ts_out, bool checksum=False, Tensor? temp_mem=None, Tensor? out_status=None" in s
    assert "temp_mem=67108864" in str(torch.ops.dietgpu.compress_data_simple.default._schema)
    assert ops.max_any_compressed_size(1 << 20) == 1868320
    assert ops.max_float_compressed_size(torch.empty(0, dtype=torch.bfloat16), 524288) == 1737264
    with pytest.raises(RuntimeError):
        ops.compress_data(False, [torch.zeros(16, dtype=torch.uint8)])  # CPU tensor


def _run_codec(ops, as_float, dev, ts, checksum, temp_mem=None):
    comp, sizes, _ = ops.compress_data(as_float, ts, checksum, temp_mem)
    # truncate to exactly the reported sizes (ans_test.py:21-26)
    trunc = [t.narrow(0, 0, s.item()).clone() for s, t in zip(sizes, [*comp])]
    outs = [torch.empty(t.size(), dtype=t.dtype, device=t.device) for t in ts]
    if temp_mem is not None:
        st = torch.empty([len(ts)], dtype=torch.uint8, device=dev)
        sz = torch.empty([len(ts)], dtype=torch.int32, device=dev)
        ops.decompress_data(as_float, trunc, outs, checksum, temp_mem, st, sz)
        for t, s, z in zip(ts, st, sz):
            assert s.item()
            assert z.item() == (t.numel() if as_float else t.numel() * t.element_size())
    else:
        ops.decompress_data(as_float, trunc, outs, checksum)
    for a, b in zip(ts, outs):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_ans_codec(ops):
    dev = torch.device("cuda:0")
    temp_mem = torch.empty([64 * 1024 * 1024], dtype=torch.uint8, device=dev)
    for tm in (False, True):
        for checksum in (False, True):
            ts = [torch.normal(0, 1.0, [n], dtype=torch.float32, device=dev) for n in (10000, 100000, 1000000)]
            _run_codec(ops, False, dev, ts, checksum, temp_mem if tm else None)


@pytest.mark.gpu
def test_float_codec_and_large(ops):
    dev = torch.device("cuda:0")
    temp_mem = torch.empty([64 * 1024 * 1024], dtype=torch.uint8, device=dev)
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        for tm in (False, True):
            ts = [torch.normal(0, 1.0, [n], dtype=dt, device=dev) for n in (10000, 100000, 1000000)]
            _run_codec(ops, True, dev, ts, True, temp_mem if tm else None)
    # float_test.py:66-76
    ts = [torch.normal(0, 1.0, [123456789], dtype=torch.float16, device=dev)]
    _run_codec(ops, True, dev, ts, False)


@pytest.mark.gpu
def test_simple_and_empty(ops):
    dev = torch.device("cuda:0")
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        ts = [torch.normal(0, 1.0, [10000], dtype=dt, device=dev), torch.normal(0, 1.0, [100000], dtype=dt, device=dev)]
        comp = ops.compress_data_simple(True, ts, True)
        for c, t in zip(comp, ts):
            assert c.numel() < t.numel() * t.element_size()  # float_test.py:88-92
        for a, b in zip(ts, ops.decompress_data_simple(True, comp, True)):
            assert torch.equal(a, b)
        e = [torch.empty([0], dtype=dt, device=dev)]
        ce = ops.compress_data_simple(True, e, True)
        assert ce[0].numel() > 0
        assert torch.equal(e[0], ops.decompress_data_simple(True, ce, True)[0])
    e = [torch.empty([0], dtype=torch.uint8, device=dev)]
    ce = ops.compress_data_simple(False, e, True)
    assert ce[0].numel() > 0
    assert torch.equal(e[0], ops.decompress_data_simple(False, ce, True)[0])


@pytest.mark.gpu
def test_split_compress_and_decompress(ops):
    dev = torch.device("cuda:0")
    temp_mem = torch.empty([64 * 1024 * 1024], dtype=torch.uint8, device=dev)
    random.seed(7)
    for _ in range(3):
        sizes = []
        for _ in range(random.randrange(1, 15)):
            s = random.randrange(1, 10000)
            sizes.append(s + 4 - (s % 4))
        t = torch.randint(0, 65, [sum(sizes)], dtype=torch.uint8, device=dev)
        sizes_t = torch.IntTensor(sizes)
        splits = torch.split(t, sizes)
        comp_ts, _, _ = ops.compress_data_split_size(False, t, sizes_t, True, temp_mem)
        for a, b in zip(splits, ops.decompress_data_simple(False, comp_ts, True)):
            assert torch.equal(a, b)
        comp_ts = ops.compress_data_simple(False, splits, True)
        out = torch.empty([sum(sizes)], dtype=torch.uint8, device=dev)
        ops.decompress_data_split_size(False, comp_ts, out, sizes_t, True, temp_mem)
        assert torch.equal(t, out)
    for dt in (torch.bfloat16, torch.float32):
        sizes = [random.randrange(1, 10000) for _ in range(6)]
        t = torch.normal(0, 1.0, [sum(sizes)], dtype=dt, device=dev)
        sizes_t = torch.IntTensor(sizes)
        comp_ts, _, _ = ops.compress_data_split_size(True, t, sizes_t, True, temp_mem)
        out = torch.empty_like(t)
        ops.decompress_data_split_size(True, [c.clone() for c in comp_ts], out, sizes_t, True, temp_mem)
        assert torch.equal(t, out)


@pytest.mark.gpu
@pytest.mark.parametrize("prob_bits", [9, 11])
def test_package_ops_reach_the_op_library_at_every_precision(ops, prob_bits):
    # dietgpu_amd.compress_data(..., prob_bits=9 / 11) goes through torch.ops.dietgpu.* with the library's thread-local
    # precision set around the call (dietgpu_amd::set_precision), not through ctypes; archives equal the oracle's at that
    # precision, the default precision is back afterwards, and a bad value is refused.
    import numpy as np

    import dietgpu_amd as dg
    import oracle as O
    from dietgpu_amd import ops as pkg_ops

    dg.prefer_torch_ops(True)
    assert isinstance(pkg_ops._fast_ops(prob_bits), pkg_ops._OpsAtPrecision)
    rng = np.random.default_rng(prob_bits)
    words = (rng.standard_normal(3 * 4096 + 77).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    t = torch.from_numpy(words.view(np.int16)).cuda().view(torch.bfloat16)
    comp, sizes, _ = dg.compress_data(True, [t], prob_bits=prob_bits)
    n = int(sizes[0].item())
    want = O.float_compress(O.BFLOAT16, words, prob_bits)
    assert n == want.size and (comp[0, :n].cpu().numpy() == want).all()
    out = torch.empty_like(t)
    dg.decompress_data(True, [comp[0, :n]], [out], prob_bits=prob_bits)
    assert torch.equal(out.view(torch.int16), t.view(torch.int16))
    # the registered ops are back at the reference's precision
    c10, s10, _ = ops.compress_data(True, [t])
    want10 = O.float_compress(O.BFLOAT16, words, 10)
    assert int(s10[0].item()) == want10.size and (c10[0, : want10.size].cpu().numpy() == want10).all()
    with pytest.raises(RuntimeError, match="probBits"):
        torch.ops.dietgpu_amd.set_precision(12)
