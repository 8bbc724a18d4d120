"""The C ABI calls are pure stream work once warm (parameter block cached on the device, per-stream state created,
no malloc / memcpy / synchronise per call): a compress + decompress pair can be captured into a HIP graph and
replayed on new data.  (The reference's API cannot: StackDeviceMemory's overflow path and the per-call pointer
uploads are host work, DietGpu.cpp:246-270.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("as_float", [True, False])
def test_encode_decode_captured_in_a_hip_graph(as_float):
    import dietgpu_amd
    from dietgpu_amd import ops

    dietgpu_amd.lib()  # fails loudly if the HIP extension is missing
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, n = 12, 70_000
    g0 = torch.Generator(device="cpu").manual_seed(5)

    def fresh():
        if as_float:
            return [torch.randn(n + 8 * i, generator=g0).to(torch.bfloat16).to(dev) for i in range(B)]
        return [(torch.randn(n + 8 * i, generator=g0) * 20).to(torch.int8).view(torch.uint8).to(dev) for i in range(B)]

    xs = fresh()
    rows, cols = (ops.max_float_compressed_output_size(xs) if as_float else ops.max_any_compressed_output_size(xs))
    comp = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
    sizes = torch.zeros((rows,), dtype=torch.int32, device=dev)
    outs = [torch.empty_like(x) for x in xs]
    status = torch.zeros((B,), dtype=torch.uint8, device=dev)
    osz = torch.zeros((B,), dtype=torch.int32, device=dev)
    temp = torch.empty((128 << 20,), dtype=torch.uint8, device=dev)

    def roundtrip():
        ops.compress_data(as_float, xs, False, temp, comp, sizes)
        ops.decompress_data(as_float, [comp[i] for i in range(B)], outs, False, temp, status, osz)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):  # warm: uploads the parameter blocks, creates this stream's library state
            roundtrip()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        roundtrip()
    torch.cuda.synchronize()

    for trial in range(3):
        new = fresh()
        for x, y in zip(xs, new):
            x.copy_(y)
        for o in outs:
            o.zero_()
        status.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert bool(status.all())
        assert osz.cpu().tolist() == [x.numel() for x in xs]
        for x, o in zip(xs, outs):
            assert torch.equal(x.view(torch.uint8), o.view(torch.uint8)), trial
        # the archives of the replay are what a plain call produces
        ref_comp, ref_sizes, _ = ops.compress_data(as_float, xs, False, temp)
        torch.cuda.synchronize()
        assert torch.equal(ref_sizes, sizes)
        for i in range(B):
            k = int(sizes[i])
            assert torch.equal(ref_comp[i, :k], comp[i, :k])


def test_graph_survives_parameter_cache_churn_and_reports_unwarmed_capture():
    """a captured call bakes the device copy of its pointer / size arrays (and the stream's counters)
    into the graph; a replay never passes through the library, so nothing refreshes the cache entry.  The entry is
    pinned at capture time: 40 other distinct pointer-list calls (the cache holds 16 blocks) must not evict it.
    And a call whose arrays are NOT resident cannot be captured: it fails with a message that says so."""
    import dietgpu_amd
    from dietgpu_amd import ops

    L = dietgpu_amd.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g0 = torch.Generator(device="cpu").manual_seed(11)
    B, n = 6, 40_000
    # ragged sizes + a shuffled order: a genuine pointer list (no arithmetic progression), so the call needs its
    # parameter block on the device
    xs = [torch.randn(n + 24 * ((i * 5) % B), generator=g0).to(torch.bfloat16).to(dev) for i in range(B)]
    rows, cols = ops.max_float_compressed_output_size(xs)
    comp = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
    sizes = torch.zeros((rows,), dtype=torch.int32, device=dev)
    outs = [torch.empty_like(x) for x in xs]
    status = torch.zeros((B,), dtype=torch.uint8, device=dev)
    temp = torch.empty((128 << 20,), dtype=torch.uint8, device=dev)
    order = [3, 0, 5, 1, 4, 2]

    def roundtrip():
        ops.compress_data(True, [xs[i] for i in order], False, temp, comp, sizes)
        ops.decompress_data(True, [comp[k] for k in range(B)], [outs[i] for i in order], False, temp, status, None)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    # (1) capture WITHOUT warm-up: refused with a clear message, the capture is abandoned cleanly
    torch.cuda.synchronize()
    graph0 = torch.cuda.CUDAGraph()
    failed = None
    try:
        with torch.cuda.graph(graph0, stream=s):
            roundtrip()
    except Exception as e:  # the library's error surfaces through ops.check()
        failed = str(e)
    assert failed is not None and "before capturing" in failed, failed
    torch.cuda.synchronize()
    # (2) warm up, capture, churn the cache, replay
    with torch.cuda.stream(s):
        roundtrip()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        roundtrip()
    torch.cuda.synchronize()
    for k in range(40):
        ys = [torch.randn(3000 + 16 * ((k + 7 * j) % 9), generator=g0).to(torch.bfloat16).to(dev) for j in range(3)]
        c2, s2, _ = ops.compress_data(True, [ys[2], ys[0], ys[1]], False, temp)
    torch.cuda.synchronize()
    for x in xs:
        x.copy_(torch.randn(x.numel(), generator=g0).to(torch.bfloat16))
    for o in outs:
        o.zero_()
    status.zero_()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    assert bool(status.all())
    for x, o in zip(xs, outs):
        assert torch.equal(x.view(torch.int16), o.view(torch.int16))
    del graph
    assert L.dgpu_release_graph_state() >= 1  # the pinned block(s) become evictable again


def test_graph_survives_growth_of_the_overflow_slab():
    """a call captured with temp_mem=None bakes the address of the stream's library-owned overflow
    slab into the graph.  A later, LARGER call on the same stream makes the slab grow: the old slab is retired -- and
    must not be freed by the call after that, because the graph still replays into it."""
    import dietgpu_amd
    from dietgpu_amd import ops

    L = dietgpu_amd.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g0 = torch.Generator(device="cpu").manual_seed(23)
    B, n = 4, 60_000
    xs = [torch.randn(n + 8 * i, generator=g0).to(torch.bfloat16).to(dev) for i in range(B)]
    rows, cols = ops.max_float_compressed_output_size(xs)
    comp = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
    sizes = torch.zeros((rows,), dtype=torch.int32, device=dev)
    outs = [torch.empty_like(x) for x in xs]
    status = torch.zeros((B,), dtype=torch.uint8, device=dev)

    def roundtrip():
        ops.compress_data(True, xs, False, None, comp, sizes)
        ops.decompress_data(True, [comp[i] for i in range(B)], outs, False, None, status, None)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        roundtrip()  # warm: parameter blocks resident, the stream's slab exists (small: this batch needs a few MiB)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        roundtrip()
    torch.cuda.synchronize()
    # a much larger batch on the same stream, twice, without temp memory: the first call grows the slab (retiring
    # the one the graph knows), the second is the call that used to free retired slabs
    big = [torch.randn(3_000_000, generator=g0).to(torch.bfloat16).to(dev) for _ in range(24)]
    with torch.cuda.stream(s):
        for _ in range(2):
            c2, s2, _ = ops.compress_data(True, big, False, None)
            torch.cuda.synchronize()
        # memory freed under the graph would be handed out again here
        scratch = [torch.full((8 << 20,), 0xAB, dtype=torch.uint8, device=dev) for _ in range(24)]
    torch.cuda.synchronize()
    for trial in range(2):
        for x in xs:
            x.copy_(torch.randn(x.numel(), generator=g0).to(torch.bfloat16))
        for o in outs:
            o.zero_()
        status.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert bool(status.all())
        for x, o in zip(xs, outs):
            assert torch.equal(x.view(torch.int16), o.view(torch.int16)), trial
        assert all(bool((t == 0xAB).all()) for t in scratch)  # ... and the replay wrote nowhere else
    del graph, scratch
    assert L.dgpu_release_graph_state() >= 1


def test_checksum_verification_cannot_be_captured():
    """decode with checksum verification is host work behind a synchronise; under stream capture the call
    is refused with a message instead of invalidating the capture with an opaque HIP error."""
    import dietgpu_amd
    from dietgpu_amd import ops

    dietgpu_amd.lib()
    dev = torch.device("cuda", 0)
    x = (torch.arange(50_000, device=dev) % 37).to(torch.uint8)
    comp, sizes, _ = ops.compress_data(False, [x], True)
    out = torch.empty_like(x)
    ops.decompress_data(False, [comp[0]], [out], True)  # plain call: fine
    assert torch.equal(out, x)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    failed = None
    try:
        with torch.cuda.graph(graph, stream=s):
            ops.decompress_data(False, [comp[0]], [out], True)
    except Exception as e:
        failed = str(e)
    assert failed is not None and "checksum verification" in failed, failed
    torch.cuda.synchronize()
    out.zero_()
    ops.decompress_data(False, [comp[0]], [out], True)  # the library is usable afterwards
    assert torch.equal(out, x)
