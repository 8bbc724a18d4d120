"""What a batch whose elements differ widely in size costs: the grids are laid out for the largest element (as upstream's
are), so the workgroups beyond a small element's end start, find nothing and leave.  One large bf16 tensor alone, next to
255 small ones, and the small ones alone.  Usage (GPU box): python tools/ragged_probe.py [--raw]"""
import sys, time, torch
sys.path.insert(0, ".")
import dietgpu_amd as dg
dg.load_torch_ops()
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(5)
RAW = "--raw" in sys.argv
KIND = "u8" if RAW else "bf16"  # byte tensors through the raw rANS ops instead of bf16 through the float codec


def tensor(n):
    if RAW:
        return (torch.randn(n, generator=g, device=dev) * 20).to(torch.int8).view(torch.uint8)
    return torch.randn(n, generator=g, device=dev).to(torch.bfloat16)


def rate(ts, reps=50):
    comp, sizes, _ = dg.compress_data(not RAW, ts, False)
    rows = [comp[i] for i in range(len(ts))]
    outs = [torch.empty_like(t) for t in ts]
    dg.decompress_data(not RAW, rows, outs, False)
    assert all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(ts, outs))
    temp = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    res = []
    for fn in (lambda: dg.compress_data(not RAW, ts, False, temp, comp, sizes), lambda: dg.decompress_data(not RAW, rows, outs, False, temp)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / reps * 1e6)
    return res


def rate_cabi(ns, reps=20):
    """compress / decompress us per call of a batch of tensors of `ns` words through the C ABI with pointer arrays: every
    member has an archive buffer of its own maximum size (the tensor API's output is ONE matrix with a row of the
    largest member's maximum per member -- terabytes for one 32 Mi-word tensor next to 32768 small ones)."""
    import ctypes as C

    L = dg.lib()
    ft = 0 if RAW else 2
    wb = 1 if RAW else 2
    B = len(ns)
    g.manual_seed(5)  # (the same words for the same list of sizes)
    flat = tensor(sum((n * wb + 15) // 16 * 16 // wb for n in ns) + 16)
    out = torch.empty_like(flat)
    cap_of = (lambda n: int(L.dgpu_ans_max_compressed_size(n))) if RAW else (lambda n: int(L.dgpu_float_max_compressed_size(ft, n)))
    caps = [(cap_of(n) + 15) // 16 * 16 for n in ns]
    comp = torch.empty(sum(caps), dtype=torch.uint8, device=dev)
    offs, coffs, o, c = [], [], 0, 0
    for n, cp in zip(ns, caps):
        offs.append(o)
        coffs.append(c)
        o += (n * wb + 15) // 16 * 16 // wb  # every member 16-byte aligned
        c += cp
    assert o <= flat.numel()
    in_ptrs = (C.c_void_p * B)(*[flat.data_ptr() + x * wb for x in offs])
    out_ptrs = (C.c_void_p * B)(*[out.data_ptr() + x * wb for x in offs])
    comp_ptrs = (C.c_void_p * B)(*[comp.data_ptr() + x for x in coffs])
    sizes_h = (C.c_uint32 * B)(*ns)
    sizes = torch.zeros(B, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.uint8, device=dev)
    osz = torch.zeros(B, dtype=torch.int32, device=dev)
    temp = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    err = C.c_int32(-1)
    tp, tb = C.c_void_p(temp.data_ptr()), temp.numel()

    def enc():
        if RAW:
            rc = L.dgpu_ans_encode_batch_pointer(tp, tb, None, 10, 0, B, in_ptrs, sizes_h, None, comp_ptrs, C.c_void_p(sizes.data_ptr()), stream)
        else:
            rc = L.dgpu_float_compress(tp, tb, None, ft, 10, 0, B, in_ptrs, sizes_h, comp_ptrs, C.c_void_p(sizes.data_ptr()), stream)
        assert rc == 0, L.dgpu_last_error().decode()

    def dec():
        if RAW:
            rc = L.dgpu_ans_decode_batch_pointer(tp, tb, None, 10, 0, B, comp_ptrs, out_ptrs, sizes_h, C.c_void_p(status.data_ptr()), C.c_void_p(osz.data_ptr()), stream, C.byref(err))
        else:
            rc = L.dgpu_float_decompress(tp, tb, None, ft, 10, 0, B, comp_ptrs, out_ptrs, sizes_h, C.c_void_p(status.data_ptr()), C.c_void_p(osz.data_ptr()), stream, C.byref(err))
        assert rc == 0, L.dgpu_last_error().decode()

    enc()
    dec()
    torch.cuda.synchronize()
    assert bool(status.all().item()) and osz.tolist() == list(ns)
    vi, vo = flat.view(torch.uint8), out.view(torch.uint8)
    for x, n in list(zip(offs, ns))[:: max(1, B // 64)]:
        assert torch.equal(vi[x * wb : (x + n) * wb], vo[x * wb : (x + n) * wb])
    res = []
    for fn in (enc, dec):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        t_host = time.perf_counter() - t0  # the host's share: what it takes to ENQUEUE the calls
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / reps * 1e6)
        res.append(t_host / reps * 1e6)
    return [res[0], res[2], res[1], res[3]], comp, sizes, coffs


def classes_case(label, groups):
    """groups: [(count, words)]: the classes alone, their sum, and the whole batch in one call with the size classes off
    (one geometry for the call, from its largest member) and on (the library's policy); the archives of both forms are
    compared byte for byte."""
    alone = [rate_cabi([n] * c)[0] for c, n in groups]
    whole = [n for c, n in groups for _ in range(c)]
    L = dg.lib()
    L.dgpu_debug_set_size_classes(0)
    off, comp0, sizes0, coffs = rate_cabi(whole)
    keep = [comp0[o : o + int(s)].clone() for o, s in list(zip(coffs, sizes0.tolist()))[:: max(1, len(whole) // 128)]]
    L.dgpu_debug_set_size_classes(-1)
    on, comp1, sizes1, coffs = rate_cabi(whole)
    assert torch.equal(sizes0, sizes1)
    same = all(torch.equal(k, comp1[o : o + int(s)]) for k, (o, s) in zip(keep, list(zip(coffs, sizes1.tolist()))[:: max(1, len(whole) // 128)]))
    sa, sb = sum(a[0] for a in alone), sum(a[1] for a in alone)
    print(f"{label} {KIND}: compress / decompress us   classes alone " + " + ".join(f"{a[0]:.0f}" for a in alone) + f" = {sa:.0f} / " +
          " + ".join(f"{a[1]:.0f}" for a in alone) + f" = {sb:.0f}   one call, one geometry {off[0]:8.1f} {off[1]:8.1f}   one call, size classes {on[0]:8.1f} {on[1]:8.1f}"
          f"   (classes / sum of alone: {on[0] / sa:.2f} {on[1] / sb:.2f}; archives identical: {same}; host time to enqueue a call: one geometry {off[2]:.0f} / {off[3]:.0f}, size classes {on[2]:.0f} / {on[3]:.0f})", flush=True)


if "--classes" in sys.argv:
    classes_case("1 x 32 Mi + 32768 x 2048", [(1, 32 << 20), (32768, 2048)])
    classes_case("model-like: 8 x 4 Mi + 64 x 64 Ki + 4096 x 1 Ki", [(8, 4 << 20), (64, 64 << 10), (4096, 1 << 10)])
    classes_case("1 x 32 Mi + 255 x 2048 (too few small ones: not split)", [(1, 32 << 20), (255, 2048)])
    classes_case("256 x 512 Ki + 2048 x 6000", [(256, 512 << 10), (2048, 6000)])
    sys.exit(0)

for big_n in (32 << 20, 4 << 20):
    big = tensor(big_n)
    for small_n in (2048, 20000):
        small = [tensor(small_n) for _ in range(255)]
        a, b, c = rate([big]), rate([big] + small), rate(small)
        print(f"1 x {big_n} + 255 x {small_n} {KIND}: compress / decompress us   large alone {a[0]:8.1f} {a[1]:8.1f}   together {b[0]:8.1f} {b[1]:8.1f}   small alone {c[0]:8.1f} {c[1]:8.1f}")

# sizes that merely vary (uniform in [1/16, 1] of 1 Mi words; 10 % / 30 % / 60 % of the rectangle empty)
for lo in (0.85, 0.5, 1.0 / 16):
    ns = [int(n) for n in torch.randint(int(lo * (1 << 20)), 1 << 20, (256,), generator=torch.Generator().manual_seed(3)).tolist()]
    ts = [tensor(n) for n in ns]
    a = rate(ts)
    print(f"256 tensors of {lo:.2f} .. 1 Mi words {KIND} ({sum(ns) * (1 if RAW else 2) / 1e6:.0f} MB): compress / decompress us {a[0]:8.1f} {a[1]:8.1f}")
