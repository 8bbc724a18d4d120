"""Cost of the corners of the call surface next to the plain call: checksums, precisions, many tiny tensors.
Microseconds per compress / decompress call through dietgpu_amd.ops.  Usage (GPU box): python tools/surface_probe.py"""
import sys, time, torch
sys.path.insert(0, ".")
import dietgpu_amd as dg
dg.load_torch_ops()
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(5)


def rate(ts, checksum=False, prob_bits=10, reps=30):
    comp, sizes, _ = dg.compress_data(True, ts, checksum, prob_bits=prob_bits)
    rows = [comp[i] for i in range(len(ts))]
    outs = [torch.empty_like(t) for t in ts]
    dg.decompress_data(True, rows, outs, checksum, prob_bits=prob_bits)
    assert all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(ts[:8], outs[:8]))
    temp = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    res = []
    for fn in (lambda: dg.compress_data(True, ts, checksum, temp, comp, sizes, prob_bits=prob_bits),
               lambda: dg.decompress_data(True, rows, outs, checksum, temp, prob_bits=prob_bits)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / reps * 1e6)
    return res


big = [t for t in torch.randn(256, 524288, generator=g, device=dev).to(torch.bfloat16)]
for ck in (False, True):
    for p in (9, 10, 11):
        a = rate(big, ck, p)
        print(f"256 x 512 Ki bf16  checksum={int(ck)} probBits={p}: compress {a[0]:8.1f} us  decompress {a[1]:8.1f} us")
for n, count in ((64, 65535), (1000, 65535), (4096, 20000), (5000, 20000)):
    ts = [t for t in torch.randn(count, n, generator=g, device=dev).to(torch.bfloat16)]
    a = rate(ts, reps=5)
    print(f"{count} x {n} bf16 ({count * n * 2 / 1e6:.0f} MB): compress {a[0]:8.1f} us  decompress {a[1]:8.1f} us")
