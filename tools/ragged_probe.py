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
