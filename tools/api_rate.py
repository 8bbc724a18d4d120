"""End-to-end rate of the tensor API (BASELINE config 3): torch.ops.dietgpu.* and the ctypes mirror,
with and without caller-supplied temp memory.  Usage (GPU box): python tools/api_rate.py"""
import time, torch, sys
sys.path.insert(0, ".")
import dietgpu_amd as dg
dg.load_torch_ops()
dev = "cuda:0"
ts = [t for t in torch.randn(256, 524288, device=dev).to(torch.bfloat16)]
def ctypes_route(fn):
    def call(*a):
        dg.prefer_torch_ops(False)
        try:
            return fn(*a)
        finally:
            dg.prefer_torch_ops(True)
    return call


for name, fn_c, fn_d in (("torch.ops.dietgpu", torch.ops.dietgpu.compress_data, torch.ops.dietgpu.decompress_data),
                         ("dietgpu_amd.ops (default route)", dg.compress_data, dg.decompress_data),
                         ("dietgpu_amd.ops (ctypes route)", ctypes_route(dg.compress_data), ctypes_route(dg.decompress_data))):
    for temp in (None, torch.empty(128 << 20, dtype=torch.uint8, device=dev)):
        comp, sizes, _ = fn_c(True, ts, False, temp)
        outs = [torch.empty_like(t) for t in ts]
        rows = [comp[i] for i in range(len(ts))]
        fn_d(True, rows, outs, False, temp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            comp, sizes, _ = fn_c(True, ts, False, temp, comp, sizes)
            fn_d(True, rows, outs, False, temp)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name:32s} temp={'given' if temp is not None else 'None ':5s}  {dt*1e6:8.1f} us per compress+decompress  ({2*256*2**20/dt/1e9:.0f} GB/s)")
        assert all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(ts[:4], outs[:4]))
