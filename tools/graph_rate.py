"""Launch-bound batches: compress + decompress as plain C-ABI calls (through dietgpu_amd.ops) vs the same pair
captured once into a HIP graph and replayed.  Usage (GPU box): python tools/graph_rate.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
from dietgpu_amd import ops

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
temp = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
for B, n in ((1, 524288), (8, 65536), (64, 65536), (256, 524288)):
    xs = [torch.randn(n, device=dev).to(torch.bfloat16) for _ in range(B)]
    rows, cols = ops.max_float_compressed_output_size(xs)
    comp = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
    sizes = torch.zeros((rows,), dtype=torch.int32, device=dev)
    outs = [torch.empty_like(x) for x in xs]
    crow = [comp[i] for i in range(B)]

    def roundtrip():
        ops.compress_data(True, xs, False, temp, comp, sizes)
        ops.decompress_data(True, crow, outs, False, temp)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            roundtrip()
        torch.cuda.synchronize()
        reps = 300
        t0 = time.perf_counter()
        for _ in range(reps):
            roundtrip()
        torch.cuda.synchronize()
        plain = (time.perf_counter() - t0) / reps
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        roundtrip()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / reps
    assert all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(xs, outs))
    print(f"{B:4d} x {n:7d} bf16: plain calls {plain * 1e6:8.1f} us   graph replay {graph * 1e6:8.1f} us per compress+decompress")
