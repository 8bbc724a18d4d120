"""How much of the small-element histogram kernel is the normalisation?  Times the stand-alone normalisation
(dgpu_ans_calc_weights = k_normalize, one 256-thread workgroup per element) on the exponent histograms of
B x 4096 bf16 N(0,1).  Usage (GPU box): python tools/norm_rate.py"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
import dietgpu_amd as dg

dev = torch.device("cuda", 0)
L = dg.lib()
for B, n in ((32768, 4096), (8192, 16384)):
    x = torch.randn(B, n, device=dev).to(torch.bfloat16).view(torch.int16).to(torch.int32)
    exp = (x >> 7) & 0xff
    hist = torch.zeros((B, 256), dtype=torch.int32, device=dev)
    hist.scatter_add_(1, exp.to(torch.int64), torch.ones_like(exp))
    sizes = torch.full((B,), n, dtype=torch.int32, device=dev)
    table = torch.zeros((B, 256, 4), dtype=torch.int32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for P in (10,):
        for _ in range(5):
            L.dgpu_ans_calc_weights(B, P, C.c_void_p(sizes.data_ptr()), 0, C.c_void_p(hist.data_ptr()), C.c_void_p(table.data_ptr()), st)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            L.dgpu_ans_calc_weights(B, P, C.c_void_p(sizes.data_ptr()), 0, C.c_void_p(hist.data_ptr()), C.c_void_p(table.data_ptr()), st)
        e.record()
        torch.cuda.synchronize()
        q = (hist.to(torch.float32) / n * (1 << P)).floor().clamp(min=0)
        q = torch.where((hist > 0) & (q == 0), torch.ones_like(q), q)
        deficit = (q.sum(1) > (1 << P)).float().mean().item()
        print(f"B={B} n={n} P={P}: k_normalize {s.elapsed_time(e) / 50 * 1e3:.1f} us per call; elements in the deficit branch: {deficit * 100:.0f} %")
