"""Latency of the stand-alone normalisation of ONE element (k_normalize) next to the smallest kernel there is, events around
each launch.  Round 6 on MI355X: 6.2 us both -- the floor of a launch, not work.  Usage (GPU box): python tools/norm_latency.py"""
import ctypes as C, sys, statistics, torch
sys.path.insert(0, ".")
import dietgpu_amd as dg
dev = torch.device("cuda", 0); L = dg.lib()
x = torch.randn(1, 1 << 20, device=dev).to(torch.bfloat16).view(torch.int16).to(torch.int32)
exp = (x >> 7) & 0xff
hist = torch.zeros((1, 256), dtype=torch.int32, device=dev); hist.scatter_add_(1, exp.to(torch.int64), torch.ones_like(exp))
sizes = torch.full((1,), 1 << 20, dtype=torch.int32, device=dev); table = torch.zeros((1, 256, 4), dtype=torch.int32, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call = lambda: L.dgpu_ans_calc_weights(1, 10, C.c_void_p(sizes.data_ptr()), 0, C.c_void_p(hist.data_ptr()), C.c_void_p(table.data_ptr()), st)
empty = torch.zeros(1, device=dev)
for fn, name in ((call, "k_normalize B=1"), (lambda: empty.add_(1), "a one-element torch kernel")):
    for _ in range(20): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(200):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    print(name, "median us", round(statistics.median(ts), 2), "min", round(min(ts), 2))
