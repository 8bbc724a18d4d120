#!/usr/bin/env python3
"""Which buffer's placement decides the encoder's mode (tools/mode_probe.py: ~80 or ~86 us for bf16 256 x 512 Ki)?
All buffers are carved out of one arena; one of {input, archive rows, temp} is moved at a time."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dietgpu_amd as dg
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
wl = sys.argv[1] if len(sys.argv) > 1 else "bf16"
src, ft, _, P, desc = bench.make_workload(wl, 256, 1234, dev)
MiB = 1 << 20
arena = torch.empty((3072 * MiB,), dtype=torch.uint8, device=dev)
base = arena.data_ptr()
base_al = (base + 64 * MiB - 1) // (64 * MiB) * (64 * MiB) - base   # 64 MiB aligned start inside the arena
nbytes = src.numel() * src.element_size()

def run(off_in, off_comp, off_temp, off_out=None, steps=60):
    data = arena[base_al + off_in: base_al + off_in + nbytes].view(src.dtype).view(src.shape)
    data.copy_(src)
    c = bench.Codec(dg, data, ft, P)
    B = c.B
    comp = arena[base_al + off_comp: base_al + off_comp + B * c.row_cap].view(B, c.row_cap)
    temp = arena[base_al + off_temp: base_al + off_temp + c.temp.numel()]
    spans = sorted([(off_in, off_in + nbytes), (off_comp, off_comp + B * c.row_cap), (off_temp, off_temp + c.temp.numel())] +
                   ([(off_out, off_out + nbytes)] if off_out is not None else []))
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and base_al + spans[-1][1] <= arena.numel(), "buffers overlap"
    if os.environ.get("PROBE_SYNC"):
        print("  in %x comp %x..%x temp %x..%x out +%x  arena end %x" % (data.data_ptr(), comp.data_ptr(), comp.data_ptr() + comp.numel(),
              temp.data_ptr(), temp.data_ptr() + temp.numel(), off_out or 0, base + arena.numel()), flush=True)
    c.comp, c.temp = comp, temp
    c.comp_ptrs = (C.c_void_p * B)(*[comp.data_ptr() + i * c.row_cap for i in range(B)])
    if off_out is not None:
        out = arena[base_al + off_out: base_al + off_out + nbytes].view(src.dtype).view(src.shape)
        c.out = out
        row_in = data.stride(0) * data.element_size()
        c.out_ptrs = (C.c_void_p * B)(*[out.data_ptr() + i * row_in for i in range(B)])
    for it in range(10):
        c.encode()
        if os.environ.get("PROBE_SYNC"):
            torch.cuda.synchronize(); print("  encode", it, "ok", flush=True)
        c.decode()
        if os.environ.get("PROBE_SYNC"):
            torch.cuda.synchronize(); print("  decode", it, "ok", flush=True)
    torch.cuda.synchronize()
    prof = bench.kernel_profile(c, steps, lambda i: (c.encode(), c.decode()))
    t = {n[6:]: round(r["total_ms"] / max(r["launches"], 1) * 1e3, 1) for n, r in prof.items()}
    assert torch.equal(c.out.view(torch.uint8), data.view(torch.uint8))
    return t

IN0, COMP0, TEMP0, OUT0 = 0, 512 * MiB, 1280 * MiB, 1536 * MiB  # (the archive rows take B x row_cap = 445 MB)
print("arena %x aligned start +%x" % (base, base_al))
print("baseline", run(IN0, COMP0, TEMP0, OUT0))
deltas = [4096, 65536, 1 * MiB, 2 * MiB, 6 * MiB, 16 * MiB, 34 * MiB, 130 * MiB]
if os.environ.get("PROBE_DELTAS"):
    deltas = [int(x) for x in os.environ["PROBE_DELTAS"].split(",")]
for name in os.environ.get("PROBE_BUFFERS", "in,comp,temp,out").split(","):
    for d in deltas:
        o = {"in": IN0, "comp": COMP0, "temp": TEMP0, "out": OUT0}
        o[name] += d
        print("%-5s +%9d  %s" % (name, d, run(o["in"], o["comp"], o["temp"], o["out"])), flush=True)
print("baseline again", run(IN0, COMP0, TEMP0, OUT0))
