"""How the first timed regions of a fresh process run: the headline loop of bench.py (4 rotating buffer sets of BASELINE
config 3) in consecutive regions of 5 + 20 steps, wall time per step and GPU time per step (events), with the host's time
to ENQUEUE a step beside it.  Is the slow first region the GPU's clocks or the host?  Usage (GPU box): python tools/ramp_probe.py"""
import sys, time, torch
sys.path.insert(0, ".")
import bench, dietgpu_amd as dg

dev = torch.device("cuda:0")
sets = []
for r in range(4):
    d, ft, _, P, _ = bench.make_workload("bf16", 256, 1234 + 1000 * r, dev, 512 * 1024)
    c = bench.Codec(dg, d, ft, P)
    if sets:
        c.temp = sets[0].temp
    c.step()
    c.verify()
    sets.append(c)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for region in range(8):
    for i in range(5):
        sets[i % 4].step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(20):
        sets[(5 + i) % 4].step()
    t_enq = time.perf_counter() - t0
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"region {region}: wall {wall / 20 * 1e3:.4f} ms/step   gpu (events) {ev0.elapsed_time(ev1) / 20:.4f} ms/step   host enqueue {t_enq / 20 * 1e3:.4f} ms/step")
    if region == 3:
        time.sleep(0.5)
        print("(0.5 s idle)")
