#!/usr/bin/env python3
"""Two-step (temporal blocking) kernel probe: time per time step vs chunk length (development tool)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import parallel_finite_difference_computation_amd as F
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]
for n in sizes:
    for xchunk in (12, 22, 32, 42, 72, 102, 132):
        ctx = F.FDWave(8, n, n, 64, 64, 100, 0.75, 10.0, 10.0, 0.001, compat=False)
        ctx.set_tuning(xchunk=xchunk)
        bufs = [torch.randn((n, ctx.pitch), device=dev) * 1e-3 for _ in range(4)]
        v2 = (1500.0 + 2500.0 * torch.rand((n, ctx.pitch), device=dev)) ** 2
        st = {"ip": 0, "ipp": 1}
        def go():
            st["ip"], st["ipp"] = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), None, 0, 0, 0, 8, True, st["ip"], st["ipp"], stream=s)
        ms = min(timeit(go) for _ in range(2)) / 8
        g = n * n / ms / 1e6
        print(f"TB2 n={n} xchunk={xchunk:3d}: {ms*1e3:8.1f} us/step  {g:7.1f} Gpt/s  {g*16/1e3:5.2f} TB/s-equivalent ({g*16/8000*100:5.1f}% of 8 TB/s at 16 B/pt)", flush=True)
        del bufs, v2, ctx
