#!/usr/bin/env python3
"""One-step kernel on slab-shaped grids (what each GPU of an N-way decomposition of 8192^2 runs)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import parallel_finite_difference_computation_amd as F
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
n = 8192
for nxl in (1056, 2080, 4128):
    for two in (-1, 1):
        for xchunk in ((2, 4, 8, 10, 16, 24) if two < 0 else (12, 22, 32)):
            ctx = F.FDWave(8, n, n, 64, 64, 100, 0.75, 10.0, 10.0, 0.001, compat=False, slab=(2048, nxl))
            ctx.set_tuning(xchunk=xchunk, two_step=two)
            bufs = [torch.randn((nxl, ctx.pitch), device=dev) * 1e-3 for _ in range(4)]
            v2 = (1500.0 + 2500.0 * torch.rand((nxl, ctx.pitch), device=dev)) ** 2
            st = {"ip": 0, "ipp": 1}
            def go():
                st["ip"], st["ipp"] = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), None, 0, 0, 0, 8, True, st["ip"], st["ipp"], stream=s)
            ms = min(timeit(go) for _ in range(2)) / 8
            print(f"slab {nxl}x{n} {'two-step' if two > 0 else 'one-step'} xchunk={xchunk:2d}: {ms*1e3:7.1f} us/step  {nxl*n/ms/1e6:6.1f} Gpt/s", flush=True)
