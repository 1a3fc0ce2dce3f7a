#!/usr/bin/env python3
"""Slab-sized grids (what one rank of an 8-way split of 8192^2 steps): one-step vs forced two-step kernel (development tool)."""
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
shapes = [(1056, 8192), (2080, 8192), (4128, 8192)]
for (nx, nz) in shapes:
    for two, chunks in ((-1, (0, 4, 6, 8, 12)), (1, (2, 12, 22, 32))):
        for xchunk in chunks:
            ctx = F.FDWave(8, nx, nz, 64, 64, 100, 0.75, 10.0, 10.0, 0.001, compat=False)
            ctx.set_tuning(xchunk=xchunk, two_step=two)
            bufs = [torch.randn((nx, ctx.pitch), device=dev) * 1e-3 for _ in range(4)]
            v2 = (1500.0 + 2500.0 * torch.rand((nx, ctx.pitch), device=dev)) ** 2
            st = {"ip": 0, "ipp": 1}
            def go():
                st["ip"], st["ipp"] = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), None, 0, 0, 0, 16, True, st["ip"], st["ipp"], stream=s)
            ms = min(timeit(go) for _ in range(3)) / 16
            g = nx * nz / ms / 1e6
            print(f"{nx}x{nz} {'two-step' if two > 0 else 'one-step'} xchunk={xchunk:3d}: {ms*1e3:7.2f} us/step  {g:7.1f} Gpt/s", flush=True)
            del bufs, v2, ctx
