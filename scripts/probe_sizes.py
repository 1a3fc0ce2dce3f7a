#!/usr/bin/env python3
"""Square grids of several sizes: automatic policy vs forced one-step / two-step / four-wave pipeline (development tool)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import parallel_finite_difference_computation_amd as F
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
sizes = [(int(a), int(a)) for a in sys.argv[1:]] or [(415, 295), (1024, 1024), (2048, 2048), (4096, 4096)]
for (nx, nz) in sizes:
    for two, chunks in ((0, (0,)), (-1, (0,)), (1, (0,)), (4, (13, 23, 43, 83, 173))):
        for xchunk in chunks:
            ctx = F.FDWave(8, nx, nz, 50, 50, 100, 0.75, 10.0, 10.0, 0.001, compat=False)
            ctx.set_tuning(xchunk=xchunk, two_step=two)
            bufs = [torch.randn((nx, ctx.pitch), device=dev) * 1e-3 for _ in range(4)]
            v2 = (1500.0 + 2500.0 * torch.rand((nx, ctx.pitch), device=dev)) ** 2
            st = {"ip": 0, "ipp": 1}
            def go():
                st["ip"], st["ipp"] = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), None, 0, 0, 0, 32, True, st["ip"], st["ipp"], stream=s)
            ms = min(timeit(go) for _ in range(2)) / 32
            name = {0: "auto", -1: "one-step", 1: "two-step", 4: "pipe-4"}[two]
            print(f"{nx}x{nz} {name:8s} xchunk={xchunk:3d}: {ms*1e3:7.2f} us/step  {nx*nz/ms/1e6:7.1f} Gpt/s", flush=True)
            del bufs, v2, ctx
