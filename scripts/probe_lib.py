#!/usr/bin/env python3
"""Time the fused step of whatever library FDW_LIB points at, a few sizes/chunks (development tool)."""
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
tag = os.path.basename(os.environ.get("FDW_LIB", "default"))
for n in (4096, 8192, 16384):
    for xchunk in (0, 10, 30, 60):
        ctx = F.FDWave(8, n, n, 64, 64, 100, 0.75, 10.0, 10.0, 0.001, compat=False)
        ctx.set_tuning(xchunk=xchunk, wz=4)
        p = torch.randn((n, ctx.pitch), device=dev) * 1e-3
        pp = torch.randn((n, ctx.pitch), device=dev) * 1e-3
        v2 = (1500.0 + 2500.0 * torch.rand((n, ctx.pitch), device=dev)) ** 2
        ms = min(timeit(lambda: ctx.dev_steps(p.data_ptr(), pp.data_ptr(), v2.data_ptr(), None, 0, 0, 0, 2, True, stream=s)) / 2 for _ in range(3))
        print(f"{tag} n={n} xchunk={xchunk:3d}: {ms*1e3:8.1f} us  {n*n/ms/1e6:7.1f} Gpt/s  {n*n/ms/1e6*16/1e3:5.2f} TB/s", flush=True)
        del p, pp, v2, ctx
