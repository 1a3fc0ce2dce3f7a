#!/usr/bin/env python3
"""How slow is the general (edge) body?  Same launch, every wave forced through it."""
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
n = 8192
for xchunk in (16, 64):
    for fe in (False, True):
        ctx = F.FDWave(8, n, n, 64, 64, 100, 0.75, 10.0, 10.0, 0.001, compat=False)
        ctx.set_tuning(xchunk=xchunk, wz=4, force_edge=fe)
        p = torch.randn((n, ctx.pitch), device=dev) * 1e-3
        pp = torch.randn((n, ctx.pitch), device=dev) * 1e-3
        v2 = (1500.0 + 2500.0 * torch.rand((n, ctx.pitch), device=dev)) ** 2
        ms = timeit(lambda: ctx.dev_steps(p.data_ptr(), pp.data_ptr(), v2.data_ptr(), None, 0, 0, 0, 2, True, stream=s)) / 2
        print(f"xchunk={xchunk} force_edge={fe}: {ms*1e3:8.1f} us  {n*n/ms/1e6:7.1f} Gpt/s", flush=True)
