#!/usr/bin/env python3
"""Profiling driver (development tool): a few fused steps at one size, for rocprofv3.
usage: prof_step.py size steps xchunk prefetch [mode]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import parallel_finite_difference_computation_amd as F
n, steps, xchunk, pf = (int(a) for a in sys.argv[1:5])
mode = sys.argv[5] if len(sys.argv) > 5 else "step"
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
ctx = F.FDWave(8, n, n, 64, 64, 100, 0.75, 10.0, 10.0, 0.001, compat=False)
ctx.set_tuning(xchunk=xchunk, wz=4, prefetch=pf)
p = torch.randn((n, ctx.pitch), device=dev) * 1e-3
pp = torch.randn((n, ctx.pitch), device=dev) * 1e-3
v2 = (1500.0 + 2500.0 * torch.rand((n, ctx.pitch), device=dev)) ** 2
torch.cuda.synchronize()
if mode == "lap":
    for _ in range(steps): ctx.dev_laplacian(p.data_ptr(), pp.data_ptr(), stream=s)
else:
    ctx.dev_steps(p.data_ptr(), pp.data_ptr(), v2.data_ptr(), None, 0, 0, 0, steps, True, stream=s)
torch.cuda.synchronize()
print("done")
