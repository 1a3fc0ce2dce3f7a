#!/usr/bin/env python3
"""One configuration of the forward loop, for profiling: probe_one.py nx nz two_step(-1|0|1) xchunk [launches]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import parallel_finite_difference_computation_amd as F
nx, nz, two, xchunk = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
ctx = F.FDWave(8, nx, nz, 64, 64, 100, 0.75, 10.0, 10.0, 0.001, compat=False)
ctx.set_tuning(xchunk=xchunk, two_step=two)
bufs = [torch.randn((nx, ctx.pitch), device=dev) * 1e-3 for _ in range(4)]
v2 = (1500.0 + 2500.0 * torch.rand((nx, ctx.pitch), device=dev)) ** 2
ip, ipp = 0, 1
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ip, ipp = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), None, 0, 0, 0, 2 * reps, True, ip, ipp, stream=s)
e1.record(); torch.cuda.synchronize()
print(f"{nx}x{nz} two={two} xchunk={xchunk}: {e0.elapsed_time(e1) / (2 * reps) * 1e3:.2f} us/step")
