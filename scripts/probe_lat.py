#!/usr/bin/env python3
"""Per-march-step latency of the pipeline kernel on an almost empty machine (development tool)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import parallel_finite_difference_computation_amd as F
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
for (nx, nz, mode, xchunk) in ((173, 224, 4, 173), (173, 224 * 8, 4, 173), (173, 8192, 4, 173), (173 * 4, 8192, 4, 173), (173 * 8, 8192, 4, 173), (173*16, 8192, 4, 173), (173*32, 8192, 4, 173),
                               (172, 240, 1, 172), (172, 8192, 1, 172), (172 * 8, 8192, 1, 172)):
    ctx = F.FDWave(8, nx, nz, 16, 16, 100, 0.75, 10.0, 10.0, 0.001, compat=False)
    ctx.set_tuning(xchunk=xchunk, two_step=mode)
    bufs = [torch.randn((nx, ctx.pitch), device=dev) * 1e-3 for _ in range(4)]
    v2 = (1500.0 + 2500.0 * torch.rand((nx, ctx.pitch), device=dev)) ** 2
    ip, ipp = 0, 1
    nlaunch = 20
    per = 4 if mode == 4 else 2
    for rep in range(2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ip, ipp = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), None, 0, 0, 0, per * nlaunch, True, ip, ipp, stream=s)
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / nlaunch * 1e3
    iters = (xchunk + 27 + 9) // 10 * 10 if mode == 4 else (xchunk + 8 + 9) // 10 * 10
    print(f"mode {mode} grid {nx}x{nz}: {us:8.1f} us per launch, {iters} march steps -> {us / iters * 1e3:7.1f} ns per march step ({us/iters*2.4e3:6.0f} cycles at 2.4 GHz)", flush=True)
