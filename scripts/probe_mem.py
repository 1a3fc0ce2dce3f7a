#!/usr/bin/env python3
"""Memory-system probe (development tool): copy ceiling, Laplacian-only, step with padded pitch."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import parallel_finite_difference_computation_amd as F

dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream

def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for mb in (64, 256, 1024):
    a = torch.empty(mb * 2**18, device=dev); b = torch.empty_like(a); a.normal_()
    ms = timeit(lambda: b.copy_(a))
    print(f"torch copy {mb} MiB: {ms*1e3:7.1f} us  {2*mb*2**20/ms/1e9:7.2f} TB/s (read+write)", flush=True)
    c = torch.empty_like(a); c.normal_()
    ms = timeit(lambda: torch.add(a, c, out=b))
    print(f"torch add  {mb} MiB: {ms*1e3:7.1f} us  {3*mb*2**20/ms/1e9:7.2f} TB/s (2 reads+write)", flush=True)
    del a, b, c

def step_probe(n, xchunk, pf, lap=False):
    ctx = F.FDWave(8, n, n, 64, 64, 100, 0.75, 10.0, 10.0, 0.001, compat=False)
    ctx.set_tuning(xchunk=xchunk, wz=4, prefetch=pf)
    p = torch.randn((n, ctx.pitch), device=dev) * 1e-3
    pp = torch.randn((n, ctx.pitch), device=dev) * 1e-3
    v2 = (1500.0 + 2500.0 * torch.rand((n, ctx.pitch), device=dev)) ** 2
    if lap:
        ms = timeit(lambda: ctx.dev_laplacian(p.data_ptr(), pp.data_ptr(), stream=s), n=40)
        return ms, n * n / ms / 1e6, 8
    ms = timeit(lambda: ctx.dev_steps(p.data_ptr(), pp.data_ptr(), v2.data_ptr(), None, 0, 0, 0, 2, True, stream=s), n=40) / 2
    return ms, n * n / ms / 1e6, 16

pad = os.environ.get("FDW_PITCH_PAD", "0")
for n in (8192,):
    for lap in (True, False):
        for xchunk in (12, 24, 60):
            ms, g, bpp = step_probe(n, xchunk, 2, lap)
            print(f"pad={pad} n={n} {'LAP ' if lap else 'STEP'} xchunk={xchunk:3d}: {ms*1e3:8.1f} us  {g:7.1f} Gpt/s  {g*bpp/1e3:6.2f} TB/s", flush=True)
