#!/usr/bin/env python3
"""Laplacian-only kernel (stencil_code path) and RTM backward iteration at large sizes (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import parallel_finite_difference_computation_amd as F
from parallel_finite_difference_computation_amd import MODE_PLAIN, MODE_RECV
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
for n in (4096, 8192, 16384):
    ctx = F.FDWave(8, n, n, 64, 64, 100, 0.75, 10.0, 10.0, 0.001, compat=False)
    f = [torch.randn((n, ctx.pitch), device=dev) * 1e-3 for _ in range(4)]
    v2 = (1500.0 + 2500.0 * torch.rand((n, ctx.pitch), device=dev)) ** 2
    img = torch.zeros((n, ctx.pitch), device=dev)
    dobs = torch.randn((n - 128,), device=dev)
    ms = timeit(lambda: ctx.dev_laplacian(f[0].data_ptr(), f[1].data_ptr(), stream=s))
    print(f"n={n} Laplacian only: {ms*1e3:8.1f} us  {n*n/ms/1e6:7.1f} Gpt/s  {8*n*n/ms/1e9:6.2f} TB/s of 8 B/pt ({8*n*n/ms/1e9/8*100:.1f}% of 8 TB/s)", flush=True)
    def back_iter():
        ctx.dev_step(MODE_PLAIN, f[0].data_ptr(), f[1].data_ptr(), v2.data_ptr(), stream=s)
        ctx.dev_step(MODE_RECV, f[2].data_ptr(), f[3].data_ptr(), v2.data_ptr(), d_inj=dobs.data_ptr(), inj_z=70, d_psrc=f[1].data_ptr(), d_img=img.data_ptr(), stream=s)
    ms = timeit(back_iter)
    print(f"n={n} backward iteration (PLAIN + RECV): {ms*1e3:8.1f} us  {n*n/ms/1e6:7.1f} Gpt/s  {44*n*n/ms/1e9:6.2f} TB/s of 44 B/pt ({44*n*n/ms/1e9/8*100:.1f}% of 8 TB/s)", flush=True)
    del f, v2, img, ctx
