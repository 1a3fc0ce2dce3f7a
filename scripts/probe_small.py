#!/usr/bin/env python3
"""Small-deck probe (development tool): per-step time of the forward loop on the new_mod grid (415x295)."""
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
for (nxe, nze) in ((415, 295), (1024, 1024), (2048, 2048)):
    for pf in (2, 1):
        for xchunk in (1, 2, 3, 4, 6, 8, 16):
            ctx = F.FDWave(8, nxe, nze, 50, 50, 2000, 0.75, 10.0, 10.0, 0.001, compat=True)
            ctx.set_tuning(xchunk=xchunk, prefetch=pf)
            p = torch.zeros((nxe, ctx.pitch), device=dev); pp = torch.zeros_like(p)
            v2 = torch.full_like(p, 2500.0 ** 2)
            srce = torch.from_numpy(F.ricker_wavelet(2000, 0.001, 20.0)).to(dev)
            ms = timeit(lambda: ctx.dev_steps(p.data_ptr(), pp.data_ptr(), v2.data_ptr(), srce.data_ptr(), nxe // 2, nze // 2, 0, 100, True, stream=s)) / 100
            print(f"{nxe}x{nze} pf={pf} xchunk={xchunk:2d}: {ms*1e3:6.2f} us/step  {nxe*nze/ms/1e6:6.1f} Gpt/s", flush=True)
