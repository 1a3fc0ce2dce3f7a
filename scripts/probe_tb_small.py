#!/usr/bin/env python3
"""Two-step kernel on small decks: us per time step vs chunk length (development tool)."""
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
for (nxe, nze) in ((415, 295), (1024, 1024), (2048, 2048), (4096, 4096)):
    for xchunk in (2, 4, 12, 22, 32):
        ctx = F.FDWave(8, nxe, nze, 50, 50, 2000, 0.75, 10.0, 10.0, 0.001, compat=True)
        ctx.set_tuning(xchunk=xchunk)
        bufs = [torch.zeros((nxe, ctx.pitch), device=dev) for _ in range(4)]
        v2 = torch.full((nxe, ctx.pitch), 2500.0 ** 2, device=dev)
        srce = torch.from_numpy(F.ricker_wavelet(2000, 0.001, 20.0)).to(dev)
        st = {"ip": 0, "ipp": 1}
        def go():
            st["ip"], st["ipp"] = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), srce.data_ptr(), nxe // 2, nze // 2, 0, 100, True, st["ip"], st["ipp"], stream=s)
        ms = timeit(go) / 100
        print(f"TB2 {nxe}x{nze} xchunk={xchunk:2d}: {ms*1e3:6.2f} us/step  {nxe*nze/ms/1e6:6.1f} Gpt/s", flush=True)
