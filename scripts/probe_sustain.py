#!/usr/bin/env python3
"""Does the two-step kernel's rate depend on how long it runs / on the data?  (development tool)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, numpy as np
import parallel_finite_difference_computation_amd as F
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
n = 8192
ctx = F.FDWave(8, n, n, 64, 64, 5000, 0.75, 10.0, 10.0, 0.001, compat=False)
def t(bufs, v2, srce, steps):
    st = {"ip": 0, "ipp": 1}
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    st["ip"], st["ipp"] = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), srce, n // 2, n // 2, 0, steps, True, 0, 1, stream=s)
    e1.record(); torch.cuda.synchronize()
    return n * n * steps / (e0.elapsed_time(e1) * 1e-3) / 1e9
z = torch.arange(n, device=dev, dtype=torch.float32)[None, :]
x = torch.arange(n, device=dev, dtype=torch.float32)[:, None]
v2_smooth = ((1500.0 + 2500.0 * z / (n - 1)) * (1.0 + 0.03 * torch.sin(2.0 * np.pi * 8.0 * x / n))) ** 2
v2_rand = (1500.0 + 2500.0 * torch.rand((n, n), device=dev)) ** 2
srce = torch.from_numpy(F.ricker_wavelet(5000, 0.001, 20.0)).to(dev)
for name, v2 in (("smooth v2", v2_smooth), ("random v2", v2_rand)):
    for init in ("noise", "zeros"):
        for steps in (16, 100, 400, 1000):
            bufs = [torch.zeros((n, n), device=dev) for _ in range(4)]
            if init == "noise":
                bufs[0].normal_(); bufs[0].mul_(1e-3); bufs[1].normal_(); bufs[1].mul_(1e-3)
            t(bufs, v2, srce.data_ptr(), 16)
            print(f"{name:10s} init={init:5s} steps={steps:4d}: {t(bufs, v2, srce.data_ptr(), steps):6.1f} Gpt/s", flush=True)
