#!/usr/bin/env python3
"""Two-step (temporal blocking) kernel probe: time per time step vs chunk length (development tool)."""
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
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]
mode = int(os.environ.get("TB_MODE", "1"))     # 1: two steps per pass, 4: four-wave pipeline
chunks = [int(c) for c in os.environ["TB_CHUNKS"].split(",")] if "TB_CHUNKS" in os.environ else ((12, 22, 32, 42, 72, 102, 132) if mode == 1 else (23, 43, 63, 83, 123, 173, 253))
for n in sizes:
    for xchunk in chunks:
        ctx = F.FDWave(8, n, n, 64, 64, 100, 0.75, 10.0, 10.0, 0.001, compat=False)
        ctx.set_tuning(xchunk=xchunk, two_step=mode)
        bufs = [torch.randn((n, ctx.pitch), device=dev) * 1e-3 for _ in range(4)]
        v2 = (1500.0 + 2500.0 * torch.rand((n, ctx.pitch), device=dev)) ** 2
        st = {"ip": 0, "ipp": 1}
        def go():
            st["ip"], st["ipp"] = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), None, 0, 0, 0, 8, True, st["ip"], st["ipp"], stream=s)
        ms = min(timeit(go) for _ in range(2)) / 8
        g = n * n / ms / 1e6
        print(f"TB{mode if mode > 1 else 2} n={n} xchunk={xchunk:3d}: {ms*1e3:8.1f} us/step  {g:7.1f} Gpt/s  {g*16/1e3:5.2f} TB/s-equivalent ({g*16/8000*100:5.1f}% of 8 TB/s at 16 B/pt)", flush=True)
        del bufs, v2, ctx
