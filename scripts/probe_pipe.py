#!/usr/bin/env python3
"""Wave-pipeline kernel (four steps per pass) on the bench grid, a slab of an 8-way split and 4096^2 (development tool; FDW_LIB selects a variant build)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import parallel_finite_difference_computation_amd as F
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); s = ts.cuda_stream
def timeit(fn, n=12, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cases = [((8192, 8192), (173, 123)), ((1152, 8192), (43, 33, 63)), ((4096, 4096), (83,)), ((16384, 16384), (253,))]
if os.environ.get("PIPE_SHAPE"):        # e.g. PIPE_SHAPE=4128x8192 PIPE_CHUNKS=83,103,123
    nx_, nz_ = (int(v) for v in os.environ["PIPE_SHAPE"].split("x"))
    cases = [((nx_, nz_), (43, 83, 123, 173))]
    sys.argv = sys.argv[:1]
if os.environ.get("PIPE_CHUNKS"):
    cases = [(c[0], tuple(int(x) for x in os.environ["PIPE_CHUNKS"].split(","))) for c in cases]
if len(sys.argv) > 1:
    cases = [c for c in cases if str(c[0][0]) in sys.argv[1:]]
for (nx, nz), chunks in cases:
    for xchunk in chunks:
        ctx = F.FDWave(8, nx, nz, 64, 64, 100, 0.75, 10.0, 10.0, 0.001, compat=False, numerics=int(os.environ.get("PIPE_NUMERICS", "0")))
        ctx.set_tuning(xchunk=xchunk, two_step=int(os.environ.get("PIPE_MODE", "4")))
        bufs = [torch.randn((nx, ctx.pitch), device=dev) * 1e-3 for _ in range(4)]
        for b in bufs:
            b[:, nz:] = 0
        v2 = torch.zeros((nx, ctx.pitch), device=dev)
        v2[:, :nz] = (1500.0 + 2500.0 * torch.rand((nx, nz), device=dev)) ** 2
        st = {"ip": 0, "ipp": 1}
        def go():
            st["ip"], st["ipp"] = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), None, 0, 0, 0, 32, True, st["ip"], st["ipp"], stream=s)
        ms = min(timeit(go) for _ in range(3)) / 32
        print(f"{os.path.basename(F.LIB_PATH):28s} {nx}x{nz} xchunk={xchunk:3d}: {ms*1e3:7.2f} us/step  {nx * nz / ms / 1e6:7.1f} Gpt/s", flush=True)
        del bufs, v2, ctx
