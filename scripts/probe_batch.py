#!/usr/bin/env python3
"""Throughput of new_mod-sized RTM shots against the batch size of fdw_shot_batch (development tool; numbers in DESIGN.md section 6e)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import parallel_finite_difference_computation_amd as F
nx, nz, nxb, nzb, nt = 415, 295, 40, 40, 1700
nxe, nze = nx + 2 * nxb, nz + 2 * nzb
rng = np.random.default_rng(0)
ctx = F.FDWave(8, nxe, nze, nxb, nzb, nt, 0.75, 10.0, 10.0, 1e-3, compat=True)
ctx.model_resident((1500 + 2500 * rng.random((nx, nz))).astype(np.float32))
srce = F.ricker_wavelet(nt, 1e-3, 25.0)
print("shot_batch_max:", ctx.shot_batch_max(), flush=True)
for n in (1, 2, 4, 8, 16, 32):
    d_obs = rng.standard_normal((n, nx, nt)).astype(np.float32)
    ctx.shot_batch(n, nxb + 5, 3, nzb + 1, nzb + 2, srce, d_obs)
    t0 = time.perf_counter(); ctx.shot_batch(n, nxb + 5, 3, nzb + 1, nzb + 2, srce, d_obs); dt = time.perf_counter() - t0
    print(f"batch {n:2d}: {dt * 1e3:8.2f} ms = {dt / n * 1e3:6.2f} ms/shot, {3.0 * nt * nxe * nze * n / dt / 1e9:7.1f} Gpoints/s", flush=True)
if len(sys.argv) > 1:       # chunk length sweep at fixed batch sizes
    for n in (8, 16, 32):
        d_obs = rng.standard_normal((n, nx, nt)).astype(np.float32)
        for xchunk in (1, 2, 3, 4, 6, 8, 12):
            ctx.set_tuning(xchunk=xchunk)
            ctx.shot_batch(n, nxb + 5, 3, nzb + 1, nzb + 2, srce, d_obs)
            t0 = time.perf_counter(); ctx.shot_batch(n, nxb + 5, 3, nzb + 1, nzb + 2, srce, d_obs); dt = time.perf_counter() - t0
            print(f"batch {n:2d} xchunk {xchunk:2d}: {dt / n * 1e3:6.2f} ms/shot", flush=True)
