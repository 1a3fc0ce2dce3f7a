#!/usr/bin/env python3
"""mod_main-sized modelling shots (151x151 interior, borders 50, nt = 1001): one shot per call against fdw_model_shot_batch."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import parallel_finite_difference_computation_amd as F
nx, nz, nxb, nzb, nt = 151, 151, 50, 50, 1001
nxe, nze = nx + 2 * nxb, nz + 2 * nzb
v2 = np.zeros((nxe, nze), np.float32); v2[nxb:nxb + nx, nzb:nzb + nz] = 2000.0 ** 2
v2 = F.mod_extendvel(v2, nx, nz, nxb, nzb)
srce = F.mod_ricker_wavelet(nt, 0.001, 30.0)
ctx = F.FDWave(8, nxe, nze, nxb, nzb, nt, 0.01, 10.0, 10.0, 0.001, dialect=1)
print("shot_batch_max:", ctx.shot_batch_max(), flush=True)
ctx.model_shot(v2, nxb + 3, nzb, nzb, srce)
t0 = time.perf_counter(); ctx.model_shot(v2, nxb + 3, nzb, nzb, srce); a = time.perf_counter() - t0
print(f"one shot per call: {a * 1e3:7.2f} ms/shot", flush=True)
for n in (4, 16, 32, 64):
    ctx.model_shot_batch(n, v2, nxb + 3, 2, nzb, nzb, srce)
    t0 = time.perf_counter(); ctx.model_shot_batch(n, v2, nxb + 3, 2, nzb, nzb, srce); b = time.perf_counter() - t0
    print(f"batch {n:2d}: {b / n * 1e3:7.2f} ms/shot ({nt * nxe * nze * n / b / 1e9:6.1f} Gpoints/s)", flush=True)
