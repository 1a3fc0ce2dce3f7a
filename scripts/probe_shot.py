#!/usr/bin/env python3
"""A few whole shots of the new_mod-sized deck through fdw_shot (profiling target for the launch-bound small-deck regime)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import parallel_finite_difference_computation_amd as F
nx, nz, nxb, nzb, nt = 415, 295, 40, 40, 1700
nxe, nze = nx + 2 * nxb, nz + 2 * nzb
rng = np.random.default_rng(0)
vp = (1500 + 2500 * rng.random((nx, nz))).astype(np.float32)
srce = F.ricker_wavelet(nt, 1e-3, 25.0)
d_obs = rng.standard_normal((nx, nt)).astype(np.float32)
ctx = F.FDWave(8, nxe, nze, nxb, nzb, nt, 0.75, 10.0, 10.0, 1e-3, compat=True)
ctx.model_resident(vp)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    t0 = time.perf_counter(); ctx.dev_extendvel_linear(rep * ctx.border_draws()); ctx.shot_resident(nxb + 5, nzb + 1, nzb + 2, srce, d_obs)
    print(f"shot {(time.perf_counter() - t0) * 1e3:8.2f} ms", flush=True)
