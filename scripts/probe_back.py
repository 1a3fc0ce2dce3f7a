#!/usr/bin/env python3
"""Backward (source-field reconstruction + receiver step + imaging) time per iteration at 8192^2 through fdw_back (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import parallel_finite_difference_computation_amd as F
n, nb = (int(sys.argv[1]) if len(sys.argv) > 1 else 8192), 64
nt = 400
ctx = F.FDWave(8, n, n, nb, nb, nt, 0.75, 10.0, 10.0, 1e-3, compat=False)
ctx.set_tuning(two_step=int(os.environ.get("TB_MODE", "0")), xchunk=int(os.environ.get("XCHUNK", "0")))    # -1: one-step kernels only (fused backward iteration at any size)
rng = np.random.default_rng(0)
v2 = np.full((n, n), 2500.0 ** 2, np.float32)
s0 = (1e-3 * rng.standard_normal((n, n))).astype(np.float32); s1 = (1e-3 * rng.standard_normal((n, n))).astype(np.float32)
d_obs = rng.standard_normal((n - 2 * nb, nt)).astype(np.float32)
def t(ns):
    t0 = time.perf_counter(); ctx.back(v2, s0, s1, d_obs, nb + 2, nsteps=ns); return time.perf_counter() - t0
t(8)
for rep in range(3):
    a, b = t(8), t(nt)
    print(f"backward iteration: {(b - a) / (nt - 8) * 1e6:7.1f} us", flush=True)
