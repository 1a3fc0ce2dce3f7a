#!/usr/bin/env python3
"""A new_mod-sized shot (415 x 295, nt = 1700) through fdw_forward / fdw_back with the one-step, two-step and pipeline kernels forced in turn:
which kernel the launch-bound small-deck regime wants (development tool; the one-step kernel: 8.3 + 10.2 ms, two-step 16 + 29, pipeline 28 + 46)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import parallel_finite_difference_computation_amd as F
nxe, nze, nb, nt = 415, 295, 40, 1700
rng = np.random.default_rng(0)
v2 = ((1500 + 2500 * rng.random((nxe, nze))) ** 2).astype(np.float32)
srce = F.ricker_wavelet(nt, 1e-3, 20.0)
dobs = rng.standard_normal((nxe - 2 * nb, nt)).astype(np.float32)
for mode in (-1, 0, 1, 4):
    ctx = F.FDWave(8, nxe, nze, nb, nb, nt, 0.75, 10.0, 10.0, 1e-3, compat=True)
    ctx.set_tuning(two_step=mode)
    ctx.forward(v2, 200, 45, srce)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); P, PP = ctx.forward(v2, 200, 45, srce); ts.append(time.perf_counter() - t0)
    tb = []
    for _ in range(3):
        t0 = time.perf_counter(); img = ctx.back(v2, P, PP, dobs, 45); tb.append(time.perf_counter() - t0)
    print(f"two_step={mode:2d} steps_per_pass={ctx.steps_per_pass()}: forward {min(ts)*1e3:6.2f} ms  back {min(tb)*1e3:6.2f} ms", flush=True)
