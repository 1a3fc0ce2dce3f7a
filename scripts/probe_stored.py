#!/usr/bin/env python3
"""The stored-wavefield RTM of the CPU-serial sibling (fdw_rtm_stored_shot) on a 3lay_mod-sized deck (251 x 251 extended,
nt = 1001) and a larger one: wall time per shot; under `rocprofv3 --kernel-trace --stats` the share of the per-step field copy (swf[it] = P)
against the step kernels (development tool; VERDICT r1 weak point 9)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import parallel_finite_difference_computation_amd as F

for nx, nb, nt in ((151, 50, 1001), (1024, 64, 400)):
    nxe = nx + 2 * nb
    rng = np.random.default_rng(nx)
    v2 = np.full((nxe, nxe), 2500.0 ** 2, np.float32)
    srce = F.mod_ricker_wavelet(nt, 0.001, 30.0)
    dobs = rng.standard_normal((1, nx, nt)).astype(np.float32)
    ctx = F.FDWave(8, nxe, nxe, nb, nb, nt, 0.01, 10.0, 10.0, 0.001, dialect=2)
    ctx.rtm_stored_shot(v2, nxe // 2, nb + 1, nb + 2, srce, dobs, shot=0)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.rtm_stored_shot(v2, nxe // 2, nb + 1, nb + 2, srce, dobs, shot=0)
    dt = (time.perf_counter() - t0) / 3
    print(f"{nxe}x{nxe} nt={nt}: {dt * 1e3:8.2f} ms per shot = {dt / (2 * nt) * 1e6:6.2f} us per step (forward + backward)", flush=True)
