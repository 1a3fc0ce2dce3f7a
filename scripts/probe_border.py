#!/usr/bin/env python3
"""Per-shot cost of the border velocity model: host loop (extendvel_linear + square + upload inside fdw_shot) against the device generator
(fdw_dev_extendvel_linear), and whole shots either way (development tool; numbers quoted in DESIGN.md section 6e)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import parallel_finite_difference_computation_amd as F

nx, nz, nxb, nzb, nt = (int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (415, 295, 40, 40, 1700)))
nxe, nze = nx + 2 * nxb, nz + 2 * nzb
rng = np.random.default_rng(0)
vp = (1500 + 2500 * rng.random((nx, nz))).astype(np.float32)
srce = F.ricker_wavelet(nt, 1e-3, 25.0)
d_obs = rng.standard_normal((nx, nt)).astype(np.float32)
ctx = F.FDWave(8, nxe, nze, nxb, nzb, nt, 0.75, 10.0, 10.0, 1e-3, compat=True)
ctx.model_resident(vp)
T = ctx.border_draws()
vpe = np.zeros((nxe, nze), np.float32); vpe[nxb:nxb + nx, nzb:nzb + nz] = vp
ctx.dev_extendvel_linear(0, want_vel=True)

def host_model():
    F.extendvel_linear(vpe, nx, nz, nxb, nzb)
    return vpe * vpe

reps = 20
t0 = time.perf_counter()
for s in range(reps): host_model()
th = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for s in range(reps): ctx.dev_extendvel_linear(s * T)
ctx.rand_stream(0, 1)                                   # drains the context's stream
td = (time.perf_counter() - t0) / reps
print(f"{nx}x{nz} border {nxb}/{nzb}: {T} draws per shot; host model {th * 1e6:8.1f} us, device model {td * 1e6:8.1f} us", flush=True)
for rep in range(3):
    t0 = time.perf_counter(); ctx.shot(host_model(), nxb + 5, nzb + 1, nzb + 2, srce, d_obs); a = time.perf_counter() - t0
    t0 = time.perf_counter(); ctx.dev_extendvel_linear(rep * T); ctx.shot_resident(nxb + 5, nzb + 1, nzb + 2, srce, d_obs); b = time.perf_counter() - t0
    print(f"  shot with host model + upload {a * 1e3:8.2f} ms, with the device model {b * 1e3:8.2f} ms", flush=True)
