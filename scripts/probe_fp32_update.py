#!/usr/bin/env python3
"""What the fp64 leap-frog costs and buys (development tool).  Runs the bench's forward loop at 8192^2 with the library named by FDW_LIB and
writes the final field to gpurun_out/; with a second argument compares two such dumps.
    FDW_LIB=ablate/libfdwave_fp32upd.so python3 scripts/probe_fp32_update.py fp32     (a variant built with the update in fp32)
    python3 scripts/probe_fp32_update.py fp64
    python3 scripts/probe_fp32_update.py compare fp32 fp64"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
os.makedirs(out, exist_ok=True)
if sys.argv[1] == "compare":
    for K in (200, 1000, 1700):
        a, b = (np.load(os.path.join(out, f"fp32probe_{t}_{K}.npy")) for t in sys.argv[2:4])
        d = np.abs(a.astype(np.float64) - b)
        print(f"{K:5d} steps: max |{sys.argv[2]} - {sys.argv[3]}| = {d.max():.3e}, max |field| = {np.abs(b).max():.3e}, max-norm relative {d.max() / np.abs(b).max():.3e}, "
              f"rms relative {np.sqrt((d ** 2).mean()) / np.sqrt((b.astype(np.float64) ** 2).mean()):.3e}")
    sys.exit(0)
import parallel_finite_difference_computation_amd as F
from bench import DT, DX, FAC, FPEAK, NB, ORDER, synthetic_velocity_rows
tag = sys.argv[1]
n, dev = 8192, torch.device("cuda:0")
ctx = F.FDWave(ORDER, n, n, NB, NB, 1700, FAC, DX, DX, DT, compat=False)
v2 = torch.zeros((n, ctx.pitch), device=dev)
v2[:, :n] = synthetic_velocity_rows(n, 0, n, dev)
srce = torch.from_numpy(F.ricker_wavelet(1700, DT, FPEAK)).to(dev)
for K in (200, 1000, 1700):
    bufs = [torch.zeros((n, ctx.pitch), device=dev) for _ in range(4)]      # at rest + source: the physical case (a Ricker pulse spreading)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.Stream()
    e0.record(st)
    ip, ipp = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), srce.data_ptr(), n // 2, n // 2, 0, K, False, 0, 1, stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    np.save(os.path.join(out, f"fp32probe_{tag}_{K}.npy"), bufs[ipp][4000:4200, :n].cpu().numpy())      # a band of rows through the source
    print(f"{os.path.basename(F.LIB_PATH)} {K:5d} steps: {ms / K * 1e3:7.2f} us/step = {n * n * K / ms / 1e6:6.1f} Gpoints/s", flush=True)
