#!/usr/bin/env python3
"""A grid whose fields exceed 2 GiB (the multi-step kernels address a field through one buffer descriptor and stay off there): the one-step
kernels on a 24576 x 24576 grid (2.25 GiB per field) against the oracle, a few steps (development tool; the test of the same name does this)."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import parallel_finite_difference_computation_amd as F
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
nb, nt = 64, 3
rng = np.random.default_rng(1)
v2 = ((1500.0 + 2500.0 * rng.random((n, n), dtype=np.float32)) ** 2).astype(np.float32)
p0 = 1e-3 * rng.standard_normal((n, n), dtype=np.float32)
pp0 = 1e-3 * rng.standard_normal((n, n), dtype=np.float32)
srce = (O.ricker_wavelet(nt, 0.001, 30.0) + 0.25).astype(np.float32)
sx, sz = n - 200, n // 3
ctx = F.FDWave(8, n, n, nb, nb, nt, 0.75, 10.0, 10.0, 0.001, compat=False)
print("steps per pass", ctx.steps_per_pass(), "pitch", ctx.pitch, flush=True)
t0 = time.time()
P, PP = ctx.forward(v2, sx, sz, srce, p0, pp0)
print("gpu forward", time.time() - t0, flush=True)
orc = O.Oracle(8, n, n, nb, nb, nt, 0.75, 10.0, 10.0, 0.001, compat=False, omp=True)
t0 = time.time()
oP, oPP = orc.forward(v2, sx, sz, srce, p0, pp0)
print("oracle forward", time.time() - t0, flush=True)
print("PP equal:", np.array_equal(PP.view(np.uint32), oPP.view(np.uint32)), " P equal:", np.array_equal(P.view(np.uint32), oP.view(np.uint32)),
      " max", float(np.abs(oPP).max()), flush=True)
if not np.array_equal(PP.view(np.uint32), oPP.view(np.uint32)):
    bad = np.argwhere(PP.view(np.uint32) != oPP.view(np.uint32))
    print("differing cells", len(bad), "first", bad[:5], "last", bad[-5:])
