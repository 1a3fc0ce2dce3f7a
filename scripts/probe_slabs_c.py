#!/usr/bin/env python3
"""What ONE rank of an N-way x-slab decomposition of the 8192^2 bench grid costs without its links (development tool): the C slab driver
(fdw_slabs_dev_forward / fdw_slabs_dev_back) on the geometry of a middle rank with a communicator whose exchanges move nothing, so the
figure is compute + host enqueue + stream choreography -- the ceiling ideal links would give.  Whole-job Gpoints/s = n^2 K / t.
    python3 scripts/probe_slabs_c.py [n] [K]        (PROBE_NUMERICS=1: FAST numerics)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import parallel_finite_difference_computation_amd as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 320
dev = torch.device("cuda:0")
NB = 64


def run(world, ksteps, back):
    comm = F.Comm.stub(world // 2, world) if world > 1 else None
    sl = F.Slabs(8, n, n, NB, NB, K, 0.75, 10.0, 10.0, 1e-3, comm=comm, compat=False, ksteps=ksteps, numerics=int(os.environ.get("PROBE_NUMERICS", "0")))
    nfb, nrb = sl.back_buffers()
    fl = [1e-3 * torch.randn((sl.nxl, sl.pitch), device=dev) for _ in range(max(sl.nbuf, nfb) + nrb)]
    for f in fl:
        f[:, n:] = 0
    v2 = torch.zeros((sl.nxl, sl.pitch), device=dev)
    v2[:, :n] = 2500.0 ** 2
    img = torch.zeros((sl.nxl, sl.pitch), device=dev)
    srce = torch.zeros(K, device=dev)
    smp = torch.randn((K, n - 2 * NB), device=dev)
    ptrs = [f.data_ptr() for f in fl[:sl.nbuf]]
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        if back:
            sl.dev_back([f.data_ptr() for f in fl[:max(sl.nbuf, nfb)]], [f.data_ptr() for f in fl[max(sl.nbuf, nfb):]], v2.data_ptr(), smp.data_ptr(), NB + 3, img.data_ptr(), 2, K - 2)
        else:
            sl.dev_forward(ptrs, v2.data_ptr(), srce.data_ptr(), n // 2, n // 2, 0, K, True, 0, 1)
        sl.synchronize()
        best = min(best, time.perf_counter() - t0)
    steps = K - 2 if back else K
    tag = f"N={world} rows/rank={sl.own1 - sl.own0} nxl={sl.nxl} ksteps={sl.ksteps} nbuf={sl.nbuf}"
    print(f"{'backward' if back else 'forward '} {tag}: {best / steps * 1e6:8.1f} us/step  whole job {n * n * steps / best / 1e9:8.1f} Gpoints/s", flush=True)
    sl.close()
    if comm is not None:
        comm.close()
    return n * n * steps / best / 1e9


for back in ((True,) if os.environ.get("ONLY_BACK") else (False, True)):
    base = run(1, 0, back)
    for world in (tuple(int(x) for x in os.environ["ONLY_WORLD"].split(",")) if os.environ.get("ONLY_WORLD") else (2, 4, 8)):
        for k in (tuple(int(x) for x in os.environ["KSTEPS"].split(",")) if os.environ.get("KSTEPS") else ((0,) if not back else (16,))):
            v = run(world, k, back)
            print(f"    -> {v / base:5.2f} x the one-GPU figure", flush=True)
