#!/usr/bin/env python3
"""One rank's share of an N-way slab run, WITHOUT the transfers (exchange is a no-op): compute + host-enqueue time per step of the
slab driver, classic one-step cycles vs four-steps-per-pass cycles (development tool; projects the N-GPU rate an ideal link would give)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))      # decomp_harness: the Python slab drivers are a test harness
import torch
import parallel_finite_difference_computation_amd as F
from decomp_harness import HipSlabStepper, SlabForward, SlabGeometry
dev = torch.device("cuda:0")
n = 8192
for world in (2, 4, 8):
    for pipe in (False, True):
        for k in ((8, 16) if world == 8 else (8,)):
            rank = world // 2 - 1 if world > 2 else 0
            geom = SlabGeometry(rank, world, n, 4, k)
            ctx = F.FDWave(8, n, n, 64, 64, 4000, 0.75, 10.0, 10.0, 1e-3, compat=False, slab=(geom.x_off, geom.nxl))
            nb = 4 if pipe else 2
            fields = [torch.randn((geom.nxl, ctx.pitch), device=dev) * 1e-3 for _ in range(nb)]
            v2 = (1500.0 + 2500.0 * torch.rand((geom.nxl, ctx.pitch), device=dev)) ** 2
            srce = torch.from_numpy(F.ricker_wavelet(4000, 1e-3, 20.0)).to(dev)
            fw = SlabForward(geom, HipSlabStepper(ctx), fields, v2, srce, n // 2, n // 2, overlap=True, pipe_ctx=ctx if pipe else None)
            fw.g.world = 2 if world == 1 else world
            def fake_exchange(wait_compute=True, fw=fw):
                fw.comm.wait_stream(fw.compute)          # the dependency structure stays, the transfer is free
                fw.fresh = True
            fw.exchange = fake_exchange
            fw.run(2 * k); fw.synchronize()
            steps = 20 * k
            t0 = time.perf_counter(); fw.run(steps); t_host = time.perf_counter() - t0; fw.synchronize(); t = time.perf_counter() - t0
            print(f"N={world} rank {rank} slab {geom.nxl}x{n} k={k:2d} {'pipe-4  ' if pipe else 'one-step'}: {t/steps*1e6:6.2f} us/step (host enqueue {t_host/steps*1e6:6.2f}) -> "
                  f"{n*n/(t/steps)/1e9:7.1f} Gpt/s whole job if the links were free", flush=True)
            del fw, fields, v2, ctx
