import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import parallel_finite_difference_computation_amd as F
from parallel_finite_difference_computation_amd.decomp import HipSlabStepper, SlabForward, SlabGeometry
n, K = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
geom = SlabGeometry(0, 1, n, 4, 1)
ctx = F.FDWave(8, n, n, 64, 64, K + 100, 0.75, 10.0, 10.0, 1e-3, compat=False)
a = torch.zeros((n, ctx.pitch), device=dev); b = torch.zeros_like(a); v2 = torch.full_like(a, 4.0e6)
srce = torch.from_numpy(F.ricker_wavelet(K + 100, 1e-3, 20.0)).to(dev)
fw = SlabForward(geom, HipSlabStepper(ctx), (a, b), v2, srce, n // 2, n // 2)
fw.run(50); torch.cuda.synchronize()
for rep in range(3):
    e0, e1, em = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0 = time.perf_counter()
    e0.record(fw.compute)
    fw.run(1); em.record(fw.compute)
    fw.run(K - 1)
    e1.record(fw.compute)
    t1 = time.perf_counter()
    while not e1.query(): pass
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"K={K} enqueue {1e3*(t1-t0):.2f} ms  done {1e3*(t2-t0):.2f} ms  synced {1e3*(t3-t0):.2f} ms  dev {e0.elapsed_time(e1):.2f} ms  first step {e0.elapsed_time(em)*1e3:.1f} us")
