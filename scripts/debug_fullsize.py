import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import parallel_finite_difference_computation_amd as F
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
nb, nt = 64, 16
ctx = F.FDWave(8, n, n, nb, nb, nt, 0.75, 10.0, 10.0, 0.001, compat=False)
pitch = ctx.pitch
g = torch.Generator(device=dev); g.manual_seed(7)
init = [torch.zeros((n, pitch), device=dev) for _ in range(2)]
for t in init: t[:, :n] = 1e-3 * torch.randn((n, n), device=dev, generator=g)
v2 = torch.zeros((n, pitch), device=dev); v2[:, :n] = (1500.0 + 2500.0 * torch.rand((n, n), device=dev, generator=g)) ** 2
srce = torch.full((nt,), 0.25, device=dev)
def run(mode, xchunk=0):
    ctx.set_tuning(two_step=mode, xchunk=xchunk)
    bufs = [torch.zeros((n, pitch), device=dev) for _ in range(4)]
    bufs[0].copy_(init[0]); bufs[1].copy_(init[1])
    torch.cuda.synchronize()   # the library launches on its own non-blocking stream
    ip, ipp = ctx.dev_steps2([b.data_ptr() for b in bufs], v2.data_ptr(), srce.data_ptr(), n // 2 + 3, n // 3, 0, nsteps, False, 0, 1)
    torch.cuda.synchronize()
    return bufs[ip], bufs[ipp]
rp, rpp = run(-1)
for mode, xc in ((1, 0), (1, 22), (1, 12), (4, 0), (4, 43)):
    p, pp = run(mode, xc)
    for name, a, b in (("p", p, rp), ("pp", pp, rpp)):
        d = (a != b)
        if d.any():
            rows = d.any(dim=1).nonzero().flatten(); cols = d.any(dim=0).nonzero().flatten()
            print(f"mode {mode} xchunk {xc} {name}: {int(d.sum())} cells differ; rows {int(rows.min())}..{int(rows.max())} ({rows.numel()}), cols {int(cols.min())}..{int(cols.max())} ({cols.numel()})")
        else:
            print(f"mode {mode} xchunk {xc} {name}: identical")
