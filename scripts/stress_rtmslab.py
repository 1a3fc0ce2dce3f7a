#!/usr/bin/env python3
"""Repeats the single-rank shot of `bench.py --workload rtm-slab` (C slab driver: forward K steps, hand-over, backward K iterations with
imaging) from the same start and checks after every phase that the fields / the image are finite and IDENTICAL to the first repetition's
(development tool: a race or a read of memory nobody wrote shows up as a repetition that differs).
    python3 scripts/stress_rtmslab.py [n] [K] [reps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import parallel_finite_difference_computation_amd as F
from bench import DT, DX, FAC, FPEAK, NB, ORDER, synthetic_velocity_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda:0")
nx, gz, sx, sz = n - 2 * NB, NB + 3, n // 2, NB + 2
sl = F.Slabs(ORDER, n, n, NB, NB, K, FAC, DX, DX, DT, comm=None, compat=False, numerics=int(os.environ.get("STRESS_NUMERICS", "0")))      # STRESS_NUMERICS=1: the FAST instantiations
nfb, nrb = sl.back_buffers()
nsrc = max(sl.nbuf, nfb)
srce = torch.from_numpy(F.ricker_wavelet(K, DT, FPEAK)).to(dev)
g = torch.Generator(device=dev)
g.manual_seed(5)
samples = torch.randn((K, nx), device=dev, generator=g)
noise = [1e-3 * torch.randn((n, n), device=dev, generator=g) for _ in range(2)]
fld = [torch.zeros((n, sl.pitch), device=dev) for _ in range(nsrc + nrb)]
v2 = torch.zeros((n, sl.pitch), device=dev)
v2[:, :n] = synthetic_velocity_rows(n, 0, n, dev)
img = torch.zeros((n, sl.pitch), device=dev)
first = {}
bad = 0
for rep in range(reps):
    for f in fld:
        f.zero_()
    fld[0][:, :n] = noise[0]
    fld[1][:, :n] = noise[1]
    img.zero_()
    torch.cuda.synchronize()
    ip, ipp = sl.dev_forward([f.data_ptr() for f in fld[:sl.nbuf]], v2.data_ptr(), srce.data_ptr(), sx, sz, 0, K, True, 0, 1)
    sl.taper_finalize(fld[ip].data_ptr())
    sl.synchronize()
    state = {"P": fld[ip].clone(), "PP": fld[ipp].clone()}
    rcv = fld[nsrc:]
    role = sl.dev_back([f.data_ptr() for f in fld[:nsrc]], [r.data_ptr() for r in rcv], v2.data_ptr(), samples.data_ptr(), gz, img.data_ptr(), 0, K,
                       role=(ip, ipp, 0, 1))
    sl.synchronize()
    torch.cuda.synchronize()
    state["img"] = img.clone()
    state["F1"], state["F0"], state["R1"], state["R0"] = fld[role[0]].clone(), fld[role[1]].clone(), rcv[role[2]].clone(), rcv[role[3]].clone()
    msg = []
    for k, t in state.items():
        fin = bool(torch.isfinite(t).all().item())
        if rep == 0:
            first[k] = t
            same = True
        else:
            same = bool(torch.equal(t, first[k])) if fin else False
        if not fin or not same:
            where = torch.nonzero(~torch.isfinite(t) if not fin else (t != first[k]))
            dd = (t - first[k]).abs()
            d = float(dd.max().item()) if fin and rep else float("nan")
            at = divmod(int(dd.argmax().item()), t.shape[1]) if fin and rep else None
            msg.append(f"{k}: {where.shape[0]} cells differ, rows {int(where[:, 0].min())}..{int(where[:, 0].max())} cols {int(where[:, 1].min())}..{int(where[:, 1].max())}, "
                       f"max |diff| {d:.3e} at {at} (value there {float(first[k][at].item()) if at else 0:.3e}, max |value| {float(first[k].abs().max().item()):.3e})")
    bad += bool(msg)
    print(f"rep {rep}: " + ("ok" if not msg else " | ".join(msg)), flush=True)
print("FAILED" if bad else "all repetitions identical")
sys.exit(1 if bad else 0)
