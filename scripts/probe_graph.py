#!/usr/bin/env python3
"""Single small shot: the forward loop of a new_mod-sized deck (415 x 295, 1700 dependent launches) eager vs captured in a HIP graph
(development tool).  Capture + instantiate are timed too: a shot's launches differ from the next shot's (source row,
model), so a graph is built per shot unless its nodes are updated."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import parallel_finite_difference_computation_amd as F
dev = torch.device("cuda:0")
nxe, nze, nb, nt = 415, 295, 50, 1700
ctx = F.FDWave(8, nxe, nze, nb, nb, nt, 0.75, 10.0, 10.0, 1e-3, compat=True)
p = torch.zeros((nxe, ctx.pitch), device=dev); pp = torch.zeros_like(p)
v2 = torch.zeros_like(p); v2[:, :nze] = 2500.0 ** 2
srce = torch.from_numpy(F.ricker_wavelet(nt, 1e-3, 20.0)).to(dev)
s = torch.cuda.Stream()
torch.cuda.synchronize()
def loop():
    ctx.dev_steps(p.data_ptr(), pp.data_ptr(), v2.data_ptr(), srce.data_ptr(), nb + 150, nb, 0, nt, False, stream=s.cuda_stream)
def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); s.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
t_eager = timed(loop)
print(f"eager: {nt} launches in {t_eager * 1e3:.2f} ms = {t_eager / nt * 1e6:.2f} us per step", flush=True)
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.graph(g, stream=s):
    loop()
torch.cuda.synchronize()
t_build = time.perf_counter() - t0
def replay():
    with torch.cuda.stream(s):
        g.replay()
t_first = timed(replay, 1)
t_replay = timed(replay)
print(f"graph: capture + instantiate {t_build * 1e3:.2f} ms, first replay {t_first * 1e3:.2f} ms, replay {t_replay * 1e3:.2f} ms = {t_replay / nt * 1e6:.2f} us per step", flush=True)
print(f"one shot built and run once: {(t_build + t_first) * 1e3:.2f} ms vs eager {t_eager * 1e3:.2f} ms", flush=True)
