#!/usr/bin/env python3
"""Wall time of ./rtm_code on a new_mod-sized synthetic job (415x295 extended, nt = 1700, 12 shots) against the number of shot workers
(host threads, each with its own context and stream on the one GPU).  Development tool; numbers in DESIGN.md section 6e."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
exe = os.path.join(ROOT, "parallel_finite_difference_computation_amd", "bin", "rtm_code")
nx, nz, nxb, nzb, nt, ns = 315, 195, 50, 50, 1700, int(sys.argv[1]) if len(sys.argv) > 1 else 12
with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "gpurun_out")) as d:
    os.makedirs(os.path.join(d, "out"))
    rng = np.random.default_rng(0)
    (1500 + 2500 * rng.random((nx, nz))).astype(np.float32).tofile(os.path.join(d, "vp.bin"))
    rng.standard_normal((ns, nx, nt)).astype(np.float32).tofile(os.path.join(d, "dobs.bin"))
    open(os.path.join(d, "input.dat"), "w").write(
        f"tmpdir=./out\nvpfile=./vp.bin\ndatfile=./dobs.bin\nnz={nz}\nnx={nx}\nnt={nt}\ndz=10\ndx=10\ndt=0.001\nfpeak=25.\nns={ns}\nsz=1\nfsx=5\nds=20\ngz=2\n"
        f"nxb={nxb}\nnzb={nzb}\nrnd=1\nfac=0.75\norder=8\n")
    ref = None
    for workers in (1, 2, 4, 6, 8, 12):
        best = 1e9
        for rep in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "./input.dat"], cwd=d, capture_output=True, text=True, env=dict(os.environ, FDW_SHOT_WORKERS=str(workers), FDW_TIMING="1"))
            best = min(best, time.perf_counter() - t0)
            assert r.returncode == 0, r.stderr
        img = open(os.path.join(d, "out", "dir.image"), "rb").read()
        ref = ref or img
        print(f"workers {workers:2d}: {best:6.3f} s for {ns} shots = {best / ns * 1e3:6.2f} ms/shot, image identical: {img == ref}  {r.stderr.strip()}", flush=True)
