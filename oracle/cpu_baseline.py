"""CPU baseline of bench.py (TEST INFRASTRUCTURE, like everything under oracle/): the oracle's fused forward loop (fdw_oracle.c
orc_fused_steps: same arithmetic per point as the GPU step, 16 B/point) timed on the host, one thread and all threads (OpenMP build of
the same file).  Run as its own process so that no other threading runtime (torch's) competes for the cores:
    python oracle/cpu_baseline.py <n> [seconds_per_leg]      -> one JSON object on stdout
    python oracle/cpu_baseline.py rtm <m> [seconds_per_leg]  -> the same for one RTM shot (orc_fd_forward + orc_fd_back with imaging, one pass
                                                              per reference kernel) on an m x m grid: bench.py --workload rtm-slab"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as O  # noqa: E402


def host_threads():
    """Threads this process may really use: the affinity mask, the cgroup CPU quota if there is one, and at most FDW_CPU_THREADS
    (default 16 = the CPU share of a one-GPU box; the machine itself may show hundreds of hardware threads)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("FDW_CPU_THREADS", "16"))))


def rtm_main():
    """One shot -- nts forward steps, the snapshot hand-over, nts backward iterations with receiver injection and imaging -- through the
    oracle's reference-shaped passes on an m x m grid; nts is sized for ~`target` seconds per leg from a 4-step calibration run.
    value = field updates per second (three per time index), the unit of bench.py --workload rtm-slab."""
    os.environ["OMP_NUM_THREADS"] = str(host_threads())
    m = int(sys.argv[2])
    target = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
    nb, dt = 64, 1e-3
    rng = np.random.default_rng(0)
    v2 = ((1500.0 + 2500.0 * np.linspace(0.0, 1.0, m, dtype=np.float32))[None, :] ** 2 * np.ones((m, 1), np.float32)).astype(np.float32)

    def leg(omp, cap):
        def run(nts):
            orc = O.Oracle(8, m, m, nb, nb, nts, 0.75, 10.0, 10.0, dt, compat=False, omp=omp)
            srce = O.ricker_wavelet(nts, dt, 20.0)
            d_obs = rng.standard_normal((m - 2 * nb, nts)).astype(np.float32)
            t0 = time.perf_counter()
            P, PP = orc.forward(v2, m // 2, nb + 2, srce)
            orc.back(v2, P, PP, d_obs, nb + 3)
            return time.perf_counter() - t0
        run(2)                                                # thread team start-up, page faults
        one = run(4) / 4
        nts = int(max(4, min(cap, round(target / max(one, 1e-4)))))
        el = run(nts)
        return 3.0 * m * m * nts / el / 1e9, nts, el

    g1, s1, t1 = leg(False, 60)
    out = {"value": round(g1, 4), "unit": "Gpoints/s", "cores": 1, "kind": "port",
           "sample": f"one RTM shot on a {m}x{m} fp32 grid cut to {s1} forward + {s1} backward iterations with imaging (oracle/fdw_oracle.c orc_fd_forward + "
                     f"orc_fd_back: one pass per reference kernel, gcc -O2 -ffp-contract=off), single thread, {t1:.1f} s"}
    if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liborc_omp.so")) and host_threads() > 1:
        gm, sm, tm = leg(True, 2000)
        out = {"value": round(gm, 4), "unit": "Gpoints/s", "cores": host_threads(), "kind": "port",
               "sample": f"one RTM shot on a {m}x{m} fp32 grid cut to {sm} forward + {sm} backward iterations with imaging (oracle/fdw_oracle.c orc_fd_forward + "
                         f"orc_fd_back: one pass per reference kernel, rows shared out with OpenMP), {host_threads()} threads, {tm:.1f} s; single thread: "
                         f"{g1:.4f} Gpoints/s ({s1} + {s1} iterations, {t1:.1f} s)"}
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "rtm":
        return rtm_main()
    os.environ["OMP_NUM_THREADS"] = str(host_threads())      # read by libgomp when the OpenMP build is loaded
    n = int(sys.argv[1])
    target = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
    order, dx, dt = 8, 10.0, 1e-3
    L = O.lib()
    cx, cz = O.scaled_coefs(order, dx, dx)
    rng = np.random.default_rng(0)
    p0 = (1e-3 * rng.standard_normal((n, n))).astype(np.float32)
    pp0 = (1e-3 * rng.standard_normal((n, n))).astype(np.float32)
    v2 = ((1500.0 + 2500.0 * np.linspace(0.0, 1.0, n, dtype=np.float32))[None, :] ** 2 * np.ones((n, 1), np.float32)).astype(np.float32)
    p, pp = p0.copy(), pp0.copy()

    def leg(lib, cap):
        # warm up for ~2 s of wall time (a freshly started thread team runs at a fraction of its speed for the first second), then
        # take the per-step time of the last warm-up call
        one, t_start = None, time.perf_counter()
        while one is None or time.perf_counter() - t_start < 2.0:
            p[:], pp[:] = p0, pp0
            t0 = time.perf_counter()
            lib.orc_fused_steps(order, n, n, p, pp, v2, cx, cz, dt * dt, 2)
            one = (time.perf_counter() - t0) / 2
        p[:], pp[:] = p0, pp0
        steps = int(max(1, min(cap, round(target / max(one, 1e-4)))))
        t0 = time.perf_counter()
        lib.orc_fused_steps(order, n, n, p, pp, v2, cx, cz, dt * dt, steps)
        el = time.perf_counter() - t0
        return n * n * steps / el / 1e9, steps, el

    g1, s1, t1 = leg(L, 40)
    out = {"value": round(g1, 4), "unit": "Gpoints/s", "cores": 1, "kind": "port",
           "sample": f"{n}x{n} fp32 grid, {s1} fused steps (oracle/fdw_oracle.c orc_fused_steps, gcc -O2 -ffp-contract=off), single thread, {t1:.1f} s"}
    omp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liborc_omp.so")
    if os.path.exists(omp):
        M = C.CDLL(omp)
        M.orc_fused_steps.argtypes = L.orc_fused_steps.argtypes
        M.orc_max_threads.restype = C.c_int
        nthr = int(M.orc_max_threads())
        if nthr > 1:
            gm, sm, tm = leg(M, 2000)
            out = {"value": round(gm, 4), "unit": "Gpoints/s", "cores": nthr, "kind": "port",
                   "sample": f"{n}x{n} fp32 grid, {sm} fused steps (oracle/fdw_oracle.c orc_fused_steps, gcc -O2 -ffp-contract=off -fopenmp), "
                             f"{nthr} threads, {tm:.1f} s; single thread: {g1:.4f} Gpoints/s ({s1} steps, {t1:.1f} s)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
