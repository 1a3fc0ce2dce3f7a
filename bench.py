#!/usr/bin/env python3
"""Headline benchmark: fused 8th-order 2-D acoustic time step (taper + Laplacian + leap-frog + source,
the body of the reference's fd_forward loop, cuda_reference_RTM/src/fd-code.cu:259-267) on a synthetic
grid, device resident.  One JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 8192] [--ksteps 4] [--no-cpu-baseline]

N = 1 : the whole grid on one MI355X.
N > 1 : one rank per GPU; the SAME grid is split into x slabs with deep-halo exchange over RCCL inside
        libfdwave.so (csrc/fdw_slabs.cpp, csrc/fdw_comm.cpp) -- strong scaling, as BASELINE.json's
        "RTM domain decomposition, 8192^2 grid, 2->4->8" config asks.  `python bench.py --gpus N` starts its
        N ranks itself (child processes, before anything touches HIP: self_launch below); launched by
        `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it uses the ranks it is given.
        The line then carries rccl_ranks (world size the communicator reports), halo_exchange (the path
        the halo rows took) and exposed_comm_fraction (the same window re-timed with the transfers
        switched off: 1 - t_stub / t).

Metric: Gpoints/s = nxe*nze*K / wall (barrier + synchronize on both sides, max over ranks).
Timing: the K-step window (barrier + synchronize on both sides) is repeated until 0.3 s have been measured; the MEDIAN window is
reported, so --steps 20 and --steps 1000 give the same Gpoints/s.
roofline: bound = HBM.  `achieved` = HBM-side bytes one launch moves / its duration (HIP events on the launch stream): the bytes are the
rocprofv3 --pmc figure of profiles/traffic.json when that profile was taken on exactly this kernel build (source hash), otherwise the
minimum one launch must move (12 B in + 8 B out per point for a multi-step pass).  SURVEY.md 8(d)'s 16 B/point/step model -- which a
temporally blocked kernel beats by construction -- is reported beside it as `algorithmic_16B_model`, never as the fraction.
cpu_baseline: the oracle's fused C loop (same arithmetic), 1 thread, bounded sample, rank 0, N = 1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _gpus_on_command_line(argv):
    """--gpus N / --gpus=N as the driver passes it (parsed by hand: nothing heavier than the standard library is imported yet)."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of this one (fresh interpreters, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set as torch.distributed.run would) and wait for them.  This runs before torch or libfdwave.so is imported, so the
    parent never touches HIP; nothing is exec'ed over a process that has.  Rank 0 inherits this process's stdout, so its JSON line is the last
    line this command prints; the other ranks' stdout goes to stderr.  A rank that fails takes the others down (they would wait in a barrier
    for ever); exit status = the first non-zero one."""
    import signal
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    limit = float(os.environ.get("FDW_BENCH_LAUNCH_TIMEOUT", "1500"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FDW_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr, start_new_session=True))
    t0, rc, failed_at = time.time(), 0, None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and failed_at is None:
            failed_at, rc = time.time(), bad[0].returncode
        late = time.time() - t0 > limit
        if late or (failed_at is not None and time.time() - failed_at > 15.0):
            for p in procs:                      # exactly the processes started above, by PID (their own sessions: nothing else is signalled)
                if p.poll() is None:
                    try:
                        os.killpg(p.pid, signal.SIGKILL)
                    except OSError:
                        pass
            if late and rc == 0:
                rc = 124
                print(f"[bench] self-launch: the {n} ranks did not finish within {limit:.0f} s; killed", file=sys.stderr, flush=True)
            break
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pass
    for p in procs:
        if rc == 0 and p.returncode not in (0, None):
            rc = p.returncode
    return rc if rc >= 0 else 128 - rc


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and _gpus_on_command_line(sys.argv[1:]) > 1:
    sys.exit(self_launch(_gpus_on_command_line(sys.argv[1:])))

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)

import parallel_finite_difference_computation_amd as F  # noqa: E402
from parallel_finite_difference_computation_amd.decomp import SlabGeometry  # noqa: E402

ALGO_BYTES_PER_POINT = 16.0   # read p, pp, v2 + write pp -- the ONE-step-per-pass byte model
MIN_BYTES_PER_POINT_PER_LAUNCH = 20.0   # what any launch of the forward kernels must move: read u^n, u^{n-1}, v2 (12 B), write two fields (8 B)
#                                         (the one-step kernel: 12 + 4 = the 16 B of the model above)
HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves
ORDER, NB, FAC, DX, DT, FPEAK = 8, 64, 0.75, 10.0, 1.0e-3, 20.0
MIN_TIMED_SECONDS = 0.3       # the K-step window is repeated until this much time has been measured; the median window is reported


KERNEL_SOURCES = {      # the translation units behind each workload's dominant kernel (+ the shared device header and argument structs)
    "forward": ("fdw_stepn.hip", "fdw_step2.hip", "fdw_device.h", "fdw_kernels.h"),
    "forward-fast": ("fdw_stepn.hip", "fdw_step2.hip", "fdw_device.h", "fdw_kernels.h"),      # the same sources, NUM = 1 instantiations (--numerics fast)
    "model": ("fdw_stepn.hip", "fdw_device.h", "fdw_kernels.h"),
    "model-fast": ("fdw_stepn.hip", "fdw_device.h", "fdw_kernels.h"),
    "rtm-slab-fast": ("fdw_stepn.hip", "fdw_device.h", "fdw_kernels.h"),
    "rtm-slab": ("fdw_stepn.hip", "fdw_device.h", "fdw_kernels.h"),
    "stencil": ("fdw_step1.hip", "fdw_device.h", "fdw_kernels.h"),
}


def kernel_source_hash(workload="forward"):
    """Identifies the built kernel: profiles/traffic.json (rocprofv3 --pmc passes taken offline) is only quoted for the sources it was taken on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "parallel_finite_difference_computation_amd", "csrc")
    for f in KERNEL_SOURCES.get(workload, KERNEL_SOURCES["forward"]):
        h.update(f.encode())
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def offline_counters(workload, n, steps_per_launch):
    """The entry of profiles/traffic.json for this workload / size / kernel build, or None (a stale profile is never quoted)."""
    try:
        entries = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return None
    if isinstance(entries, dict):
        entries = [entries]
    want = kernel_source_hash(workload)
    for e in entries:
        if e.get("workload", "forward") == workload and e.get("size") == n and e.get("steps_per_launch", 1) == steps_per_launch and e.get("source_hash") == want:
            return e
    return None


def traffic_source_note(prof):
    return (f"offline profile {prof.get('source')} taken on kernel sources {prof.get('source_hash')} (= this build)" if prof else
            "none: profiles/traffic.json holds no entry for this size / kernel build (a stale profile is never quoted)")


def reference_sibling_cpu(m, nti):
    """The reference's OWN CPU path for this stencil -- fd.c / ptsrc.c / taper.c of dpct_gpu_rtm_domain_division/src, compiled unmodified into
    oracle/_ref/libref_dd.so -- driven through mod_main's loop (mod_main.cpp:147-164: fd_step, ptsrc, taper_apply(PP), taper_apply(P), swap) on an
    m x m grid for nti steps, one thread.  None when that library was not built (reference tree absent at build time)."""
    import ctypes as C
    from oracle import oracle as O
    R = O.ref_dd_lib()
    if R is None:
        return None
    fp = C.POINTER(C.c_float)
    hv2 = np.full((m, m), 2500.0 ** 2, np.float32)
    hs = F.mod_ricker_wavelet(nti, DT, FPEAK)
    P, PP = np.zeros((m, m), np.float32), np.zeros((m, m), np.float32)
    rows = lambda a: (fp * a.shape[0])(*[C.cast(a[i].ctypes.data, fp) for i in range(a.shape[0])])
    rP, rPP, rV = rows(P), rows(PP), rows(hv2)
    R._Z7fd_initiiifff(ORDER, m, m, C.c_float(DX), C.c_float(DX), C.c_float(DT))
    R._Z10taper_initiif(NB, NB, C.c_float(0.01))
    t0 = time.perf_counter()
    for it in range(nti):
        R._Z7fd_stepiPPfS0_S0_ii(ORDER, rP, rPP, rV, m, m)
        R._Z5ptsrciiiifPPf(m // 2, m // 2, m, m, C.c_float(float(hs[it])), rPP)
        R._Z11taper_applyPPfiiii(rPP, m - 2 * NB, m - 2 * NB, NB, NB)
        R._Z11taper_applyPPfiiii(rP, m - 2 * NB, m - 2 * NB, NB, NB)
        rP, rPP = rPP, rP
    dt = time.perf_counter() - t0
    R._Z10fd_destroyv()
    R._Z13taper_destroyv()
    return {"value": round(m * m * nti / dt / 1e9, 4), "unit": "Gpoints/s", "cores": 1, "kind": "reference",
            "sample": f"{m}x{m} fp32 grid, {nti} steps of mod_main's loop through the reference's own fd_step / ptsrc / taper_apply "
                      f"(dpct_gpu_rtm_domain_division/src, g++ -O3 as its Makefiles build them: oracle/_ref/libref_dd.so), single thread, {dt:.1f} s"}


MAX_WINDOWS = [400]      # --max-windows: debugging aid (a short, fixed sequence of shots)


def timed_windows(window, sync_all, world, dev, max_windows=None):
    """Times `window()` (EXACTLY K steps, enqueue only) bracketed by barrier + synchronize on both sides, max over ranks; repeats the window
    until MIN_TIMED_SECONDS have been measured and returns (median wall seconds, median device ms between HIP events, windows).  A 20-step
    window at 8192^2 lasts 2.6 ms -- shorter than the clock ramp and than anything that samples the GPU; its median over ~100 windows equals
    what a 1000-step window gives."""
    import torch.distributed as dist
    max_windows = min(max_windows or MAX_WINDOWS[0], MAX_WINDOWS[0])
    walls, devs = [], []
    total = 0.0
    while True:
        sync_all()
        t0 = time.perf_counter()
        dev_ms = window()
        sync_all()
        wall = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([wall], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())            # every rank sees the same number, so every rank stops after the same window
        walls.append(wall)
        devs.append(dev_ms)
        total += wall
        if total >= MIN_TIMED_SECONDS or len(walls) >= max_windows:
            break
    return float(np.median(walls)), float(np.median(devs)), len(walls)


def shm_communicator(rank, world, local_rank, n):
    """--backend shm: this rank of the library's process transport (fdw_comm_init_shm).  The segment is named after the rendezvous port, which
    every rank of this run shares and no other run does; a box holds the largest message: four fields x 64 ghost rows (16 steps per exchange)."""
    name = f"/fdw_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getuid()}"
    return F.Comm.shm(name, rank, world, local_rank, box_bytes=4 * 64 * ((n + 63) // 64 * 64) * 4)


HALO_PATHS = {"rccl": "RCCL ncclSend/ncclRecv groups issued by libfdwave.so on the communication stream",
              "shm": "libfdwave.so's process transport (fdw_comm_init_shm): halo blocks staged through shared memory, ranks may share a GPU -- a rehearsal, not xGMI",
              "local": "device copies between ranks-as-threads inside libfdwave.so"}


def rccl_rehearsal_main(uid_hex):
    """`python bench.py --rccl-rehearsal <unique id, hex>` (RANK / WORLD_SIZE / LOCAL_RANK in the environment): a SHORT slab run over RCCL in a
    process of its own -- communicator, the self-addressed message, 32 forward steps of a 4096-row grid through the C slab driver (the halo
    exchange of the measurement itself), one all-reduce.  Prints RCCL-REHEARSAL-OK.  The ranks of the measurement start one each
    (rccl_or_fallback) so that a collective library that hangs or crashes on this machine costs the rehearsal, not the measurement."""
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FDW_BENCH_REHEARSAL_HANG") == "1":       # tests: a rehearsal that never comes back
        time.sleep(3600)
    if os.environ.get("FDW_BENCH_SHARE_GPU") == "1":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = F.Comm.rccl(bytes.fromhex(uid_hex), rank, world, local_rank)
    comm.selftest()
    n, nt = 4096, 32
    sl = F.Slabs(ORDER, n, n, NB, NB, nt, FAC, DX, DX, DT, comm=comm, compat=False, device=local_rank)
    fld = [torch.zeros((sl.nxl, sl.pitch), device=dev) for _ in range(sl.nbuf)]
    fld[0][:, :n] = 1e-3 * torch.randn((sl.nxl, n), device=dev)
    v2 = torch.zeros((sl.nxl, sl.pitch), device=dev)
    v2[:, :n] = synthetic_velocity_rows(n, sl.x_off, sl.nxl, dev)
    srce = torch.from_numpy(F.ricker_wavelet(nt, DT, FPEAK)).to(dev)
    torch.cuda.synchronize()
    ip, _ = sl.dev_forward([f_.data_ptr() for f_ in fld], v2.data_ptr(), srce.data_ptr(), n // 2, n // 2, 0, nt)
    sl.synchronize()
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(fld[ip]).all().item())
    total = comm.allreduce(1.0 if ok else 0.0)
    sl.close()
    comm.close()
    if total != float(world):
        sys.exit(f"rccl rehearsal: {total} of {world} ranks finished with finite fields")
    print("RCCL-REHEARSAL-OK", flush=True)


def rccl_rehearsal(rank, world, local_rank, dist):
    """None if a rehearsal of the RCCL halo exchange (rccl_rehearsal_main, one child process per rank, bounded in time) went through on every
    rank, else the reason it did not.  FDW_BENCH_NO_REHEARSAL=1 skips it."""
    import signal
    import subprocess
    if os.environ.get("FDW_BENCH_NO_REHEARSAL") == "1":
        return None
    limit = float(os.environ.get("FDW_BENCH_REHEARSAL_TIMEOUT", "240"))
    why, uid = None, [None]
    if rank == 0:
        try:
            uid = [F.Comm.unique_id().hex()]
        except F.FdwError as e:
            why = f"rank 0 could not create the RCCL unique id: {e}"
    dist.broadcast_object_list(uid, src=0)
    if uid[0] is not None:
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rccl-rehearsal", uid[0]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                 text=True, start_new_session=True)
        try:
            out, _ = child.communicate(timeout=limit)
            if child.returncode != 0 or "RCCL-REHEARSAL-OK" not in out:
                why = f"the rehearsal of rank {rank} ended with status {child.returncode}: {(out.strip().splitlines() or [''])[-1][-300:]}"
        except subprocess.TimeoutExpired:
            try:
                os.killpg(child.pid, signal.SIGKILL)             # exactly the process started above (its own session)
            except OSError:
                pass
            child.wait()
            why = f"the rehearsal of rank {rank} did not finish within {limit:.0f} s"
    reasons = [None] * world
    dist.all_gather_object(reasons, why)
    bad = [r for r in reasons if r]
    return bad[0] if bad else None


def rccl_or_fallback(args, rank, world, local_rank, n, dist):
    """(communicator, reason) of a run on `world` > 1 ranks: RCCL inside libfdwave.so (ncclGetUniqueId on rank 0, the bytes to every rank through
    the control-plane group, ncclCommInitRank, one message to the own rank as a check) after its rehearsal in child processes went through.
    Should RCCL not be usable from the C library on this machine -- an error, a crash or a hang of the rehearsal -- every rank falls back,
    together, to the library's process transport (halo blocks staged through shared memory: slow, flagged in the line; reason = why) rather
    than lose the measurement altogether."""
    if args.backend == "shm":
        return shm_communicator(rank, world, local_rank, n), None
    why = rccl_rehearsal(rank, world, local_rank, dist)
    comm = None
    if why is None:
        err, uid = None, [None]
        if rank == 0:
            try:
                uid = [F.Comm.unique_id()]
            except F.FdwError as e:
                err = e
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is not None:
            try:
                comm = F.Comm.rccl(uid[0], rank, world, local_rank)
                comm.selftest()
            except F.FdwError as e:
                err = e
        else:
            err = err or RuntimeError("rank 0 could not create the RCCL unique id")
        reasons = [None] * world
        dist.all_gather_object(reasons, None if err is None else str(err))
        bad = [r for r in reasons if r]
        why = bad[0] if bad else None
    if why is None:
        return comm, None
    print(f"[bench] rank {rank}: RCCL inside libfdwave.so is not usable here ({why}); falling back to the library's process transport (shared-memory staging)",
          file=sys.stderr, flush=True)
    if comm is not None:
        comm.close()
    return shm_communicator(rank, world, local_rank, n), why


def synthetic_velocity_rows(n, row0, rows, device):
    """v(ix,iz) = 1500 + 2500*iz/(nze-1) with a 3 % lateral sinusoid (m/s)."""
    z = torch.arange(n, device=device, dtype=torch.float32)[None, :]
    x = torch.arange(row0, row0 + rows, device=device, dtype=torch.float32)[:, None]
    v = (1500.0 + 2500.0 * z / (n - 1)) * (1.0 + 0.03 * torch.sin(2.0 * np.pi * 8.0 * x / n))
    return v * v


def cpu_baseline(n, seconds_target=12.0, rtm=False):
    """The oracle's fused loop on the same grid size (rtm: one RTM shot through its reference-shaped passes on an n x n grid), one thread and
    all host threads, ~10-20 s per leg, in a process of its own (oracle/cpu_baseline.py) so that torch's thread pool does not compete for the cores."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py")] + (["rtm"] if rtm else []) + [str(n), str(seconds_target)],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("cpu_baseline failed: " + r.stderr[-2000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def forward_line(n, K, W, dev, numerics=0, init="noise"):
    """A compact forward-loop measurement on one GPU (the headline workload at another size): K steps per window after W warm-up steps from
    the seeded-noise start, median window, HIP-event launch time, roofline from profiles/traffic.json when it holds this size and build."""
    nt = K + W
    ctx = F.FDWave(ORDER, n, n, NB, NB, nt, FAC, DX, DX, DT, compat=False, device=dev.index or 0, numerics=numerics)
    pitch = ctx.pitch
    v2 = torch.zeros((n, pitch), device=dev)
    v2[:, :n] = synthetic_velocity_rows(n, 0, n, dev)
    srce = torch.from_numpy(F.ricker_wavelet(nt, DT, FPEAK)).to(dev)
    bufs = [torch.zeros((n, pitch), device=dev) for _ in range(4)]
    if init == "noise":
        g = torch.Generator(device=dev)
        g.manual_seed(0x5EED0001)
        for b_ in bufs[:2]:
            b_[:, :n] = 1e-3 * torch.randn((n, n), device=dev, generator=g)
    ptrs = [b.data_ptr() for b in bufs]
    stream = torch.cuda.Stream()
    roles = {"ip": 0, "ipp": 1}

    def run(it0, nsteps):
        roles["ip"], roles["ipp"] = ctx.dev_steps2(ptrs, v2.data_ptr(), srce.data_ptr(), n // 2, n // 2, it0, nsteps, it0 > 0, roles["ip"], roles["ipp"],
                                                   stream=stream.cuda_stream)

    torch.cuda.synchronize()
    run(0, W)
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def window():
        e0.record(stream)
        run(W, K)
        e1.record(stream)
        while not e1.query():
            pass
        stream.synchronize()
        return e0.elapsed_time(e1)

    wall, dev_ms, nwin = timed_windows(window, torch.cuda.synchronize, 1, dev)
    newest = bufs[roles["ipp"]]
    ok = bool(torch.isfinite(newest).all().item()) and float(newest.abs().max().item()) > 0.0
    spl = ctx.steps_per_pass()
    rem = K % spl
    launches = K // spl + (rem // 2 + rem % 2 if ctx.two_step_active() else rem)
    launch_s = dev_ms * 1e-3 / launches
    prof = offline_counters("forward-fast" if numerics else "forward", n, spl)
    traffic = prof["hbm_bytes_per_launch"] if prof else None
    min_bytes = (ALGO_BYTES_PER_POINT if spl == 1 else MIN_BYTES_PER_POINT_PER_LAUNCH) * n * n
    basis = traffic if traffic else min_bytes
    return {"value": round(n * n * K / wall / 1e9, 3), "unit": "Gpoints/s", "ms_per_step": round(wall * 1e3 / K, 6), "steps": K, "warmup": W, "windows": nwin,
            "grid": [n, n], "numerics": "fast" if numerics else "exact", "result_finite_nonzero": ok,
            "roofline": {"bound": "hbm", "achieved": round(basis / launch_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(basis / launch_s / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "basis": "measured HBM-side bytes per launch" if traffic else "minimum bytes one launch must move",
                         "traffic_source": traffic_source_note(prof), "launch_us": round(launch_s * 1e6, 2), "steps_per_launch": spl}}


def run_model_workload(args):
    """`--workload model`: the forward-modelling producer (dialect MOD) -- one launch per time step of
    fd_step + ptsrc + both taper_apply + trace recording, device resident; same metric, roofline and CPU baseline fields."""
    import ctypes as C
    n, K, W = args.size, args.steps, args.warmup
    nt = K + W
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    NUM = 1 if args.numerics == "fast" else 0
    ctx = F.FDWave(ORDER, n, n, NB, NB, nt, 0.01, DX, DX, DT, dialect=1, numerics=NUM)
    pitch = ctx.pitch
    v2 = torch.zeros((n, pitch), device=dev)
    v2[:, :n] = synthetic_velocity_rows(n, 0, n, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0003)
    p = torch.zeros((n, pitch), device=dev)
    pp = torch.zeros((n, pitch), device=dev)
    p[:, :n] = 1e-3 * torch.randn((n, n), device=dev, generator=g)
    pp[:, :n] = 1e-3 * torch.randn((n, n), device=dev, generator=g)
    srce = torch.from_numpy(F.mod_ricker_wavelet(nt, DT, FPEAK)).to(dev)
    rec = torch.zeros((nt, n - 2 * NB), device=dev)
    stream = torch.cuda.Stream()
    bufs = [p, pp]
    torch.cuda.synchronize()           # fills ran on torch's default stream

    def run(it0, nsteps):
        ctx.dev_model_steps(bufs[0].data_ptr(), bufs[1].data_ptr(), v2.data_ptr(), srce.data_ptr(), n // 2, n // 2, NB, rec.data_ptr(), it0, nsteps,
                            stream=stream.cuda_stream)
        if nsteps % 2:
            bufs.reverse()

    run(0, W)
    stream.synchronize()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if K % 2:
        sys.exit("bench --workload model: use an even --steps (the two caller-owned buffers swap roles every step)")

    def window():
        e0.record(stream)
        run(W, K)
        e1.record(stream)
        while not e1.query():
            pass
        stream.synchronize()
        return e0.elapsed_time(e1)

    wall, dev_ms, nwin = timed_windows(window, torch.cuda.synchronize, 1, dev)
    spl = ctx.steps_per_pass()                        # 4: wave pipeline (large grids), 1: one-step kernel
    launches = K // spl + K % spl
    launch_s = dev_ms * 1e-3 / launches
    newest = bufs[0]
    finite = bool(torch.isfinite(newest).all().item()) and float(newest.abs().max().item()) > 0.0 and float(rec.abs().max().item()) > 0.0
    algo = ALGO_BYTES_PER_POINT * n * n * spl
    min_bytes = (ALGO_BYTES_PER_POINT if spl == 1 else MIN_BYTES_PER_POINT_PER_LAUNCH) * n * n
    prof = offline_counters("model-fast" if NUM else "model", n, spl)
    traffic = prof["hbm_bytes_per_launch"] if prof else None
    achieved = (traffic if traffic else min_bytes) / launch_s / 1e9
    out = {"metric": "Gpoints/s (stencil updates) + achieved HBM GB/s vs peak", "value": round(n * n * K / wall / 1e9, 3), "unit": "Gpoints/s",
           "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": round(wall * 1e3 / K, 6), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded noise wavefield)",
           "config": {"workload": f"forward-modelling producer (mod_main of the CPU-serial sibling): fd_step + 7x7 Gaussian source + four-sided taper + "
                                  f"trace recording fused in one launch per step, {n}x{n} fp32 extended grid, nxb=nzb={NB}, {K} steps"
                                  + (", FAST numerics (weights carry their spacing, symmetric sums + fused multiply-adds)" if NUM else ""),
                      "grid": [n, n], "order": ORDER, "parallelism": "single"},
           "result_finite_nonzero": finite, "numerics": "fast" if NUM else "exact",
           "timing": {"windows": nwin, "window_steps": K, "statistic": "median window"},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "basis": "measured HBM-side bytes per launch" if traffic else "minimum bytes one launch must move",
                        "traffic_source": traffic_source_note(prof),
                        "kernel": f"fdw::fdw_stepn_kernel<4,4,true,3,2,true,0,{NUM}> (four time steps per launch)" if spl == 4 else "fdw::fdw_step_kernel<4,true,3,false,false,2,true>",
                        "launch_us": round(launch_s * 1e6, 2), "steps_per_launch": spl, "min_bytes_per_launch": min_bytes,
                        "algorithmic_16B_model": {"bytes_per_launch": algo, "achieved": round(algo / launch_s / 1e9, 1),
                                                  "ratio_to_peak": round(algo / launch_s / 1e9 / HBM_PEAK_GBS, 4),
                                                  "note": "SURVEY.md 8(d)'s one-pass-per-step byte model; can exceed 1 for a temporally blocked launch -- not a roofline fraction"}}}
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        m = min(n, 2048)                       # bounded sample on an m x m grid
        hv2 = np.full((m, m), 2500.0 ** 2, np.float32)
        ref = reference_sibling_cpu(m, 200)
        if ref is not None:
            out["cpu_baseline"] = ref
        else:
            nti = 400                              # ~10 s of one core
            hs = F.mod_ricker_wavelet(nti, DT, FPEAK)
            t0 = time.perf_counter()
            O.mod_shot(ORDER, m - 2 * NB, m - 2 * NB, NB, NB, DX, DX, DT, 0.01, hv2, m // 2, m // 2, NB, hs)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": round(m * m * nti / dt / 1e9, 4), "unit": "Gpoints/s", "cores": 1, "kind": "port",
                                   "sample": f"{m}x{m} fp32 grid, {nti} steps of oracle/fdw_oracle_mod.c orc_mod_shot (gcc -O2 -ffp-contract=off), single thread, {dt:.1f} s"}
    print(json.dumps(out), flush=True)
    if not finite:
        sys.exit("bench: result is not finite / all zero")


def run_stencil_workload(args):
    """`--workload stencil`: the whole device work of `stencil_code` (BASELINE.json config 1's program on a synthetic grid): ONE 8th-order
    Laplacian of a field (kernel_lap, fd-source-code.cu:110-135, launched once at S:325), device resident.  A "step" is one Laplacian of the grid;
    algorithmic traffic 8 B/point (read p, write lap)."""
    n, K, W = args.size, args.steps, args.warmup
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ctx = F.FDWave(ORDER, n, n, NB, NB, 0, 1.0, DX, DX, DT, compat=False, coef_cxx=True)
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0004)
    p = torch.zeros((n, ctx.pitch), device=dev)
    p[:, :n] = torch.randn((n, n), device=dev, generator=g)
    lap = torch.zeros((n, ctx.pitch), device=dev)
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(W):
        ctx.dev_laplacian(p.data_ptr(), lap.data_ptr(), stream=stream.cuda_stream)
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def window():
        e0.record(stream)
        for _ in range(K):
            ctx.dev_laplacian(p.data_ptr(), lap.data_ptr(), stream=stream.cuda_stream)
        e1.record(stream)
        while not e1.query():
            pass
        stream.synchronize()
        return e0.elapsed_time(e1)

    wall, dev_ms, nwin = timed_windows(window, torch.cuda.synchronize, 1, dev)
    launch_s = dev_ms * 1e-3 / K
    algo = 8.0 * n * n
    prof = offline_counters("stencil", n, 1)
    traffic = prof["hbm_bytes_per_launch"] if prof else None
    achieved = (traffic if traffic else algo) / launch_s / 1e9
    finite = bool(torch.isfinite(lap).all().item()) and float(lap.abs().max().item()) > 0.0
    out = {"metric": "Gpoints/s (stencil updates) + achieved HBM GB/s vs peak", "value": round(n * n * K / wall / 1e9, 3), "unit": "Gpoints/s",
           "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": round(wall * 1e3 / K, 6), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded noise field)",
           "config": {"workload": f"stencil_code's device work: one 8th-order Laplacian (kernel_lap) of a {n}x{n} fp32 field per step, {K} steps",
                      "grid": [n, n], "order": ORDER, "parallelism": "single"},
           "result_finite_nonzero": finite, "timing": {"windows": nwin, "window_steps": K, "statistic": "median window"},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "basis": "measured HBM-side bytes per launch" if traffic else "algorithmic bytes: 8 B/point (read p, write lap)",
                        "traffic_source": traffic_source_note(prof), "algorithmic_frac": round(algo / launch_s / 1e9 / HBM_PEAK_GBS, 4),
                        "kernel": "fdw::fdw_step_kernel<4,false,0,false,true,2> (Laplacian only)", "launch_us": round(launch_s * 1e6, 2), "steps_per_launch": 1,
                        "algorithmic_bytes_per_launch": algo}}
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        m = min(n, 4096)
        hp = np.random.default_rng(0).standard_normal((m, m), dtype=np.float32)
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0:
            O.stencil(ORDER, m, m, DX, DX, hp)
            reps += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(m * m * reps / dt / 1e9, 4), "unit": "Gpoints/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} Laplacians of a {m}x{m} field, oracle/fdw_oracle.c orc_stencil (gcc -O2 -ffp-contract=off), single thread, {dt:.1f} s"}
    print(json.dumps(out), flush=True)


def run_rtm_workload(args):
    """`--workload rtm`: BASELINE.json's third configuration -- one RTM shot (fd_forward + fd_back with imaging, fd-code.cu:496-518) on the
    reference's own deck size (models/new_mod: 415 x 295 extended grid, nt = 1700) as the drop-in `rtm_code` runs it: the shot's random-border
    models drawn on the device, the gathers uploaded, the images downloaded, and -- a 122k-point grid fills a few percent of the chip -- a batch
    of shots advanced through each launch (fdw_shot_batch, bit-identical to the shots one by one).  value = field updates per second: per time index one
    forward step, one source-field reconstruction step and one receiver step."""
    nxe, nze, nxb, nzb, nt = 415, 295, 50, 50, 1700
    K, W = max(1, args.steps // 100), 1                 # shots timed / warm-up shots (a shot is 3 * nt kernel launches)
    ctx = F.FDWave(ORDER, nxe, nze, nxb, nzb, nt, 0.75, DX, DX, DT, compat=True)
    rng = np.random.default_rng(0)
    ctx.model_resident((1500.0 + 2500.0 * rng.random((nxe - 2 * nxb, nze - 2 * nzb))).astype(np.float32))
    draws = ctx.border_draws()
    srce = F.ricker_wavelet(nt, DT, FPEAK)
    d_obs = rng.standard_normal((nxe - 2 * nxb, nt)).astype(np.float32)

    B = max(1, min(ctx.shot_batch_max(), 32))           # shots the library advances through one launch per time step (small decks)
    K = max(B, (K + B - 1) // B * B)                    # whole batches
    gathers = np.ascontiguousarray(np.broadcast_to(d_obs, (B,) + d_obs.shape))

    def batch(k):                                       # shots k .. k + B - 1 of the reference's loop: sx = fsx + is * ds (fd-code.cu:405-407)
        return ctx.shot_batch(B, nxb + 20, 5, nzb, nzb, srce, gathers, draw_offset=k * draws)

    for k in range(W):
        img = batch(0)
    t0 = time.perf_counter()
    for k in range(0, K, B):
        img = batch(k)
    wall = time.perf_counter() - t0
    upd = 3.0 * nt * nxe * nze * K
    out = {"metric": "Gpoints/s (stencil updates) + achieved HBM GB/s vs peak", "value": round(upd / wall / 1e9, 3), "unit": "Gpoints/s", "n_gpus": 1,
           "steps": K, "warmup": W, "ms_per_step": round(wall * 1e3 / K, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic (random velocity model and gather)",
           "config": {"workload": f"one RTM shot per step (forward {nt} + backward {nt} iterations with imaging) on the new_mod deck size, "
                                  f"{nxe}x{nze} extended grid, border model drawn on the device, gathers in / images out, {B} shots per batch through one launch per time step (fdw_shot_batch), {K} shots", "grid": [nxe, nze], "order": ORDER,
                      "parallelism": "single"},
           "result_finite_nonzero": bool(np.isfinite(img).all() and np.abs(img).max() > 0),
           "roofline": {"bound": "hbm", "achieved": round(upd * 16 / wall / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(upd * 16 / wall / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                        "kernel": "fdw::fdw_step_kernel<4,...> (one-step kernels: a 122k-point grid alone is launch/latency bound, so a batch of shots shares each launch; "
                                  "2 launches per time index: forward step, fused backward iteration)",
                        "launch_us": round(wall * 1e6 / (2.0 * nt * K / B), 2), "steps_per_launch": 1, "shots_per_launch": B,
                        "algorithmic_bytes_per_launch": 16.0 * nxe * nze * B}}
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        nts = 250                                      # bounded sample: forward + backward + imaging of one shot cut to 250 time steps (~10 s of one core)
        orc = O.Oracle(ORDER, nxe, nze, nxb, nzb, nts, 0.75, DX, DX, DT, compat=True)
        hv2 = np.full((nxe, nze), 2500.0 ** 2, np.float32)
        hs = F.ricker_wavelet(nts, DT, FPEAK)
        t0 = time.perf_counter()
        P, PP = orc.forward(hv2, nxb + 20, nzb, hs)
        orc.back(hv2, P, PP, np.ascontiguousarray(d_obs[:, :nts]), nzb)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(3.0 * nts * nxe * nze / dt / 1e9, 4), "unit": "Gpoints/s", "cores": 1, "kind": "port",
                               "sample": f"one shot of the same deck size cut to {nts} forward + {nts} backward iterations with imaging, oracle/fdw_oracle.c "
                                         f"orc_fd_forward + orc_fd_back (one pass per reference kernel, gcc -O2 -ffp-contract=off), single thread, {dt:.1f} s"}
    print(json.dumps(out), flush=True)


def run_rtm_slab_workload(args):
    """`--workload rtm-slab`: BASELINE.json's fourth configuration -- the RTM loops under domain decomposition: ONE shot on an n x n grid split
    into x slabs over the N ranks: K forward steps (fd_forward, fd-code.cu:259-267), the snapshot hand-over, then K backward iterations
    with source-field reconstruction, receiver injection and imaging (fd_back, fd-code.cu:302-339), halo exchanges of two / four fields
    overlapped with the interior rows.  A "step" is one time index of the shot = three field updates; value = 3 n^2 K / wall.
    N = 1 runs the same driver on one slab (the whole grid).  Everything runs inside libfdwave.so (fdw_slabs_*): --backend nccl over RCCL, one
    rank per GPU; --backend shm over the library's process transport, ranks may share one GPU (rehearsals).  The image gathered from the
    slabs is compared bitwise with a single-domain run on rank 0 unless --no-check."""
    from parallel_finite_difference_computation_amd.decomp import slab_bounds
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        if world == 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if args.backend != "nccl" or os.environ.get("FDW_BENCH_SHARE_GPU") == "1":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n, K, W = args.size, args.steps, args.warmup
    nt = max(K, W)
    nx = n - 2 * NB
    gz = NB + 3
    sx, sz = n // 2, NB + 2
    c_driver = True
    NUM = 1 if args.numerics == "fast" else 0
    srce = torch.from_numpy(F.ricker_wavelet(nt, DT, FPEAK)).to(dev)
    gsmp = torch.Generator(device=dev)
    gsmp.manual_seed(0x5EED0005)
    samples = torch.randn((nt, nx), device=dev, generator=gsmp)

    def make_rank(comm_, geom_world, geom_rank):
        """Driver state of one rank: fields with the decomposition-independent noise start, v2, image."""
        sl = F.Slabs(ORDER, n, n, NB, NB, nt, FAC, DX, DX, DT, comm=comm_, compat=False, ksteps=args.ksteps, device=local_rank, numerics=NUM)
        g = SlabGeometry(geom_rank, geom_world, n, ORDER // 2, sl.ksteps)
        assert (g.x_off, g.nxl, g.o0, g.o1) == (sl.x_off, sl.nxl, sl.own0, sl.own1)
        pitch, nbuf = sl.pitch, sl.nbuf
        nfb, nrb = sl.back_buffers()
        nsrc = max(nbuf, nfb)                                                                  # forward buffers double as the backward loop's source-field buffers
        st = dict(sl=sl, g=g, pitch=pitch, nsrc=nsrc)
        st["fld"] = [torch.zeros((g.nxl, pitch), device=dev) for _ in range(nsrc + nrb)]       # ... + the receiver buffers
        gen = torch.Generator(device=dev)
        for f_, seed in ((st["fld"][0], 0x5EED0001), (st["fld"][1], 0x5EED0002)):
            gen.manual_seed(seed)
            f_[:, :n] = (1e-3 * torch.randn((n, 64), device=dev, generator=gen))[g.x_off:g.x_off + g.nxl].repeat(1, n // 64)
        st["v2"] = torch.zeros((g.nxl, pitch), device=dev)
        st["v2"][:, :n] = synthetic_velocity_rows(n, g.x_off, g.nxl, dev)
        st["img"] = torch.zeros((g.nxl, pitch), device=dev)
        st["nbuf"], st["ip"], st["ipp"] = nbuf, 0, 1
        return st

    def shot(st, nsteps):
        """forward nsteps, snapshot hand-over, backward nsteps with imaging (enqueue; the caller synchronises)."""
        fld, rcv = st["fld"], st["fld"][st["nsrc"]:]
        sl = st["sl"]
        st["ip"], st["ipp"] = sl.dev_forward([f_.data_ptr() for f_ in fld[:st["nbuf"]]], st["v2"].data_ptr(), srce.data_ptr(), sx, sz, 0, nsteps, True,
                                             st["ip"], st["ipp"])
        sl.taper_finalize(fld[st["ip"]].data_ptr())                               # the damped d_p the reference hands over (R:285)
        with torch.cuda.stream(torch.cuda.ExternalStream(sl.stream)):
            rcv[0].zero_()
            rcv[1].zero_()                                                        # R:513-514
        sl.dev_back([f_.data_ptr() for f_ in fld[:st["nsrc"]]], [r_.data_ptr() for r_ in rcv], st["v2"].data_ptr(), samples.data_ptr(),
                    gz, st["img"].data_ptr(), 0, nsteps, role=(st["ip"], st["ipp"], 0, 1))

    def sync(st):
        st["sl"].synchronize()
        torch.cuda.synchronize()

    comm = rccl_fallback = None
    if world > 1:
        comm, rccl_fallback = rccl_or_fallback(args, rank, world, local_rank, n, dist)
    me = make_rank(comm, world, rank)
    torch.cuda.synchronize()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    shot(me, W)
    sync(me)

    def window():
        t0 = time.perf_counter()
        shot(me, K)
        sync(me)
        return (time.perf_counter() - t0) * 1e3

    wall, _, nwin = timed_windows(window, sync_all, world, dev)
    g = me["g"]
    img_own = me["img"][g.g_lo:g.nxl - g.g_hi, :n]
    finite = bool(torch.isfinite(img_own).all().item())
    nonzero = float(img_own.abs().max().item()) > 0.0
    if world > 1:
        f = torch.tensor([1.0 if finite else 0.0, 1.0 if nonzero else 0.0])
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        nonzero = bool(f[1].item() > 0.5)
        f2 = torch.tensor([1.0 if finite else 0.0])
        dist.all_reduce(f2, op=dist.ReduceOp.MIN)
        finite = bool(f2.item() > 0.5)
    check = None
    check_failed = False
    if world > 1 and not args.no_check:
        own = img_own.contiguous().cpu()
        if rank == 0:
            parts = [torch.empty(((b1 - b0), n)) for (b0, b1) in slab_bounds(n, world)]
            parts[0].copy_(own)
            for r in range(1, world):
                dist.recv(parts[r], src=r)
            full = torch.cat(parts).to(dev)
            ref = make_rank(None, 1, 0)                    # the same driver on ONE slab = the whole grid, same sequence of shots
            torch.cuda.synchronize()                       # its fields were filled on torch's stream; the library's streams are non-blocking and would not wait for that
            shot(ref, W)
            sync(ref)
            for _ in range(nwin):
                shot(ref, K)
                sync(ref)
            same = bool(torch.equal(full, ref["img"][:, :n]))
            print(f"[check] decomposed image ({world} slabs) == single domain, bitwise: {same}", file=sys.stderr, flush=True)
            if not same:      # where: rows / columns / slabs of the differing cells (a halo that came late shows up next to a band edge)
                bad = (full != ref["img"][:, :n]).nonzero()
                rws, cls = bad[:, 0], bad[:, 1]
                edges = [b0 for (b0, _b1) in slab_bounds(n, world)]
                print(f"[check] {bad.shape[0]} cells differ: rows {int(rws.min())}..{int(rws.max())}, columns {int(cls.min())}..{int(cls.max())}; band edges at rows {edges}; "
                      f"max |diff| {float((full - ref['img'][:, :n]).abs().max()):.3e} of max {float(ref['img'].abs().max()):.3e}; "
                      f"rows with differences per band: {[int(((rws >= b0) & (rws < b1)).sum()) for (b0, b1) in slab_bounds(n, world)]}", file=sys.stderr, flush=True)
            check = "image bitwise equal to a single-domain run of the same shots" if same else "image DIFFERS from the single-domain run"
            check_failed = not same        # the line is still printed (flagged), the exit code says so at the end
        else:
            dist.send(own, dst=0)
    exposed = None
    if c_driver and world > 1 and not args.no_exposed:
        me["sl"].set_stub(True)            # the same shots with the transfers switched off; results are wrong from here on
        shot(me, K)
        sync(me)
        wall_stub, _, _ = timed_windows(window, sync_all, world, dev, max_windows=max(2, nwin // 4))
        me["sl"].set_stub(False)
        exposed = {"value": round(max(0.0, 1.0 - wall_stub / wall), 4), "ms_per_step_with_transfers": round(wall * 1e3 / K, 6),
                   "ms_per_step_transfers_off": round(wall_stub * 1e3 / K, 6),
                   "method": "median window re-timed with fdw_slabs_set_stub(1): exchanges enqueue nothing, everything else unchanged"}
    # N = 1 through the C driver: the dominant kernel is the fused backward pass (four iterations per launch); its launch time is measured
    # here, on the slab's stream, as the difference of two backward loops that differ by 32 passes
    back_launch_s = None
    if world == 1 and c_driver and me["sl"].back_buffers() == (6, 4) and nt >= 2 + 4 * 40:
        sl, fld, rcv = me["sl"], me["fld"], me["fld"][me["nsrc"]:]
        ext = torch.cuda.ExternalStream(sl.stream)
        scratch_img = torch.zeros_like(me["img"])
        torch.cuda.synchronize()                           # filled on torch's stream; the library's streams are non-blocking

        def back_ms(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            sl.dev_back([f_.data_ptr() for f_ in fld[:me["nsrc"]]], [r_.data_ptr() for r_ in rcv], me["v2"].data_ptr(), samples.data_ptr(), gz,
                        scratch_img.data_ptr(), 0, iters, role=(me["ip"], me["ipp"], 0, 1))
            e1.record(ext)
            sl.synchronize()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1)
        back_ms(2 + 4 * 8)
        ts = sorted(back_ms(2 + 4 * 40) - back_ms(2 + 4 * 8) for _ in range(5))
        back_launch_s = ts[len(ts) // 2] * 1e-3 / 32
    if rank == 0:
        upd = 3.0 * n * n * K
        bytes_min = (16.0 + 16.0 + 28.0) * n * n * K         # forward step 16 B/point + backward iteration 44 B/point (40 + the image read)
        piped = c_driver and me["sl"].back_buffers() == (6, 4)
        bytes_job = (20.0 / 4 + 44.0 / 4) * n * n * K if piped else bytes_min
        out = {"metric": "Gpoints/s (stencil updates) + achieved HBM GB/s vs peak", "value": round(upd / wall / 1e9, 3), "unit": "Gpoints/s", "n_gpus": world,
               "steps": K, "warmup": W, "ms_per_step": round(wall * 1e3 / K, 6), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic (seeded noise wavefield and gather)",
               "config": {"workload": f"RTM domain decomposition: one shot on a {n}x{n} fp32 grid in {world} x-slab(s): {K} forward steps + {K} backward iterations "
                                      f"with imaging, {me['g'].ksteps if world > 1 else 0} steps per halo exchange", "grid": [n, n], "order": ORDER,
                          "parallelism": f"slab{world}" if world > 1 else "single"},
               "result_finite_nonzero": finite and nonzero, "numerics": "fast" if NUM else "exact",
               "halo_exchange": (HALO_PATHS[comm.kind] + (f" -- FALLBACK: RCCL was not usable from libfdwave.so ({rccl_fallback})" if rccl_fallback else "")) if world > 1 else None,
               "decomposition_check": check,
               "rccl_ranks": comm.world if (comm is not None and comm.kind == "rccl") else None, "comm_ranks": comm.world if comm is not None else world,
               "exposed_comm_fraction": exposed["value"] if exposed else None, "exposed_comm": exposed,
               "launched_by": "bench.py itself (child processes)" if os.environ.get("FDW_BENCH_SELF_LAUNCHED") else ("an external launcher (torch.distributed.run)" if world > 1 else None),
               "timing": {"windows": nwin, "window_steps": K, "statistic": "median window (barrier + synchronize on both sides, max over ranks)"},
               "roofline": {"bound": "hbm", "achieved": round(bytes_job / wall / 1e9, 1), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                            "frac": round(bytes_job / wall / 1e9 / (HBM_PEAK_GBS * world), 4), "traffic": None,
                            "basis": ("minimum bytes the job's launches must move per time index (" + ("four-step passes: forward 20 B/point per 4 steps, backward 44 B/point per "
                                      "4 iterations" if piped else "one-step kernels: forward 16 B/point, backward iteration 44 B/point") + "; owned rows only) / wall time of the "
                                      "median window, against N x the HBM peak"),
                            "one_step_model": {"bytes": bytes_min, "achieved": round(bytes_min / wall / 1e9, 1),
                                               "note": "SURVEY.md 8(d)'s byte model of the one-step kernels (16 + 44 B/point per time index): a throughput figure, NOT a roofline fraction"}}}
        if back_launch_s:
            prof = offline_counters("rtm-slab-fast" if NUM else "rtm-slab", n, 1)
            traffic = prof["hbm_bytes_per_launch"] if prof else None
            min_bytes = 44.0 * n * n                        # 6 fields in (F_k, F_k-1, r, r', v2, image) + 5 out per point and launch
            basis = traffic if traffic else min_bytes
            out["roofline"] = {"bound": "hbm", "achieved": round(basis / back_launch_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(basis / back_launch_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "basis": ("HBM-side bytes per launch measured with rocprofv3 --pmc (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE) / launch time of this run"
                                         if traffic else "minimum bytes one launch must move (six fields in, five out) / launch time of this run"),
                               "traffic_source": (f"offline profile {prof.get('source')} taken on kernel sources {prof.get('source_hash')} (= this build)" if prof else
                                                  "none: profiles/traffic.json holds no entry for this size / kernel build (a stale profile is never quoted)"),
                               "kernel": "fdw::fdw_back4_kernel<4,4,2> (four backward iterations per launch: source-field and receiver-field pipelines of four waves each, imaging fused)",
                               "launch_us": round(back_launch_s * 1e6, 2), "iterations_per_launch": 4, "min_bytes_per_launch": min_bytes,
                               "min_bytes_frac": round(min_bytes / back_launch_s / 1e9 / HBM_PEAK_GBS, 4),
                               "algorithmic_44B_model": {"bytes_per_launch": 4 * 44.0 * n * n, "achieved": round(4 * 44.0 * n * n / back_launch_s / 1e9, 1),
                                                         "ratio_to_peak": round(4 * 44.0 * n * n / back_launch_s / 1e9 / HBM_PEAK_GBS, 4),
                                                         "note": "SURVEY.md 8(d)'s one-pass-per-iteration byte model (44 B/point/iteration x 4); a throughput figure in bytes, NOT a roofline fraction"},
                               "whole_shot_one_step_model": {"bytes": bytes_min, "achieved": round(bytes_min / wall / 1e9, 1),
                                                             "note": "forward 16 + backward 44 B/point per time index over the wall time (the model of the one-step kernels)"}}
            if prof and prof.get("valu_busy") is not None:
                out["roofline"]["issue"] = {"valu_busy": prof["valu_busy"], "salu_per_valu": prof.get("salu_per_valu"), "source": prof.get("sq_source")}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(min(n, 2048), 8.0, rtm=True)      # bounded sample: a 2048^2 grid, iterations sized for ~8 s per leg
        print(json.dumps(out), flush=True)
    me["sl"].close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()
    if not (finite and nonzero):
        sys.exit(f"bench: image is not finite / all zero (finite {finite}, nonzero {nonzero})")
    if check_failed:
        sys.exit("bench: the decomposed image differs from the single-domain image (see decomposition_check in the line above)")


def poison_free_memory():
    """Development aid (FDW_BENCH_POISON=1): fill the free device memory with NaNs and release it again, so that anything the run reads
    without having written it shows up as a NaN result instead of depending on what the previous process left behind."""
    free, _ = torch.cuda.mem_get_info()
    chunks = []
    left = int(free * 0.9)
    while left > (1 << 28):
        nb = min(left, 8 << 30)
        chunks.append(torch.full((nb // 4,), float("nan"), device="cuda"))
        left -= nb
    torch.cuda.synchronize()
    del chunks
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--ksteps", type=int, default=0, help="time steps per halo exchange (N > 1); 0 = auto")
    ap.add_argument("--pipe", choices=("auto", "on", "off"), default="auto",
                    help="N > 1: four-steps-per-pass wave-pipeline kernel inside the slabs (auto = where the library would pick it)")
    ap.add_argument("--workload", choices=("forward", "model", "rtm", "stencil", "rtm-slab"), default="forward",
                    help="forward: the headline fused forward step of rtm_code (default); model: the forward-modelling producer; "
                         "rtm: whole RTM shots on the reference's new_mod deck size (both N = 1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rest-line", action="store_true", help="forward workload, N = 1: skip the extra measurement from BASELINE.md's zero initial fields")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "shm"),
                    help="N > 1: nccl = halo exchange over RCCL inside libfdwave.so, one rank per GPU (default); shm = the same C drivers over the "
                         "library's process transport (shared-memory staging), ranks may share one GPU (rehearsals on a one-GPU box)")
    ap.add_argument("--check", action="store_true", help="N > 1: compare the decomposed result with a single-domain run on rank 0, bitwise (always on; kept for old command lines)")
    ap.add_argument("--no-check", action="store_true", help="N > 1: skip that comparison")
    ap.add_argument("--no-exposed", action="store_true", help="N > 1, --backend nccl: skip the re-timing with the transfers off (exposed_comm_fraction)")
    ap.add_argument("--numerics", choices=["exact", "fast"], default="exact",
                    help="exact (default, what `value` always reports in the default run): the reference's arithmetic operation for operation, bit-identical "
                         "to the oracle; fast: fdw_params.numerics = FDW_NUMERICS_FAST (symmetric sums + fused multiply-adds in the Laplacian, within "
                         "1e-5 of exact) for the whole run -- profiling and scaling runs of that mode; the default run prints it beside the headline")
    ap.add_argument("--max-windows", type=int, default=400, help="cap on the repetitions of the timed window (debugging aid: a short, fixed sequence)")
    ap.add_argument("--no-extra", action="store_true", help="forward workload, N = 1, default size: skip the extra 4096^2 / 1000-step line (BASELINE.json configs[1])")
    ap.add_argument("--no-fast-line", action="store_true", help="forward workload, N = 1: skip the extra measurement in FAST numerics")
    ap.add_argument("--init", choices=["noise", "rest"], default="noise",
                    help="initial wavefield: seeded noise (default; every cache line carries real data) or at rest (zeros + source)")
    ap.add_argument("--rccl-rehearsal", default=None, metavar="UNIQUE_ID_HEX",
                    help="internal: what each rank of an N > 1 run starts as a child process before it opens RCCL itself (rccl_rehearsal_main)")
    args = ap.parse_args()
    if args.rccl_rehearsal is not None:
        if not torch.cuda.is_available():
            sys.exit("bench: no GPU visible (the product has no CPU path)")
        return rccl_rehearsal_main(args.rccl_rehearsal)
    MAX_WINDOWS[0] = max(1, args.max_windows)
    if os.environ.get("FDW_BENCH_POISON") == "1":
        poison_free_memory()
    if args.workload == "rtm-slab":
        if not torch.cuda.is_available():
            sys.exit("bench: no GPU visible (the product has no CPU path)")
        return run_rtm_slab_workload(args)
    if args.workload != "forward":
        if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
            sys.exit(f"bench --workload {args.workload} runs on one GPU")
        if not torch.cuda.is_available():
            sys.exit("bench: no GPU visible (the product has no CPU path)")
        return {"model": run_model_workload, "rtm": run_rtm_workload, "stencil": run_stencil_workload}[args.workload](args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (the product has no CPU path)")
    if args.backend != "nccl" or os.environ.get("FDW_BENCH_SHARE_GPU") == "1":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)     # rehearsal: several ranks may share one GPU (FDW_BENCH_SHARE_GPU: tests of the RCCL fallback)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane (barriers, the max over ranks of the window time, gathering the slabs for the check): a gloo group on the host.
        # The data plane -- the halo rows -- travels inside libfdwave.so: over RCCL / xGMI (--backend nccl, the default, one rank per GPU) or,
        # for rehearsals of several ranks on ONE GPU (RCCL refuses duplicate devices), through the library's process transport (--backend shm:
        # the same C drivers, halo blocks staged through shared memory).
        dist.init_process_group("gloo", rank=rank, world_size=world)

    n, K, W = args.size, args.steps, args.warmup
    nt = K + W
    NUM = 1 if args.numerics == "fast" else 0
    srce = torch.from_numpy(F.ricker_wavelet(nt, DT, FPEAK)).to(dev)
    sx, sz = n // 2, n // 2
    c_driver = world > 1 or os.environ.get("FDW_FORCE_SLAB_DRIVER") == "c"      # the latter: the multi-GPU code path on one rank (tests)
    slabs = comm = None
    use_pipe = False
    rccl_fallback = None
    if c_driver:
        # the communicator: ncclGetUniqueId on rank 0, the bytes to every rank through the control-plane group, ncclCommInitRank (one rank per GPU),
        # and one message to the own rank as a check.  Should librccl not be usable from the C library on this machine, every rank falls back
        # -- together -- to the library's process transport (halo blocks staged through shared memory: slow, flagged in the line) rather than
        # lose the measurement altogether.
        if world > 1:
            comm, rccl_fallback = rccl_or_fallback(args, rank, world, local_rank, n, dist)
        else:                                   # FDW_FORCE_SLAB_DRIVER=c: the slab driver on one rank, over a one-rank RCCL communicator if there is one
            try:
                comm = F.Comm.rccl(F.Comm.unique_id(), 0, 1, local_rank)
                comm.selftest()
            except F.FdwError as e:
                print(f"[bench] RCCL inside libfdwave.so is not usable here ({e}); one slab without a communicator", file=sys.stderr, flush=True)
                comm = None
        if args.pipe != "auto":
            os.environ["FDW_SLAB_PIPE"] = "1" if args.pipe == "on" else "0"
        if args.no_overlap:
            os.environ["FDW_SLAB_NO_OVERLAP"] = "1"
        slabs = F.Slabs(ORDER, n, n, NB, NB, nt, FAC, DX, DX, DT, comm=comm, compat=False, ksteps=args.ksteps, numerics=NUM)
        args.ksteps, use_pipe = slabs.ksteps, slabs.nbuf == 4
        geom = SlabGeometry(rank, world, n, ORDER // 2, slabs.ksteps)
        assert (geom.x_off, geom.nxl, geom.o0, geom.o1) == (slabs.x_off, slabs.nxl, slabs.own0, slabs.own1)
        ctx, pitch = None, slabs.pitch
    else:
        geom = SlabGeometry(0, 1, n, ORDER // 2, 1)
        ctx = F.FDWave(ORDER, n, n, NB, NB, nt, FAC, DX, DX, DT, compat=False, device=local_rank, numerics=NUM)
        if os.environ.get("FDW_XCHUNK"):      # tuning experiments only
            ctx.set_tuning(xchunk=int(os.environ["FDW_XCHUNK"]))
        pitch = ctx.pitch
    v2 = torch.zeros((geom.nxl, pitch), device=dev)
    v2[:, :n] = synthetic_velocity_rows(n, geom.x_off, geom.nxl, dev)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def slab_noise_fields(count):
        """`count` local fields; the first two carry the same GLOBAL noise field on every decomposition (seeded per global row)."""
        fl = [torch.zeros((geom.nxl, pitch), device=dev) for _ in range(count)]
        if args.init == "noise":
            g = torch.Generator(device=dev)
            for f_, seed in ((fl[0], 0x5EED0001), (fl[1], 0x5EED0002)):
                g.manual_seed(seed)
                full_rows = 1e-3 * torch.randn((n, 64), device=dev, generator=g)      # cheap, decomposition-independent pattern
                f_[:, :n] = full_rows[geom.x_off:geom.x_off + geom.nxl].repeat(1, n // 64)
        return fl

    rest_line = fast_line = None
    if not c_driver:
        # One GPU: the library's forward loop over four rotating buffers (fdw_dev_steps2): pairs of time steps go
        # through the two-step kernel (temporal blocking) where it pays, everything enqueued by ONE library call.
        skew = int(os.environ.get("FDW_ALLOC_SKEW", "0"))     # tuning experiments only: bytes of padding between the field buffers
        bufs, pads = [], []
        for _ in range(4):
            if skew:
                pads.append(torch.empty(skew, dtype=torch.uint8, device=dev))
            bufs.append(torch.zeros((n, pitch), device=dev))
        if args.init == "noise":   # non-trivial bit patterns everywhere (a quiescent field is mostly zeros for the whole run)
            g = torch.Generator(device=dev)
            g.manual_seed(0x5EED0001)
            for b_ in bufs[:2]:
                b_[:, :n] = 1e-3 * torch.randn((n, n), device=dev, generator=g)
        ptrs = [b.data_ptr() for b in bufs]
        stream = torch.cuda.Stream()
        roles = {"ip": 0, "ipp": 1}

        def run(it0, nsteps):
            roles["ip"], roles["ipp"] = ctx.dev_steps2(ptrs, v2.data_ptr(), srce.data_ptr(), sx, sz, it0, nsteps, it0 > 0,
                                                       roles["ip"], roles["ipp"], stream=stream.cuda_stream)

        torch.cuda.synchronize()       # the fills above ran on torch's default stream; `stream` does not wait for it by itself
        run(0, W)
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def window():
            e0.record(stream)
            run(W, K)
            e1.record(stream)
            while not e1.query():      # spin: hipStreamSynchronize naps in ms-sized steps
                pass
            stream.synchronize()
            return e0.elapsed_time(e1)

        wall, dev_ms, nwin = timed_windows(window, sync_all, world, dev)
        newest = bufs[roles["ipp"]].clone()
        if args.init == "noise" and not args.no_rest_line:
            # BASELINE.md section 4's literal initial condition beside the headline: zero fields + the Ricker source at the grid centre, the
            # same K steps from time index 0.  It runs faster (a mostly-zero wavefield draws less power and the issue-bound kernel clocks
            # higher), which is why the headline value is the noise-filled case.
            for b_ in bufs:
                b_.zero_()
            roles["ip"], roles["ipp"] = 0, 1
            torch.cuda.synchronize()

            def window_rest():
                for b_ in bufs:
                    b_.zero_()
                roles["ip"], roles["ipp"] = 0, 1
                torch.cuda.synchronize()
                e0.record(stream)
                run(0, K)
                e1.record(stream)
                while not e1.query():
                    pass
                stream.synchronize()
                return e0.elapsed_time(e1)

            window_rest()
            r_wall, r_dev_ms, r_nwin = timed_windows(window_rest, sync_all, world, dev, max_windows=8)
            rest_line = {"init": "rest: zero fields + Ricker source at the grid centre (BASELINE.md section 4), steps 0 .. K-1", "value": round(n * n * K / (r_dev_ms * 1e-3) / 1e9, 3),
                         "unit": "Gpoints/s", "ms_per_step": round(r_dev_ms / K, 6), "windows": r_nwin,
                         "timing": "HIP events around the K enqueued steps (the fills between windows are outside them)"}
        if NUM == 0 and not args.no_fast_line:
            # The same K steps from the same noise start in FAST numerics (include/fdwave.h: symmetric sums + fused multiply-adds in the Laplacian,
            # <= 1e-5 from the exact arithmetic -- tests/test_fast_numerics.py), beside the headline: `value` stays the exact mode's.
            fctx = F.FDWave(ORDER, n, n, NB, NB, nt, FAC, DX, DX, DT, compat=False, device=local_rank, numerics=1)
            g = torch.Generator(device=dev)

            def refill():
                for b_ in bufs:
                    b_.zero_()
                if args.init == "noise":
                    g.manual_seed(0x5EED0001)
                    for b_ in bufs[:2]:
                        b_[:, :n] = 1e-3 * torch.randn((n, n), device=dev, generator=g)
                roles["ip"], roles["ipp"] = 0, 1
                torch.cuda.synchronize()

            def frun(it0, nsteps):
                roles["ip"], roles["ipp"] = fctx.dev_steps2(ptrs, v2.data_ptr(), srce.data_ptr(), sx, sz, it0, nsteps, it0 > 0,
                                                            roles["ip"], roles["ipp"], stream=stream.cuda_stream)

            refill()
            frun(0, W)
            stream.synchronize()

            def window_fast():
                e0.record(stream)
                frun(W, K)
                e1.record(stream)
                while not e1.query():
                    pass
                stream.synchronize()
                return e0.elapsed_time(e1)

            f_wall, f_dev_ms, f_nwin = timed_windows(window_fast, sync_all, world, dev)
            f_new = bufs[roles["ipp"]]
            f_ok = bool(torch.isfinite(f_new).all().item()) and float(f_new.abs().max().item()) > 0.0
            f_spl = fctx.steps_per_pass()
            f_rem = K % f_spl
            f_launches = K // f_spl + (f_rem // 2 + f_rem % 2 if fctx.two_step_active() else f_rem)
            f_launch_s = f_dev_ms * 1e-3 / f_launches
            f_prof = offline_counters("forward-fast", n, f_spl)
            f_traffic = f_prof["hbm_bytes_per_launch"] if f_prof else None
            f_min = (ALGO_BYTES_PER_POINT if f_spl == 1 else MIN_BYTES_PER_POINT_PER_LAUNCH) * n * n
            f_basis = f_traffic if f_traffic else f_min
            fast_line = {"numerics": "FAST (fdw_params.numerics = 1): Laplacian as symmetric sums + fused multiply-adds, fp64 leap-frog kept; <= 1e-5 max-norm-relative "
                                     "from the exact arithmetic over 1 700 steps (tests/test_fast_numerics.py)",
                         "value": round(n * n * K / f_wall / 1e9, 3), "unit": "Gpoints/s", "ms_per_step": round(f_wall * 1e3 / K, 6), "windows": f_nwin,
                         "result_finite_nonzero": f_ok,
                         "roofline": {"bound": "hbm", "achieved": round(f_basis / f_launch_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(f_basis / f_launch_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": f_traffic,
                                      "basis": "measured HBM-side bytes per launch" if f_traffic else "minimum bytes one launch must move",
                                      "traffic_source": traffic_source_note(f_prof), "launch_us": round(f_launch_s * 1e6, 2), "steps_per_launch": f_spl,
                                      "kernel": "fdw::fdw_stepn_kernel<4,4,true,1,2,false,0,1> (NUM = 1)" if f_spl == 4 else "FAST instantiation of the kernel above"}}
            if f_prof and f_prof.get("valu_busy") is not None:
                fast_line["roofline"]["issue"] = {"valu_busy": f_prof["valu_busy"], "salu_per_valu": f_prof.get("salu_per_valu"), "source": f_prof.get("sq_source")}
            del fctx
    else:
        # N GPUs, one rank each: the whole K-step window -- passes, boundary strips, halo exchange over RCCL on the communication stream,
        # interior rows beside the transfer -- is enqueued by ONE call into the C library (fdw_slabs_dev_forward)
        fields = slab_noise_fields(slabs.nbuf)
        ptrs = [f_.data_ptr() for f_ in fields]
        roles = {"ip": 0, "ipp": 1}

        def run(it0, nsteps):
            roles["ip"], roles["ipp"] = slabs.dev_forward(ptrs, v2.data_ptr(), srce.data_ptr(), sx, sz, it0, nsteps, it0 > 0, roles["ip"], roles["ipp"])

        torch.cuda.synchronize()
        run(0, W)
        slabs.synchronize()

        def window():
            t0 = time.perf_counter()
            run(W, K)
            slabs.synchronize()
            return (time.perf_counter() - t0) * 1e3

        wall, dev_ms, nwin = timed_windows(window, sync_all, world, dev)
        newest = fields[roles["ipp"]][geom.g_lo:geom.nxl - geom.g_hi]
    finite = bool(torch.isfinite(newest).all().item()) and float(newest.abs().max().item()) > 0.0
    if world > 1:
        f = torch.tensor([1.0 if finite else 0.0])
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        finite = bool(f.item() > 0.5)

    check = None
    check_failed = False
    if world > 1 and not args.no_check:
        # the decomposed field against a single-domain run of the same step sequence (warm-up, then `nwin` windows that replay the
        # source samples W .. W+K-1) on rank 0, bitwise: a halo that arrives late or not at all cannot hide behind a plausible number
        own = newest[:, :n].contiguous().cpu()
        from parallel_finite_difference_computation_amd.decomp import slab_bounds
        if rank == 0:
            parts = [torch.empty(((b1 - b0), n)) for (b0, b1) in slab_bounds(n, world)]
            parts[0].copy_(own)
            for r in range(1, world):
                dist.recv(parts[r], src=r)
            full = torch.cat(parts).to(dev)
            ref_ctx = F.FDWave(ORDER, n, n, NB, NB, nt, FAC, DX, DX, DT, compat=False, device=local_rank, numerics=NUM)
            rb = [torch.zeros((n, ref_ctx.pitch), device=dev) for _ in range(4)]
            if args.init == "noise":
                g = torch.Generator(device=dev)
                for f_, seed in ((rb[0], 0x5EED0001), (rb[1], 0x5EED0002)):
                    g.manual_seed(seed)
                    f_[:, :n] = (1e-3 * torch.randn((n, 64), device=dev, generator=g)).repeat(1, n // 64)
            rv2 = torch.zeros((n, ref_ctx.pitch), device=dev)
            rv2[:, :n] = synthetic_velocity_rows(n, 0, n, dev)
            rptrs = [b_.data_ptr() for b_ in rb]
            torch.cuda.synchronize()
            ip, ipp = ref_ctx.dev_steps2(rptrs, rv2.data_ptr(), srce.data_ptr(), sx, sz, 0, W, False, 0, 1, stream=None)
            for _ in range(nwin):
                ip, ipp = ref_ctx.dev_steps2(rptrs, rv2.data_ptr(), srce.data_ptr(), sx, sz, W, K, True, ip, ipp, stream=None)
            torch.cuda.synchronize()
            same = bool(torch.equal(full, rb[ipp][:, :n]))
            print(f"[check] decomposed ({world} slabs) == single domain, bitwise: {same}", file=sys.stderr, flush=True)
            check = "bitwise equal to a single-domain run of the same step sequence" if same else "DIFFERS from the single-domain run"
            check_failed = not same        # the line is still printed (flagged), the exit code says so at the end
        else:
            dist.send(own, dst=0)
    exposed = None
    if c_driver and world > 1 and not args.no_exposed:
        # the same window with the transfers switched off (launches and stream hand-overs stay): what the exchange costs the cycle.
        # Results are wrong from here on, so this comes after `newest` and the check.
        slabs.set_stub(True)
        run(W, K)
        slabs.synchronize()
        wall_stub, _, _ = timed_windows(window, sync_all, world, dev, max_windows=max(3, nwin // 4))
        slabs.set_stub(False)
        exposed = {"value": round(max(0.0, 1.0 - wall_stub / wall), 4), "ms_per_step_with_transfers": round(wall * 1e3 / K, 6),
                   "ms_per_step_transfers_off": round(wall_stub * 1e3 / K, 6),
                   "method": "median window re-timed with fdw_slabs_set_stub(1): exchanges enqueue nothing, everything else unchanged"}
    if rank == 0:
        gpts = n * n * K / wall / 1e9
        # dominant kernel: the fused step.  At N = 1 one launch updates the whole grid; its average
        # duration is the event time over K back-to-back launches on the launch stream.
        pts_per_launch = n * n if world == 1 else None
        out = {
            "metric": "Gpoints/s (stencil updates) + achieved HBM GB/s vs peak",
            "value": round(gpts, 3), "unit": "Gpoints/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(wall * 1e3 / K, 6), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (seeded noise wavefield)" if args.init == "noise" else " (field at rest + Ricker source)"),
            "config": {"workload": f"2D 8th-order acoustic stencil, fused forward step (taper+Laplacian+leap-frog+source), "
                                   f"{n}x{n} fp32 extended grid, nxb=nzb={NB}, {K} steps"
                                   + (f", x-slab decomposition over {world} GPUs, {args.ksteps} steps per halo exchange" + (", four steps per pass inside the slabs" if use_pipe else "") if world > 1 else ""),
                       "grid": [n, n], "order": ORDER, "parallelism": f"slab{world}" if world > 1 else "single"},
            "result_finite_nonzero": finite,
        }
        if world > 1:
            out["halo_exchange"] = HALO_PATHS[comm.kind] + (f" -- FALLBACK: RCCL was not usable from libfdwave.so ({rccl_fallback})" if rccl_fallback else "")
            out["decomposition_check"] = check
            out["rccl_ranks"] = comm.world if comm.kind == "rccl" else None
            out["comm_ranks"] = comm.world
            out["exposed_comm_fraction"] = exposed["value"] if exposed else None
            out["exposed_comm"] = exposed
            out["launched_by"] = "bench.py itself (child processes)" if os.environ.get("FDW_BENCH_SELF_LAUNCHED") else "an external launcher (torch.distributed.run)"
        if rest_line is not None:
            out["baseline_md_initial_condition"] = rest_line
        if fast_line is not None:
            out["fast_numerics"] = fast_line
        if world == 1 and not c_driver and n == 8192 and not args.no_extra:
            # BASELINE.json's second configuration verbatim -- 4096^2 fp32, 1000 steps, one MI355X -- recorded by the same invocation, so that the
            # driver's one default run holds it too (the headline stays the 8192^2 grid the north-star target is quoted on)
            del bufs
            torch.cuda.empty_cache()
            out["extra"] = {"baseline_config_2": dict(forward_line(4096, 1000, 50, dev, numerics=NUM, init=args.init),
                                                      config="2D 8th-order acoustic stencil, 4096^2 fp32, 1000 steps, 1 x MI355X (BASELINE.json configs[1])")}
        out["numerics"] = "fast" if NUM else "exact"
        if NUM:
            out["config"]["workload"] += ", FAST numerics (symmetric sums + fused multiply-adds in the Laplacian; <= 1e-5 from the reference's arithmetic)"
        out["timing"] = {"windows": nwin, "window_steps": K, "statistic": "median window (each bracketed by barrier + synchronize, max over ranks)",
                         "measured_seconds_min": MIN_TIMED_SECONDS}
        model_note = ("SURVEY.md 8(d)'s one-pass-per-step byte model (16 B/point/step x the steps one launch advances); a temporally blocked launch "
                      "moves a fraction of it, so this ratio can exceed 1 -- it is a throughput figure in bytes, NOT a roofline fraction")
        if world == 1 and ctx is not None:
            steps_per_launch = ctx.steps_per_pass()          # 4: wave pipeline, 2: two-step kernel, 1: one-step kernel
            rem = K % steps_per_launch
            launches = K // steps_per_launch + (rem // 2 + rem % 2 if ctx.two_step_active() else rem)
            kname = {4: "fdw::fdw_stepn_kernel<4,4,true,1,2,false> (four time steps per launch: one wave per time level, rows handed through LDS)",
                     2: "fdw::fdw_step2_kernel<4,true,1,false,2> (two time steps per launch)", 1: "fdw::fdw_step_kernel<4,true,1,false,false,2>"}[steps_per_launch]
            launch_s = dev_ms * 1e-3 / launches
            algo = ALGO_BYTES_PER_POINT * pts_per_launch * steps_per_launch      # 16 B/point/step x steps in one launch
            min_bytes = (ALGO_BYTES_PER_POINT if steps_per_launch == 1 else MIN_BYTES_PER_POINT_PER_LAUNCH) * pts_per_launch
            prof = offline_counters("forward-fast" if NUM else "forward", n, steps_per_launch)
            if NUM:
                kname = kname.replace("> (", ",0,1> (FAST numerics; ") if "> (" in kname else kname + " with NUM = 1 (FAST numerics)"
            traffic = prof["hbm_bytes_per_launch"] if prof else None
            basis_bytes = traffic if traffic else min_bytes
            achieved = basis_bytes / launch_s / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "basis": ("HBM-side bytes per launch measured with rocprofv3 --pmc (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE) / launch time of this run"
                                         if traffic else "minimum bytes one launch must move (read u^n, u^(n-1), v2; write the fields it produces) / launch time of this run"),
                               "traffic_source": (f"offline profile {prof.get('source')} taken on kernel sources {prof.get('source_hash')} (= this build)" if prof else
                                                  "none: profiles/traffic.json holds no entry for this size / kernel build (a stale profile is never quoted)"),
                               "kernel": kname, "launch_us": round(launch_s * 1e6, 2), "steps_per_launch": steps_per_launch,
                               "min_bytes_per_launch": min_bytes, "min_bytes_frac": round(min_bytes / launch_s / 1e9 / HBM_PEAK_GBS, 4),
                               "algorithmic_16B_model": {"bytes_per_launch": algo, "achieved": round(algo / launch_s / 1e9, 1),
                                                         "ratio_to_peak": round(algo / launch_s / 1e9 / HBM_PEAK_GBS, 4), "note": model_note}}
            if prof and prof.get("valu_busy") is not None:
                out["roofline"]["issue"] = {"bound": "memory-path" if NUM else "valu-issue", "valu_busy": prof["valu_busy"], "salu_per_valu": prof.get("salu_per_valu"),
                                            "source": prof.get("sq_source"),
                                            "note": ("SQ counters: with half the Laplacian's instructions the vector units idle; the kernel is bound by the memory path (ablations in "
                                                     "profiles/r03_fast_ablations.txt, DESIGN.md section 3e)") if NUM else
                                                    "SQ counters: the kernel saturates the vector issue slots well below the HBM line; see DESIGN.md section 4"}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(n)
                sib = reference_sibling_cpu(min(n, 2048), 100)      # the reference's own CPU code for the same stencil (its two-pass fd_step), beside our port
                if sib is not None:
                    out["cpu_baseline"]["reference_cpu_sibling"] = sib
        else:
            passes_bytes = (MIN_BYTES_PER_POINT_PER_LAUNCH / 4.0 if use_pipe else ALGO_BYTES_PER_POINT) * n * n * K      # whole job, ghost rows not counted
            algo = ALGO_BYTES_PER_POINT * n * n * K
            out["roofline"] = {"bound": "hbm", "achieved": round(passes_bytes / wall / 1e9, 1), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                               "frac": round(passes_bytes / wall / 1e9 / (HBM_PEAK_GBS * world), 4), "traffic": None,
                               "basis": "minimum bytes the job's launches must move (owned rows only) / wall time of the median window, against N x the HBM peak",
                               "algorithmic_16B_model": {"bytes": algo, "achieved": round(algo / wall / 1e9, 1),
                                                         "ratio_to_peak": round(algo / wall / 1e9 / (HBM_PEAK_GBS * world), 4), "note": model_note}}
        print(json.dumps(out), flush=True)
    if slabs is not None:
        slabs.close()
        comm.close()
    if world > 1:
        dist.destroy_process_group()
    if not finite:
        sys.exit("bench: result is not finite / all zero")
    if check_failed:
        sys.exit("bench: the decomposed result differs from the single-domain result (see decomposition_check in the line above)")


if __name__ == "__main__":
    main()
