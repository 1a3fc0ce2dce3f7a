"""ctypes front end of the CPU oracle (oracle/fdw_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package.  Builds liborc.so on first use (gcc -O2 -ffp-contract=off, oracle/Makefile).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_OMP = None
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


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


def lib(omp=False):
    """liborc.so; omp=True: the OpenMP build of the same file (rows of the lap / time / img passes shared out over the host threads,
    per-point arithmetic unchanged -- for the full-size parity tests, where one thread would take minutes)."""
    global _LIB, _OMP
    if omp:
        if _OMP is None:
            os.environ.setdefault("OMP_NUM_THREADS", str(host_threads()))      # read by libgomp when the library is loaded
            so = os.path.join(_HERE, "liborc_omp.so")
            if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "fdw_oracle.c")):
                build()
            _OMP = _declare(C.CDLL(so))
        return _OMP
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        src = os.path.join(_HERE, "fdw_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            build()
        _LIB = _declare(C.CDLL(so))
    return _LIB


def _declare(L):
    L.orc_calc_coefs.argtypes = [C.c_int, C.c_int, f32p]
    L.orc_scaled_coefs.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, f32p, f32p]
    L.orc_taper_tables.argtypes = [C.c_int, C.c_int, C.c_float, f32p, f32p]
    L.orc_ricker_wavelet.argtypes = [C.c_int, C.c_float, C.c_float, f32p]
    L.orc_extendvel_linear.argtypes = [C.c_int] * 4 + [f32p]
    L.orc_srand.argtypes = [C.c_uint]
    L.orc_extents.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] * 3
    L.orc_init.restype = C.c_void_p
    L.orc_init.argtypes = [C.c_int] * 6 + [C.c_float] * 4 + [C.c_int]
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_set_numerics.argtypes = [C.c_void_p, C.c_int]
    L.orc_stencil_fast.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, f32p, f32p]
    L.orc_fd_forward.argtypes = [C.c_void_p, f32p, f32p, f32p, C.c_int, C.c_int, f32p, C.c_int]
    L.orc_fd_back.argtypes = [C.c_void_p, f32p, f32p, f32p, f32p, C.c_int, f32p, C.c_int]
    L.orc_slab_step.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, f32p, f32p] + [C.c_int] * 6 + [C.c_float]
    L.orc_slab_back_iter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p] + [C.c_int] * 4 + [f32p, C.c_int, f32p]
    L.orc_stencil.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, f32p, f32p]
    L.orc_fused_steps.argtypes = [C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p, C.c_float, C.c_int]
    L.orc_mod_set_numerics.argtypes = [C.c_int]
    L.orc_mod_taper_tables.argtypes = [C.c_int, C.c_int, C.c_float, f32p, f32p]
    L.orc_mod_extendvel.argtypes = [C.c_int] * 4 + [f32p]
    L.orc_mod_ricker_wavelet.argtypes = [C.c_int, C.c_float, C.c_float, f32p]
    L.orc_mod_taper_apply.argtypes = [f32p] + [C.c_int] * 4 + [f32p, f32p]
    L.orc_mod_fd_step.argtypes = [C.c_int, f32p, C.c_float, C.c_float, C.c_float, f32p, f32p, f32p, f32p, C.c_int, C.c_int]
    L.orc_mod_ptsrc.argtypes = [C.c_int] * 4 + [C.c_float, f32p]
    L.orc_mod_shot.argtypes = [C.c_int] * 6 + [C.c_float] * 4 + [f32p, C.c_int, C.c_int, C.c_int, f32p, f32p]
    L.orc_mod_taper_apply2.argtypes = [f32p] + [C.c_int] * 4 + [f32p, f32p]
    L.orc_image_laplacian.argtypes = [f32p, C.c_int, C.c_int, C.c_float, C.c_float, f32p]
    L.orc_image_compare.argtypes = [f32p, f32p, C.c_size_t, C.c_void_p, C.POINTER(C.c_double)]
    L.orc_rtm_stored_shot.argtypes = [C.c_int] * 6 + [C.c_float] * 4 + [f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_size_t, C.c_int, f32p]
    return L


def ref_lib():
    """The reference's own functions.c compiled unmodified (oracle/_ref); None when not built."""
    so = os.path.join(_HERE, "_ref", "libref_functions.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    L.calc_coefs.restype = C.POINTER(C.c_float)
    L.calc_coefs.argtypes = [C.c_int]
    L.ricker_wavelet.argtypes = [C.c_int, C.c_float, C.c_float, f32p]
    L.extendvel_linear.argtypes = [C.c_int] * 4 + [C.POINTER(C.POINTER(C.c_float))]
    return L


def calc_coefs(order, cxx=False):
    c = np.zeros(order + 1, np.float32)
    lib().orc_calc_coefs(order, int(cxx), c)
    return c


def scaled_coefs(order, dx, dz, cxx=False):
    cx = np.zeros(order + 1, np.float32)
    cz = np.zeros(order + 1, np.float32)
    lib().orc_scaled_coefs(order, dx, dz, int(cxx), cx, cz)
    return cx, cz


def taper_tables(nxb, nzb, fac):
    tx = np.zeros(max(nxb, 1), np.float32)
    tz = np.zeros(max(nzb, 1), np.float32)
    lib().orc_taper_tables(nxb, nzb, fac, tx, tz)
    return tx[:nxb], tz[:nzb]


def ricker_wavelet(nt, dt, fpeak):
    s = np.zeros(nt, np.float32)
    lib().orc_ricker_wavelet(nt, dt, fpeak, s)
    return s


def extendvel_linear(vpe, nx, nz, nxb, nzb, seed=None):
    """In place on vpe[nxe][nze].  seed=None keeps the process's current glibc rand() state."""
    assert vpe.shape == (nx + 2 * nxb, nz + 2 * nzb) and vpe.dtype == np.float32
    if seed is not None:
        lib().orc_srand(seed)
    lib().orc_extendvel_linear(nx, nz, nxb, nzb, vpe)
    return vpe


def extents(nxe, nze, nzb, compat):
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib().orc_extents(nxe, nze, nzb, int(compat), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def stencil(order, nxe, nze, dx, dz, field, numerics=0):
    out = np.zeros((nxe, nze), np.float32)
    fn = lib().orc_stencil_fast if numerics else lib().orc_stencil
    fn(order, nxe, nze, dx, dz, np.ascontiguousarray(field, np.float32).reshape(nxe, nze), out)
    return out


class Oracle:
    """State of one reference fd_init (R:200-224) on the CPU."""

    def __init__(self, order, nxe, nze, nxb, nzb, nt, fac, dx, dz, dt, compat=True, omp=False, numerics=0):
        """numerics 0: the reference's arithmetic (what every parity pin refers to); 1: the product's FAST tolerance mode restated
        (fdw_oracle.c orc_lap_fast) -- the checker of the FAST kernels' own formula, not a statement about the reference."""
        self.shape = (nxe, nze)
        self.nx, self.nz, self.nt = nxe - 2 * nxb, nze - 2 * nzb, nt
        self._lib = lib(omp)
        self._h = self._lib.orc_init(order, nxe, nze, nxb, nzb, nt, fac, dx, dz, dt, int(compat))
        if not self._h:
            raise ValueError("orc_init rejected the parameters")
        if numerics:
            self._lib.orc_set_numerics(self._h, int(numerics))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.orc_free(self._h)
            self._h = None

    def forward(self, v2, sx, sz, srce, p=None, pp=None, nsteps=None):
        """R:247-288.  Returns (P, PP) = (d_p, d_pp) after nsteps iterations."""
        p = np.zeros(self.shape, np.float32) if p is None else np.array(p, np.float32, order="C")
        pp = np.zeros(self.shape, np.float32) if pp is None else np.array(pp, np.float32, order="C")
        nsteps = len(srce) if nsteps is None else nsteps
        self._lib.orc_fd_forward(self._h, p, pp, np.ascontiguousarray(v2, np.float32), sx, sz,
                             np.ascontiguousarray(srce, np.float32), nsteps)
        return p, pp

    def slab_step(self, x_off, p, pp, v2, r0, r1, sx, sz, srce_it, taper_rows=None):
        """One forward iteration on rows [r0,r1) of a slab (local arrays, in place).  Tests only.
        taper_rows: (t0, t1) rows damped in place by this call (default: every local row)."""
        nxl = p.shape[0]
        assert p.shape == pp.shape == v2.shape and p.dtype == np.float32 and p.flags.c_contiguous
        t0, t1 = (0, nxl) if taper_rows is None else taper_rows
        self._lib.orc_slab_step(self._h, x_off, nxl, p, pp, v2, r0, r1, t0, t1, sx, sz, srce_it)

    def slab_back_iter(self, x_off, step_source, f1, f0, pr, ppr, v2, r0, r1, samples, gz, img, taper_rows=None):
        """One iteration of fd_back's loop on rows [r0,r1) of a slab (local arrays, in place).  Tests only."""
        nxl = pr.shape[0]
        t0, t1 = (0, nxl) if taper_rows is None else taper_rows
        self._lib.orc_slab_back_iter(self._h, x_off, nxl, int(step_source), f1, f0, pr, ppr, v2, r0, r1, t0, t1, np.ascontiguousarray(samples, np.float32), gz, img)

    def back(self, v2, snap0, snap1, d_obs, gz, imloc=None, nsteps=None):
        """R:290-341.  d_obs is [nx][nt]; returns imloc [nx][nz]."""
        imloc = np.zeros((self.nx, self.nz), np.float32) if imloc is None else np.array(imloc, np.float32, order="C")
        nsteps = self.nt if nsteps is None else nsteps
        d_obs = np.ascontiguousarray(d_obs, np.float32)
        assert d_obs.shape == (self.nx, self.nt)
        self._lib.orc_fd_back(self._h, np.ascontiguousarray(v2, np.float32), np.ascontiguousarray(snap0, np.float32),
                          np.ascontiguousarray(snap1, np.float32), d_obs, gz, imloc, nsteps)
        return imloc


# ---- forward-modelling producer of the CPU-serial sibling (oracle/fdw_oracle_mod.c) ---------------------------------
def mod_taper_tables(nxb, nzb, fac):
    tx, tz = np.ones(max(nxb, 1), np.float32), np.ones(max(nzb, 1), np.float32)
    lib().orc_mod_taper_tables(nxb, nzb, fac, tx, tz)
    return tx[:nxb], tz[:nzb]


def mod_extendvel(vel, nx, nz, nxb, nzb):
    vel = np.ascontiguousarray(vel, np.float32)
    lib().orc_mod_extendvel(nx, nz, nxb, nzb, vel)
    return vel


def mod_ricker_wavelet(nt, dt, fpeak):
    s = np.zeros(nt, np.float32)
    lib().orc_mod_ricker_wavelet(nt, dt, fpeak, s)
    return s


def mod_numerics(numerics):
    """Switch the sibling-dialect loops of the oracle (mod_shot, mod_steps, rtm_stored_shot) to the product's FAST formula (1) or back to
    the sibling's own arithmetic (0, what every parity pin refers to).  Module-level state: tests reset it."""
    lib().orc_mod_set_numerics(int(numerics))


def mod_shot(order, nx, nz, nxb, nzb, dx, dz, dt, fac, vel2, sx, sz, gz, srce, numerics=0):
    """mod_main's loop for one shot (mod_main.cpp:140-174): data[nx][nt]."""
    srce = np.ascontiguousarray(srce, np.float32)
    data = np.zeros((nx, srce.size), np.float32)
    mod_numerics(numerics)
    try:
        lib().orc_mod_shot(order, nx, nz, nxb, nzb, srce.size, dx, dz, dt, fac, np.ascontiguousarray(vel2, np.float32), sx, sz, gz, srce, data)
    finally:
        mod_numerics(0)
    return data


def mod_steps(order, nx, nz, nxb, nzb, dx, dz, dt, fac, vel2, sx, sz, gz, srce, P, PP):
    """nsteps = len(srce) iterations of mod_main's loop (mod_main.cpp:147-164) from the given P / PP (as the loop holds them on entry):
    returns (P, PP, data[nx][nsteps]) as the loop holds them after the last swap."""
    L = lib()
    nxe, nze = nx + 2 * nxb, nz + 2 * nzb
    P, PP = np.array(P, np.float32, order="C"), np.array(PP, np.float32, order="C")
    vel2 = np.ascontiguousarray(vel2, np.float32)
    srce = np.ascontiguousarray(srce, np.float32)
    tx, tz = np.ones(max(nxb, 1), np.float32), np.ones(max(nzb, 1), np.float32)
    L.orc_mod_taper_tables(nxb, nzb, fac, tx, tz)
    coefs = calc_coefs(order, cxx=True)
    f = np.float32
    dx2inv, dz2inv, dt2 = f((1. / dx) * (1. / dx)), f((1. / dz) * (1. / dz)), f(f(dt) * f(dt))      # fd.c:13-15
    lap = np.zeros((nxe, nze), np.float32)
    data = np.zeros((nx, srce.size), np.float32)
    for it in range(srce.size):
        L.orc_mod_fd_step(order, coefs, dx2inv, dz2inv, dt2, P, PP, vel2, lap, nze, nxe)
        L.orc_mod_ptsrc(sx, sz, nxe, nze, srce[it], PP)
        L.orc_mod_taper_apply(PP, nx, nz, nxb, nzb, tx, tz)
        L.orc_mod_taper_apply(P, nx, nz, nxb, nzb, tx, tz)
        data[:, it] = P[nxb:nxb + nx, gz]
        P, PP = PP, P
    return P, PP, data


def mod_taper_apply(field, nx, nz, nxb, nzb, fac, times=1):
    """taper_apply (taper.c:46-66) `times` times, on a copy."""
    tx, tz = np.ones(max(nxb, 1), np.float32), np.ones(max(nzb, 1), np.float32)
    lib().orc_mod_taper_tables(nxb, nzb, fac, tx, tz)
    out = np.array(field, np.float32, order="C")
    for _ in range(times):
        lib().orc_mod_taper_apply(out, nx, nz, nxb, nzb, tx, tz)
    return out


def rtm_stored_shot(order, nx, nz, nxb, nzb, dx, dz, dt, fac, vel2, sx, sz, gz, srce, dobs, shot=0, numerics=0):
    """rtm_main's loop for one shot (rtm_main.cpp:158-240): imloc[nx][nz]; dobs is the whole gather [ns][nx][nt]."""
    srce = np.ascontiguousarray(srce, np.float32)
    dobs = np.ascontiguousarray(dobs, np.float32).ravel()
    imloc = np.zeros((nx, nz), np.float32)
    mod_numerics(numerics)
    try:
        lib().orc_rtm_stored_shot(order, nx, nz, nxb, nzb, srce.size, dx, dz, dt, fac, np.ascontiguousarray(vel2, np.float32), sx, sz, gz, srce,
                                  dobs, dobs.size, shot, imloc)
    finally:
        mod_numerics(0)
    return imloc


def image_laplacian(img, dx, dz):
    """laplace.f90:25-29 on img[nx][nz]."""
    img = np.ascontiguousarray(img, np.float32)
    out = np.zeros_like(img)
    lib().orc_image_laplacian(img, img.shape[0], img.shape[1], dx, dz, out)
    return out


def image_compare(a, b, want_diff=False):
    """The reference's `./psnr file1 file2`: (mse, rmse, snr, psnr) and optionally the difference a - b."""
    a, b = np.ascontiguousarray(a, np.float32).ravel(), np.ascontiguousarray(b, np.float32).ravel()
    st = (C.c_double * 4)()
    diff = np.zeros_like(a) if want_diff else None
    lib().orc_image_compare(a, b, a.size, diff.ctypes.data if want_diff else None, st)
    return (tuple(st), diff) if want_diff else tuple(st)


def psnr_lines(stats):
    """The four lines the reference tool prints for these values."""
    return "".join("%-10s %15e\n" % (k, v) for k, v in zip(("MSE:", "RMSE:", "SNR:", "PSNR:"), stats))


def ref_lapfilt():
    """Path of the reference's laplace.f90 built with flang (oracle/_ref/lapfilt); None when not built."""
    exe = os.path.join(_HERE, "_ref", "lapfilt")
    return exe if os.path.exists(exe) else None


def ref_dd_lib():
    """The CPU-serial sibling's fd.c / taper.c / ptsrc.c compiled unmodified with g++ (oracle/_ref/libref_dd.so); None when not built."""
    so = os.path.join(_HERE, "_ref", "libref_dd.so")
    return C.CDLL(so) if os.path.exists(so) else None
