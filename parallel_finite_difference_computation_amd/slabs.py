"""ctypes mirror of the multi-GPU part of libfdwave.so (include/fdwave.h: fdw_comm_*, fdw_slabs_*): communicators (RCCL, or ranks as
host threads of one process) and the forward / backward loops of rtm_code on one rank's slab of an x-decomposed grid, with the halo
exchange over RCCL / xGMI inside the C library.  The reference has no multi-GPU path; the decomposed result is
bit-identical to the single-domain one.  decomp.py holds the same scheme in Python as the test harness (gloo, CPU oracle stepper)."""
import ctypes as C
import sys

import numpy as np

from ._lib import Params, check, lib
from .api import FDWave, _f32

ID_BYTES = 128


def _one_rccl():
    """Before the first RCCL entry point of libfdwave.so in a Python process: load PyTorch if it is installed.  The library opens librccl.so.1
    with dlopen, and the dynamic linker hands it the copy PyTorch already holds (same SONAME) -- but not the other way round: libtorch_hip asks
    for "librccl.so" by file name next to itself, so a process that opened the system RCCL first and imports torch later ends up with TWO RCCL
    libraries whose global state collides (seen as "double free or corruption" at interpreter exit once ncclAllReduce had run in one of them).
    C programs (rtm_code slabs=N) never load PyTorch and are not concerned."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass



class Comm:
    """One rank of a communicator.  Build with Comm.rccl(...) or Comm.local(world)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def unique_id():
        """ncclGetUniqueId: call on rank 0 and hand the bytes to every rank."""
        _one_rccl()
        buf = C.create_string_buffer(ID_BYTES)
        check(lib().fdw_comm_get_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, unique_id, rank, world, device):
        _one_rccl()
        h = C.c_void_p()
        check(lib().fdw_comm_init_rank(C.create_string_buffer(bytes(unique_id), ID_BYTES), rank, world, device, C.byref(h)))
        return cls(h)

    @classmethod
    def local(cls, world, devices=None):
        """`world` ranks inside this process (drive each from its own thread); devices[r] = HIP ordinal of rank r (default: all 0)."""
        arr = (C.c_void_p * world)()
        dev = (C.c_int * world)(*devices) if devices is not None else None
        check(lib().fdw_comm_init_local(world, dev, arr))
        return [cls(C.c_void_p(arr[r])) for r in range(world)]

    @classmethod
    def stub(cls, rank, world, device=0):
        """Rank `rank` of a world whose other ranks do not exist: nothing travels.  Timing experiments only."""
        h = C.c_void_p()
        check(lib().fdw_comm_init_stub(rank, world, device, C.byref(h)))
        return cls(h)

    @classmethod
    def shm(cls, name, rank, world, device=0, box_bytes=64 << 20):
        """Rank `rank` of `world` PROCESSES that stage their halo blocks through the POSIX shared-memory segment `name` ('/...'): the test
        transport that lets the C slab drivers run as real processes on one GPU.  Collective; box_bytes = largest message of one exchange."""
        h = C.c_void_p()
        check(lib().fdw_comm_init_shm(name.encode(), rank, world, device, box_bytes, C.byref(h)))
        return cls(h)

    kind = property(lambda self: {1: "rccl", 2: "local", 3: "shm"}.get(lib().fdw_comm_kind(self._h), "stub"))
    rank = property(lambda self: lib().fdw_comm_rank(self._h))
    world = property(lambda self: lib().fdw_comm_world(self._h))
    device = property(lambda self: lib().fdw_comm_device(self._h))

    def selftest(self):
        check(lib().fdw_comm_selftest(self._h))

    def allreduce(self, value, op="sum"):
        v = C.c_double(value)
        check(lib().fdw_comm_allreduce(self._h, C.byref(v), 1 if op == "max" else 0))
        return v.value

    def barrier(self):
        check(lib().fdw_comm_barrier(self._h))

    def close(self):
        if self._h is not None and self._h.value:
            lib().fdw_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        if sys.is_finalizing():      # interpreter shutdown: the HIP runtime may already be gone, and the process frees everything anyway
            return
        try:
            self.close()
        except Exception:
            pass


class Slabs:
    """This rank's share of an x-slab decomposition (fdw_slabs_create).  Collective: every rank of `comm` constructs one."""

    def __init__(self, order, nxe, nze, nxb, nzb, nt, fac, dx, dz, dt, comm=None, compat=True, device=0, ksteps=0, numerics=0):
        self.params = Params(order, nxe, nze, nxb, nzb, nt, dx, dz, dt, fac, int(compat), 0, 0, int(numerics))
        self._h = C.c_void_p()
        self.comm = comm
        check(lib().fdw_slabs_create(C.byref(self.params), comm._h if comm is not None else None, device, ksteps, C.byref(self._h)))
        v = [C.c_int() for _ in range(6)]
        check(lib().fdw_slabs_geometry(self._h, *[C.byref(x) for x in v]))
        self.x_off, self.nxl, self.own0, self.own1, self.ksteps, self.nbuf = [x.value for x in v]
        self.nxe, self.nze, self.nxb, self.nzb, self.nt = nxe, nze, nxb, nzb, nt
        self.nx, self.nz = nxe - 2 * nxb, nze - 2 * nzb
        self.ctx = FDWave.__new__(FDWave)                    # a view of the slab's own context (not owned: never destroyed from here)
        self.ctx._h = None
        self._ctx_h = C.c_void_p(lib().fdw_slabs_ctx(self._h))
        self.pitch = lib().fdw_pitch(self._ctx_h)
        self.stream = lib().fdw_slabs_stream(self._h)

    def ctx_call(self, name, *args):
        """Call an fdw_* function of the C ABI on the slab's context (tuning, introspection)."""
        return getattr(lib(), name)(self._ctx_h, *args)

    def dev_forward(self, bufs, d_v2, d_srce, sx, sz, it0, nsteps, first_pp_twice=False, ip=0, ipp=1):
        arr = (C.c_void_p * len(bufs))(*bufs)
        a, b = C.c_int(ip), C.c_int(ipp)
        check(lib().fdw_slabs_dev_forward(self._h, arr, d_v2, d_srce, sx, sz, it0, nsteps, int(first_pp_twice), C.byref(a), C.byref(b)))
        return a.value, b.value

    def back_buffers(self):
        """(nfb, nrb): source-field and receiver-field buffers fdw_slabs_dev_back works on (6 and 4 with the wave pipeline, else 2 and 2)."""
        a, b = C.c_int(), C.c_int()
        check(lib().fdw_slabs_back_buffers(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def dev_back(self, f, r, d_v2, d_samples, gz, d_img, it0, nsteps, role=(0, 1, 0, 1)):
        """fd_back's loop on the slab; role = indices of (F_{k-1}, F_{k-2}) in f and (r^k, r^{k-1}) in r; returns the roles after the loop."""
        fa, ra = (C.c_void_p * len(f))(*f), (C.c_void_p * len(r))(*r)
        ro = (C.c_int * 4)(*role)
        check(lib().fdw_slabs_dev_back(self._h, fa, ra, d_v2, d_samples, gz, d_img, it0, nsteps, ro))
        return tuple(ro)

    def taper_finalize(self, d_f):
        check(lib().fdw_dev_taper_finalize(self._ctx_h, d_f, self.stream))

    def synchronize(self):
        check(lib().fdw_slabs_synchronize(self._h))

    def set_stub(self, on):
        """Measurement only: halo exchanges move nothing while on (bench.py's exposed-communication figure)."""
        check(lib().fdw_slabs_set_stub(self._h, int(bool(on))))

    def shot(self, v2, sx, sz, gz, srce, d_obs, imloc=None, want_fields=False):
        """One shot of rtm_code's loop on the decomposed grid (global host arrays in; this rank's OWNED rows of imloc / P / PP out)."""
        shape = (self.nxe, self.nze)
        imloc = np.zeros((self.nx, self.nz), np.float32) if imloc is None else np.array(imloc, np.float32, order="C")
        P = np.zeros(shape, np.float32) if want_fields else None
        PP = np.zeros(shape, np.float32) if want_fields else None
        check(lib().fdw_slabs_shot(self._h, _f32(v2, shape), sx, sz, gz, _f32(srce, (self.nt,)), _f32(d_obs, (self.nx, self.nt)), imloc,
                                   P.ctypes.data if want_fields else None, PP.ctypes.data if want_fields else None))
        return (imloc, P, PP) if want_fields else imloc

    def owned_interior_rows(self):
        """Rows [a, b) of the image imloc[nx][nz] this rank produces."""
        return max(self.own0, self.nxb) - self.nxb, max(min(self.own1, self.nxb + self.nx), self.nxb) - self.nxb

    def close(self):
        if self._h is not None and self._h.value:
            lib().fdw_slabs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        if sys.is_finalizing():      # interpreter shutdown: the HIP runtime may already be gone, and the process frees everything anyway
            return
        try:
            self.close()
        except Exception:
            pass


def run_ranks(fn, world):
    """Run fn(rank) on `world` host threads (the ranks of a local communicator) and return the results in rank order; the first exception
    of any rank is re-raised."""
    import threading
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:      # noqa: BLE001 -- reported to the caller
            err[r] = e

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out
