"""Host-side mirror of the reference's propagation API on top of libfdwave.so.

The reference (cuda_reference_RTM/src/fd-code.cu) exposes fd_init / fd_forward / fd_back as C
functions over file-scope globals; `FDWave` is that state as an object and keeps the reference's
argument meaning (extended-grid sizes, nz-before-nx order of the raw arrays, sx/sz/gz on the
extended grid).  All arrays are numpy float32 [nxe][nze] (x slow, z contiguous, fd-code.cu:58).
"""
import ctypes as C
import sys

import numpy as np

from . import _lib
from ._lib import Params, Slab, check, lib


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


# ---- host formulas (functions.c restated in csrc/fdw_host.c) --------------------------------------
def calc_coefs(order, cxx=False):
    """calc_coefs (functions.c:113-158); cxx=True = the stencil program's float-overload variant."""
    c = np.zeros(order + 1, np.float32)
    check(lib().fdw_calc_coefs(order, int(cxx), c))
    return c


def ricker_wavelet(nt, dt, fpeak):
    """ricker_wavelet (functions.c:328-334)."""
    s = np.zeros(nt, np.float32)
    lib().fdw_ricker_wavelet(nt, dt, fpeak, s)
    return s


def taper_tables(nxb, nzb, fac):
    """taper_x / taper_z of fd_init_cuda (fd-code.cu:159-166)."""
    tx = np.zeros(max(nxb, 1), np.float32)
    tz = np.zeros(max(nzb, 1), np.float32)
    lib().fdw_taper_tables(nxb, nzb, fac, tx.ctypes.data, tz.ctypes.data)
    return tx[:nxb], tz[:nzb]


def srand(seed):
    """Reseed the private restatement of glibc rand() behind extendvel_linear (fresh process == srand(1))."""
    lib().fdw_srand(seed)


def extendvel_linear(vpe, nx, nz, nxb, nzb):
    """extendvel_linear (functions.c:336-394), in place; draws glibc's rand() stream (private generator)."""
    if vpe.shape != (nx + 2 * nxb, nz + 2 * nzb) or vpe.dtype != np.float32 or not vpe.flags.c_contiguous:
        raise ValueError("vpe must be C-contiguous float32 [nx+2nxb][nz+2nzb]")
    lib().fdw_extendvel_linear(nx, nz, nxb, nzb, vpe)
    return vpe


def mod_extendvel(vel, nx, nz, nxb, nzb):
    """taper.c:7-23 of the CPU-serial sibling: replicate the edge values outwards, in place on [nxe][nze]."""
    vel = _f32(vel, (nx + 2 * nxb, nz + 2 * nzb))
    lib().fdw_mod_extendvel(nx, nz, nxb, nzb, vel)
    return vel


def mod_ricker_wavelet(nt, dt, fpeak):
    s = np.zeros(nt, np.float32)
    lib().fdw_mod_ricker_wavelet(nt, dt, fpeak, s)
    return s


def mod_taper_tables(nxb, nzb, fac):
    tx, tz = np.ones(max(nxb, 1), np.float32), np.ones(max(nzb, 1), np.float32)
    lib().fdw_mod_taper_tables(nxb, nzb, fac, tx, tz)
    return tx[:nxb], tz[:nzb]


def image_laplacian(img, dx, dz, device=0):
    """The reference's offline image filter (models/3lay_mod/laplace.f90:25-29) on img[nx][nz]; runs on the GPU."""
    img = np.ascontiguousarray(img, np.float32)
    out = np.zeros_like(img)
    check(lib().fdw_image_laplacian(device, img, img.shape[0], img.shape[1], dx, dz, out))
    return out


def image_compare(a, b, device=0, want_diff=False, exact_sums=False):
    """The reference's `./psnr file1 file2` (models/marmousi/psnr) on the GPU: dict(mse, rmse, snr, psnr) and, if asked, the difference a - b.
    exact_sums=False: the tool's own (serial fp32) sums, its figures digit for digit; True: a parallel reduction in double."""
    a, b = np.ascontiguousarray(a, np.float32).ravel(), np.ascontiguousarray(b, np.float32).ravel()
    if a.size != b.size:
        raise ValueError("sizes differ")
    st = (C.c_double * 4)()
    diff = np.zeros_like(a) if want_diff else None
    check(lib().fdw_image_compare(device, a, b, a.size, diff.ctypes.data if want_diff else None, st, int(exact_sums)))
    out = dict(mse=st[0], rmse=st[1], snr=st[2], psnr=st[3])
    return (out, diff) if want_diff else out


class FDWave:
    """One fd_init (fd-code.cu:200-224 / fd-source-code.cu:241-262) worth of state on one MI355X."""

    def __init__(self, order, nxe, nze, nxb=0, nzb=0, nt=0, fac=1.0, dx=1.0, dz=1.0, dt=0.0, *, compat=True,
                 coef_cxx=False, device=0, slab=None, dialect=0, numerics=0):
        """dialect 0: the CUDA programs (stencil_code / rtm_code); 1: the forward-modelling producer of the CPU-serial sibling
        (mod_main: model_shot only); 2: its stored-wavefield RTM (rtm_main: rtm_stored_shot only).
        numerics 0: the reference's arithmetic operation for operation (bit-exact); 1: FAST -- symmetric taps summed first + fused
        multiply-adds in the Laplacian (fdwave.h), within 1e-5 of the former, RTM dialect only."""
        self.params = Params(order, nxe, nze, nxb, nzb, nt, dx, dz, dt, fac, int(compat), int(coef_cxx), int(dialect), int(numerics))
        self._h = C.c_void_p()
        if slab is None:
            check(lib().fdw_create(C.byref(self.params), device, C.byref(self._h)))
            self.x_off, self.nxl = 0, nxe
        else:
            self.x_off, self.nxl = slab
            s = Slab(self.x_off, self.nxl)
            check(lib().fdw_create_slab(C.byref(self.params), C.byref(s), device, C.byref(self._h)))
        self.order, self.nxe, self.nze, self.nxb, self.nzb, self.nt = order, nxe, nze, nxb, nzb, nt
        self.nx, self.nz = nxe - 2 * nxb, nze - 2 * nzb
        self.device = device
        self.pitch = lib().fdw_pitch(self._h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().fdw_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        if sys.is_finalizing():      # interpreter shutdown: the HIP runtime may already be gone, and the process frees everything anyway
            return
        try:
            self.close()
        except Exception:
            pass

    # ---- introspection --------------------------------------------------------------------------
    def tables(self):
        cx = np.zeros(self.order + 1, np.float32)
        cz = np.zeros(self.order + 1, np.float32)
        tx = np.zeros(max(self.nxb, 1), np.float32)
        tz = np.zeros(max(self.nzb, 1), np.float32)
        check(lib().fdw_get_tables(self._h, cx.ctypes.data, cz.ctypes.data, tx.ctypes.data, tz.ctypes.data))
        return cx, cz, tx[:self.nxb], tz[:self.nzb]

    def extents(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(lib().fdw_get_extents(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_tuning(self, xchunk=0, wz=0, use_generic=False, prefetch=0, two_step=0, force_edge=False):
        """two_step: 0 auto, 1 two steps per pass, 4 the four-wave pipeline, -1 never (temporal blocking in forward loops).  force_edge is obsolete and ignored."""
        check(lib().fdw_set_tuning(self._h, xchunk, wz, int(use_generic), prefetch, int(two_step)))

    def two_step_active(self):
        return bool(lib().fdw_two_step_active(self._h))

    def steps_per_pass(self):
        """Time steps one launch of the forward loops advances on this grid: 4 (wave pipeline), 2 (two-step kernel) or 1."""
        return int(lib().fdw_steps_per_pass(self._h))

    def selftest(self):
        check(lib().fdw_selftest(self._h))

    # ---- host-array API (the reference's L2 seam) -----------------------------------------------
    def laplacian(self, p):
        """stencil_code's single kernel_lap launch (fd-source-code.cu:320-333)."""
        p = _f32(p, (self.nxe, self.nze))
        out = np.empty_like(p)
        check(lib().fdw_laplacian(self._h, p, out))
        return out

    def forward(self, v2, sx, sz, srce, p=None, pp=None, nsteps=None):
        """fd_forward (fd-code.cu:247-288): returns (P, PP) = (d_p, d_pp) after the loop."""
        shape = (self.nxe, self.nze)
        p = np.zeros(shape, np.float32) if p is None else np.array(p, np.float32, order="C")
        pp = np.zeros(shape, np.float32) if pp is None else np.array(pp, np.float32, order="C")
        srce = _f32(srce)
        nsteps = len(srce) if nsteps is None else nsteps
        if nsteps > len(srce):
            raise ValueError("srce shorter than nsteps")
        check(lib().fdw_forward(self._h, p, pp, _f32(v2, shape), sx, sz, srce, nsteps))
        return p, pp

    def back(self, v2, snap0, snap1, d_obs, gz, imloc=None, nsteps=None):
        """fd_back (fd-code.cu:290-341): d_obs [nx][nt]; returns imloc [nx][nz]."""
        shape = (self.nxe, self.nze)
        imloc = np.zeros((self.nx, self.nz), np.float32) if imloc is None else np.array(imloc, np.float32, order="C")
        nsteps = self.nt if nsteps is None else nsteps
        check(lib().fdw_back(self._h, _f32(v2, shape), _f32(snap0, shape), _f32(snap1, shape),
                             _f32(d_obs, (self.nx, self.nt)), gz, imloc, nsteps))
        return imloc

    def shot(self, v2, sx, sz, gz, srce, d_obs, imloc=None, want_fields=False):
        """One shot of rtm_code's loop (fd-code.cu:496-518), device resident."""
        shape = (self.nxe, self.nze)
        imloc = np.zeros((self.nx, self.nz), np.float32) if imloc is None else np.array(imloc, np.float32, order="C")
        P = np.zeros(shape, np.float32) if want_fields else None
        PP = np.zeros(shape, np.float32) if want_fields else None
        check(lib().fdw_shot(self._h, _f32(v2, shape), sx, sz, gz, _f32(srce, (self.nt,)),
                             _f32(d_obs, (self.nx, self.nt)), imloc,
                             P.ctypes.data if want_fields else None, PP.ctypes.data if want_fields else None))
        return (imloc, P, PP) if want_fields else imloc

    # ---- random-border model generated on the device ----
    def border_draws(self):
        """rand() calls one extendvel_linear (functions.c:336-394) consumes on this geometry."""
        return int(lib().fdw_border_draws(self.nx, self.nz, self.nxb, self.nzb))

    def model_resident(self, vp):
        """Upload the interior velocity model vp[nx][nz] (not squared) once; the per-shot borders are then drawn on the device."""
        check(lib().fdw_model_resident(self._h, _f32(vp, (self.nx, self.nz))))

    def dev_extendvel_linear(self, draw_offset, want_vel=False):
        """extendvel_linear + vel2 = vpe * vpe (fd-code.cu:486-494) in HBM, from draws [draw_offset, draw_offset + border_draws()) of the
        unseeded glibc rand() stream.  want_vel: also return the extended model [nxe][nze]."""
        vel = np.zeros((self.nxe, self.nze), np.float32) if want_vel else None
        check(lib().fdw_dev_extendvel_linear(self._h, int(draw_offset), vel.ctypes.data if want_vel else None))
        return vel

    def shot_resident(self, sx, sz, gz, srce, d_obs, imloc=None, want_fields=False):
        """shot() on the squared model dev_extendvel_linear left in HBM."""
        shape = (self.nxe, self.nze)
        imloc = np.zeros((self.nx, self.nz), np.float32) if imloc is None else np.array(imloc, np.float32, order="C")
        P = np.zeros(shape, np.float32) if want_fields else None
        PP = np.zeros(shape, np.float32) if want_fields else None
        check(lib().fdw_shot_resident(self._h, sx, sz, gz, _f32(srce, (self.nt,)), _f32(d_obs, (self.nx, self.nt)), imloc,
                                      P.ctypes.data if want_fields else None, PP.ctypes.data if want_fields else None))
        return (imloc, P, PP) if want_fields else imloc

    def shot_batch(self, nshots, sx0, dsx, sz, gz, srce, d_obs, v2_all=None, draw_offset=0, imloc=None):
        """`nshots` consecutive shots of rtm_code's loop (fd-code.cu:480-520) through one launch per time step.  d_obs[nshots][nx][nt];
        v2_all[nshots][nxe][nze], or None = border models drawn on the device from the resident interior model at draws
        draw_offset + b * border_draws().  Returns imloc[nshots][nx][nz]."""
        imloc = np.zeros((nshots, self.nx, self.nz), np.float32) if imloc is None else np.array(imloc, np.float32, order="C")
        v2p = None if v2_all is None else _f32(v2_all, (nshots, self.nxe, self.nze)).ctypes.data
        check(lib().fdw_shot_batch(self._h, nshots, v2p, int(draw_offset), sx0, dsx, sz, gz, _f32(srce, (self.nt,)),
                                   _f32(d_obs, (nshots, self.nx, self.nt)), imloc))
        return imloc

    def shot_batch_max(self):
        return int(lib().fdw_shot_batch_max(self._h))

    def rand_stream(self, draw_offset, n):
        """Draws [draw_offset, draw_offset + n) of the unseeded glibc rand() stream as the device generator produces them."""
        out = np.zeros(n, np.int32)
        check(lib().fdw_rand_stream(self._h, int(draw_offset), n, out.ctypes.data))
        return out

    def model_shot(self, vel2, sx, sz, gz, srce):
        """One shot of mod_main's loop (dpct_gpu_rtm_domain_division/src/mod_main.cpp:140-174): the gather data[nx][nt]."""
        srce = _f32(srce)
        data = np.zeros((self.nx, srce.size), np.float32)
        check(lib().fdw_model_shot(self._h, _f32(vel2, (self.nxe, self.nze)), sx, sz, gz, srce, srce.size, data))
        return data

    def model_shot_batch(self, nshots, vel2, sx0, dsx, sz, gz, srce):
        """`nshots` consecutive shots of mod_main's loop through one launch per time step: data[nshots][nx][nt]."""
        srce = _f32(srce)
        data = np.zeros((nshots, self.nx, srce.size), np.float32)
        check(lib().fdw_model_shot_batch(self._h, nshots, _f32(vel2, (self.nxe, self.nze)), sx0, dsx, sz, gz, srce, srce.size, data))
        return data

    def rtm_stored_shot(self, vel2, sx, sz, gz, srce, dobs, shot=0):
        """One shot of the sibling's stored-wavefield RTM (dpct_gpu_rtm_domain_division/src/rtm_main.cpp:158-240); dobs is the WHOLE
        gather [ns][nx][nt] (the reference reads one sample past each trace, see fdwave.h).  Returns imloc[nx][nz]."""
        srce = _f32(srce)
        dobs = np.ascontiguousarray(dobs, np.float32).ravel()
        imloc = np.zeros((self.nx, self.nz), np.float32)
        check(lib().fdw_rtm_stored_shot(self._h, _f32(vel2, (self.nxe, self.nze)), sx, sz, gz, srce, srce.size, dobs, dobs.size, shot, imloc))
        return imloc

    def dev_check_field(self, d_f, stream=None):
        """The precondition of the lazy damping checked on a DEVICE array (compat extents: the damped strip must be zero on the rows the
        reference never time-steps; fdwave.h).  Raises FdwError if it is violated; synchronises the stream."""
        check(lib().fdw_dev_check_field(self._h, d_f, stream))

    def set_store_budget(self, nbytes):
        """Bytes the stored source fields of rtm_stored_shot may occupy (0: no limit of our own).  Below nt + 1 fields the shot checkpoints and
        recomputes (fdwave.h); the image stays bit-identical."""
        check(lib().fdw_set_store_budget(self._h, int(nbytes)))

    def store_segments(self):
        """Segments the last rtm_stored_shot was cut into (1: every field was kept)."""
        return int(lib().fdw_store_segments(self._h))

    def field_bytes(self):
        return int(lib().fdw_field_bytes(self._h))

    def dev_model_steps(self, d_p, d_pp, d_v2, d_srce, sx, sz, gz, d_rec, it0, nsteps, stream=None):
        check(lib().fdw_dev_model_steps(self._h, d_p, d_pp, d_v2, d_srce, sx, sz, gz, d_rec, it0, nsteps, stream))

    # ---- device-array API (raw pointers; see device.py for torch helpers) -----------------------
    def dev_step(self, mode, d_p, d_pp, d_v2, r0=0, r1=None, pp_twice=True, d_inj=None, inj_x=-1, inj_z=0,
                 d_psrc=None, d_img=None, stream=None):
        r1 = self.nxl if r1 is None else r1
        check(lib().fdw_dev_step(self._h, mode, d_p, d_pp, d_v2, r0, r1, int(pp_twice), d_inj, inj_x, inj_z,
                                 d_psrc, d_img, stream))

    def dev_back_iter(self, step_source, d_f1, d_f0, d_pr, d_ppr, d_v2, r0, r1, pp_twice, d_samples, gz, d_img, stream=None):
        """One iteration of fd_back's loop (fd-code.cu:302-339) on local rows [r0, r1): see fdwave.h."""
        check(lib().fdw_dev_back_iter(self._h, int(step_source), d_f1, d_f0, d_pr, d_ppr, d_v2, r0, r1, int(pp_twice), d_samples, gz, d_img, stream))

    def dev_steps(self, d_p, d_pp, d_v2, d_srce, sx, sz, it0, nsteps, first_pp_twice=False, stream=None):
        check(lib().fdw_dev_steps(self._h, d_p, d_pp, d_v2, d_srce, sx, sz, it0, nsteps, int(first_pp_twice), stream))

    def dev_steps_shrink(self, d_p, d_pp, d_v2, d_srce, sx, sz, it0, nsteps, first_pp_twice, j0, shrink_lo, shrink_hi, stream=None):
        check(lib().fdw_dev_steps_shrink(self._h, d_p, d_pp, d_v2, d_srce, sx, sz, it0, nsteps, int(first_pp_twice), j0,
                                         int(shrink_lo), int(shrink_hi), stream))

    def dev_step2(self, d_p, d_pp, d_v2, d_out1, d_out2, pp_twice=True, d_srce_it=None, sx=-1, sz=0, stream=None):
        """Two forward iterations in one pass (temporal blocking); d_p is the NEWEST field."""
        check(lib().fdw_dev_step2(self._h, d_p, d_pp, d_v2, d_out1, d_out2, int(pp_twice), d_srce_it, sx, sz, stream))

    def dev_step4(self, d_p, d_pp, d_v2, d_out1, d_out2, pp_twice=True, d_srce_it=None, sx=-1, sz=0, r0=0, r1=-1, r0b=0, r1b=0, xchunk=0, stream=None):
        """Four forward iterations in one pass (wave pipeline) on local rows [r0, r1) (+ [r0b, r1b)); r1 < 0 = all rows."""
        check(lib().fdw_dev_step4(self._h, d_p, d_pp, d_v2, d_out1, d_out2, int(pp_twice), d_srce_it, sx, sz, r0, r1, r0b, r1b, xchunk, stream))

    def dev_steps2(self, bufs, d_v2, d_srce, sx, sz, it0, nsteps, first_pp_twice=False, ip=0, ipp=1, stream=None):
        """nsteps iterations over four rotating device buffers (pairs via the two-step kernel).
        Returns the indices (ip, ipp) of the reference's (d_p, d_pp) after the loop."""
        arr = (C.c_void_p * 4)(*bufs)
        a, b = C.c_int(ip), C.c_int(ipp)
        check(lib().fdw_dev_steps2(self._h, arr, d_v2, d_srce, sx, sz, it0, nsteps, int(first_pp_twice), C.byref(a), C.byref(b), stream))
        return a.value, b.value

    def dev_taper_finalize(self, d_f, stream=None):
        check(lib().fdw_dev_taper_finalize(self._h, d_f, stream))

    def dev_laplacian(self, d_p, d_lap, stream=None):
        check(lib().fdw_dev_laplacian(self._h, d_p, d_lap, stream))

    def upload(self, d_dst, h_src):
        check(lib().fdw_upload_field(self._h, d_dst, _f32(h_src, (self.nxl, self.nze))))

    def download(self, d_src):
        out = np.empty((self.nxl, self.nze), np.float32)
        check(lib().fdw_download_field(self._h, out, d_src))
        return out


# ---- literal reference names --------------------------------------------------------------------
_state = {"ctx": None}


def fd_init(order, nxe, nze, nxb, nzb, nt, ns, fac, dx, dz, dt):
    """fd_init (fd-code.cu:200): creates the process-global propagation state like the reference."""
    if _state["ctx"] is not None:
        _state["ctx"].close()
    _state["ctx"] = FDWave(order, nxe, nze, nxb, nzb, nt, fac, dx, dz, dt, compat=True)
    return _state["ctx"]


def fd_forward(order, p, pp, v2, nz, nx, nt, is_, sz, sx, srce, propag=0):
    """fd_forward (fd-code.cu:247): note the reference's nz-before-nx order; p/pp are updated in place."""
    ctx = _state["ctx"]
    if ctx is None:
        raise _lib.FdwError(_lib.FDW_ESTATE, "fd_init has not been called")
    P, PP = ctx.forward(v2, int(sx[is_]), sz, srce, p, pp, nt)
    p[...] = P
    pp[...] = PP


def fd_back(order, p, pp, pr, ppr, v2, nz, nx, nt, is_, sz, gz, snaps, imloc, d_obs):
    """fd_back (fd-code.cu:290): d_obs[is] is the [nx][nt] gather; imloc is accumulated in place."""
    ctx = _state["ctx"]
    if ctx is None:
        raise _lib.FdwError(_lib.FDW_ESTATE, "fd_init has not been called")
    imloc[...] = ctx.back(v2, snaps[0], snaps[1], np.asarray(d_obs[is_]).reshape(ctx.nx, ctx.nt), gz, imloc, nt)
