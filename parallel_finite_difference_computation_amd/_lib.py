"""ctypes binding of libfdwave.so (include/fdwave.h).  No fallback: a missing library is an error."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDW_LIB") or os.path.join(_HERE, "libfdwave.so")   # FDW_LIB: experiment builds only

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
vp = C.c_void_p

FDW_OK, FDW_EINVAL, FDW_ENODEVICE, FDW_EHIP, FDW_ENOMEM, FDW_ESTATE = 0, -1, -2, -3, -4, -5
MODE_FWD, MODE_PLAIN, MODE_RECV = 0, 1, 2


class FdwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfdwave error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """struct fdw_params (fdwave.h) == the arguments of the reference's fd_init (fd-code.cu:200)."""
    _fields_ = [("order", C.c_int), ("nxe", C.c_int), ("nze", C.c_int), ("nxb", C.c_int), ("nzb", C.c_int),
                ("nt", C.c_int), ("dx", C.c_float), ("dz", C.c_float), ("dt", C.c_float), ("fac", C.c_float),
                ("compat", C.c_int), ("coef_cxx", C.c_int), ("dialect", C.c_int), ("numerics", C.c_int)]


class Slab(C.Structure):
    _fields_ = [("x_off", C.c_int), ("nxl", C.c_int)]


# every symbol fdwave.h declares: (name, restype, argtypes)
SIGNATURES = [
    ("fdw_last_error", C.c_char_p, []),
    ("fdw_version", C.c_int, []),
    ("fdw_create", C.c_int, [C.POINTER(Params), C.c_int, C.POINTER(vp)]),
    ("fdw_device_count", C.c_int, []),
    ("fdw_device_usable", C.c_int, [C.c_int]),
    ("fdw_create_slab", C.c_int, [C.POINTER(Params), C.POINTER(Slab), C.c_int, C.POINTER(vp)]),
    ("fdw_destroy", None, [vp]),
    ("fdw_laplacian", C.c_int, [vp, f32p, f32p]),
    ("fdw_forward", C.c_int, [vp, f32p, f32p, f32p, C.c_int, C.c_int, f32p, C.c_int]),
    ("fdw_back", C.c_int, [vp, f32p, f32p, f32p, f32p, C.c_int, f32p, C.c_int]),
    ("fdw_shot", C.c_int, [vp, f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, vp, vp]),
    ("fdw_pitch", C.c_int, [vp]),
    ("fdw_field_bytes", C.c_size_t, [vp]),
    ("fdw_dev_step", C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp]),
    ("fdw_dev_back_iter", C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp]),
    ("fdw_dev_back4", C.c_int, [vp] * 12 + [vp, C.c_int, C.c_int, vp] + [C.c_int] * 6 + [vp]),
    ("fdw_back_pipe_active", C.c_int, [vp]),
    ("fdw_dev_steps", C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    ("fdw_dev_steps_shrink", C.c_int, [vp, vp, vp, vp, vp] + [C.c_int] * 8 + [vp]),
    ("fdw_dev_step2", C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    ("fdw_dev_step4", C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, vp] + [C.c_int] * 7 + [vp]),
    ("fdw_dev_steps2", C.c_int, [vp, C.POINTER(vp), vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), vp]),
    ("fdw_dev_taper_finalize", C.c_int, [vp, vp, vp]),
    ("fdw_dev_check_field", C.c_int, [vp, vp, vp]),
    ("fdw_model_shot", C.c_int, [vp, f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p]),
    ("fdw_image_laplacian", C.c_int, [C.c_int, f32p, C.c_int, C.c_int, C.c_float, C.c_float, f32p]),
    ("fdw_image_compare", C.c_int, [C.c_int, f32p, f32p, C.c_size_t, vp, C.POINTER(C.c_double), C.c_int]),
    ("fdw_rtm_stored_shot", C.c_int, [vp, f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p, C.c_size_t, C.c_int, f32p]),
    ("fdw_set_store_budget", C.c_int, [vp, C.c_size_t]),
    ("fdw_store_segments", C.c_int, [vp]),
    ("fdw_border_draws", C.c_longlong, [C.c_int] * 4),
    ("fdw_model_resident", C.c_int, [vp, f32p]),
    ("fdw_dev_extendvel_linear", C.c_int, [vp, C.c_ulonglong, vp]),
    ("fdw_shot_resident", C.c_int, [vp, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, vp, vp]),
    ("fdw_shot_batch", C.c_int, [vp, C.c_int, vp, C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p]),
    ("fdw_shot_batch_max", C.c_int, [vp]),
    ("fdw_model_shot_batch", C.c_int, [vp, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p]),
    ("fdw_rand_stream", C.c_int, [vp, C.c_ulonglong, C.c_longlong, vp]),
    ("fdw_dev_model_steps", C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]),
    ("fdw_mod_extendvel", None, [C.c_int] * 4 + [f32p]),
    ("fdw_mod_ricker_wavelet", None, [C.c_int, C.c_float, C.c_float, f32p]),
    ("fdw_mod_taper_tables", None, [C.c_int, C.c_int, C.c_float, f32p, f32p]),
    ("fdw_dev_laplacian", C.c_int, [vp, vp, vp, vp]),
    ("fdw_upload_field", C.c_int, [vp, vp, f32p]),
    ("fdw_download_field", C.c_int, [vp, f32p, vp]),
    ("fdw_comm_get_unique_id", C.c_int, [C.c_char_p]),
    ("fdw_comm_init_rank", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    ("fdw_comm_init_local", C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]),
    ("fdw_comm_init_stub", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    ("fdw_comm_init_shm", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(vp)]),
    ("fdw_comm_kind", C.c_int, [vp]),
    ("fdw_comm_destroy", None, [vp]),
    ("fdw_comm_rank", C.c_int, [vp]),
    ("fdw_comm_world", C.c_int, [vp]),
    ("fdw_comm_device", C.c_int, [vp]),
    ("fdw_comm_is_local", C.c_int, [vp]),
    ("fdw_comm_allreduce", C.c_int, [vp, C.POINTER(C.c_double), C.c_int]),
    ("fdw_comm_barrier", C.c_int, [vp]),
    ("fdw_comm_selftest", C.c_int, [vp]),
    ("fdw_slabs_create", C.c_int, [C.POINTER(Params), vp, C.c_int, C.c_int, C.POINTER(vp)]),
    ("fdw_slabs_destroy", None, [vp]),
    ("fdw_slabs_ctx", vp, [vp]),
    ("fdw_slabs_geometry", C.c_int, [vp] + [C.POINTER(C.c_int)] * 6),
    ("fdw_slabs_stream", vp, [vp]),
    ("fdw_slabs_synchronize", C.c_int, [vp]),
    ("fdw_slabs_dev_forward", C.c_int, [vp, C.POINTER(vp), vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("fdw_slabs_back_buffers", C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("fdw_slabs_dev_back", C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), vp, vp, C.c_int, vp, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    ("fdw_slabs_shot", C.c_int, [vp, f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, vp, vp]),
    ("fdw_slabs_set_stub", C.c_int, [vp, C.c_int]),
    ("fdw_set_tuning", C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("fdw_get_tables", C.c_int, [vp, vp, vp, vp, vp]),
    ("fdw_get_extents", C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("fdw_two_step_active", C.c_int, [vp]),
    ("fdw_steps_per_pass", C.c_int, [vp]),
    ("fdw_selftest", C.c_int, [vp]),
    ("fdw_trace_active", C.c_int, []),
    ("fdw_calc_coefs", C.c_int, [C.c_int, C.c_int, f32p]),
    ("fdw_ricker_wavelet", None, [C.c_int, C.c_float, C.c_float, f32p]),
    ("fdw_taper_tables", None, [C.c_int, C.c_int, C.c_float, vp, vp]),
    ("fdw_extendvel_linear", None, [C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
    ("fdw_srand", None, [C.c_uint]),
]

_lib = None


def lib():
    """Load libfdwave.so.  Raises (never falls back) if it has not been built: run
    `python -c 'import __graft_entry__ as g; g.build()'` or `make -C parallel_finite_difference_computation_amd/csrc`."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                              "(hipcc, gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, res, args in SIGNATURES:
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise FdwError(rc, lib().fdw_last_error().decode(errors="replace"))
