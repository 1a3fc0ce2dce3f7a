"""parallel_finite_difference_computation_amd -- MI355X (gfx950) 2-D acoustic finite-difference /
RTM hot path behind the reference's own interfaces.  The compute lives in libfdwave.so
(csrc/: hand-written HIP kernels + C ABI, include/fdwave.h); this package is the thin host side.
Importing the package does not touch the GPU; using it without libfdwave.so raises ImportError.
"""
from ._lib import FdwError, LIB_PATH, MODE_FWD, MODE_PLAIN, MODE_RECV, lib  # noqa: F401
from .api import (FDWave, calc_coefs, extendvel_linear, fd_back, fd_forward, fd_init, image_compare, image_laplacian, mod_extendvel,  # noqa: F401
                  mod_ricker_wavelet, mod_taper_tables, ricker_wavelet, srand, taper_tables)

from .slabs import Comm, Slabs, run_ranks  # noqa: F401,E402

__all__ = ["Comm", "Slabs", "run_ranks", "FDWave", "FdwError", "calc_coefs", "ricker_wavelet", "taper_tables", "extendvel_linear", "srand",
           "fd_init", "fd_forward", "fd_back", "mod_extendvel", "mod_ricker_wavelet", "mod_taper_tables", "image_laplacian", "image_compare", "lib", "LIB_PATH", "MODE_FWD", "MODE_PLAIN", "MODE_RECV"]
