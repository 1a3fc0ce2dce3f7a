"""Geometry of the x-slab domain decomposition (which rows a rank owns, how many ghost rows it carries, which blocks travel).

The decomposition itself -- the forward and backward loops on a slab with deep-halo exchange over RCCL / xGMI -- lives in the C library
(csrc/fdw_slabs.cpp, csrc/fdw_comm.cpp; ctypes mirror: slabs.py).  This module only restates the row bookkeeping of fdw_slabs_create for
the hosts that need it next to the C driver (bench.py gathers the slabs for its bitwise check with it; the assertion there holds it to
fdw_slabs_geometry).  The Python drivers that used to live here (SlabForward / SlabBack over torch.distributed) are a TEST HARNESS now:
tests/decomp_harness.py (gloo on the CPU with the oracle as stepper).

The reference has no multi-GPU path.  Rows (x, the slow axis) are split into `world` contiguous slabs; a ghost
row is `pitch` contiguous floats, so a halo is one contiguous block.  Deep halos: with half order h and `ksteps` steps per exchange every
interior side carries G = h*ksteps ghost rows.  Right after an exchange all local rows are valid; step j of the cycle (j = 1..ksteps)
can only update rows [h*j, nxl - h*j) on the interior sides, so the valid region shrinks by h per step and is exactly the owned rows
after ksteps steps."""
from dataclasses import dataclass


def slab_bounds(nxe, world):
    """Owned global row ranges [o0, o1) per rank: as even as possible, multiples of 4 where possible."""
    base = [(nxe * r) // world for r in range(world + 1)]
    for r in range(1, world):
        base[r] = (base[r] // 4) * 4
    return [(base[r], base[r + 1]) for r in range(world)]


@dataclass
class SlabGeometry:
    rank: int
    world: int
    nxe: int
    h: int
    ksteps: int

    def __post_init__(self):
        self.G = self.h * self.ksteps
        self.o0, self.o1 = slab_bounds(self.nxe, self.world)[self.rank]
        self.has_lo = self.rank > 0
        self.has_hi = self.rank < self.world - 1
        self.g_lo = self.G if self.has_lo else 0
        self.g_hi = self.G if self.has_hi else 0
        self.x_off = self.o0 - self.g_lo            # global row of local row 0
        self.nxl = (self.o1 - self.o0) + self.g_lo + self.g_hi
        if self.o1 - self.o0 < self.G:
            raise ValueError(f"slab of {self.o1 - self.o0} rows is thinner than the ghost width {self.G}")

    def update_range(self, j):
        """Local rows [r0, r1) that step j (1-based) of a cycle may update."""
        r0 = self.h * j if self.has_lo else 0
        r1 = self.nxl - (self.h * j if self.has_hi else 0)
        return r0, r1

    # local row ranges of the halo blocks
    def send_lo(self):  # my first G owned rows -> left neighbour's right ghost
        return self.g_lo, self.g_lo + self.G

    def send_hi(self):  # my last G owned rows -> right neighbour's left ghost
        return self.nxl - self.g_hi - self.G, self.nxl - self.g_hi

    def recv_lo(self):
        return 0, self.g_lo

    def recv_hi(self):
        return self.nxl - self.g_hi, self.nxl
