import os
import sys


def _host_threads():
    """Threads this process may really use (affinity mask, cgroup CPU quota), at most 16 -- the CPU share of a one-GPU box, whose kernel shows
    hundreds of hardware threads.  The OpenMP build of the oracle (and torch's own pool) would otherwise start one thread per hardware thread
    it sees, and the fine-grained parallel loops of a small grid then crawl (the six-shot new_mod test: 137 s instead of 3 s)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


os.environ.setdefault("OMP_NUM_THREADS", str(_host_threads()))      # before numpy / torch / liborc_omp.so are loaded

import numpy as np  # noqa: E402
import pytest  # noqa: E402

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def golden_field(name, shape):
    return np.fromfile(os.path.join(GOLDEN, name), np.float32).reshape(shape)


@pytest.fixture(scope="session")
def tables():
    return np.load(os.path.join(GOLDEN, "host_tables.npz"))


@pytest.fixture(scope="session")
def new_mod():
    """The shipped new_mod deck (cuda_reference_RTM/models/new_mod/input.dat) + shot-5 velocity."""
    vel = golden_field("new_mod_vel_ext_shot5.f32", (415, 295))
    return dict(order=8, nxe=415, nze=295, nxb=50, nzb=50, nt=1700, fac=0.75, dx=10.0, dz=10.0, dt=0.001,
                fpeak=20.0, sx=7 + 5 * 60 + 50, sz=50, gz=50, v2=(vel * vel).astype(np.float32),
                golden_P=golden_field("stencil_input_415x295.f32", (415, 295)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    bad = np.flatnonzero(bits(a).ravel() != bits(b).ravel())
    if bad.size:
        k = bad[0]
        raise AssertionError(f"{what}: {bad.size} of {a.size} values differ bitwise; first at flat index {k} "
                             f"(coords {np.unravel_index(k, a.shape)}): {a.ravel()[k]!r} vs {b.ravel()[k]!r}")


def rel_max(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def make_deck(nxe, nze, nxb, nzb, nt, seed=0, order=8, fac=0.75, compat=True, fpeak=20.0, dx=10.0, dz=10.0):
    """Small seeded synthetic deck: layered velocity with noise, stable dt.  dx != dz as in the reference's marmousi deck
    (cuda_reference_RTM/models/marmousi/input.dat:7-8: dz=8, dx=25) exercises the separate x / z weight pairs of the packed kernels."""
    rng = np.random.default_rng(seed)
    z = np.arange(nze, dtype=np.float32)[None, :]
    vel = (1500.0 + 2000.0 * z / max(nze - 1, 1) + 200.0 * rng.standard_normal((nxe, nze))).astype(np.float32)
    vel = np.clip(vel, 1200.0, 4200.0 * min(1.0, min(dx, dz) / 10.0)).astype(np.float32)     # keeps v dt / min(dx, dz) where it was
    return dict(order=order, nxe=nxe, nze=nze, nxb=nxb, nzb=nzb, nt=nt, fac=fac, dx=float(dx), dz=float(dz), dt=0.001,
                fpeak=fpeak, compat=compat, v2=(vel * vel).astype(np.float32),
                sx=nxb + (nxe - 2 * nxb) // 3, sz=nzb + 2, gz=nzb + 1)


def random_fields(deck, seed, amp=1.0):
    """Random initial p/pp honouring the compat precondition (rows >= xlim of the damped strip are zero)."""
    rng = np.random.default_rng(seed)
    shape = (deck["nxe"], deck["nze"])
    p = (amp * rng.standard_normal(shape)).astype(np.float32)
    pp = (amp * rng.standard_normal(shape)).astype(np.float32)
    if deck.get("compat", True):
        xlim, ztap = 8 * (deck["nxe"] // 8), 8 * (deck["nzb"] // 8)
        p[xlim:, :ztap] = 0
        pp[xlim:, :ztap] = 0
    return p, pp


@pytest.fixture(scope="session", autouse=True)
def _poison_free_device_memory():
    """FDW_TEST_POISON=1: before the first test, fill the free device memory with NaNs and release it, so that anything a kernel reads without
    it having been written shows up as a NaN mismatch instead of depending on what earlier processes left in HBM (development aid)."""
    if os.environ.get("FDW_TEST_POISON") == "1":
        import torch
        if torch.cuda.is_available():
            free, _ = torch.cuda.mem_get_info()
            left, chunks = int(free * 0.9), []
            while left > (1 << 28):
                nb = min(left, 8 << 30)
                chunks.append(torch.full((nb // 4,), float("nan"), device="cuda"))
                left -= nb
            torch.cuda.synchronize()
            del chunks
            torch.cuda.empty_cache()
    yield
