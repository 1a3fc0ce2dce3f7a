"""The two drop-in executables (csrc/stencil_code.c, csrc/rtm_code.c) and their deck reader."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import parallel_finite_difference_computation_amd as F
from conftest import GOLDEN, ROOT, assert_bit_equal, golden_field
from oracle import oracle as O

BIN = os.path.join(ROOT, "parallel_finite_difference_computation_amd", "bin")
DECKS = os.path.join(GOLDEN, "decks")


def _deck(path):
    L = F.lib()
    L.fdw_deck_read.restype = C.c_void_p
    L.fdw_deck_read.argtypes = [C.c_char_p]
    L.fdw_deck_int.argtypes = [C.c_void_p, C.c_char_p]
    L.fdw_deck_float.argtypes = [C.c_void_p, C.c_char_p]
    L.fdw_deck_float.restype = C.c_float
    L.fdw_deck_str.argtypes = [C.c_void_p, C.c_char_p]
    L.fdw_deck_str.restype = C.c_char_p
    L.fdw_deck_free.argtypes = [C.c_void_p]
    h = L.fdw_deck_read(path.encode())
    assert h
    return L, h


def test_deck_reader_on_every_shipped_input_dat():
    """Parse table for the reference's own decks (values read off the files themselves)."""
    expect = {
        "stencil.dat": dict(tmpdir="./input.bin", nz=195, nx=315, dz=10.0, dx=10.0, nxb=50, nzb=50, order=8),
        "new_mod.dat": dict(tmpdir="./output", vpfile="./models/new_mod/vel-koslov.1", datfile="./models/new_mod/dobs.6",
                            vel_ext_file="./models/new_mod/vel_ext_rnd.6", nz=195, nx=315, nt=1700, dz=10.0, dx=10.0, dt=0.001,
                            fpeak=20.0, ns=6, iss=0, sz=0, fsx=7, ds=60, gz=0, nxb=50, nzb=50, rnd=1, fac=0.75, order=8),
        "marmousi.dat": dict(tmpdir="./", vpfile="model-375.cwp", datfile="dado_teste.bin", nz=375, nx=369, nt=3004, dz=8.0, dx=25.0,
                             dt=0.001, fpeak=6.5, ns=1, fsx=179, ds=6, nxb=40, nzb=40, fac=0.75, order=8),
        "1lay_mod.dat": dict(vpfile="vp_101x201.bin", nz=101, nx=201, nt=401, fpeak=40.0, ns=4, sz=50, fsx=100, ds=40, fac=0.010),
        "3lay_mod.dat": dict(vpfile="3layer_151x151.bin", nz=151, nx=151, nt=1001, fpeak=30.0, ns=4, ds=50, fac=0.010),
    }
    for name, kv in expect.items():
        L, h = _deck(os.path.join(DECKS, name))
        for k, v in kv.items():
            if isinstance(v, str):
                assert L.fdw_deck_str(h, k.encode()).decode() == v, (name, k)
            elif isinstance(v, int):
                assert L.fdw_deck_int(h, k.encode()) == v, (name, k)
            else:
                assert abs(L.fdw_deck_float(h, k.encode()) - v) < 1e-6 * max(1.0, abs(v)), (name, k)
        # the reference's sentinels for absent keys (functions.c:64,86,109)
        assert L.fdw_deck_int(h, b"no_such_key") == -1
        assert L.fdw_deck_float(h, b"no_such_key") == -1.0
        assert L.fdw_deck_str(h, b"no_such_key") is None
        L.fdw_deck_free(h)
    L, h = _deck(os.path.join(DECKS, "1lay_mod.dat"))
    assert L.fdw_deck_str(h, b"datfile") is None and L.fdw_deck_str(h, b"vel_ext_file") is None   # decks without data
    L.fdw_deck_free(h)


def test_deck_reader_details(tmp_path):
    p = tmp_path / "d.dat"
    p.write_text("# comment\n  nz = 12 \r\nnzb=3\nnx=7 # trailing\nname=./a b/c.bin\nnz=99\n\nbad line\n")
    L, h = _deck(str(p))
    assert L.fdw_deck_int(h, b"nz") == 12          # first occurrence wins; nzb does not shadow nz
    assert L.fdw_deck_int(h, b"nzb") == 3 and L.fdw_deck_int(h, b"nx") == 7
    assert L.fdw_deck_str(h, b"name") == b"./a b/c.bin"
    L.fdw_deck_free(h)
    L.fdw_deck_read.restype = C.c_void_p
    assert L.fdw_deck_read(str(tmp_path / "missing.dat").encode()) is None


def test_programs_are_built_and_refuse_bad_usage():
    for exe in ("stencil_code", "rtm_code", "mod_main", "rtm_main"):   # lapfilt has defaults for everything, psnr exits 0 like the reference tool
        path = os.path.join(BIN, exe)
        assert os.access(path, os.X_OK), f"{path} missing: run `make -C parallel_finite_difference_computation_amd/csrc`"
        assert subprocess.run([path], capture_output=True).returncode != 0
        assert subprocess.run([path, "/nonexistent/input.dat"], capture_output=True).returncode != 0


@pytest.mark.gpu
def test_stencil_code_program_reproduces_the_committed_output(tmp_path):
    """./stencil_code ./input.dat with the reference's own deck and input field -> output_teste.bin, bit for bit."""
    shutil.copy(os.path.join(DECKS, "stencil.dat"), tmp_path / "input.dat")
    shutil.copy(os.path.join(GOLDEN, "stencil_input_415x295.f32"), tmp_path / "input.bin")
    r = subprocess.run([os.path.join(BIN, "stencil_code"), "./input.dat"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Input reading was successful." in r.stdout and "Output writing was successful." in r.stdout and "order = 8" in r.stdout
    out = np.fromfile(tmp_path / "output_teste.bin", np.float32).reshape(415, 295)
    assert_bit_equal(out, golden_field("stencil_lap_415x295.f32", (415, 295)), "stencil_code output")


def _small_rtm_case(tmp_path, with_vel_ext, ns=3, ds=20):
    """A small synthetic rtm_code job on disk; returns its arrays."""
    nx, nz, nxb, nzb, nt = 61, 47, 17, 13, 90
    nxe, nze = nx + 2 * nxb, nz + 2 * nzb
    rng = np.random.default_rng(11)
    vp = (1500 + 2500 * np.linspace(0, 1, nz, dtype=np.float32)[None, :] + 100 * rng.standard_normal((nx, nz))).astype(np.float32)
    d_obs = rng.standard_normal((ns, nx, nt)).astype(np.float32)
    (tmp_path / "models").mkdir()
    (tmp_path / "output").mkdir()
    vp.tofile(tmp_path / "models" / "vp.bin")
    d_obs.tofile(tmp_path / "models" / "dobs.bin")
    deck = ("tmpdir=./output\nvpfile=./models/vp.bin\ndatfile=./models/dobs.bin\n"
            f"nz={nz}\nnx={nx}\nnt={nt}\ndz=10\ndx=10\ndt=0.001\nfpeak=25.\nns={ns}\nsz=1\nfsx=5\nds={ds}\ngz=2\n"
            f"nxb={nxb}\nnzb={nzb}\nrnd=1\nfac=0.75\norder=8\n")
    vel_ext = None
    if with_vel_ext:
        vel_ext = (1500 + 2000 * rng.random((ns, nxe, nze))).astype(np.float32)
        vel_ext.tofile(tmp_path / "models" / "velext.bin")
        deck = deck.replace("vpfile=", "vel_ext_file=./models/velext.bin\nvpfile=")
    (tmp_path / "input.dat").write_text(deck)
    return nx, nz, nxb, nzb, nt, ns, vp, d_obs, vel_ext


@pytest.mark.gpu
@pytest.mark.parametrize("with_vel_ext", [False, True])
def test_rtm_code_program_vs_oracle_pipeline(tmp_path, with_vel_ext):
    """./rtm_code on a small synthetic deck: dir.image / image.num / empty side files against the oracle's
    restatement of main()'s shot loop (fd-code.cu:480-542), including the unseeded rand() border model."""
    nx, nz, nxb, nzb, nt, ns, vp, d_obs, vel_ext = _small_rtm_case(tmp_path, with_vel_ext)
    nxe, nze = nx + 2 * nxb, nz + 2 * nzb
    r = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert f"## nz = {nz}, nx = {nx}, nt = {nt} " in r.stdout and "** source 3, at (45,1) " in r.stdout and "> Exec time" in r.stdout

    # oracle pipeline
    orc = O.Oracle(8, nxe, nze, nxb, nzb, nt, 0.75, 10.0, 10.0, 0.001, compat=True)
    srce = O.ricker_wavelet(nt, 0.001, 25.0)
    vpe = np.zeros((nxe, nze), np.float32)
    vpe[nxb:nxb + nx, nzb:nzb + nz] = vp
    img = np.zeros((nx, nz), np.float32)
    for s in range(ns):
        if with_vel_ext:
            v = vel_ext[s]
        else:
            O.extendvel_linear(vpe, nx, nz, nxb, nzb, seed=1 if s == 0 else None)   # rand() is never seeded by the reference
            v = vpe
        v2 = (v * v).astype(np.float32)
        P, PP = orc.forward(v2, 5 + s * 20 + nxb, 1 + nzb, srce)
        img = img + orc.back(v2, P, PP, d_obs[s], 2 + nzb)
    got = np.fromfile(tmp_path / "output" / "dir.image", np.float32).reshape(nx, nz)
    assert_bit_equal(got, img, "dir.image")
    assert np.abs(got).max() > 0
    lap = np.fromfile(tmp_path / "output" / "dir.image_lap", np.float32)
    assert lap.size == nx * nz and not lap.any()
    if not with_vel_ext:  # that run drew the border models on the device and took the three shots as ONE batch (a launch per time step for all);
        #                   one shot at a time (FDW_NO_SHOT_BATCH=1) and the host border loop (FDW_HOST_BORDER=1) give the same file
        for env in ({"FDW_NO_SHOT_BATCH": "1"}, {"FDW_HOST_BORDER": "1"}, {"FDW_NO_SHOT_BATCH": "1", "FDW_SHOT_WORKERS": "1"}):
            r2 = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=tmp_path, capture_output=True, text=True, env=dict(os.environ, **env))
            assert r2.returncode == 0, r2.stderr
            assert_bit_equal(np.fromfile(tmp_path / "output" / "dir.image", np.float32).reshape(nx, nz), img, f"dir.image with {env}")
    if with_vel_ext:      # that run took the three shots as one batch on the host-given models; one shot at a time gives the same file
        r2 = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=tmp_path, capture_output=True, text=True,
                            env=dict(os.environ, FDW_NO_SHOT_BATCH="1"))
        assert r2.returncode == 0, r2.stderr
        assert_bit_equal(np.fromfile(tmp_path / "output" / "dir.image", np.float32).reshape(nx, nz), img, "dir.image, one shot at a time")
    if with_vel_ext:      # opt-in deck key: dir.image_lap = the reference's offline Laplacian filter of the stacked image
        (tmp_path / "input.dat").write_text((tmp_path / "input.dat").read_text() + "image_lap=1\n")
        r2 = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=tmp_path, capture_output=True, text=True)
        assert r2.returncode == 0, r2.stderr
        lap = np.fromfile(tmp_path / "output" / "dir.image_lap", np.float32).reshape(nx, nz)
        assert_bit_equal(lap, O.image_laplacian(img, 10.0, 10.0), "dir.image_lap with image_lap=1")
        assert_bit_equal(np.fromfile(tmp_path / "output" / "dir.image", np.float32).reshape(nx, nz), img, "dir.image unchanged")
    for name in ("dir.snaps", "dir.snaps_rec", "dir.snapr"):
        assert os.path.getsize(tmp_path / "output" / name) == 0
    lines = (tmp_path / "image.num").read_text().splitlines()
    assert len(lines) == ns * (1 + nx * nz) and lines[0] == "======== 0 ========" and lines[1 + nx * nz] == "======== 1 ========"
    last = np.array([float(x) for x in lines[-nx * nz:]], np.float32).reshape(nz, nx).T   # iz outer, ix inner
    assert np.allclose(last, img, rtol=1e-5, atol=1e-5 * np.abs(img).max())


@pytest.mark.gpu
def test_rtm_code_program_on_a_large_deck(tmp_path):
    """./rtm_code where one shot fills the chip (2 980 x 2 484 extended grid, extents that are no multiples of 8: the reference's truncated
    kernel ranges at scale): the border model drawn on the device from the rand() stream (two shots, 0.85 M draws each), the resident-model
    shot through the four-steps-per-pass kernels the library picks from this size on (rtm_code prints nothing about it: the size is chosen
    past fdw_api.cpp's kPipeAutoStripRows), the stack -- dir.image against the oracle's restatement of main()'s loop bit for bit, and the
    last block of image.num (7 M lines per shot, as the reference writes it) against it as text."""
    nx, nz, nxb, nzb, nt, ns, fsx, ds = 2900, 2404, 40, 40, 14, 2, 1400, 60
    nxe, nze = nx + 2 * nxb, nz + 2 * nzb
    rng = np.random.default_rng(23)
    vp = (1500 + 2500 * np.linspace(0, 1, nz, dtype=np.float32)[None, :] + 100 * rng.standard_normal((nx, nz), dtype=np.float32)).astype(np.float32)
    d_obs = rng.standard_normal((ns, nx, nt), dtype=np.float32)
    (tmp_path / "models").mkdir()
    (tmp_path / "output").mkdir()
    vp.tofile(tmp_path / "models" / "vp.bin")
    d_obs.tofile(tmp_path / "models" / "dobs.bin")
    (tmp_path / "input.dat").write_text("tmpdir=./output\nvpfile=./models/vp.bin\ndatfile=./models/dobs.bin\n"
                                        f"nz={nz}\nnx={nx}\nnt={nt}\ndz=10\ndx=10\ndt=0.001\nfpeak=25.\nns={ns}\nsz=1\nfsx={fsx}\nds={ds}\ngz=2\n"
                                        f"nxb={nxb}\nnzb={nzb}\nrnd=1\nfac=0.75\norder=8\n")
    r = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    orc = O.Oracle(8, nxe, nze, nxb, nzb, nt, 0.75, 10.0, 10.0, 0.001, compat=True, omp=True)
    srce = O.ricker_wavelet(nt, 0.001, 25.0)
    vpe = np.zeros((nxe, nze), np.float32)
    vpe[nxb:nxb + nx, nzb:nzb + nz] = vp
    img = np.zeros((nx, nz), np.float32)
    for s in range(ns):
        O.extendvel_linear(vpe, nx, nz, nxb, nzb, seed=1 if s == 0 else None)
        v2 = (vpe * vpe).astype(np.float32)
        P, PP = orc.forward(v2, fsx + s * ds + nxb, 1 + nzb, srce)
        img = img + orc.back(v2, P, PP, d_obs[s], 2 + nzb)
    got = np.fromfile(tmp_path / "output" / "dir.image", np.float32).reshape(nx, nz)
    assert_bit_equal(got, img, "dir.image")
    assert np.abs(got).max() > 0
    with open(tmp_path / "image.num") as f:
        lines = f.read().splitlines()
    assert len(lines) == ns * (1 + nx * nz) and lines[1 + nx * nz] == "======== 1 ========"
    last = np.array(lines[-nx * nz:], dtype=np.float64).astype(np.float32).reshape(nz, nx).T
    assert np.allclose(last, img, rtol=1e-5, atol=1e-5 * np.abs(img).max())
    del lines, last
    # the same job with every shot cut into three bands of rows (ranks as host threads on this GPU, device copies for the halos)
    r = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=tmp_path, capture_output=True, text=True,
                       env=dict(os.environ, FDW_SLABS="3", FDW_SLABS_LOCAL="1"))
    assert r.returncode == 0, r.stderr + r.stdout
    assert_bit_equal(np.fromfile(tmp_path / "output" / "dir.image", np.float32).reshape(nx, nz), img, "dir.image, slabs=3")


def test_bench_self_launch_without_a_gpu_fails_fast_and_clean(tmp_path):
    """The launcher of `python bench.py --gpus N` on a box WITHOUT a GPU (this container): it starts its N ranks before touching HIP itself, every
    rank finds no device and exits non-zero, the launcher reports that status without a JSON line and without waiting for a timeout.  (On a GPU
    box the -m gpu tests run the same launcher to a result.)"""
    import sys
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by test_bench_starts_its_own_ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "shm", "--size", "1024", "--steps", "8", "--warmup", "4",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=tmp_path, env=env)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "no GPU visible" in r.stderr and time.time() - t0 < 120


def test_rtm_code_refuses_more_slabs_than_gpus(tmp_path):
    """slabs=N over RCCL needs N GPUs, one rank each: with fewer visible (none in the authoring container, one on a GPU box) the program must
    say so and exit at once -- before it creates a communicator or a thread, because a rank that never arrives would leave the others
    waiting inside ncclCommInitRank for ever."""
    _small_rtm_case(tmp_path, False, ns=1)
    (tmp_path / "input.dat").write_text((tmp_path / "input.dat").read_text() + "slabs=64\n")
    env = {k: v for k, v in os.environ.items() if k != "FDW_SLABS_LOCAL"}
    r = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=tmp_path, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "needs 64 GPUs" in r.stderr, r.stderr + r.stdout


@pytest.mark.gpu
def test_rtm_code_fast_numerics_deck_key(tmp_path):
    """Deck key numerics=1 (our extension; absent = the reference's arithmetic): the image stays within 1e-5 of the exact program's, and is
    not the same bits."""
    a, b = tmp_path / "exact", tmp_path / "fast"
    a.mkdir()
    nx, nz, *_ = _small_rtm_case(a, True, ns=2, ds=14)
    shutil.copytree(a, b)
    (b / "input.dat").write_text((b / "input.dat").read_text() + "numerics=1\n")
    for d in (a, b):
        r = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout
    assert "numerics = FAST" in r.stdout
    ia, ib = (np.fromfile(d / "output" / "dir.image", np.float32) for d in (a, b))
    assert np.abs(ia - ib).max() / np.abs(ia).max() < 1e-5 and (ia != ib).any()


@pytest.mark.gpu
@pytest.mark.parametrize("with_vel_ext", [False, True])
def test_rtm_code_slab_mode_matches_the_serial_program(tmp_path, with_vel_ext):
    """`rtm_code` with every shot decomposed over 3 ranks (deck key slabs=3 / FDW_SLABS; host threads driving fdw_slabs_shot, here all on
    this box's one GPU through the in-process communicator, FDW_SLABS_LOCAL=1): dir.image, image.num and dir.image_lap are the serial
    program's byte for byte, for the random-border and the vel_ext_file deck."""
    a, b = tmp_path / "serial", tmp_path / "slabs"
    a.mkdir()
    nx, nz, *_ = _small_rtm_case(a, with_vel_ext, ns=3, ds=14)
    shutil.copytree(a, b)
    r = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=a, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    (b / "input.dat").write_text((b / "input.dat").read_text() + "slabs=3\n")
    r2 = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=b, capture_output=True, text=True, env=dict(os.environ, FDW_SLABS_LOCAL="1"))
    assert r2.returncode == 0, r2.stderr + r2.stdout
    img = (a / "output" / "dir.image").read_bytes()
    assert len(img) == 4 * nx * nz and np.frombuffer(img, np.float32).any()
    assert (b / "output" / "dir.image").read_bytes() == img
    assert (b / "image.num").read_bytes() == (a / "image.num").read_bytes()
    assert r2.stdout.replace("\r", "").splitlines()[:-1] == r.stdout.replace("\r", "").splitlines()[:-1]          # all but the "> Exec time" line
    # FDW_GPUS=1 (shots dealt to GPUs) is the plain program
    r3 = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=a, capture_output=True, text=True, env=dict(os.environ, FDW_GPUS="1"))
    assert r3.returncode == 0 and (a / "output" / "dir.image").read_bytes() == img


@pytest.mark.gpu
def test_rtm_new_mod_shot5_full_length(new_mod):
    """BASELINE config 3 at full length: forward + backward + imaging of shot 5 of models/new_mod (nt = 1700, the
    reference's velocity with random border) on a seeded synthetic gather; image bit-identical to the oracle."""
    d = new_mod
    nx, nz = 315, 195
    rng = np.random.default_rng(2024)
    t = np.arange(d["nt"], dtype=np.float32) * d["dt"]
    d_obs = np.zeros((nx, d["nt"]), np.float32)
    for k, (t0, amp) in enumerate(((0.35, 1.0), (0.7, -0.6), (1.1, 0.4))):      # three hyperbolic Ricker events + noise
        off = (np.arange(nx, dtype=np.float32) - (d["sx"] - 50)) * 10.0
        tt = np.sqrt(t0 * t0 + (off / (2500.0 + 400.0 * k)) ** 2)
        x = np.pi * 20.0 * (t[None, :] - tt[:, None])
        d_obs += (amp * (1 - 2 * x * x) * np.exp(-x * x)).astype(np.float32)
    d_obs += (0.01 * rng.standard_normal(d_obs.shape)).astype(np.float32)
    ctx = F.FDWave(8, 415, 295, 50, 50, d["nt"], 0.75, 10.0, 10.0, 0.001, compat=True)
    img, P, PP = ctx.shot(d["v2"], d["sx"], d["sz"], d["gz"], F.ricker_wavelet(d["nt"], d["dt"], d["fpeak"]), d_obs, want_fields=True)
    orc = O.Oracle(8, 415, 295, 50, 50, d["nt"], 0.75, 10.0, 10.0, 0.001, compat=True)
    oP, oPP = orc.forward(d["v2"], d["sx"], d["sz"], O.ricker_wavelet(d["nt"], d["dt"], d["fpeak"]))
    oimg = orc.back(d["v2"], oP, oPP, d_obs, d["gz"])
    assert_bit_equal(P, oP, "new_mod shot 5 P")
    assert_bit_equal(img, oimg, "new_mod shot 5 image")
    assert np.abs(img).max() > 0


def _hyperbolic_gather(nx, nt, dt, sx_interior, dx, fpeak, seed):
    """A seeded synthetic shot gather [nx][nt]: three hyperbolic Ricker events around the source column + a little noise (the
    reference's own dobs.6 / dado_teste.bin are listed in its .MISSING_LARGE_BLOBS)."""
    rng = np.random.default_rng(seed)
    t = np.arange(nt, dtype=np.float32) * np.float32(dt)
    g = np.zeros((nx, nt), np.float32)
    off = (np.arange(nx, dtype=np.float32) - np.float32(sx_interior)) * np.float32(dx)
    for k, (t0, amp) in enumerate(((0.35, 1.0), (0.7, -0.6), (1.1, 0.4))):
        tt = np.sqrt(t0 * t0 + (off / (2500.0 + 400.0 * k)) ** 2)
        x = np.pi * fpeak * (t[None, :] - tt[:, None])
        g += (amp * (1 - 2 * x * x) * np.exp(-x * x)).astype(np.float32)
    return g + (0.01 * rng.standard_normal(g.shape)).astype(np.float32)


@pytest.mark.gpu
def test_rtm_code_on_the_reference_new_mod_deck_all_six_shots(tmp_path):
    """BASELINE config 3 as the reference runs it: `./rtm_code ./models/new_mod/input.dat` with the reference's OWN deck, velocity model
    and six per-shot extended models (vel_ext_rnd.6, R:412-418, R:483-484), nt = 1700, on a seeded synthetic dobs.6 -- the stacked
    dir.image against the oracle's restatement of main()'s shot loop (R:480-542), bit for bit; every shot's image is non-trivial."""
    d = tmp_path / "models" / "new_mod"
    d.mkdir(parents=True)
    (tmp_path / "output").mkdir()
    shutil.copy(os.path.join(DECKS, "new_mod.dat"), d / "input.dat")
    shutil.copy(os.path.join(GOLDEN, "new_mod_vel_koslov.f32"), d / "vel-koslov.1")
    vel = np.load(os.path.join(GOLDEN, "new_mod_vel_ext_rnd6.npz"))["vel"]
    assert vel.shape == (6, 415, 295)
    vel.tofile(d / "vel_ext_rnd.6")
    nx, nz, nxb, nzb, nt, ns, fsx, ds = 315, 195, 50, 50, 1700, 6, 7, 60
    dobs = np.stack([_hyperbolic_gather(nx, nt, 0.001, fsx + s * ds, 10.0, 20.0, 100 + s) for s in range(ns)])
    dobs.tofile(d / "dobs.6")
    r = subprocess.run([os.path.join(BIN, "rtm_code"), "./models/new_mod/input.dat"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "## nz = 195, nx = 315, nt = 1700 " in r.stdout and "** source 6, at (307,0) " in r.stdout, r.stdout
    got = np.fromfile(tmp_path / "output" / "dir.image", np.float32).reshape(nx, nz)
    srce = O.ricker_wavelet(nt, 0.001, 20.0)

    def oracle_shot(s):      # one oracle state per shot (no shared state in fdw_oracle.c's RTM passes; ctypes releases the GIL): the six shots side by side
        orc = O.Oracle(8, 415, 295, nxb, nzb, nt, 0.75, 10.0, 10.0, 0.001, compat=True)
        v2 = (vel[s] * vel[s]).astype(np.float32)
        P, PP = orc.forward(v2, fsx + s * ds + nxb, nzb, srce)
        return orc.back(v2, P, PP, dobs[s], nzb)

    import concurrent.futures
    with concurrent.futures.ThreadPoolExecutor(max_workers=ns) as pool:
        imlocs = list(pool.map(oracle_shot, range(ns)))
    img = np.zeros((nx, nz), np.float32)
    for imloc in imlocs:                                   # stacked in shot order, as main() does (R:522-528)
        assert np.abs(imloc).max() > 0
        img = img + imloc
    assert_bit_equal(got, img, "dir.image of the six-shot new_mod deck")
    lines = (tmp_path / "image.num").read_text().splitlines()
    assert len(lines) == ns * (1 + nx * nz) and lines[5 * (1 + nx * nz)] == "======== 5 ========"


@pytest.mark.gpu
def test_rtm_code_on_the_reference_marmousi_deck(tmp_path):
    """The reference's other runnable deck (models/marmousi/input.dat: 369 x 375 model, dx = 25, dz = 8, nt = 3004, one shot, random
    border drawn from the unseeded rand() stream, outputs in the working directory) with its own velocity model and a seeded
    synthetic dado_teste.bin: dir.image against the oracle pipeline, bit for bit."""
    shutil.copy(os.path.join(DECKS, "marmousi.dat"), tmp_path / "input.dat")
    vp = np.load(os.path.join(GOLDEN, "marmousi_model_375.npz"))["vp"]
    assert vp.shape == (369, 375)
    vp.tofile(tmp_path / "model-375.cwp")
    nx, nz, nb, nt = 369, 375, 40, 3004
    dobs = _hyperbolic_gather(nx, nt, 0.001, 179, 25.0, 6.5, 7)
    dobs.tofile(tmp_path / "dado_teste.bin")
    r = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    got = np.fromfile(tmp_path / "dir.image", np.float32).reshape(nx, nz)
    nxe, nze = nx + 2 * nb, nz + 2 * nb
    orc = O.Oracle(8, nxe, nze, nb, nb, nt, 0.75, 25.0, 8.0, 0.001, compat=True, omp=True)
    vpe = np.zeros((nxe, nze), np.float32)
    vpe[nb:nb + nx, nb:nb + nz] = vp
    O.extendvel_linear(vpe, nx, nz, nb, nb, seed=1)
    v2 = (vpe * vpe).astype(np.float32)
    srce = O.ricker_wavelet(nt, 0.001, 6.5)
    P, PP = orc.forward(v2, 179 + nb, nb, srce)
    img = orc.back(v2, P, PP, dobs, nb)
    assert np.isfinite(img).all() and np.abs(img).max() > 0
    assert_bit_equal(got, img, "dir.image of the marmousi deck")


@pytest.mark.gpu
def test_reference_signature_compat_library():
    """libfdwave_rtm_compat.so: fd_init / fd_forward / fd_back with the reference's own argument lists
    (float** SU-style arrays, nz-before-nx order), driven the way main() drives them (fd-code.cu:452-518)."""
    so = os.path.join(ROOT, "parallel_finite_difference_computation_amd", "libfdwave_rtm_compat.so")
    L = C.CDLL(so)
    fpp = C.POINTER(C.POINTER(C.c_float))
    nx, nz, nxb, nzb, nt = 48, 40, 12, 11, 30
    nxe, nze = nx + 2 * nxb, nz + 2 * nzb
    rng = np.random.default_rng(3)
    vel = (1500 + 2000 * rng.random((nxe, nze))).astype(np.float32)
    v2 = vel * vel
    srce = O.ricker_wavelet(nt, 0.001, 30.0)
    d_obs = rng.standard_normal((1, nx, nt)).astype(np.float32)

    def rows(a):   # alloc2float layout: row pointers into one contiguous block
        return (C.POINTER(C.c_float) * a.shape[0])(*[C.cast(a[i].ctypes.data, C.POINTER(C.c_float)) for i in range(a.shape[0])])

    P, PP = np.zeros((nxe, nze), np.float32), np.zeros((nxe, nze), np.float32)
    L.fd_init.argtypes = [C.c_int] * 7 + [C.c_float] * 4
    L.fd_init(8, nxe, nze, nxb, nzb, nt, 1, 0.75, 10.0, 10.0, 0.001)
    sx = (C.c_int * 1)(nxb + 20)
    L.fd_forward.argtypes = [C.c_int, fpp, fpp, fpp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int]
    L.fd_forward(8, rows(P), rows(PP), rows(v2), nze, nxe, nt, 0, nzb + 1, sx, srce.ctypes.data_as(C.POINTER(C.c_float)), 0)
    orc = O.Oracle(8, nxe, nze, nxb, nzb, nt, 0.75, 10.0, 10.0, 0.001, compat=True)
    oP, oPP = orc.forward(v2, nxb + 20, nzb + 1, srce)
    assert_bit_equal(P, oP, "compat fd_forward P")
    assert_bit_equal(PP, oPP, "compat fd_forward PP")
    snaps = np.stack([P, PP])
    snap_rows = [rows(snaps[0]), rows(snaps[1])]
    snaps_pp = (fpp * 2)(C.cast(snap_rows[0], fpp), C.cast(snap_rows[1], fpp))
    imloc = np.zeros((nx, nz), np.float32)
    dobs_rows = (C.POINTER(C.c_float) * 1)(C.cast(d_obs[0].ctypes.data, C.POINTER(C.c_float)))
    z = np.zeros((nxe, nze), np.float32)
    L.fd_back.argtypes = [C.c_int, fpp, fpp, fpp, fpp, fpp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(fpp), fpp, fpp]
    L.fd_back(8, rows(z), rows(z), rows(z), rows(z), rows(v2), nze, nxe, nt, 0, nzb + 1, nzb + 2, snaps_pp, rows(imloc), dobs_rows)
    assert_bit_equal(imloc, orc.back(v2, oP, oPP, d_obs[0], nzb + 2), "compat fd_back imloc")
    L.fd_free()


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,opts", [(2, ["--ksteps", "4"]), (3, ["--ksteps", "0", "--steps", "25"]),
                                        (2, ["--ksteps", "8", "--pipe", "on"]), (3, ["--ksteps", "4", "--pipe", "on", "--steps", "26"])],
                         ids=["2ranks-k4", "3ranks-auto-odd", "2ranks-pipeline-k8", "3ranks-pipeline-k4-leftover"])
def test_multi_process_bench_rehearsal_on_one_gpu(ranks, opts, tmp_path):
    """bench.py's N > 1 path end to end under an external launcher (torch.distributed.run): one process per slab, the C slab driver
    (fdw_slabs_dev_forward) with its overlapped deep-halo exchange (one-step kernels, and four steps per pass through the wave-pipeline
    kernel on four rotating buffers) -- over the library's process transport (--backend shm) so that all ranks can share this box's single
    GPU (RCCL refuses duplicate devices).  The slabs are gathered and compared bitwise with a single-domain run."""
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--backend", "shm", "--size", "1024",
           "--steps", "24", "--warmup", "6", "--check", "--no-cpu-baseline"] + opts
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    assert f"decomposed ({ranks} slabs) == single domain, bitwise: True" in r.stderr
    import json
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == ranks and out["scaling"] == "strong" and out["result_finite_nonzero"] and out["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("workload,backend", [("forward", "shm"), ("rtm-slab", "shm")])
def test_bench_starts_its_own_ranks(workload, backend, tmp_path):
    """`python bench.py --gpus 2 ...` exactly as the driver calls it -- no launcher, no RANK / WORLD_SIZE in the environment: the parent starts
    the two ranks itself (before anything touches HIP), rank 0's JSON line is the LAST line on stdout, n_gpus = 2, exit status 0, and the
    decomposed result equals the single-domain one bitwise.  --backend shm: the C slab driver (fdw_slabs_*) over the process transport of
    fdw_comm.cpp -- the code RCCL drives on a multi-GPU node, here with the two ranks sharing this box's GPU."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", backend, "--size", "1024", "--steps", "24", "--warmup", "6",
           "--check", "--no-cpu-baseline", "--workload", workload]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp_path, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    last = [ln for ln in r.stdout.splitlines() if ln.strip()][-1]
    out = json.loads(last)
    assert out["n_gpus"] == 2 and out["result_finite_nonzero"] and out["value"] > 0
    assert out["launched_by"].startswith("bench.py itself")
    assert "bitwise equal" in out["decomposition_check"]
    if backend == "shm":
        assert out["rccl_ranks"] is None and out["comm_ranks"] == 2 and "process transport" in out["halo_exchange"]
        assert 0.0 <= out["exposed_comm_fraction"] <= 1.0


@pytest.mark.gpu
def test_bench_rccl_rehearsal_on_one_rank(tmp_path):
    """What every rank of an N > 1 run starts before it opens RCCL itself (bench.py --rccl-rehearsal): communicator, self-addressed message,
    32 steps through the C slab driver, all-reduce -- here as a world of one rank, the most this box's single GPU lets RCCL do."""
    import sys
    import parallel_finite_difference_computation_amd as F
    uid = F.Comm.unique_id().hex()
    env = dict({k: v for k, v in os.environ.items() if k not in ("MASTER_ADDR", "MASTER_PORT")}, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-rehearsal", uid], capture_output=True, text=True, timeout=600, cwd=tmp_path, env=env)
    assert r.returncode == 0 and "RCCL-REHEARSAL-OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["refused", "hangs"])
def test_bench_falls_back_when_the_rccl_rehearsal_fails(how, tmp_path):
    """`bench.py --gpus 2` with the default back end where RCCL cannot work: both ranks on this box's one GPU (RCCL refuses duplicate devices:
    the rehearsal children fail), and a rehearsal that never returns (killed at its time limit).  Either way the ranks fall back together to
    the process transport, the line says so, and the decomposed result still equals the single-domain one."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["FDW_BENCH_SHARE_GPU"] = "1"
    if how == "hangs":
        env.update(FDW_BENCH_REHEARSAL_HANG="1", FDW_BENCH_REHEARSAL_TIMEOUT="8")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "1024", "--steps", "24", "--warmup", "6", "--check",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=tmp_path, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] is None and "FALLBACK" in out["halo_exchange"] and "bitwise equal" in out["decomposition_check"]
    if how == "hangs":
        assert "did not finish within 8 s" in out["halo_exchange"]


@pytest.mark.gpu
def test_bench_self_launch_reports_a_failing_rank(tmp_path):
    """A rank that dies must not leave the launcher waiting: non-zero exit, no JSON line."""
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "shm", "--size", "1024", "--steps", "8", "--warmup", "4",
                        "--no-cpu-baseline", "--ksteps", "200"], capture_output=True, text=True, timeout=600, cwd=tmp_path, env=env)      # ghost rows wider than a band
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_bench_multi_gpu_code_path_on_one_rank(tmp_path):
    """bench.py's --backend nccl path (communicator over librccl, fdw_slabs_dev_forward: the code the 8-GPU run executes) forced onto one
    rank: the JSON line must come out and agree with the plain single-GPU path on the field it ends with (finite, non-zero)."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "1024", "--steps", "24", "--warmup", "6", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=tmp_path, env=dict(os.environ, FDW_FORCE_SLAB_DRIVER="c"))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["result_finite_nonzero"] and out["value"] > 0 and out["roofline"]["frac"] <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,opts", [(1, []), (2, ["--backend", "shm", "--ksteps", "4"]), (3, ["--backend", "shm", "--ksteps", "3", "--steps", "11"])],
                         ids=["1rank-c-driver", "2ranks-shm-k4", "3ranks-shm-k3-leftover"])
def test_bench_rtm_slab_workload(ranks, opts, tmp_path):
    """bench.py --workload rtm-slab: forward + backward + imaging of one shot under the slab decomposition through the C driver (fdw_slabs_*).
    One rank: the whole grid; 2 and 3 ranks: real processes sharing this GPU over the library's process transport, with the gathered image
    compared bitwise with a single-domain run of the same shots."""
    import json
    import sys
    base = [os.path.join(ROOT, "bench.py"), "--workload", "rtm-slab", "--gpus", str(ranks), "--size", "1024", "--steps", "12", "--warmup", "4", "--no-cpu-baseline"] + opts
    if ranks == 1:
        cmd = [sys.executable] + base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + base
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    if ranks > 1:
        assert f"decomposed image ({ranks} slabs) == single domain, bitwise: True" in r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == ranks and out["result_finite_nonzero"] and out["value"] > 0 and out["roofline"]["frac"] <= 1.0


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,with_vel_ext", [(1, False), (2, False), (3, True)])
def test_shot_parallel_driver_matches_the_serial_program(ranks, with_vel_ext, tmp_path):
    """python -m <package>.rtm deals the shots of rtm_code's loop to `ranks` processes (gloo here: the ranks share this box's one
    GPU) and stacks the images in shot order: dir.image and image.num must be byte-identical to the serial C program's,
    including the replayed unseeded-rand() border stream."""
    import sys
    a, b = tmp_path / "serial", tmp_path / "parallel"
    a.mkdir()
    nx, nz, *_ = _small_rtm_case(a, with_vel_ext, ns=5, ds=12)
    shutil.copytree(a, b)
    r = subprocess.run([os.path.join(BIN, "rtm_code"), "./input.dat"], cwd=a, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), FDW_DIST_BACKEND="gloo")
    mod = "parallel_finite_difference_computation_amd.rtm"
    if ranks == 1:
        cmd = [sys.executable, "-m", mod, "./input.dat"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "-m", mod, "./input.dat"]
    r = subprocess.run(cmd, cwd=b, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    assert "** source 5, at (53,1) " in r.stdout and "> Exec time" in r.stdout, r.stdout
    img = (a / "output" / "dir.image").read_bytes()
    assert len(img) == 4 * nx * nz and np.frombuffer(img, np.float32).any()
    assert (b / "output" / "dir.image").read_bytes() == img
    assert (b / "image.num").read_bytes() == (a / "image.num").read_bytes()
    assert (b / "output" / "dir.image_lap").read_bytes() == (a / "output" / "dir.image_lap").read_bytes()
    for name in ("dir.snaps", "dir.snaps_rec", "dir.snapr"):
        assert os.path.getsize(b / "output" / name) == 0


def test_python_deck_reader_applies_the_reference_defaults(tmp_path):
    """rtm.read_deck = fdw_deck_* plus init_args' defaults (fd-code.cu:343-378); no GPU involved."""
    from parallel_finite_difference_computation_amd.rtm import read_deck
    d = read_deck(os.path.join(DECKS, "new_mod.dat"))
    assert (d["nx"], d["nz"], d["nt"], d["ns"], d["fsx"], d["ds"], d["nxb"], d["order"]) == (315, 195, 1700, 6, 7, 60, 50, 8)
    assert d["vel_ext_file"] == "./models/new_mod/vel_ext_rnd.6" and abs(d["fac"] - 0.75) < 1e-7
    (tmp_path / "min.dat").write_text("tmpdir=.\nvpfile=v\ndatfile=d\nnz=3\nnx=4\nnt=5\ndz=1\ndx=1\ndt=0.001\nfpeak=10\n")
    d = read_deck(str(tmp_path / "min.dat"))
    assert (d["ns"], d["sz"], d["fsx"], d["ds"], d["gz"], d["order"], d["nzb"], d["nxb"]) == (1, 0, 0, 1, 0, 8, 40, 40)
    assert np.float32(d["fac"]) == np.float32(0.7) and d["vel_ext_file"] is None
    with pytest.raises(FileNotFoundError):
        read_deck(str(tmp_path / "absent.dat"))


@pytest.mark.gpu
def test_mod_main_program_reproduces_the_reference_gather(tmp_path):
    """./mod_main par=input.dat on the CPU-serial sibling's own deck and model (build/3lay_mod): the datfile it writes is its
    committed dobs.bin byte for byte."""
    shutil.copy(os.path.join(GOLDEN, "dd_3lay_mod_vp_151x151.f32"), tmp_path / "3layer_151x151.bin")
    shutil.copy(os.path.join(DECKS, "dd_3lay_mod.dat"), tmp_path / "input.dat")
    r = subprocess.run([os.path.join(BIN, "mod_main"), "par=input.dat"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "## nz = 151, nx = 151, nt = 1001 " in r.stdout and "** source 1, at (0,0) " in r.stdout and "F = 0.010000" in r.stdout
    assert (tmp_path / "dobs.bin").read_bytes() == open(os.path.join(GOLDEN, "dd_3lay_mod_dobs.f32"), "rb").read()


@pytest.mark.gpu
def test_modelled_gather_feeds_the_rtm_program(tmp_path):
    """The pipeline end to end on a small synthetic model: mod_main writes the gather (ns shots), rtm_code migrates it."""
    nx, nz, nt, ns = 61, 47, 260, 2
    vp = np.full((nx, nz), 2000.0, np.float32)
    vp[:, 25:] = 3000.0                                   # one reflector
    vp.tofile(tmp_path / "vp.bin")
    (tmp_path / "out").mkdir()
    (tmp_path / "mod.dat").write_text(f"tmpdir=./out\nvpfile=vp.bin\ndatfile=dobs.bin\nnz={nz}\nnx={nx}\nnt={nt}\ndz=10\ndx=10\ndt=0.001\n"
                                      f"fpeak=30.\nns={ns}\nsz=0\nfsx=20\nds=20\ngz=0\nnxb=20\nnzb=20\nfac=0.02\norder=8\n")
    r = subprocess.run([os.path.join(BIN, "mod_main"), "par=mod.dat"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    d = np.fromfile(tmp_path / "dobs.bin", np.float32).reshape(ns, nx, nt)
    assert np.isfinite(d).all() and np.abs(d).max() > 0
    (tmp_path / "rtm.dat").write_text((tmp_path / "mod.dat").read_text().replace("fac=0.02", "fac=0.75") + "rnd=1\n")
    r = subprocess.run([os.path.join(BIN, "rtm_code"), "./rtm.dat"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    img = np.fromfile(tmp_path / "out" / "dir.image", np.float32).reshape(nx, nz)
    assert np.isfinite(img).all() and np.abs(img).max() > 0


@pytest.mark.gpu
def test_rtm_main_program_reproduces_the_reference_image(tmp_path):
    """The sibling's whole workflow on its own 3lay_mod deck: ./mod_main writes dobs.bin, ./rtm_main migrates it; dir.image and
    dir.img are its committed images byte for byte."""
    shutil.copy(os.path.join(GOLDEN, "dd_3lay_mod_vp_151x151.f32"), tmp_path / "3layer_151x151.bin")
    shutil.copy(os.path.join(DECKS, "dd_3lay_mod.dat"), tmp_path / "input.dat")
    for exe in ("mod_main", "rtm_main"):
        r = subprocess.run([os.path.join(BIN, exe), "par=input.dat"], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout
    assert "** backward propagation 1, at (0,0) " in r.stdout and "Execution Time:" in r.stdout
    gold = open(os.path.join(GOLDEN, "dd_3lay_mod_dir_image.f32"), "rb").read()
    assert (tmp_path / "dir.image").read_bytes() == gold
    assert (tmp_path / "dir.img").read_bytes() == gold          # one shot: the per-shot image is the stack


@pytest.mark.gpu
def test_lapfilt_program_reproduces_the_reference_output(tmp_path):
    shutil.copy(os.path.join(GOLDEN, "dd_3lay_mod_dir_image.f32"), tmp_path / "dir.image")
    r = subprocess.run([os.path.join(BIN, "lapfilt")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "dir.imalap").read_bytes() == open(os.path.join(GOLDEN, "dd_3lay_mod_dir_imalap.f32"), "rb").read()


@pytest.mark.gpu
def test_psnr_program_against_the_reference_tool(tmp_path):
    """bin/psnr (the reference's offline image comparer models/marmousi/psnr, an ELF without source) against the output of that very binary
    (tests/golden/psnr_reference_output.json): stdout identical character for character -- usage / error messages, and the MSE, RMSE, SNR,
    PSNR lines, whose serial fp32 sums one GPU lane reproduces -- exit status identical, ./dir.output (the difference) identical byte for
    byte; the parallel reduction in double (exact_sums) agrees with the oracle's restatement to 1e-4 (the tool's own accumulated rounding
    reaches 3e-5 on these images)."""
    import hashlib
    import json
    g = json.load(open(os.path.join(GOLDEN, "psnr_reference_output.json")))
    img = golden_field("dd_3lay_mod_dir_image.f32", (151 * 151,))
    noisy = (img + np.float32(0.05) * np.abs(img).max() * np.random.default_rng(7).standard_normal(img.size).astype(np.float32)).astype(np.float32)
    noisy.tofile(tmp_path / "noisy.f32")
    path = lambda n: str(tmp_path / n) if n in ("noisy.f32", "nonexistent") else os.path.join(GOLDEN, n)
    exe = os.path.join(BIN, "psnr")
    for c in g["cases"]:
        argv = [path(c["a"]), path(c["b"])] if "a" in c else [path(x) for x in c["argv"]]
        r = subprocess.run([exe] + argv, cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        if "a" not in c:
            assert r.stdout == c["stdout"], c["argv"]
            continue
        assert hashlib.sha256((tmp_path / "dir.output").read_bytes()).hexdigest() == c["dir_output_sha256"]
        assert r.stdout == c["stdout"], (c["a"], c["b"])
        fa, fb = np.fromfile(path(c["a"]), np.float32), np.fromfile(path(c["b"]), np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ost = O.image_compare(fa, fb)
        st, st2 = F.image_compare(fa, fb), F.image_compare(fa, fb, exact_sums=True)
        for k, v in zip(("mse", "rmse", "snr", "psnr"), ost):
            assert st[k] == v, (k, st[k], v)
            assert st2[k] == v or abs(st2[k] - v) <= 1e-4 * max(abs(v), 1.0), (k, st2[k], v)


def test_committed_bench_lines_keep_the_contract():
    """The bench lines committed under profiles/ (what DESIGN.md quotes) carry every field the driver's contract names, a roofline fraction that
    is a fraction, and a traffic figure taken on the kernel build the profile names."""
    import glob
    import json
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    files = sorted(glob.glob(os.path.join(root, "r02_bench_*.json")) + glob.glob(os.path.join(root, "r03_bench_*.json")))
    assert len(files) >= 16
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        if "ranks_shm_rehearsal" in f:      # functional rehearsals of the N > 1 path on one GPU: the line's shape, not its value
            assert d["n_gpus"] in (2, 4) and d["rccl_ranks"] is None and d["comm_ranks"] == d["n_gpus"] and "bitwise equal" in d["decomposition_check"]
            assert 0.0 <= d["exposed_comm_fraction"] <= 1.0 and d["launched_by"].startswith("bench.py itself")
            continue
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in d, f"{os.path.basename(f)}: {key} missing"
        assert d["unit"] == "Gpoints/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["n_gpus"] == 1
        assert "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in r, f"{os.path.basename(f)}: roofline.{key} missing"
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert 0.0 < r["frac"] <= 1.0, f"{os.path.basename(f)}: roofline.frac = {r['frac']} is not a fraction"
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        if r["traffic"] is not None:
            assert "profiles/" in r.get("traffic_source", ""), os.path.basename(f)
            if os.path.basename(f).startswith("r03_"):      # a FAST line quotes the counter profile of the FAST kernels, an EXACT line that of the EXACT ones
                assert ("fast.txt" in r["traffic_source"]) == (d.get("numerics") == "fast"), f"{os.path.basename(f)}: {d.get('numerics')} line quotes {r['traffic_source']}"
        if "cpu_baseline" in d:
            for key in ("value", "unit", "cores", "kind", "sample"):
                assert key in d["cpu_baseline"], f"{os.path.basename(f)}: cpu_baseline.{key} missing"
        if os.path.basename(f) in ("r03_bench_8192_steps20.json", "r03_bench_8192_steps1000.json"):      # the driver's line: headline EXACT, the rest beside it
            assert d["numerics"] == "exact" and d["roofline"]["traffic"] and d["cpu_baseline"]["cores"] >= 1
            fl, ex = d["fast_numerics"], d["extra"]["baseline_config_2"]
            assert fl["value"] > d["value"] and fl["roofline"]["traffic"] and 0.0 < fl["roofline"]["frac"] <= 1.0
            assert ex["grid"] == [4096, 4096] and ex["steps"] == 1000 and ex["value"] > 0
    entries = json.load(open(os.path.join(root, "traffic.json")))
    assert {e["workload"] for e in entries} >= {"forward", "stencil", "rtm-slab", "model"}
    for e in entries:
        assert os.path.exists(os.path.join(root, "..", e["source"])), e["source"]


def test_committed_counter_profiles_belong_to_the_current_kernel_sources():
    """profiles/traffic.json (what bench.py's roofline.traffic quotes) was taken on exactly the kernel sources in the tree: every entry carries
    the hash of the translation units behind its workload, and bench.py drops an entry whose hash is stale -- this test makes a kernel edit
    that forgets to re-profile visible here instead of silently turning `traffic` into null on the driver's run."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_for_hash", os.path.join(ROOT, "bench.py"))
    # bench.py imports torch at module level; the hash helper itself needs nothing of it
    src = open(os.path.join(ROOT, "bench.py")).read()
    ns = {"os": os, "ROOT": ROOT}
    start, end = src.index("KERNEL_SOURCES = {"), src.index("def offline_counters(")
    exec(src[start:end], ns)
    entries = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert {e["workload"] for e in entries} >= {"forward", "forward-fast", "rtm-slab", "model", "stencil"}
    for e in entries:
        assert e["source_hash"] == ns["kernel_source_hash"](e["workload"]), f"profiles/traffic.json: the {e['workload']} {e['size']} entry ({e['source']}) is stale: re-profile (scripts/profile_all.sh)"
        assert os.path.exists(os.path.join(ROOT, e["source"])), e["source"]


def test_built_library_has_no_store_followed_by_a_write_of_its_data():
    """scripts/lint_store_hazard.py on the built libfdwave.so: no 96 / 128-bit vector store (buffer with or without a register soffset, global,
    flat, scratch) is followed within two wait states -- along straight-line code or across a branch -- by a VALU write of its data
    registers (the gfx950 hazard of csrc/fdw_device.h, f4_store_arr; the check reads the disassembly of the embedded gfx950 code objects, so
    it needs the ROCm tools but no GPU)."""
    import importlib.util
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    lib = os.path.join(root, "parallel_finite_difference_computation_amd", "libfdwave.so")
    if not (os.path.exists(lib) and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump")):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    spec = importlib.util.spec_from_file_location("lint_store_hazard", os.path.join(root, "scripts", "lint_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    findings, nstores, nsymbols, notes, nbranches = mod.check(lib)
    assert nsymbols > 20 and nstores >= 1000 and nbranches > 0, "the disassembly was not read (no kernels / no stores / no branch inside a window found)"
    assert not findings, "\n".join(findings)
    # the walker itself: a store at the end of a loop body whose data register is rewritten at the loop head, and the padded form
    loop = """
0000000000001000 <k>:
	v_mov_b32_e32 v40, v1                                      // 000000001000: 7E500301
	s_nop 0                                                    // 000000001004: BF800000
	buffer_store_dwordx4 v[40:43], v48, s[8:11], s18 offen     // 000000001008: E07C1000 12022830
	s_cbranch_scc1 65532                                       // 000000001010: BF85FFFC
	s_endpgm                                                   // 000000001014: BF810000
"""
    insts = mod.parse(loop)
    assert mod.branch_target(insts[4]) == 0x1004 and insts[3][2] == "buffer_store_dwordx4"
    orig = mod.disassemble
    try:
        mod.disassemble = lambda _lib: iter([(0, loop.replace("65532", "65531"))])           # back to the v_mov: one wait state (the branch) in between
        found = mod.check("x")[0]
        assert len(found) == 1 and "v_mov_b32" in found[0]
        mod.disassemble = lambda _lib: iter([(0, loop)])                                       # back to the s_nop behind it: two wait states, clean
        assert mod.check("x")[0] == []
    finally:
        mod.disassemble = orig
