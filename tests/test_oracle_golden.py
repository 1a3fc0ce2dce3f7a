"""CPU: the oracle (oracle/fdw_oracle.c) against every known answer the reference ships.
This is what pins the oracle; the GPU tests then compare the HIP path with the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_bit_equal, golden_field, rel_max
from oracle import oracle as O


def test_laplacian_known_answer_bit_exact():
    # dpct_migrated_stencil_computation/output_teste.bin == kernel_lap(input.bin), order 8, dx=dz=10
    inp = golden_field("stencil_input_415x295.f32", (415, 295))
    gold = golden_field("stencil_lap_415x295.f32", (415, 295))
    assert_bit_equal(O.stencil(8, 415, 295, 10.0, 10.0, inp), gold, "oracle kernel_lap vs output_teste.bin")
    assert not gold[:4].any() and not gold[-4:].any() and not gold[:, :4].any() and not gold[:, -4:].any()


def test_forward_loop_known_answer(new_mod):
    # cuda_reference_stencil_computation/input.bin is P of fd_forward, shot 5 of new_mod, nt=1700, produced on
    # real hardware.  Tolerance: 1e-5 max-norm-relative (north star); the FMA/no-FMA noise floor is ~4e-6.
    d = new_mod
    orc = O.Oracle(d["order"], d["nxe"], d["nze"], d["nxb"], d["nzb"], d["nt"], d["fac"], d["dx"], d["dz"], d["dt"], compat=True)
    P, PP = orc.forward(d["v2"], d["sx"], d["sz"], O.ricker_wavelet(d["nt"], d["dt"], d["fpeak"]))
    g = d["golden_P"]
    assert rel_max(P, g) < 1e-5
    assert np.linalg.norm(P - g) / np.linalg.norm(g) < 1e-5
    # rows/cols the truncated launch grids never touch stay exactly zero in both (fd-code.cu:185-195)
    assert not P[408:].any() and not P[:, 288:].any() and not g[408:].any() and not g[:, 288:].any()
    assert PP[:408, :288].any()


def test_fast_numerics_restatement_against_the_reference_goldens(new_mod):
    """The oracle's restatement of the product's FAST tolerance mode (orc_lap_fast: symmetric sums + fmaf, one chain per point) -- not
    the reference's arithmetic, so what it owes the reference is the north star's tolerance: the Laplacian golden within 1e-6 and the
    hardware forward-loop golden (input.bin, 1 700 steps) within 1e-5, max norm and L2; the untouched rows and columns stay zero."""
    inp = golden_field("stencil_input_415x295.f32", (415, 295))
    gold = golden_field("stencil_lap_415x295.f32", (415, 295))
    lap = O.stencil(8, 415, 295, 10.0, 10.0, inp, numerics=1)
    assert rel_max(lap, gold) < 1e-6 and (lap != gold).any()
    assert not lap[:4].any() and not lap[-4:].any() and not lap[:, :4].any() and not lap[:, -4:].any()
    d = new_mod
    orc = O.Oracle(d["order"], d["nxe"], d["nze"], d["nxb"], d["nzb"], d["nt"], d["fac"], d["dx"], d["dz"], d["dt"], compat=True, numerics=1)
    P, _ = orc.forward(d["v2"], d["sx"], d["sz"], O.ricker_wavelet(d["nt"], d["dt"], d["fpeak"]))
    g = d["golden_P"]
    assert rel_max(P, g) < 1e-5 and np.linalg.norm(P - g) / np.linalg.norm(g) < 1e-5
    assert not P[408:].any() and not P[:, 288:].any()


def test_fast_numerics_formula_spelled_out():
    """orc_lap_fast against the formula of include/fdwave.h written out with exact rational arithmetic: acc = c0 p, then per tap k one
    fused multiply-add of the z pair sum and one of the x pair sum (each fma rounded once from the exact value)."""
    from fractions import Fraction as Fr
    f32 = np.float32
    order, nxe, nze = 8, 13, 12
    p = np.random.default_rng(3).standard_normal((nxe, nze)).astype(f32)
    cx, cz = O.scaled_coefs(order, 7.5, 12.5, cxx=True)
    h = order // 2
    rnd = lambda q: f32(float(q))                        # Fraction -> nearest double -> nearest float (no tie within reach of these values)
    want = np.zeros_like(p)
    for i in range(h, nxe - h):
        for j in range(h, nze - h):
            acc = f32(f32(cz[h] + cx[h]) * p[i, j])
            for k in range(1, h + 1):
                acc = rnd(Fr(float(f32(p[i, j - k] + p[i, j + k]))) * Fr(float(cz[h - k])) + Fr(float(acc)))
                acc = rnd(Fr(float(f32(p[i - k, j] + p[i + k, j]))) * Fr(float(cx[h - k])) + Fr(float(acc)))
            want[i, j] = acc
    assert_bit_equal(O.stencil(order, nxe, nze, 7.5, 12.5, p, numerics=1), want, "orc_lap_fast vs the formula")


def test_full_extent_mode_differs_from_compat(new_mod):
    # guards the compat switch: with full launch extents the late-time field is a different answer
    d = new_mod
    srce = O.ricker_wavelet(300, d["dt"], d["fpeak"])
    a = O.Oracle(8, 415, 295, 50, 50, 300, 0.75, 10.0, 10.0, 0.001, compat=True).forward(d["v2"], d["sx"], d["sz"], srce)[0]
    b = O.Oracle(8, 415, 295, 50, 50, 300, 0.75, 10.0, 10.0, 0.001, compat=False).forward(d["v2"], d["sx"], d["sz"], srce)[0]
    assert not a[408:].any() and b[408:].any()


def test_host_tables_match_reference_library(tables):
    for order in (2, 4, 6, 8, 10, 12, 14, 16, 20, 32):
        assert_bit_equal(O.calc_coefs(order), tables[f"coefs_{order}"], f"calc_coefs({order})")
    for key in ("ricker_1700_20.0", "ricker_64_20.0", "ricker_401_40.0", "ricker_3004_6.5"):
        _, nt, fp = key.split("_")
        assert_bit_equal(O.ricker_wavelet(int(nt), 0.001, float(fp)), tables[key], key)


def test_extendvel_matches_reference_library(tables):
    small = tables["extvel_small_in"]
    for seed in (1, 42):
        v = np.zeros((24 + 12, 20 + 10), np.float32)
        v[6:30, 5:25] = small
        assert_bit_equal(O.extendvel_linear(v, 24, 20, 6, 5, seed=seed), tables[f"extvel_small_seed{seed}"], f"extendvel seed {seed}")
    vk = golden_field("new_mod_vel_koslov.f32", (315, 195))
    v = np.zeros((415, 295), np.float32)
    v[50:365, 50:245] = vk
    assert_bit_equal(O.extendvel_linear(v, 315, 195, 50, 50, seed=1), tables["extvel_new_mod_seed1"], "extendvel new_mod")


def test_oracle_against_live_reference_library():
    """When oracle/_ref (the reference's functions.c compiled unmodified) is present, compare live too."""
    L = O.ref_lib()
    if L is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    for order in (2, 8, 18, 24):
        r = L.calc_coefs(order)
        assert_bit_equal(O.calc_coefs(order), np.array([r[i] for i in range(order + 1)], np.float32), f"live calc_coefs({order})")
    s = np.zeros(500, np.float32)
    L.ricker_wavelet(500, 0.002, 12.5, s)
    assert_bit_equal(O.ricker_wavelet(500, 0.002, 12.5), s, "live ricker")


def test_taper_tables_shape_and_values():
    tx, tz = O.taper_tables(50, 50, 0.75)
    assert tx.shape == (50,) and np.all(np.diff(tx) > 0) and tx[-1] < 1.0
    # exp(-(sqrt(-ln F)/nb * nb)^2) = F at i = 0
    assert abs(tx[0] - 0.75) < 1e-6 and abs(tz[0] - 0.75) < 1e-6


def test_extents():
    assert O.extents(415, 295, 50, True) == (408, 288, 48)
    assert O.extents(415, 295, 50, False) == (415, 295, 50)
    assert O.extents(4096, 4096, 64, True) == (4096, 4096, 64)


def test_back_is_linear_in_the_data():
    # property of fd_back (no reference golden exists for the image): imloc is linear in d_obs
    from conftest import make_deck
    d = make_deck(72, 64, 12, 12, 40, seed=3)
    orc = O.Oracle(8, 72, 64, 12, 12, 40, 0.75, 10.0, 10.0, 0.001, compat=True)
    srce = O.ricker_wavelet(40, 0.001, 20.0)
    P, PP = orc.forward(d["v2"], d["sx"], d["sz"], srce)
    rng = np.random.default_rng(1)
    da = rng.standard_normal((48, 40)).astype(np.float32)
    ia = orc.back(d["v2"], P, PP, da, d["gz"])
    i2 = orc.back(d["v2"], P, PP, 2 * da, d["gz"])
    assert ia.any()
    assert rel_max(i2, 2 * ia) < 1e-5


# ---- forward-modelling producer of the CPU-serial sibling (oracle/fdw_oracle_mod.c) -------------------
def dd_3lay_mod():
    """dpct_gpu_rtm_domain_division/build/3lay_mod: deck values (decks/dd_3lay_mod.dat), model, and the gather its mod_main wrote."""
    nx = nz = 151
    nxb = nzb = 40
    nt = 1001
    vp = golden_field("dd_3lay_mod_vp_151x151.f32", (nx, nz))
    v2 = np.zeros((nx + 2 * nxb, nz + 2 * nzb), np.float32)
    v2[nxb:nxb + nx, nzb:nzb + nz] = vp * vp                     # mod_main.cpp:121-125
    return dict(order=8, nx=nx, nz=nz, nxb=nxb, nzb=nzb, nt=nt, dx=10.0, dz=10.0, dt=0.001, fac=0.010, fpeak=30.0, sx=0 + nxb, sz=0 + nzb,
                gz=0 + nzb, v2=v2, dobs=golden_field("dd_3lay_mod_dobs.f32", (nx, nt)))


def test_modelling_producer_known_answer_bit_exact():
    d = dd_3lay_mod()
    v2 = O.mod_extendvel(d["v2"], d["nx"], d["nz"], d["nxb"], d["nzb"])
    srce = O.mod_ricker_wavelet(d["nt"], d["dt"], d["fpeak"])
    assert srce[:66].any() and not srce[67:].any()             # cut off after 2/fpeak (ptsrc.c:92-94)
    data = O.mod_shot(d["order"], d["nx"], d["nz"], d["nxb"], d["nzb"], d["dx"], d["dz"], d["dt"], d["fac"], v2, d["sx"], d["sz"], d["gz"], srce)
    assert_bit_equal(data, d["dobs"], "oracle mod_main loop vs build/3lay_mod/dobs.bin")
    assert np.abs(d["dobs"]).max() > 1.0


def test_modelling_passes_match_the_reference_functions():
    """Each pass of the restatement against the sibling's own fd.c / taper.c / ptsrc.c compiled unmodified (oracle/_ref/libref_dd.so,
    C++ symbols because its Makefiles build them with g++)."""
    import ctypes as C
    R = O.ref_dd_lib()
    if R is None:
        pytest.skip("oracle/_ref/libref_dd.so not built (reference tree absent)")
    rng = np.random.default_rng(3)
    nx, nz, nxb, nzb, order = 37, 29, 9, 7, 8
    nxe, nze = nx + 2 * nxb, nz + 2 * nzb
    fp = C.POINTER(C.c_float)

    def rows(a):      # float** view of a C-contiguous [n][m] array
        return (fp * a.shape[0])(*[C.cast(a[i].ctypes.data, fp) for i in range(a.shape[0])])

    # taper tables + taper_apply
    R._Z10taper_initiif(nxb, nzb, C.c_float(0.05))
    a = rng.standard_normal((nxe, nze)).astype(np.float32)
    b = a.copy()
    R._Z11taper_applyPPfiiii(rows(a), nx, nz, nxb, nzb)
    tx, tz = O.mod_taper_tables(nxb, nzb, 0.05)
    O.lib().orc_mod_taper_apply(b, nx, nz, nxb, nzb, np.ascontiguousarray(tx), np.ascontiguousarray(tz))
    assert_bit_equal(b, a, "taper_apply")
    # extendvel
    a = rng.random((nxe, nze)).astype(np.float32)
    b = a.copy()
    R._Z9extendveliiiiPf(nx, nz, nxb, nzb, a.ctypes.data_as(fp))
    assert_bit_equal(O.mod_extendvel(b, nx, nz, nxb, nzb), a, "extendvel")
    # ricker with cut-off
    for nt, dt, fpk in ((300, 0.001, 30.0), (500, 0.0007, 17.5)):
        s = np.zeros(nt, np.float32)
        R._Z14ricker_waveletiffPf(nt, C.c_float(dt), C.c_float(fpk), s.ctypes.data_as(fp))
        assert_bit_equal(O.mod_ricker_wavelet(nt, dt, fpk), s, f"ricker_wavelet({nt},{dt},{fpk})")
    # ptsrc at the corner, an edge and the middle
    for (xs, zs) in ((0, 0), (2, 20), (nxe - 1, nze - 2), (20, 15)):
        a = rng.standard_normal((nxe, nze)).astype(np.float32)
        b = a.copy()
        R._Z5ptsrciiiifPPf(xs, zs, nxe, nze, C.c_float(1.7), rows(a))
        O.lib().orc_mod_ptsrc(xs, zs, nxe, nze, 1.7, b)
        assert_bit_equal(b, a, f"ptsrc at ({xs},{zs})")
    # fd_step for the table orders
    for order in (2, 4, 6, 8):
        R._Z7fd_initiiifff(order, nxe, nze, C.c_float(10.0), C.c_float(12.5), C.c_float(0.001))
        p = rng.standard_normal((nxe, nze)).astype(np.float32)
        pp = rng.standard_normal((nxe, nze)).astype(np.float32)
        v2 = (1500 + 2000 * rng.random((nxe, nze)).astype(np.float32)) ** 2
        pp_ref = pp.copy()
        R._Z7fd_stepiPPfS0_S0_ii(order, rows(p), rows(pp_ref), rows(v2), nze, nxe)
        coefs = O.calc_coefs(order, cxx=True)
        lap = np.zeros((nxe, nze), np.float32)
        dx2inv = np.float32((1. / 10.0) * (1. / 10.0))
        dz2inv = np.float32((1. / 12.5) * (1. / 12.5))
        O.lib().orc_mod_fd_step(order, np.ascontiguousarray(coefs), dx2inv, dz2inv, np.float32(0.001) * np.float32(0.001), p, pp, v2, lap, nze, nxe)
        assert_bit_equal(pp, pp_ref, f"fd_step order {order}")
        R._Z10fd_destroyv()
    R._Z13taper_destroyv()


def test_stored_wavefield_rtm_known_answer_bit_exact():
    """oracle restatement of the sibling's rtm_main against its committed image build/3lay_mod/dir.image."""
    d = dd_3lay_mod()
    v2 = O.mod_extendvel(d["v2"], d["nx"], d["nz"], d["nxb"], d["nzb"])
    srce = O.mod_ricker_wavelet(d["nt"], d["dt"], d["fpeak"])
    img = O.rtm_stored_shot(d["order"], d["nx"], d["nz"], d["nxb"], d["nzb"], d["dx"], d["dz"], d["dt"], d["fac"], v2, d["sx"], d["sz"], d["gz"], srce, d["dobs"])
    assert_bit_equal(img, golden_field("dd_3lay_mod_dir_image.f32", (d["nx"], d["nz"])), "oracle rtm_main loop vs build/3lay_mod/dir.image")


def test_taper_apply2_matches_the_reference_function():
    import ctypes as C
    R = O.ref_dd_lib()
    if R is None:
        pytest.skip("oracle/_ref/libref_dd.so not built (reference tree absent)")
    rng = np.random.default_rng(5)
    nx, nz, nxb, nzb = 37, 29, 9, 7
    fp = C.POINTER(C.c_float)
    R._Z10taper_initiif(nxb, nzb, C.c_float(0.05))
    a = rng.standard_normal((nx + 2 * nxb, nz + 2 * nzb)).astype(np.float32)
    b = a.copy()
    R._Z12taper_apply2PPfiiii((fp * a.shape[0])(*[C.cast(a[i].ctypes.data, fp) for i in range(a.shape[0])]), nx, nz, nxb, nzb)
    tx, tz = O.mod_taper_tables(nxb, nzb, 0.05)
    O.lib().orc_mod_taper_apply2(b, nx, nz, nxb, nzb, np.ascontiguousarray(tx), np.ascontiguousarray(tz))
    assert_bit_equal(b, a, "taper_apply2")
    R._Z13taper_destroyv()


def test_image_laplacian_known_answer_bit_exact():
    """Row f3: the oracle's restatement of laplace.f90 against the committed output of the reference program (built with flang), and,
    where that binary is present, against a fresh run of it."""
    import subprocess
    img = golden_field("dd_3lay_mod_dir_image.f32", (151, 151))
    want = golden_field("dd_3lay_mod_dir_imalap.f32", (151, 151))
    got = O.image_laplacian(img, 10.0, 10.0)
    assert_bit_equal(got, want, "oracle image Laplacian vs laplace.f90's dir.imalap")
    assert not want[0].any() and not want[-1].any() and not want[:, 0].any() and not want[:, -1].any() and np.abs(want).max() > 1
    exe = O.ref_lapfilt()
    if exe is not None:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            img.tofile(os.path.join(td, "dir.image"))
            subprocess.check_call([exe], cwd=td)
            assert_bit_equal(np.fromfile(os.path.join(td, "dir.imalap"), np.float32).reshape(151, 151), want, "fresh run of the reference program")


def test_image_compare_restatement_reproduces_the_reference_tool_output():
    """orc_image_compare against the stdout of the reference's own psnr binary (tests/golden/psnr_reference_output.json, written by
    make_golden.py from runs of models/marmousi/psnr): the same printed lines, character for character, and the same dir.output."""
    import hashlib
    import json
    g = json.load(open(os.path.join(GOLDEN, "psnr_reference_output.json")))
    img = golden_field("dd_3lay_mod_dir_image.f32", (151 * 151,))
    files = {"dd_3lay_mod_dir_image.f32": img, "dd_3lay_mod_dir_imalap.f32": golden_field("dd_3lay_mod_dir_imalap.f32", (151 * 151,)),
             "noisy.f32": (img + np.float32(0.05) * np.abs(img).max() * np.random.default_rng(7).standard_normal(img.size).astype(np.float32)).astype(np.float32)}
    n = 0
    for c in g["cases"]:
        if "a" not in c:
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            st, diff = O.image_compare(files[c["a"]], files[c["b"]], want_diff=True)
        assert O.psnr_lines(st) == c["stdout"], (c["a"], c["b"])
        assert hashlib.sha256(diff.tobytes()).hexdigest() == c["dir_output_sha256"]
        n += 1
    assert n == 4
