"""CPU: the oracle (oracle/fdw_oracle.c) against every known answer the reference ships.
This is what pins the oracle; the GPU tests then compare the HIP path with the oracle."""
import numpy as np
import pytest

from conftest import assert_bit_equal, golden_field, rel_max
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
