#!/usr/bin/env python3
"""Generates the committed golden fixtures from the reference tree (run in the authoring container
only: `python tests/golden/make_golden.py`).  Fixtures are DATA: binary fields the reference ships,
and tables produced by the reference's own host library compiled unmodified (oracle/_ref, built by
oracle/Makefile from cuda_reference_RTM/lib/src/functions.c).  No reference source text is copied.

  stencil_input_415x295.f32   cuda_reference_stencil_computation/input.bin  (real-hardware output: P of
                              fd_forward, shot 5 of models/new_mod, after 1700 steps; also the input of
                              the Laplacian known-answer test)
  stencil_lap_415x295.f32     dpct_migrated_stencil_computation/output_teste.bin (8th-order Laplacian of it)
  new_mod_vel_ext_shot5.f32   slice [5] of cuda_reference_RTM/models/new_mod/vel_ext_rnd.6 (415x295)
  new_mod_vel_koslov.f32      cuda_reference_RTM/models/new_mod/vel-koslov.1 (315x195)
  new_mod_vel_ext_rnd6.npz    all six slices of cuda_reference_RTM/models/new_mod/vel_ext_rnd.6 (key "vel", [6][415][295], zlib-compressed:
                              2.9 MB -> 0.9 MB): the per-shot extended velocity models of the reference's own six-shot deck
  marmousi_model_375.npz      cuda_reference_RTM/models/marmousi/model-375.cwp (key "vp", [369][375]): the velocity model of the reference's
                              dx = 25 / dz = 8 deck (decks/marmousi.dat)
  host_tables.npz             calc_coefs / ricker_wavelet / extendvel_linear outputs of oracle/_ref
  decks/*.dat                 the reference's input.dat decks (parser fixtures)
  dd_3lay_mod_vp_151x151.f32  dpct_gpu_rtm_domain_division/build/3lay_mod/3layer_151x151.bin (velocity model of the CPU-serial sibling)
  dd_3lay_mod_dobs.f32        dpct_gpu_rtm_domain_division/build/3lay_mod/dobs.bin: the gather its mod_main produced from that
                              model and decks/dd_3lay_mod.dat (151 traces x 1001 samples) -- known answer of the modelling producer
  dd_3lay_mod_dir_image.f32   dpct_gpu_rtm_domain_division/build/3lay_mod/dir.image: the image its rtm_main formed from that gather
                              (151x151; with ns = 1 the per-shot dir.img is the same bytes) -- known answer of the stored-wavefield RTM
  psnr_reference_output.json  stdout (and the SHA-256 of ./dir.output) of the reference's own comparer models/marmousi/psnr -- an x86-64 ELF
                              shipped without source -- run here on pairs of the fixtures above: known answers of bin/psnr / fdw_image_compare
  dd_3lay_mod_dir_imalap.f32  output of the reference's laplace.f90 (built unmodified with flang: oracle/_ref/lapfilt) run on that dir.image
"""
import ctypes as C
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("FDW_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def main():
    cp = lambda src, dst: shutil.copyfile(os.path.join(REF, src), os.path.join(HERE, dst))
    cp("cuda_reference_stencil_computation/input.bin", "stencil_input_415x295.f32")
    cp("dpct_migrated_stencil_computation/output_teste.bin", "stencil_lap_415x295.f32")
    cp("cuda_reference_RTM/models/new_mod/vel-koslov.1", "new_mod_vel_koslov.f32")
    vel = np.fromfile(os.path.join(REF, "cuda_reference_RTM/models/new_mod/vel_ext_rnd.6"), np.float32).reshape(6, 415, 295)
    vel[5].tofile(os.path.join(HERE, "new_mod_vel_ext_shot5.f32"))
    np.savez_compressed(os.path.join(HERE, "new_mod_vel_ext_rnd6.npz"), vel=vel)
    np.savez_compressed(os.path.join(HERE, "marmousi_model_375.npz"),
                        vp=np.fromfile(os.path.join(REF, "cuda_reference_RTM/models/marmousi/model-375.cwp"), np.float32).reshape(369, 375))
    for name, src in [("stencil.dat", "cuda_reference_stencil_computation/input.dat"),
                      ("new_mod.dat", "cuda_reference_RTM/models/new_mod/input.dat"),
                      ("marmousi.dat", "cuda_reference_RTM/models/marmousi/input.dat"),
                      ("1lay_mod.dat", "cuda_reference_RTM/models/1lay_mod/input.dat"),
                      ("3lay_mod.dat", "cuda_reference_RTM/models/3lay_mod/input.dat"),
                      ("dd_3lay_mod.dat", "dpct_gpu_rtm_domain_division/build/3lay_mod/input.dat")]:
        cp(src, os.path.join("decks", name))
    cp("dpct_gpu_rtm_domain_division/build/3lay_mod/3layer_151x151.bin", "dd_3lay_mod_vp_151x151.f32")
    cp("dpct_gpu_rtm_domain_division/build/3lay_mod/dobs.bin", "dd_3lay_mod_dobs.f32")
    cp("dpct_gpu_rtm_domain_division/build/3lay_mod/dir.image", "dd_3lay_mod_dir_image.f32")
    lapfilt = O.ref_lapfilt()
    assert lapfilt is not None, "build oracle/_ref/lapfilt first (make -C oracle)"
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:       # the program reads ./dir.image and writes ./dir.imalap (151x151 built in)
        shutil.copyfile(os.path.join(HERE, "dd_3lay_mod_dir_image.f32"), os.path.join(td, "dir.image"))
        subprocess.check_call([lapfilt], cwd=td)
        shutil.copyfile(os.path.join(td, "dir.imalap"), os.path.join(HERE, "dd_3lay_mod_dir_imalap.f32"))
    # the reference's psnr binary on pairs of committed fixtures (it writes ./dir.output, so it runs in a scratch directory)
    import hashlib
    import json
    cases = []
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "psnr")
        shutil.copyfile(os.path.join(REF, "cuda_reference_RTM/models/marmousi/psnr"), exe)
        os.chmod(exe, 0o755)
        img = np.fromfile(os.path.join(HERE, "dd_3lay_mod_dir_image.f32"), np.float32)
        noisy = (img + np.float32(0.05) * np.abs(img).max() * np.random.default_rng(7).standard_normal(img.size).astype(np.float32)).astype(np.float32)
        noisy.tofile(os.path.join(td, "noisy.f32"))
        for a, b in (("dd_3lay_mod_dir_image.f32", "dd_3lay_mod_dir_imalap.f32"), ("dd_3lay_mod_dir_image.f32", "dd_3lay_mod_dir_image.f32"),
                     ("noisy.f32", "dd_3lay_mod_dir_image.f32"), ("dd_3lay_mod_dir_image.f32", "noisy.f32")):
            pa = os.path.join(td if a == "noisy.f32" else HERE, a)
            pb = os.path.join(td if b == "noisy.f32" else HERE, b)
            r = subprocess.run([exe, pa, pb], cwd=td, capture_output=True, text=True)
            cases.append(dict(a=a, b=b, stdout=r.stdout, dir_output_sha256=hashlib.sha256(open(os.path.join(td, "dir.output"), "rb").read()).hexdigest()))
        for argv in ([], ["/nonexistent", os.path.join(HERE, "dd_3lay_mod_dir_image.f32")], [os.path.join(HERE, "dd_3lay_mod_dir_image.f32"), "/nonexistent"],
                     [os.path.join(HERE, "dd_3lay_mod_dir_image.f32"), os.path.join(HERE, "new_mod_vel_koslov.f32")]):
            r = subprocess.run([exe] + argv, cwd=td, capture_output=True, text=True)
            cases.append(dict(argv=[os.path.basename(x) for x in argv], stdout=r.stdout, returncode=r.returncode))
    json.dump(dict(noise="noisy.f32 = image + 0.05 max|image| N(0,1), numpy default_rng(7), float32", cases=cases), open(os.path.join(HERE, "psnr_reference_output.json"), "w"), indent=1)

    L = O.ref_lib()
    assert L is not None, "build oracle/_ref first (make -C oracle)"
    out = {}
    for order in (2, 4, 6, 8, 10, 12, 14, 16, 20, 32):
        r = L.calc_coefs(order)
        out[f"coefs_{order}"] = np.array([r[i] for i in range(order + 1)], np.float32)
    for nt, dt, fp in ((1700, 0.001, 20.0), (64, 0.001, 20.0), (401, 0.001, 40.0), (3004, 0.001, 6.5)):
        s = np.zeros(nt, np.float32)
        L.ricker_wavelet(nt, dt, fp, s)
        out[f"ricker_{nt}_{fp}"] = s
    libc = C.CDLL(None)

    def ref_extend(vp, nx, nz, nxb, nzb, seed):
        a = np.zeros((nx + 2 * nxb, nz + 2 * nzb), np.float32)
        a[nxb:nxb + nx, nzb:nzb + nz] = vp
        rows = (C.POINTER(C.c_float) * a.shape[0])(*[C.cast(a[i].ctypes.data, C.POINTER(C.c_float)) for i in range(a.shape[0])])
        libc.srand(seed)
        L.extendvel_linear(nx, nz, nxb, nzb, rows)
        return a

    rng = np.random.default_rng(7)
    small = (1500 + 2500 * rng.random((24, 20))).astype(np.float32)
    out["extvel_small_in"] = small
    out["extvel_small_seed1"] = ref_extend(small, 24, 20, 6, 5, 1)
    out["extvel_small_seed42"] = ref_extend(small, 24, 20, 6, 5, 42)
    vk = np.fromfile(os.path.join(HERE, "new_mod_vel_koslov.f32"), np.float32).reshape(315, 195)
    # the reference never seeds rand(): the first extendvel_linear call of a process sees seed 1
    out["extvel_new_mod_seed1"] = ref_extend(vk, 315, 195, 50, 50, 1)
    np.savez_compressed(os.path.join(HERE, "host_tables.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
