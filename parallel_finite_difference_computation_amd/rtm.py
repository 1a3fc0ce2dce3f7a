"""Shot-parallel RTM driver: the reference's `rtm_code` shot loop (cuda_reference_RTM/src/fd-code.cu:480-542)
with the shots dealt round-robin to the GPUs of a node, one process per GPU.

    python -m torch.distributed.run --nproc-per-node N -m parallel_finite_difference_computation_amd.rtm <input.dat>
    python -m parallel_finite_difference_computation_amd.rtm <input.dat>            # one GPU

Shots are independent (the reference has no multi-GPU path; this is the embarrassingly parallel axis of
SURVEY.md section 8e).  Every rank reads the deck and the inputs, runs its shots (small decks: a contiguous block of shots per
rank, advanced in batches through one launch per time step, `fdw_shot_batch`; otherwise `fdw_shot` per shot, round-robin), and rank 0
stacks the per-shot images IN SHOT ORDER, so `dir.image` is bit-identical to the single-GPU program whatever
N is (an all-reduce would change the fp32 summation order).  The unseeded-rand() border model is generated on
the device from the resident interior model (`fdw_dev_extendvel_linear`): the stream is addressed by position, so
shot s draws exactly what the serial program would give it without any rank replaying the shots before it
(FDW_HOST_BORDER=1 or a one-cell border: the host loop, replayed on every rank).
Outputs are the reference's: <tmpdir>/dir.image, dir.image_lap (zeros), empty dir.snaps*, ./image.num.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

from . import api
from ._lib import lib


def read_deck(path):
    """Deck reader of the C programs (csrc/fdw_config.c) -> dict with the reference's defaults (fd-code.cu:343-378)."""
    L = lib()
    L.fdw_deck_read.restype = C.c_void_p
    L.fdw_deck_read.argtypes = [C.c_char_p]
    L.fdw_deck_int.argtypes = [C.c_void_p, C.c_char_p]
    L.fdw_deck_float.argtypes = [C.c_void_p, C.c_char_p]
    L.fdw_deck_float.restype = C.c_float
    L.fdw_deck_str.argtypes = [C.c_void_p, C.c_char_p]
    L.fdw_deck_str.restype = C.c_char_p
    L.fdw_deck_free.argtypes = [C.c_void_p]
    h = L.fdw_deck_read(path.encode())
    if not h:
        raise FileNotFoundError(path)
    try:
        s = lambda k: (lambda v: v.decode() if v is not None else None)(L.fdw_deck_str(h, k.encode()))
        i = lambda k: L.fdw_deck_int(h, k.encode())
        f = lambda k: float(L.fdw_deck_float(h, k.encode()))
        d = dict(tmpdir=s("tmpdir"), vpfile=s("vpfile"), datfile=s("datfile"), vel_ext_file=s("vel_ext_file"),
                 nz=i("nz"), nx=i("nx"), nt=i("nt"), ns=i("ns"), sz=i("sz"), fsx=i("fsx"), ds=i("ds"), gz=i("gz"),
                 order=i("order"), nzb=i("nzb"), nxb=i("nxb"), dz=f("dz"), dx=f("dx"), dt=f("dt"), fpeak=f("fpeak"), fac=f("fac"),
                 numerics=i("numerics"))
    finally:
        L.fdw_deck_free(h)
    for key, default in (("ns", 1), ("sz", 0), ("fsx", 0), ("ds", 1), ("gz", 0), ("order", 8), ("nzb", 40), ("nxb", 40)):
        if d[key] == -1:
            d[key] = default
    if d["fac"] == -1.0:
        d["fac"] = np.float32(0.7).item()
    if "FDW_NUMERICS" in os.environ:
        d["numerics"] = int(os.environ["FDW_NUMERICS"])
    d["numerics"] = 1 if d["numerics"] == 1 else 0      # our extension (absent = the reference's arithmetic): numerics=1 selects FAST numerics (fdwave.h)
    return d


def run(deck_path, out=sys.stdout):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("FDW_DIST_BACKEND", "nccl")
    ngpu = torch.cuda.device_count()
    if ngpu == 0:
        raise RuntimeError("no GPU visible: the product has no CPU path")
    local = int(os.environ.get("LOCAL_RANK", "0")) % ngpu
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    t0 = time.time()
    d = read_deck(deck_path)
    nx, nz, nt, ns, nxb, nzb = d["nx"], d["nz"], d["nt"], d["ns"], d["nxb"], d["nzb"]
    nxe, nze = nx + 2 * nxb, nz + 2 * nzb
    numerics = d["numerics"]
    say = (lambda *a: print(*a, file=out, flush=True)) if rank == 0 else (lambda *a: None)
    say(f"## vp = {d['vpfile']}, d_obs = {d['datfile']}, vel_ext_file = {d['vel_ext_file']}, vel_ext_flag = {int(d['vel_ext_file'] is not None)} ")
    say(f"## nz = {nz}, nx = {nx}, nt = {nt} ")
    say(f"## ns = {ns}, sz = {d['sz']}, fsx = {d['fsx']}, ds = {d['ds']}, gz = {d['gz']}  ({world} GPU(s), shots dealt round-robin)")
    srce = api.ricker_wavelet(nt, d["dt"], d["fpeak"])
    sx = [d["fsx"] + s * d["ds"] + nxb for s in range(ns)]
    sz, gz = d["sz"] + nzb, d["gz"] + nzb
    vp = np.fromfile(d["vpfile"], np.float32, nx * nz).reshape(nx, nz)
    d_obs = np.memmap(d["datfile"], np.float32, "r", shape=(ns, nx, nt))
    vel_ext = np.memmap(d["vel_ext_file"], np.float32, "r", shape=(ns, nxe, nze)) if d["vel_ext_file"] else None
    vpe = np.zeros((nxe, nze), np.float32)
    vpe[nxb:nxb + nx, nzb:nzb + nz] = vp
    # a shot of a small deck fills a few percent of one GPU: this rank's shots go through up to FDW_SHOT_WORKERS (default 4, as in rtm_code) contexts
    # (own stream and buffers each) on host threads; ctypes releases the GIL inside the library
    import concurrent.futures
    import threading
    nworkers = max(1, int(os.environ.get("FDW_SHOT_WORKERS", "4")))
    local_ctx = threading.local()

    dev_border = vel_ext is None and "FDW_HOST_BORDER" not in os.environ and nxb != 1 and nzb != 1 and nzb <= nxe

    def one_shot(s, v2):
        if not hasattr(local_ctx, "ctx"):
            local_ctx.ctx = api.FDWave(d["order"], nxe, nze, nxb, nzb, nt, d["fac"], d["dx"], d["dz"], d["dt"], compat=True, device=local, numerics=numerics)
            if dev_border:
                local_ctx.ctx.model_resident(vp)
        ctx = local_ctx.ctx
        if dev_border:          # fd-code.cu:486-494 in HBM; shot s of the serial program consumes draws [s T, (s + 1) T)
            ctx.dev_extendvel_linear(s * ctx.border_draws())
            return s, ctx.shot_resident(sx[s], sz, gz, srce, np.ascontiguousarray(d_obs[s]))
        return s, ctx.shot(v2, sx[s], sz, gz, srce, np.ascontiguousarray(d_obs[s]))

    mine, pending = {}, []
    # Small decks: a contiguous block of shots per rank, advanced in batches through ONE launch per time step (fdw_shot_batch: the shots of
    # a batch need consecutive places in the rand() stream and equally spaced sources, which consecutive shots have, fd-code.cu:405-407)
    batched = False
    if (dev_border or vel_ext is not None) and "FDW_NO_SHOT_BATCH" not in os.environ:
        ctx = api.FDWave(d["order"], nxe, nze, nxb, nzb, nt, d["fac"], d["dx"], d["dz"], d["dt"], compat=True, device=local, numerics=numerics)
        B = ctx.shot_batch_max()
        if B <= 1:
            ctx.close()        # the probing context is not needed: the per-thread contexts below do the work
        if B > 1:
            batched = True
            if dev_border:
                ctx.model_resident(vp)
            T = ctx.border_draws()
            lo, hi = rank * ns // world, (rank + 1) * ns // world
            for s0 in range(lo, hi, B):
                nb = min(B, hi - s0)
                for s in range(s0, s0 + nb):
                    print(f"** source {s + 1}, at ({sx[s] - nxb},{sz - nzb}) " + (f" [rank {rank}]" if world > 1 else ""), file=out, flush=True)
                v2_all = None if dev_border else np.square(np.asarray(vel_ext[s0:s0 + nb]), dtype=np.float32)    # fd-code.cu:484, 490-494
                imgs = ctx.shot_batch(nb, sx[s0], d["ds"], sz, gz, srce, np.ascontiguousarray(d_obs[s0:s0 + nb]), v2_all=v2_all, draw_offset=s0 * T)
                for b in range(nb):
                    mine[s0 + b] = imgs[b]
    with concurrent.futures.ThreadPoolExecutor(max_workers=nworkers) as pool:
        for s in range(0 if not batched else ns, ns):
            if vel_ext is not None:
                v = np.asarray(vel_ext[s])
            elif not dev_border:
                api.extendvel_linear(vpe, nx, nz, nxb, nzb)        # every rank replays the whole rand() stream (fd-code.cu:486)
                v = vpe
            if s % world != rank:
                continue
            print(f"** source {s + 1}, at ({sx[s] - nxb},{sz - nzb}) " + (f" [rank {rank}]" if world > 1 else ""), file=out, flush=True)
            pending.append(pool.submit(one_shot, s, None if dev_border else (v * v).astype(np.float32)))
        for f in pending:
            s, im = f.result()
            mine[s] = im
    # stack in shot order on rank 0 (fd-code.cu:522-528)
    if world > 1:
        # only rank 0 stacks: the images travel to it alone (ns x nx x nz floats through the host; a sum-reduction would reorder the
        # fp32 additions and break the byte identity with the serial program)
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        allimg = {}
        for g in gathered or ():
            allimg.update(g)
    else:
        allimg = mine
    if rank == 0:
        img = np.zeros((nx, nz), np.float32)
        with open("image.num", "w") as fnum:
            for s in range(ns):
                fnum.write(f"======== {s} ========\n")
                img = img + allimg[s]
                fnum.write("".join(f" {x:f} \n" for x in img.T.ravel()))
        for name in ("dir.snaps", "dir.snaps_rec", "dir.snapr"):
            open(os.path.join(d["tmpdir"], name), "w").close()
        img.tofile(os.path.join(d["tmpdir"], "dir.image"))
        np.zeros((nx, nz), np.float32).tofile(os.path.join(d["tmpdir"], "dir.image_lap"))
        say(f"> Exec time = {time.time() - t0:.2f} (s)")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit("usage: python -m parallel_finite_difference_computation_amd.rtm <input.dat>")
    run(sys.argv[1])
