"""One rank of a multi-PROCESS run of the C slab driver (fdw_slabs_shot over the process transport of csrc/fdw_comm.cpp), started by
tests/test_slabs_gpu.py: python slab_rank_worker.py <case.npz> <segment name> <rank> <world> <ksteps> <out.npz>.  Test infrastructure."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallel_finite_difference_computation_amd as F  # noqa: E402


def main():
    case, name, rank, world, ksteps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    z = np.load(case)
    g = {k: z[k].item() for k in ("order", "nxe", "nze", "nxb", "nzb", "nt", "fac", "dx", "dz", "dt", "compat", "sx", "sz", "gz", "numerics")}
    comm = F.Comm.shm(name, rank, world, device=0, box_bytes=4 * 4 * max(ksteps, 16) * ((int(g["nze"]) + 63) // 64 * 64) * 4)
    assert (comm.rank, comm.world, comm.kind) == (rank, world, "shm")
    assert comm.allreduce(float(rank + 1)) == world * (world + 1) / 2 and comm.allreduce(float(rank), "max") == world - 1
    s = F.Slabs(int(g["order"]), int(g["nxe"]), int(g["nze"]), int(g["nxb"]), int(g["nzb"]), int(g["nt"]), float(g["fac"]), float(g["dx"]), float(g["dz"]),
                float(g["dt"]), comm=comm, compat=bool(g["compat"]), ksteps=ksteps, numerics=int(g["numerics"]))
    img, P, PP = s.shot(z["v2"], int(g["sx"]), int(g["sz"]), int(g["gz"]), z["srce"], z["d_obs"], imloc=z["im0"], want_fields=True)
    a, b = s.owned_interior_rows()
    np.savez(out, img=img[a:b], P=P[s.own0:s.own1], PP=PP[s.own0:s.own1], own=np.array([s.own0, s.own1, a, b]), nbuf=s.nbuf, ksteps=s.ksteps,
             back_buffers=np.array(s.back_buffers()))
    s.close()
    comm.close()


if __name__ == "__main__":
    main()
