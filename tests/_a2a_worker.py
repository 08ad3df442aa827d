"""Worker for test_host.py::test_two_rank_gloo_alltoall_transpose_logic: the distributed transpose of the slab mode on
CPU tensors (gloo): rank g holds rows of a global R x C matrix, packs its column blocks in rank order, exchanges them
with alltoall_blocks (the function the product uses) and must end up with its rows of the C x R transpose."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"


def main():
    import torch
    R, C = int(sys.argv[1]), int(sys.argv[2])
    batch = importlib.import_module(PKG + ".batch")
    slab = importlib.import_module(PKG + ".slab")
    comm = batch.Comm(backend="gloo")
    X = np.arange(R * C, dtype=np.float32).reshape(R, C)
    rcnt, rdis = batch.calculate_distribution(R, comm.world)
    ccnt, cdis = batch.calculate_distribution(C, comm.world)
    lr, lc = rcnt[comm.rank], ccnt[comm.rank]
    mine = X[rdis[comm.rank]:rdis[comm.rank] + lr]
    packed = np.concatenate([mine[:, cdis[p]:cdis[p] + ccnt[p]].reshape(-1) for p in range(comm.world)]) if lr else np.zeros(0, np.float32)
    recv = slab.alltoall_blocks(comm, torch.from_numpy(packed.copy()), [lr * c for c in ccnt], [r * lc for r in rcnt]).numpy()
    got = recv.reshape(R, lc).T if lc else np.zeros((0, R), np.float32)
    ok = bool(np.array_equal(got, X.T[cdis[comm.rank]:cdis[comm.rank] + lc]))
    oks = comm.gather_objects(ok)
    if comm.rank == 0:
        print(json.dumps({"world": comm.world, "ok": all(oks)}), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
