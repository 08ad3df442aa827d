"""Worker for the single-image multi-GPU (slab) tests: every rank restores its row slab of ONE image through
..._amd/slab.py (all-to-all transposes over torch.distributed) and rank 0 compares the gathered picture with the
single-GPU parity mode of the same library (must be bit-identical) and with the CPU oracle.
usage (under torch.distributed.run): _slab_worker.py <rows> <cols> <psf_len> [--one-device] [--backend gloo|nccl] [--cropped]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"


def main():
    rows, cols, plen = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    one_device = "--one-device" in sys.argv
    backend = sys.argv[sys.argv.index("--backend") + 1] if "--backend" in sys.argv else "gloo"
    import torch
    fdr = importlib.import_module(PKG)
    batch = importlib.import_module(PKG + ".batch")
    slab = importlib.import_module(PKG + ".slab")
    from oracle import oracle as o
    local_rank = 0 if one_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    comm = batch.Comm(backend=backend, device=torch.device("cuda", local_rank))
    area = fdr.NORM_CROPPED if "--cropped" in sys.argv else fdr.NORM_PADDED
    img = o.synth_image(0x51AB, 0, rows * cols).reshape(rows, cols)
    psf = o.motion_blur_kernel(plen, 30.0)
    M = fdr.nextPowerOfTwo(rows)
    cnt, dis = batch.calculate_distribution(M, comm.world)
    first, lr = dis[comm.rank], cnt[comm.rank]
    mine = img[min(first, rows):min(first + lr, rows)]
    out = slab.wiener_slab(comm, mine, rows, cols, psf, 0.01, device=local_rank, norm_area=area)
    parts = comm.gather_objects(out)
    if comm.rank == 0:
        got = np.concatenate([p for p in parts if p.size], axis=0)
        ref_gpu = fdr.wienerDeblur_myfft(img, psf, 0.01, mode=fdr.MODE_PARITY, device=local_rank, norm_area=area)
        if area == fdr.NORM_PADDED:
            ref_cpu = o.serial_channel(img, psf, 0.01)
        else:
            padded = np.zeros((M, fdr.nextPowerOfTwo(cols)), np.float32); padded[:rows, :cols] = img
            _, raw = o.wiener(padded, psf, 0.01, want_raw=True)
            ref_cpu = o.normalize_minmax(raw[:rows, :cols])
        print(json.dumps({"world": comm.world, "shape": list(got.shape), "slab_rows": [int(p.shape[0]) for p in parts],
                          "mismatch_vs_single_gpu": int(np.count_nonzero(~(got == ref_gpu))),
                          "mismatch_vs_oracle": int(np.count_nonzero(~(got == ref_cpu))),
                          "max_abs_vs_oracle": float(np.abs(got - ref_cpu).max())}), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
