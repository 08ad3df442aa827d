#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_vectors.npz from the CPU oracle (oracle/fdr_oracle.c).

The reference repository holds no golden vectors and cannot be built in this image (it needs
OpenCV), so these vectors come from the restatement, not from reference outputs: they pin the
restatement against silent change and give the GPU tests committed data to compare with
(parity unpinned).  Inputs are seeded; everything is float32/complex64.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402


def rand_c(rng, *shape):
    return (rng.random(shape, dtype=np.float32) - 0.5 + 1j * (rng.random(shape, dtype=np.float32) - 0.5)).astype(np.complex64)


def main():
    o.build(force=True)
    out = {}
    rng = np.random.default_rng(20251205)
    for n in (2, 4, 8, 16, 64, 1024):
        x = rand_c(rng, n)
        out["fft1d_in_%d" % n] = x
        out["fft1d_fwd_%d" % n] = o.fft_radix2(x, False)
        out["fft1d_inv_%d" % n] = o.fft_radix2(x, True)
    x = rand_c(rng, 12)
    out["naive_in_12"] = x
    out["naive_fwd_12"] = o.dft_naive(x, False)
    for shape in ((8, 8), (32, 64)):
        x = rand_c(rng, *shape)
        key = "%dx%d" % shape
        out["fft2d_in_" + key] = x
        out["fft2d_fwd_" + key] = o.dft2d(x, False)
        out["fft2d_inv_" + key] = o.dft2d(x, True)
    out["twiddle_fwd_16"] = o.twiddle_recurrence(16, False)
    out["twiddle_inv_16"] = o.twiddle_recurrence(16, True)
    for size, ang in ((50, 30.0), (40, 45.0), (15, 10.0)):
        out["psf_%d_%d" % (size, int(ang))] = o.motion_blur_kernel(size, ang)
    psf = o.motion_blur_kernel(15, 30.0)
    for shape in ((64, 64), (100, 200)):
        img = o.synth_image(0x5EED0002, 0, shape[0] * shape[1]).reshape(shape)
        out["wiener_serial_%dx%d" % shape] = o.serial_channel(img, psf, 0.01)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
