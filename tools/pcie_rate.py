#!/usr/bin/env python3
"""Host-pointer (PCIe-inclusive) rate of the operator: fdr_wiener_f32 copies the image in, restores it and
copies it back, synchronously.  Reported in DESIGN.md next to the device-resident figure; never bench.py's value."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fdr = importlib.import_module("parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
img = np.random.default_rng(0).random((S, S), dtype=np.float32)
with fdr.Plan(S, S, fdr.MODE_FAST) as p:
    p.set_psf_motion(50, 30.0, 0.01)
    p.wiener(img)
    t0 = time.perf_counter(); n = 5
    for _ in range(n):
        p.wiener(img)
    dt = (time.perf_counter() - t0) / n
    print("host-pointer path %dx%d: %.2f ms per image = %.0f Mpixels/s (pageable host memory, synchronous copies)" % (S, S, dt * 1e3, S * S / 1e6 / dt))
    B = 16
    for label, alloc in (("pageable", lambda s: np.empty(s, np.float32)), ("pinned (fdr_host_alloc), DMA in place", fdr.host_alloc)):
        imgs = alloc((B, S, S)); out = alloc((B, S, S))
        imgs[...] = img
        p.wiener_batch(imgs, out)
        t0 = time.perf_counter()
        p.wiener_batch(imgs, out)
        dt = (time.perf_counter() - t0) / B
        print("host batch pipeline %dx%d x %d, %s: %.2f ms per image = %.0f Mpixels/s (%.1f GB/s each way)" % (S, S, B, label, dt * 1e3, S * S / 1e6 / dt, S * S * 4 / dt / 1e9))
