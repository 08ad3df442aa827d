"""Diagnostic: which plan / image / grouping disagrees with the one-by-one result (run on the GPU box)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
fdr = importlib.import_module("parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd")
rows, cols, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
M, N = fdr.nextPowerOfTwo(rows), fdr.nextPowerOfTwo(cols)
rng = np.random.default_rng(1)
host = rng.random((B, rows, cols), dtype=np.float32)
psf = fdr.motionBlurKernel(15, 30.0) if hasattr(fdr, "motionBlurKernel") else None
d_in = torch.from_numpy(host).cuda()
s = torch.cuda.current_stream().cuda_stream
for two in (0,):
    with fdr.Plan(M, N, fdr.MODE_FAST) as p:
        if two: p.set_option(fdr.OPT_TWO_SWEEP_NORM, two)
        p.set_psf_motion(15, 30.0, 0.01)
        one = np.stack([p.wiener(host[i]) for i in range(B)])
        for rep in range(12):
            for ns, gr in ((1, 4), (2, 2), (1, 3), (2, 4), (3, 2)):
                d_o = torch.full_like(d_in, -1.0)
                p.set_batching(ns, gr)
                p.wiener_batch_dev(d_in.data_ptr(), rows * cols, B, rows, cols, cols, d_o.data_ptr(), rows * cols, cols, fdr.NORM_PADDED, stream=s)
                torch.cuda.synchronize()
                o = d_o.cpu().numpy()
                for i in range(B):
                    bad = int(np.count_nonzero(o[i] != one[i]))
                    if bad:
                        d = np.abs(o[i] - one[i])
                        print("   max abs diff %g, mean %g; unwritten (-1) %d; equal to other images: %s; first bad rows %s" % (
                            d.max(), d.mean(), int(np.count_nonzero(o[i] == -1.0)),
                            [int(np.count_nonzero(o[i] == one[j])) for j in range(B)], np.unique(np.nonzero(o[i] != one[i])[0])[:5]))
                        big = np.nonzero((d > 0.02).any(axis=1))[0]
                        bigc = np.nonzero((d > 0.02).any(axis=0))[0]
                        print("   rows with |diff| > 0.02: %d (%s ... %s); cols: %d (%s ... %s)" % (len(big), big[:6], big[-3:], len(bigc), bigc[:6], bigc[-3:]))
                        a_, b_ = np.polyfit(one[i].ravel()[::97].astype(np.float64), o[i].ravel()[::97].astype(np.float64), 1)
                        print("   fit o = %.9f * one + %.3e" % (a_, b_))
                        print("two_sweep=%d rep %d batching %dx%d image %d: %d differ, out min/max %g %g vs %g %g" % (two, rep, ns, gr, i, bad, o[i].min(), o[i].max(), one[i].min(), one[i].max()))
print("done")
