"""Diagnostic: failure counts of batched configurations vs the one-by-one result (run on the GPU box)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
fdr = importlib.import_module("parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd")
rows, cols, B, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
M, N = fdr.nextPowerOfTwo(rows), fdr.nextPowerOfTwo(cols)
rng = np.random.default_rng(1)
host = rng.random((B, rows, cols), dtype=np.float32)
d_in = torch.from_numpy(host).cuda()
d_o = torch.empty_like(d_in)
s = torch.cuda.current_stream().cuda_stream
print("stream handle", s)
for two, flags in ((0, 0), (1, 0), (0, fdr.FLAG_FULL_SPECTRUM)):
    with fdr.Plan(M, N, fdr.MODE_FAST, flags=flags) as p:
        p.set_option(fdr.OPT_TWO_SWEEP_NORM, two)
        p.set_psf_motion(15, 30.0, 0.01)
        one = np.stack([p.wiener(host[i]) for i in range(B)])
        for ns, gr in ((2, 2), (3, 2), (2, 4), (3, 1), (2, 1), (1, 4)):
            fails = {}
            p.set_batching(ns, gr)
            for rep in range(reps):
                d_o.fill_(-1.0)
                p.wiener_batch_dev(d_in.data_ptr(), rows * cols, B, rows, cols, cols, d_o.data_ptr(), rows * cols, cols, fdr.NORM_PADDED, stream=s)
                torch.cuda.synchronize()
                o = d_o.cpu().numpy()
                for i in range(B):
                    if np.count_nonzero(o[i] != one[i]):
                        fails[i] = fails.get(i, 0) + 1
            print("two_sweep=%d flags=%d batching %dx%d: failures per image %s of %d reps" % (two, flags, ns, gr, fails, reps), flush=True)
print("done")
