"""Diagnostic (FDR_DIAG build): identity filter (delta PSF, K = 0), so the spectrum after B' is pass A's output up to
rounding -- where (rows, columns) and by how much does a failing image's pass-A output differ?"""
import sys, os, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
fdr = importlib.import_module("parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd")
lib = fdr.lib
rows, cols, B, reps, ns, gr = [int(x) for x in sys.argv[1:7]]
M, N = fdr.nextPowerOfTwo(rows), fdr.nextPowerOfTwo(cols)
rng = np.random.default_rng(1)
host = rng.random((B, rows, cols), dtype=np.float32)
d_in = torch.from_numpy(host).cuda()
d_o = torch.empty_like(d_in)
s = torch.cuda.current_stream().cuda_stream
hip = ctypes.CDLL("libamdhip64.so")
PS = M * 4 + 16

def spectrum(p, slot):
    w, r, m, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
    assert lib.fdr_debug_slot_ptrs(p._h, slot, ctypes.byref(w), ctypes.byref(r), ctypes.byref(m), ctypes.byref(n)) == 0
    work = np.empty(n.value * 2, dtype=np.float32)
    assert hip.hipMemcpy(ctypes.c_void_p(work.ctypes.data), w, ctypes.c_size_t(work.nbytes), 2) == 0
    # panel-major -> [row, column] complex
    a = work.view(np.complex64).reshape(N // 8, PS)[:, :M * 4].reshape(N // 8, M, 4)
    return np.ascontiguousarray(a.transpose(1, 0, 2).reshape(M, N // 2))

with fdr.Plan(M, N, fdr.MODE_FAST) as p:
    p.set_psf(np.ones((1, 1), dtype=np.float32), 0.0)
    p.set_batching(1, 1)
    ref, one = {}, np.empty_like(host)
    for i in range(B):
        p.wiener_dev(d_in[i].data_ptr(), rows, cols, cols, d_o[i].data_ptr(), cols, fdr.NORM_PADDED, stream=s)
        torch.cuda.synchronize()
        one[i] = d_o[i].cpu().numpy()
        ref[i] = spectrum(p, 0)
    p.set_batching(ns, gr)
    nchunks = (B + gr - 1) // gr
    shown = 0
    for rep in range(reps):
        d_o.fill_(-1.0)
        p.wiener_batch_dev(d_in.data_ptr(), rows * cols, B, rows, cols, cols, d_o.data_ptr(), rows * cols, cols, fdr.NORM_PADDED, stream=s)
        torch.cuda.synchronize()
        o = d_o.cpu().numpy()
        bad = [i for i in range(B) if np.count_nonzero(o[i] != one[i])]
        if not bad: continue
        print("rep %d: bad images %s" % (rep, bad))
        for i in bad:
            chunk, k = divmod(i, gr)
            if [c for c in range(chunk + 1, nchunks) if c % ns == chunk % ns]: continue
            sp = spectrum(p, (chunk % ns) * gr + k)
            d = np.abs(sp - ref[i])
            scale = np.abs(ref[i]).max()
            big = d > 1e-4 * scale
            rws, cls = np.nonzero(big.any(axis=1))[0], np.nonzero(big.any(axis=0))[0]
            print("  image %d: |diff| > 1e-4 * max at %d places: %d rows %s, %d columns %s; max |diff| / max %g; exact-differing places %d" % (
                i, int(big.sum()), len(rws), rws[:24], len(cls), cls[:24], float(d.max() / scale), int((d > 0).sum())))
            if len(rws):
                r0 = rws[0]
                cc = np.nonzero(big[r0])[0]
                print("    row %d: wrong columns %s" % (r0, cc[:40]))
                print("    got %s\n    ref %s" % (sp[r0, cc[:4]], ref[i][r0, cc[:4]]))
                # is the wrong value some other element of the reference (a misplaced value)?
                for c in cc[:4]:
                    hits = np.argwhere(ref[i] == sp[r0, c])
                    print("    value at (%d,%d) found in the reference at %s" % (r0, c, hits[:4].tolist()))
            shown += 1
        if shown >= 5: break
print("done")
