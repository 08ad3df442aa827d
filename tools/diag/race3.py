"""Diagnostic (FDR_DIAG build): which intermediate of a failing image is wrong -- spectrum after B', raw plane, partials."""
import sys, os, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
fdr = importlib.import_module("parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd")
lib = fdr.lib
rows, cols, B, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ns, gr = int(sys.argv[5]), int(sys.argv[6])
M, N = fdr.nextPowerOfTwo(rows), fdr.nextPowerOfTwo(cols)
rng = np.random.default_rng(1)
host = rng.random((B, rows, cols), dtype=np.float32)
d_in = torch.from_numpy(host).cuda()
d_o = torch.empty_like(d_in)
s = torch.cuda.current_stream().cuda_stream
hip = ctypes.CDLL("libamdhip64.so")

def slot_bufs(p, slot):
    w, r, m, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
    assert lib.fdr_debug_slot_ptrs(p._h, slot, ctypes.byref(w), ctypes.byref(r), ctypes.byref(m), ctypes.byref(n)) == 0
    work = np.empty(n.value * 2, dtype=np.float32); raw = np.empty(M * N, dtype=np.float32); part = np.empty(2 * (M // 4), dtype=np.float32)
    for dst, src in ((work, w), (raw, r), (part, m)):
        assert hip.hipMemcpy(ctypes.c_void_p(dst.ctypes.data), src, ctypes.c_size_t(dst.nbytes), 2) == 0
    return work, raw.reshape(M, N), part

with fdr.Plan(M, N, fdr.MODE_FAST) as p:
    p.set_psf_motion(15, 30.0, 0.01)
    p.set_batching(1, 1)
    ref = {}
    one = np.empty_like(host)
    for i in range(B):
        p.wiener_dev(d_in[i].data_ptr(), rows, cols, cols, d_o[i].data_ptr(), cols, fdr.NORM_PADDED, stream=s)
        torch.cuda.synchronize()
        one[i] = d_o[i].cpu().numpy()
        ref[i] = slot_bufs(p, 0)
    # A's output (pre-B' spectrum) of every image: a second plan whose filter is the identity (delta PSF, K = 0)
    aout = {}
    with fdr.Plan(M, N, fdr.MODE_FAST) as q:
        q.set_psf(np.ones((1, 1), dtype=np.float32), 0.0)
        for i in range(B):
            q.wiener_dev(d_in[i].data_ptr(), rows, cols, cols, d_o[i].data_ptr(), cols, fdr.NORM_PADDED, stream=s)
            torch.cuda.synchronize()
            aout[i] = slot_bufs(q, 0)[0]
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
        # the slots still hold the intermediates of the LAST chunk that used them
        for i in bad:
            chunk, k = divmod(i, gr)
            later = [c for c in range(chunk + 1, nchunks) if c % ns == chunk % ns]
            if later: continue  # overwritten by a later chunk
            slot = (chunk % ns) * gr + k
            work, raw, part = slot_bufs(p, slot)
            rw, rr, rp = ref[i]
            dw = np.nonzero(work != rw)[0]
            dr = np.nonzero((raw != rr).any(axis=1))[0]
            dp = np.nonzero(part != rp)[0]
            ps = (M * 4 + 16) * 2
            print("  image %d slot %d: spectrum floats differing %d (panels %s, rows %s); raw rows differing %d %s; partial floats differing %d" % (
                i, slot, len(dw), np.unique(dw // ps)[:8], np.unique((dw % ps) // 8)[:12], len(dr), dr[:12], len(dp)))
            if len(dw):
                j = dw[0]
                # what does the wrong spectrum hold?  compare with other images' reference spectra
                same = [int(np.count_nonzero(work[dw] == ref[q][0][dw])) for q in range(B)]
                print("    wrong spectrum values equal to image q's reference spectrum at the same place: %s of %d" % (same, len(dw)))
                pre = [int(np.count_nonzero(np.abs(work[dw] - aout[q][dw]) <= 1e-4 * (1e-3 + np.abs(aout[q][dw])))) for q in range(B)]
                print("    wrong spectrum values close to image q's PRE-B' spectrum (A output): %s" % pre)
                rel = np.abs(work - rw) / (1e-6 + np.abs(rw))
                gross = np.nonzero(rel > 1e-3)[0]
                print("    gross (rel > 1e-3) spectrum floats: %d; panels %s rows %s cols %s" % (len(gross), np.unique(gross // ps)[:10],
                      np.unique((gross % ps) // 8)[:16], np.unique(((gross % ps) % 8) // 2)))
                cols_bad = np.unique((dw // ps) * 4 + ((dw % ps) % 8) // 2)
                runs = np.split(cols_bad, np.nonzero(np.diff(cols_bad) > 1)[0] + 1)
                print("    differing COLUMNS: %d in %d runs: %s" % (len(cols_bad), len(runs), [(int(r[0]), int(r[-1])) for r in runs[:12]]))
                rowcnt = np.bincount(((dw % ps) // 8), minlength=M)
                print("    differing floats per row: min %d max %d (of %d columns * 2)" % (rowcnt.min(), rowcnt.max(), len(cols_bad)))
                pan = np.unique(dw // ps)
                full = [int(np.count_nonzero(work[pp * ps:(pp + 1) * ps] != rw[pp * ps:(pp + 1) * ps])) for pp in pan[:6]]
                print("    %d panels wrong; floats differing in the first of them: %s of %d per panel; rel err of wrong values: median %g" % (
                    len(pan), full, ps, float(np.median(np.abs(work[dw] - rw[dw]) / (1e-6 + np.abs(rw[dw]))))))
            shown += 1
        if shown >= 6: break
print("done")
