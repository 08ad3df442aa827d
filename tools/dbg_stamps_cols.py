"""Timeline of the 16-values-per-thread pass B' from a -DFDR_DEBUG_STAMPS build:
   FDR_STAMP_COLS=1 FDR_LIB_PATH=.../build_dbg/libfdr_stamps.so python tools/dbg_stamps_cols.py [size]"""
import ctypes, importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
fdr = importlib.import_module("parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nwg = S // 8
with fdr.Plan(S, S, fdr.MODE_FAST) as p:
    p.set_psf_motion(50, 30.0, 0.01)
    img = torch.rand((S, S), device="cuda")
    out = torch.empty_like(img)
    for _ in range(5):
        p.wiener_dev(img.data_ptr(), S, S, S, out.data_ptr(), S)
    torch.cuda.synchronize()
    buf = np.zeros((nwg, 32), dtype=np.uint64)
    fdr.lib.fdr_debug_dump_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert fdr.lib.fdr_debug_dump_stamps(buf.ctypes.data, nwg) == 0
t = buf[:, :8].astype(np.int64)
us = (t - t[:, 0].min()) / 100.0
names = ["start", "tile loads issued", "tile landed", "forward done", "filter applied", "inverse done", "stores issued", "stores complete"]
hw = buf[:, 8].astype(np.int64)
slot = hw & 15
for sel, label in ((slice(None), "all"), (slot % 2 == 0, "even wave slot"), (slot % 2 == 1, "odd wave slot")):
    print("--", label, int(np.count_nonzero(sel)) if not isinstance(sel, slice) else nwg)
    for k, n in enumerate(names):
        print("%-20s min %7.2f  median %7.2f  max %7.2f us" % (n, us[sel, k].min(), np.median(us[sel, k]), us[sel, k].max()))
d = np.diff(us, axis=1)
print("-- phase durations (median):", {names[k + 1]: round(float(np.median(d[:, k])), 2) for k in range(7)})
print("cu ids used:", len(np.unique((hw >> 8) & 0xFFFF)))
