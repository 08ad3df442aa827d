"""Timeline of the fused C'+E kernel from a -DFDR_DEBUG_STAMPS build (make -C <package> OBJDIR=build_dbg/stamps LIB=build_dbg/libfdr_stamps.so EXTRA=-DFDR_DEBUG_STAMPS; plan flag FDR_FLAG_FUSED_NORM):
   FDR_LIB_PATH=.../build_dbg/libfdr_stamps.so python tools/dbg_stamps.py [size]"""
import ctypes, importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
fdr = importlib.import_module("parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nwg = int(sys.argv[2]) if len(sys.argv) > 2 else 256
with fdr.Plan(S, S, fdr.MODE_FAST, flags=fdr.FLAG_FUSED_NORM) as p:
    p.set_psf_motion(50, 30.0, 0.01)
    img = torch.rand((S, S), device="cuda")
    out = torch.empty_like(img)
    for _ in range(5):
        p.wiener_dev(img.data_ptr(), S, S, S, out.data_ptr(), S)
    torch.cuda.synchronize()
    buf = np.zeros((nwg, 32), dtype=np.uint64)
    fdr.lib.fdr_debug_dump_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = fdr.lib.fdr_debug_dump_stamps(buf.ctypes.data, nwg)
    assert rc == 0
t = buf.astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = {0: "first loads issued", 1: "rounds done", 2: "published", 3: "barrier done", 4: "stores issued", 5: "stores complete"}
for r in range(4):
    names.update({8 + 4 * r: "r%d packed" % r, 9 + 4 * r: "r%d next loads issued" % r, 10 + 4 * r: "r%d transformed" % r, 11 + 4 * r: "r%d min/max" % r})
order = [0] + [8 + k for k in range(16)] + [1, 2, 3, 4, 5]
for k in order:
    if not buf[:, k].any():
        continue
    print("%-24s min %7.2f  median %7.2f  max %7.2f us" % (names[k], us[:, k].min(), np.median(us[:, k]), us[:, k].max()))
