import importlib, sys, numpy as np
sys.path.insert(0,'/root/repo')
fdr = importlib.import_module("parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd")
from oracle import oracle as o
for shape in [(128,128),(256,256),(512,512)]:
    psf = o.motion_blur_kernel(15, 30.0)
    img = o.synth_image(5, 0, shape[0]*shape[1]).reshape(shape)
    ref = o.serial_channel(img, psf, 0.01)
    for name, fl in (("default",0),("full_spectrum",fdr.FLAG_FULL_SPECTRUM),("no_pipeline",fdr.FLAG_NO_PIPELINE)):
        with fdr.Plan(shape[0], shape[1], fdr.MODE_FAST, flags=fl) as p:
            p.set_psf(psf, 0.01)
            got = p.wiener(img)
        d = np.abs(got-ref)
        bad = np.argwhere(d > 1e-3)
        print(shape, name, float(d.max()), 'nbad', len(bad), 'cols of bad:', np.unique(bad[:,1])[:12] if len(bad) else '')
