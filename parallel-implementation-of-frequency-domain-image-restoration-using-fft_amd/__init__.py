"""MI355X-native frequency-domain image restoration (Wiener deconvolution) -- Python host mirror.

Thin ctypes binding over libfdr.so (HIP kernels + C ABI, include/fdr.h) plus functions carrying
the reference's own names (utils.hpp / fft/fft.hpp of the reference) so parity tests read like
the reference's drivers.  The product path is HIP only: importing this module fails loudly when
libfdr.so is missing, and nothing here falls back to numpy or to oracle/.

The directory name contains hyphens, so import it with importlib:
    fdr = importlib.import_module("parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd")
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FDR_LIB_PATH selects an alternative build of the same library (timing-only debug builds)
LIB_PATH = os.environ.get("FDR_LIB_PATH") or os.path.join(_HERE, "libfdr.so")

MODE_PARITY = 0
MODE_FAST = 1
FLAG_SIMPLE_PATH = 1
FLAG_FULL_SPECTRUM = 32
FLAG_ANY_SIZE = 512
FLAG_TABLES_ONLY = 1024
NORM_PADDED = 1
NORM_CROPPED = 0
MAX_PASSES = 16
OPT_TWO_SWEEP_NORM = 2
OPT_BATCH_GRAPH = 3
OPT_CE_CHUNK_MB = 4
PHASES = ("alloc", "h2d", "pre", "compute", "d2h", "post")  # the reference Profiler's buckets, fft/fft_gpu.cu:17-57
BATCH_MAX_DEVICES = 16

_f32p = ctypes.POINTER(ctypes.c_float)


class BatchDesc(ctypes.Structure):
    """fdr_batch_desc of include/fdr.h"""
    _fields_ = [("n_devices", ctypes.c_int), ("devices", ctypes.POINTER(ctypes.c_int)),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("mode", ctypes.c_int), ("flags", ctypes.c_uint),
                ("psf_host", ctypes.c_void_p), ("psf_rows", ctypes.c_int), ("psf_cols", ctypes.c_int), ("psf_stride", ctypes.c_int),
                ("psf_size", ctypes.c_int), ("psf_angle_deg", ctypes.c_double), ("K", ctypes.c_float),
                ("count", ctypes.c_int), ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("stride", ctypes.c_int), ("out_stride", ctypes.c_int),
                ("imgs_host", ctypes.POINTER(ctypes.c_void_p)), ("outs_host", ctypes.POINTER(ctypes.c_void_p)),
                ("synth_seed", ctypes.c_uint64), ("steps", ctypes.c_int), ("warmup", ctypes.c_int),
                ("nstreams", ctypes.c_int), ("group", ctypes.c_int), ("norm_area", ctypes.c_int), ("bcast_filter", ctypes.c_int)]


class BatchStats(ctypes.Structure):
    """fdr_batch_stats of include/fdr.h"""
    _fields_ = [("n_devices", ctypes.c_int), ("first", ctypes.c_int * 16), ("images", ctypes.c_int * 16),
                ("elapsed_ms", ctypes.c_double * 16), ("checksum", ctypes.c_double * 16), ("status", ctypes.c_int * 16),
                ("wall_ms", ctypes.c_double), ("images_done", ctypes.c_longlong), ("mpixels_per_s", ctypes.c_double), ("filter_path", ctypes.c_int)]


class FdrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfdr error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile libfdr.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs.append(os.path.join(_HERE, "..", "include", "fdr.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-j4", "-s"])
    return LIB_PATH


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libfdr.so not found at %s: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64 and publishes
    # them in the global symbol scope.  Loaded first, libfdr.so binds to that same runtime, so device
    # pointers and hipStream_t handles can be shared with torch; loaded second, torch's runtime would
    # find the GPU already claimed ("No HIP GPUs are available").  Without torch (C++ callers, the CLI)
    # libfdr.so simply uses /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for this binding
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cf, cd, cu = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_uint
    L.fdr_version.restype = ci
    L.fdr_last_error.restype = ctypes.c_char_p
    L.fdr_device_count.argtypes = [ctypes.POINTER(ci)]
    L.fdr_next_pow2.argtypes = [ci]
    L.fdr_is_pow2.argtypes = [ci]
    L.fdr_optimal_dft_size.argtypes = [ci]
    L.fdr_plan_set_option.argtypes = [vp, ci, ctypes.c_longlong]
    L.fdr_plan_phase_times.argtypes = [vp, _f32p, ci]
    L.fdr_batch_run.argtypes = [ctypes.POINTER(BatchDesc), ctypes.POINTER(BatchStats)]
    L.fdr_slab_pad_dev.argtypes = [vp, ci, ci, ci, vp, ci, ci, vp]
    L.fdr_slab_rows_fft_dev.argtypes = [vp, vp, ci, ci, ci, vp]
    L.fdr_slab_pack_dev.argtypes = [vp, ci, ci, ci, ctypes.POINTER(ci), ci, vp, vp]
    L.fdr_slab_transpose_dev.argtypes = [vp, vp, ci, ci, ci, vp]
    L.fdr_slab_wiener_dev.argtypes = [vp, vp, vp, ctypes.c_size_t, cf, vp]
    L.fdr_slab_real_dev.argtypes = [vp, vp, ctypes.c_size_t, vp]
    L.fdr_slab_minmax_dev.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp]
    L.fdr_slab_normalize_dev.argtypes = [vp, ci, vp, vp, ci, ci, ci, vp]
    L.fdr_plan_create.argtypes = [ci, ci, ci, ci, cu, ctypes.POINTER(vp)]
    L.fdr_plan_destroy.argtypes = [vp]
    L.fdr_plan_dims.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.fdr_psf_motion.argtypes = [ci, cd, vp]
    L.fdr_psf_motion_dev.argtypes = [ci, ci, cd, vp, vp]
    L.fdr_warp_affine_f32.argtypes = [vp, ci, ci, ci, ctypes.POINTER(cd), vp, ci, ci, ci]
    L.fdr_plan_filter_bytes.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t)]
    L.fdr_plan_export_filter_dev.argtypes = [vp, vp, ctypes.c_size_t, vp]
    L.fdr_plan_import_filter_dev.argtypes = [vp, vp, ctypes.c_size_t, cf, vp]
    L.fdr_set_psf.argtypes = [vp, vp, ci, ci, ci, cf]
    L.fdr_set_psf_dev.argtypes = [vp, vp, ci, ci, ci, cf, vp]
    L.fdr_set_psf_motion.argtypes = [vp, ci, cd, cf, vp]
    L.fdr_wiener_f32.argtypes = [vp, vp, ci, ci, ci, vp, ci, ci]
    L.fdr_wiener_f32_dev.argtypes = [vp, vp, ci, ci, ci, vp, ci, ci, vp]
    L.fdr_wiener_batch_f32_dev.argtypes = [vp, vp, ctypes.c_size_t, ci, ci, ci, ci, vp, ctypes.c_size_t, ci, ci, vp]
    L.fdr_wiener_batch_f32.argtypes = [vp, vp, ctypes.c_size_t, ci, ci, ci, ci, vp, ctypes.c_size_t, ci, ci]
    L.fdr_wiener_batch_ptrs_f32.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, ci, ci, ci]
    L.fdr_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    L.fdr_host_free.argtypes = [vp]
    L.fdr_white_balance_u8.argtypes = [ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, vp, ci]
    L.fdr_white_balance_u8_dev.argtypes = [ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, vp, ci, vp]
    L.fdr_plan_set_concurrency.argtypes = [vp, ci]
    L.fdr_plan_set_batching.argtypes = [vp, ci, ci]
    L.fdr_fft2d_c2c.argtypes = [vp, vp, ci]
    L.fdr_fft2d_c2c_dev.argtypes = [vp, vp, ci, vp]
    L.fdr_fft1d_c2c.argtypes = [vp, ci, ci, ci]
    L.fdr_dft_naive_c2c.argtypes = [vp, ci, ci]
    L.fdr_synth_image_dev.argtypes = [ci, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_size_t, vp, vp]
    L.fdr_plan_profile.argtypes = [vp, ci]
    L.fdr_plan_pass_times.argtypes = [vp, ctypes.POINTER(ci), _f32p, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ci)]
    for name in ("fdr_device_count", "fdr_next_pow2", "fdr_is_pow2", "fdr_plan_create", "fdr_plan_destroy", "fdr_plan_dims",
                 "fdr_psf_motion", "fdr_psf_motion_dev", "fdr_set_psf", "fdr_set_psf_dev", "fdr_set_psf_motion",
                 "fdr_wiener_f32", "fdr_wiener_f32_dev", "fdr_wiener_batch_f32_dev", "fdr_wiener_batch_f32", "fdr_wiener_batch_ptrs_f32", "fdr_host_alloc", "fdr_host_free",
                 "fdr_white_balance_u8", "fdr_white_balance_u8_dev", "fdr_plan_set_concurrency", "fdr_plan_set_batching",
                 "fdr_fft2d_c2c", "fdr_fft2d_c2c_dev", "fdr_fft1d_c2c", "fdr_dft_naive_c2c", "fdr_synth_image_dev", "fdr_plan_profile", "fdr_plan_pass_times",
                 "fdr_optimal_dft_size", "fdr_plan_set_option", "fdr_plan_phase_times", "fdr_batch_run",
                 "fdr_slab_pad_dev", "fdr_slab_rows_fft_dev", "fdr_slab_pack_dev", "fdr_slab_transpose_dev", "fdr_slab_wiener_dev",
                 "fdr_slab_real_dev", "fdr_slab_minmax_dev", "fdr_slab_normalize_dev", "fdr_warp_affine_f32",
                 "fdr_plan_filter_bytes", "fdr_plan_export_filter_dev", "fdr_plan_import_filter_dev"):
        getattr(L, name).restype = ci
    return L


lib = _load()

EXPORTED_SYMBOLS = (
    "fdr_version", "fdr_last_error", "fdr_device_count", "fdr_next_pow2", "fdr_is_pow2", "fdr_plan_create",
    "fdr_plan_destroy", "fdr_plan_dims", "fdr_psf_motion", "fdr_psf_motion_dev", "fdr_set_psf", "fdr_set_psf_dev",
    "fdr_set_psf_motion", "fdr_wiener_f32", "fdr_wiener_f32_dev", "fdr_wiener_batch_f32_dev", "fdr_wiener_batch_f32", "fdr_wiener_batch_ptrs_f32",
    "fdr_host_alloc", "fdr_host_free", "fdr_white_balance_u8", "fdr_white_balance_u8_dev", "fdr_plan_set_concurrency",
    "fdr_plan_set_batching", "fdr_fft2d_c2c",
    "fdr_fft2d_c2c_dev", "fdr_fft1d_c2c", "fdr_dft_naive_c2c", "fdr_synth_image_dev", "fdr_plan_profile",
    "fdr_plan_pass_times", "fdr_optimal_dft_size", "fdr_plan_set_option", "fdr_plan_phase_times", "fdr_batch_run",
    "fdr_slab_pad_dev", "fdr_slab_rows_fft_dev", "fdr_slab_pack_dev", "fdr_slab_transpose_dev", "fdr_slab_wiener_dev",
    "fdr_slab_real_dev", "fdr_slab_minmax_dev", "fdr_slab_normalize_dev", "fdr_warp_affine_f32",
    "fdr_plan_filter_bytes", "fdr_plan_export_filter_dev", "fdr_plan_import_filter_dev")


def _check(rc):
    if rc != 0:
        raise FdrError(rc, lib.fdr_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _stream(stream):
    return ctypes.c_void_p(int(stream) if stream else 0)


# ---- utils.hpp mirrors ---------------------------------------------------------------------
def csrc_fingerprint():
    """sha256 (first 16 hex digits) over names and contents of csrc/*: which kernel sources a build, a profile or a bench
    line belongs to.  tools/collect_profiles.sh records it beside the PMC passes, tools/summarize_profiles.py writes it into
    profiles/traffic.json, and bench.py reports `roofline.traffic_stale` when the tree it runs from has another one."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".h")):
            h.update(name.encode() + b"\0")
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def traffic_entry(tj, key, pass_name, images, fingerprint, P, spectrum):
    """Looks a pass up in profiles/traffic.json (PMC bytes per launch, tools/summarize_profiles.py) for bench.py's roofline.
    Returns a dict: `traffic` (bytes per launch of `images` images, or None), `stale` (the counters were collected on other
    kernel sources than `fingerprint`, or the entry carries no fingerprint: the bytes are then NOT reported), `note`."""
    out = {"traffic": None, "stale": False, "note": None, "kernel": None}
    ent = tj.get(key, {}).get(pass_name)
    if ent is None:
        out["note"] = "no PMC entry for %s / %s" % (key, pass_name)
        return out
    if not isinstance(ent, dict):
        ent = {"per_launch": float(ent), "images": 1}
    out["kernel"] = ent.get("kernel")
    if ent.get("csrc") != fingerprint:
        out["stale"] = True
        out["note"] = "counters collected on csrc %s, this tree is %s" % (ent.get("csrc"), fingerprint)
        return out
    if ent["images"] == images:
        out["traffic"] = ent["per_launch"]
    else:  # a launch of another size was profiled: scale the per-image part, keep W once (pass B')
        w_once = (4 if spectrum == "half" else 8) * P if pass_name.startswith("B' cols") else 0
        out["traffic"] = (ent["per_launch"] - w_once) * images / ent["images"] + w_once
        out["note"] = "scaled from a %d-image launch" % ent["images"]
    return out


def nextPowerOfTwo(n):
    """utils.hpp:27-31"""
    return lib.fdr_next_pow2(int(n))


getNextPowerOf2 = nextPowerOfTwo  # utils.hpp:33-37


def isPowerOfTwo(n):
    """utils.hpp:50-52"""
    return bool(lib.fdr_is_pow2(int(n)))


def getOptimalDFTSize(n):
    """cv::getOptimalDFTSize as fft/fft_serial.cpp:153-154 uses it: smallest 2^a 3^b 5^c >= n"""
    return lib.fdr_optimal_dft_size(int(n))


def motionBlurKernel(size, angle):
    """utils.hpp:15-24 -- generated by the device PSF kernel, returned as a size x size float32 array."""
    out = np.empty((int(size), int(size)), dtype=np.float32)
    _check(lib.fdr_psf_motion(int(size), float(angle), _ptr(out)))
    return out


def warpAffine(src, M, dsize):
    """cv::warpAffine(src, dst, M, dsize) with its defaults (bilinear, constant 0 border) on the device: src float32
    [rows, cols], M the 2 x 3 forward matrix, dsize = (width, height) as cv::Size."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    m = (ctypes.c_double * 6)(*[float(v) for v in np.asarray(M, dtype=np.float64).reshape(6)])
    out = np.empty((int(dsize[1]), int(dsize[0])), dtype=np.float32)
    _check(lib.fdr_warp_affine_f32(_ptr(src), src.shape[0], src.shape[1], src.shape[1], m, _ptr(out), out.shape[0], out.shape[1], out.shape[1]))
    return out


def getRotationMatrix2D(center, angle, scale):
    """cv::getRotationMatrix2D (utils.hpp:20): center = (x, y) rounded to float as cv::Point2f, angle in degrees."""
    a = float(angle) * np.pi / 180.0
    alpha, beta = np.cos(a) * scale, np.sin(a) * scale
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)


def autoPadToPowerOfTwo(src):
    """utils.hpp:40-47 (host helper; the device path pads on load instead)."""
    src = np.asarray(src, dtype=np.float32)
    out = np.zeros((nextPowerOfTwo(src.shape[0]), nextPowerOfTwo(src.shape[1])), dtype=np.float32)
    out[:src.shape[0], :src.shape[1]] = src
    return out


# ---- plan -----------------------------------------------------------------------------------
class Plan:
    """One (device, M, N, mode) workspace: twiddles, spectrum buffers, filter spectrum."""

    def __init__(self, M, N, mode=MODE_PARITY, device=0, flags=0):
        h = ctypes.c_void_p()
        _check(lib.fdr_plan_create(int(device), int(M), int(N), int(mode), int(flags), ctypes.byref(h)))
        self._h = h
        self.M, self.N, self.mode, self.device = int(M), int(N), int(mode), int(device)

    def close(self):
        if getattr(self, "_h", None):
            lib.fdr_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # PSF
    def set_psf(self, psf, K=0.01):
        psf = np.ascontiguousarray(psf, dtype=np.float32)
        _check(lib.fdr_set_psf(self._h, _ptr(psf), psf.shape[0], psf.shape[1], psf.shape[1], ctypes.c_float(K)))

    def set_psf_dev(self, d_ptr, prows, pcols, pstride, K=0.01, stream=None):
        _check(lib.fdr_set_psf_dev(self._h, ctypes.c_void_p(int(d_ptr)), prows, pcols, pstride, ctypes.c_float(K),
                                   _stream(stream)))

    def set_psf_motion(self, size, angle, K=0.01, stream=None):
        _check(lib.fdr_set_psf_motion(self._h, int(size), float(angle), ctypes.c_float(K), _stream(stream)))

    # the prepared filter as an opaque block (one rank's PSF spectrum handed to the others: fft/fft_mpi.cpp:334-378)
    def filter_bytes(self):
        n = ctypes.c_size_t(0)
        _check(lib.fdr_plan_filter_bytes(self._h, ctypes.byref(n)))
        return int(n.value)

    def export_filter_dev(self, d_dst, nbytes, stream=None):
        _check(lib.fdr_plan_export_filter_dev(self._h, ctypes.c_void_p(int(d_dst)), int(nbytes), _stream(stream)))

    def import_filter_dev(self, d_src, nbytes, K=0.01, stream=None):
        _check(lib.fdr_plan_import_filter_dev(self._h, ctypes.c_void_p(int(d_src)), int(nbytes), ctypes.c_float(K), _stream(stream)))

    # operator
    def wiener(self, img, norm_area=NORM_PADDED):
        """One channel, host arrays; serial.cpp:34-39 semantics (pad -> restore -> crop)."""
        img = np.ascontiguousarray(img, dtype=np.float32)
        out = np.empty_like(img)
        _check(lib.fdr_wiener_f32(self._h, _ptr(img), img.shape[0], img.shape[1], img.shape[1], _ptr(out), img.shape[1],
                                  int(norm_area)))
        return out

    def wiener_dev(self, d_img, rows, cols, stride, d_out, out_stride, norm_area=NORM_PADDED, stream=None):
        _check(lib.fdr_wiener_f32_dev(self._h, ctypes.c_void_p(int(d_img)), rows, cols, stride, ctypes.c_void_p(int(d_out)),
                                      out_stride, int(norm_area), _stream(stream)))

    def wiener_batch_dev(self, d_imgs, img_pitch, count, rows, cols, stride, d_out, out_pitch, out_stride,
                         norm_area=NORM_PADDED, stream=None):
        _check(lib.fdr_wiener_batch_f32_dev(self._h, ctypes.c_void_p(int(d_imgs)), img_pitch, count, rows, cols, stride,
                                            ctypes.c_void_p(int(d_out)), out_pitch, out_stride, int(norm_area),
                                            _stream(stream)))

    def prepared_batch_dev(self, d_imgs, img_pitch, count, rows, cols, stride, d_out, out_pitch, out_stride,
                           norm_area=NORM_PADDED, stream=None):
        """The same call as wiener_batch_dev with its arguments converted ONCE: returns a zero-argument callable.  For callers
        that repeat one call many times (a single small image per step is a 20 us call: the per-call ctypes conversions of
        twelve arguments are then a measurable part of it)."""
        fn = lib.fdr_wiener_batch_f32_dev
        args = (self._h, ctypes.c_void_p(int(d_imgs)), ctypes.c_size_t(img_pitch), ctypes.c_int(count), ctypes.c_int(rows), ctypes.c_int(cols),
                ctypes.c_int(stride), ctypes.c_void_p(int(d_out)), ctypes.c_size_t(out_pitch), ctypes.c_int(out_stride), ctypes.c_int(int(norm_area)),
                _stream(stream))

        def call():
            rc = fn(*args)
            if rc != 0:
                _check(rc)
        return call

    def wiener_batch(self, imgs, out=None, norm_area=NORM_PADDED):
        """Host arrays [count, rows, cols] in, restored planes out; H2D / compute / D2H of consecutive images
        overlap (pinned arrays from host_alloc() are copied by DMA in place)."""
        imgs = np.ascontiguousarray(imgs, dtype=np.float32)  # (no copy when it already is: pinned arrays stay pinned)
        if out is None:
            out = np.empty_like(imgs)
        cnt, rows, cols = imgs.shape
        _check(lib.fdr_wiener_batch_f32(self._h, _ptr(imgs), rows * cols, cnt, rows, cols, cols, _ptr(out), rows * cols, cols, int(norm_area)))
        return out

    def set_concurrency(self, nstreams):
        """Batched mode: alternate images over `nstreams` private workspaces / internal streams."""
        _check(lib.fdr_plan_set_concurrency(self._h, int(nstreams)))

    def set_batching(self, nstreams, group):
        """Batched mode: `nstreams` internal streams, `group` images per pass-B' launch (fast mode)."""
        _check(lib.fdr_plan_set_batching(self._h, int(nstreams), int(group)))

    # transforms
    def fft2d(self, x, inverse=False):
        """fft_gpu::my_dft2D(Mat&, bool): unscaled, complex64 [M, N]."""
        a = np.ascontiguousarray(x, dtype=np.complex64).copy()
        assert a.shape == (self.M, self.N)
        _check(lib.fdr_fft2d_c2c(self._h, _ptr(a), int(inverse)))
        return a

    def fft2d_dev(self, d_ptr, inverse=False, stream=None):
        _check(lib.fdr_fft2d_c2c_dev(self._h, ctypes.c_void_p(int(d_ptr)), int(inverse), _stream(stream)))

    def set_option(self, option, value):
        _check(lib.fdr_plan_set_option(self._h, int(option), int(value)))

    def phase_times(self, reset=False):
        """The reference Profiler's six buckets (ms) accumulated on this plan: dict alloc/h2d/pre/compute/d2h/post."""
        ms = (ctypes.c_float * len(PHASES))()
        _check(lib.fdr_plan_phase_times(self._h, ms, int(reset)))
        return {k: float(ms[i]) for i, k in enumerate(PHASES)}

    # profiling
    def profile(self, enable=True):
        _check(lib.fdr_plan_profile(self._h, int(enable)))

    def pass_times(self):
        n = ctypes.c_int(0)
        ms = (ctypes.c_float * MAX_PASSES)()
        names = (ctypes.c_char_p * MAX_PASSES)()
        cnt = (ctypes.c_int * MAX_PASSES)()
        _check(lib.fdr_plan_pass_times(self._h, ctypes.byref(n), ms, names, cnt))
        return [(names[i].decode(), float(ms[i]), int(cnt[i])) for i in range(n.value)]


def fft1d(x, inverse=False, mode=MODE_PARITY):
    """fft_gpu::fft_radix2_kernel / transform_row_kernel: unscaled 1-D transform of a host array."""
    a = np.ascontiguousarray(x, dtype=np.complex64).copy()
    _check(lib.fdr_fft1d_c2c(_ptr(a), a.size, int(inverse), int(mode)))
    return a


def dft_naive(x, inverse=False):
    """fft_gpu::dft_naive_kernel"""
    a = np.ascontiguousarray(x, dtype=np.complex64).copy()
    _check(lib.fdr_dft_naive_c2c(_ptr(a), a.size, int(inverse)))
    return a


def synth_image_dev(d_out, count, seed, first_index=0, device=0, stream=None):
    _check(lib.fdr_synth_image_dev(int(device), ctypes.c_uint64(seed), ctypes.c_uint64(first_index), count,
                                   ctypes.c_void_p(int(d_out)), _stream(stream)))


# ---- fft/fft.hpp mirrors (fft_gpu namespace) --------------------------------------------------
def applyWhiteBalance_u8(orig_bgr, restored_bgr, device=0):
    """The drivers' colour epilogue (serial.cpp:43-54 / gpu.cpp:123-137 with utils.hpp:55-71) on the device:
    two lists of three float planes (B, G, R in [0,1]) -> uint8 [rows, cols, 3] BGR."""
    o = [np.ascontiguousarray(c, dtype=np.float32) for c in orig_bgr]
    r = [np.ascontiguousarray(c, dtype=np.float32) for c in restored_bgr]
    rows, cols = o[0].shape
    out = np.empty((rows, cols, 3), dtype=np.uint8)
    po = (ctypes.c_void_p * 3)(*[c.ctypes.data for c in o])
    pr = (ctypes.c_void_p * 3)(*[c.ctypes.data for c in r])
    _check(lib.fdr_white_balance_u8(int(device), po, pr, rows, cols, cols, ctypes.c_void_p(out.ctypes.data), 3 * cols))
    return out


def host_alloc(shape, dtype=np.float32):
    """numpy array in pinned host memory (fdr_host_alloc; the reference's cudaMallocHost buffers,
    fft/fft_gpu.cu:306-308).  Freed when the array (and every view of it) is garbage collected."""
    import weakref
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = ctypes.c_void_p()
    _check(lib.fdr_host_alloc(n, ctypes.byref(p)))
    buf = (ctypes.c_char * n).from_address(p.value)
    weakref.finalize(buf, lib.fdr_host_free, ctypes.c_void_p(p.value))
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def wienerDeblur_myfft(img, psf, K, mode=MODE_PARITY, device=0, norm_area=NORM_PADDED):
    """One channel the way the DRIVERS call the operator: pad to powers of two (serial.cpp:36 / fft_gpu.cu:287-288, on the
    device), restore, normalise (default: over the padded area, serial.cpp:34-39), crop."""
    img = np.asarray(img, dtype=np.float32)
    with Plan(nextPowerOfTwo(img.shape[0]), nextPowerOfTwo(img.shape[1]), mode, device) as p:
        p.set_psf(psf, K)
        return p.wiener(img, norm_area)


def wienerDeblur_myfft_unpadded(img, psf, K, mode=MODE_PARITY, device=0):
    """fft_serial::wienerDeblur_myfft called DIRECTLY on a channel of any size (fft/fft_serial.cpp:141-261): pad to
    getOptimalDFTSize (2^a 3^b 5^c; a non-power-of-two dimension is transformed by the naive DFT, :100-101), restore,
    crop to the input size, normalise over the cropped plane (:243-246)."""
    img = np.asarray(img, dtype=np.float32)
    M, N = getOptimalDFTSize(img.shape[0]), getOptimalDFTSize(img.shape[1])
    flags = 0 if (isPowerOfTwo(M) and isPowerOfTwo(N)) else FLAG_ANY_SIZE
    with Plan(M, N, mode, device, flags=flags) as p:
        p.set_psf(psf, K)
        return p.wiener(img, NORM_CROPPED)


def batch_run(devices, M, N, count, rows=None, cols=None, mode=MODE_FAST, flags=0, psf=None, psf_size=50, psf_angle=30.0, K=0.01,
              imgs=None, seed=0x5EED0005, steps=1, warmup=0, nstreams=0, group=0, norm_area=NORM_PADDED, bcast_filter=False):
    """fdr_batch_run: `count` independent images sharded over `devices` (ordinals, may repeat) by the reference's
    calculate_distribution rule, one host thread + plan per entry.  imgs = float32 [count, rows, cols] host array (results
    returned) or None for the device-resident synthetic run.  Returns (stats dict, outputs or None)."""
    rows = rows or M
    cols = cols or N
    d = BatchDesc()
    devs = (ctypes.c_int * len(devices))(*[int(x) for x in devices])
    d.n_devices, d.devices = len(devices), devs
    d.M, d.N, d.mode, d.flags = int(M), int(N), int(mode), int(flags)
    keep = []
    if psf is not None:
        psf = np.ascontiguousarray(psf, dtype=np.float32)
        keep.append(psf)
        d.psf_host, d.psf_rows, d.psf_cols, d.psf_stride = psf.ctypes.data, psf.shape[0], psf.shape[1], psf.shape[1]
    d.psf_size, d.psf_angle_deg, d.K = int(psf_size), float(psf_angle), float(K)
    d.count, d.rows, d.cols, d.stride, d.out_stride = int(count), int(rows), int(cols), int(cols), int(cols)
    outs = None
    if imgs is not None:
        imgs = np.ascontiguousarray(imgs, dtype=np.float32)
        assert imgs.shape == (count, rows, cols)
        outs = np.empty_like(imgs)
        pin = (ctypes.c_void_p * max(count, 1))(*[imgs[i].ctypes.data for i in range(count)])
        pout = (ctypes.c_void_p * max(count, 1))(*[outs[i].ctypes.data for i in range(count)])
        keep += [imgs, pin, pout]
        d.imgs_host, d.outs_host = pin, pout
    d.synth_seed, d.steps, d.warmup = int(seed), int(steps), int(warmup)
    d.nstreams, d.group, d.norm_area = int(nstreams), int(group), int(norm_area)
    d.bcast_filter = int(bcast_filter)  # False / True / 2 (RCCL even for one device entry)
    st = BatchStats()
    _check(lib.fdr_batch_run(ctypes.byref(d), ctypes.byref(st)))
    n = st.n_devices
    stats = {"first": list(st.first[:n]), "images": list(st.images[:n]), "elapsed_ms": list(st.elapsed_ms[:n]),
             "checksum": list(st.checksum[:n]), "status": list(st.status[:n]), "wall_ms": st.wall_ms,
             "images_done": st.images_done, "mpixels_per_s": st.mpixels_per_s,
             "filter_path": ("local", "rccl_broadcast", "peer_copy")[st.filter_path] if 0 <= st.filter_path <= 2 else st.filter_path}
    return stats, outs


def wienerDeblur_RGB_optimized(channels, psf, K, mode=MODE_PARITY, device=0, norm_area=NORM_PADDED):
    """fft_gpu::wienerDeblur_RGB_optimized (fft/fft_gpu.cu:279-394): replaces every element of
    `channels` (unpadded float32 planes of one size) in place with its restored [0,1] plane.
    One plan and one PSF spectrum serve all channels."""
    if not channels:
        return
    r, c = np.asarray(channels[0]).shape
    with Plan(nextPowerOfTwo(r), nextPowerOfTwo(c), mode, device) as p:
        p.set_psf(psf, K)
        for i in range(len(channels)):
            channels[i] = p.wiener(channels[i], norm_area)


def wienerDeblur_RGB_naive(channels, psf, K, mode=MODE_PARITY, device=0, norm_area=NORM_PADDED):
    """fft_gpu::wienerDeblur_RGB_naive (fft/fft_gpu.cu:400-512): same results, but every channel
    builds and frees its own plan and PSF spectrum, as the reference's allocation-in-loop variant."""
    for i in range(len(channels)):
        channels[i] = wienerDeblur_myfft(channels[i], psf, K, mode, device, norm_area)
