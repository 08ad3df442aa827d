"""ctypes binding of oracle/libfdr_oracle.so -- the CPU restatement of the reference's
serial path (fft/fft_serial.cpp, utils.hpp, serial.cpp:34-39 of the reference).

TEST INFRASTRUCTURE ONLY (see the header of fdr_oracle.c): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package.
Parity unpinned: the reference holds no golden vectors and cannot be built here.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfdr_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile the restatement with the recipe in oracle/Makefile (gcc -O2, no FMA)."""
    src = os.path.join(_HERE, "fdr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.fdr_oracle_fft_radix2.argtypes = [_f32p, ctypes.c_int, ctypes.c_int]
        L.fdr_oracle_dft_naive.argtypes = [_f32p, ctypes.c_int, ctypes.c_int]
        L.fdr_oracle_transform_row.argtypes = [_f32p, ctypes.c_int, ctypes.c_int]
        L.fdr_oracle_dft2d.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.fdr_oracle_twiddle_recurrence.argtypes = [ctypes.c_int, ctypes.c_int, _f32p]
        L.fdr_oracle_wiener.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_float, _f32p, _f32p, _f32p]
        L.fdr_oracle_wiener.restype = ctypes.c_int
        L.fdr_oracle_serial_channel.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_float, _f32p]
        L.fdr_oracle_serial_channel.restype = ctypes.c_int
        L.fdr_oracle_motion_blur_kernel.argtypes = [ctypes.c_int, ctypes.c_double, _f32p]
        L.fdr_oracle_normalize_minmax.argtypes = [_f32p, ctypes.c_size_t, _f32p, _f32p]
        L.fdr_oracle_minmax_scale_shift.argtypes = [ctypes.c_double, ctypes.c_double, _f32p, _f32p]
        L.fdr_oracle_synth_image.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_size_t, _f32p]
        L.fdr_oracle_next_pow2.argtypes = [ctypes.c_int]
        L.fdr_oracle_next_pow2.restype = ctypes.c_int
        L.fdr_oracle_optimal_dft_size.argtypes = [ctypes.c_int]
        L.fdr_oracle_optimal_dft_size.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_f32p)


def _c2f(x):
    """complex64 array -> contiguous float32 interleaved copy"""
    return np.ascontiguousarray(np.asarray(x, dtype=np.complex64)).view(np.float32).copy()


def fft_radix2(x, inverse=False):
    """fft_serial::fft_radix2_inplace (fft/fft_serial.cpp:40-68); x complex64, len pow2."""
    a = _c2f(x)
    lib().fdr_oracle_fft_radix2(_p(a), a.size // 2, int(inverse))
    return a.view(np.complex64)


def dft_naive(x, inverse=False):
    a = _c2f(x)
    lib().fdr_oracle_dft_naive(_p(a), a.size // 2, int(inverse))
    return a.view(np.complex64)


def transform_row(x, inverse=False):
    a = _c2f(x)
    lib().fdr_oracle_transform_row(_p(a), a.size // 2, int(inverse))
    return a.view(np.complex64)


def dft2d(x, inverse=False):
    """fft_serial::my_dft2D (fft/fft_serial.cpp:113-139); x complex64 [M,N]."""
    x = np.asarray(x, dtype=np.complex64)
    M, N = x.shape
    a = _c2f(x)
    lib().fdr_oracle_dft2d(_p(a), M, N, int(inverse))
    return a.view(np.complex64).reshape(M, N)


def twiddle_recurrence(n, inverse=False):
    out = np.zeros(2 * max(n - 1, 0), dtype=np.float32)
    lib().fdr_oracle_twiddle_recurrence(n, int(inverse), _p(out))
    return out.view(np.complex64)


def wiener(img, psf, K=0.01, want_spectrum=False, want_raw=False):
    """fft_serial::wienerDeblur_myfft (fft/fft_serial.cpp:141-261)."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    psf = np.ascontiguousarray(psf, dtype=np.float32)
    rows, cols = img.shape
    M, N = lib().fdr_oracle_optimal_dft_size(rows), lib().fdr_oracle_optimal_dft_size(cols)
    out = np.empty((rows, cols), dtype=np.float32)
    spec = np.empty(2 * M * N, dtype=np.float32) if want_spectrum else None
    raw = np.empty(M * N, dtype=np.float32) if want_raw else None
    rc = lib().fdr_oracle_wiener(_p(img), rows, cols, _p(psf), psf.shape[0], psf.shape[1], np.float32(K), _p(out),
                                 _p(spec) if want_spectrum else None, _p(raw) if want_raw else None)
    if rc != 0:
        raise ValueError("fdr_oracle_wiener rc=%d" % rc)
    res = [out]
    if want_spectrum:
        res.append(spec.view(np.complex64).reshape(M, N))
    if want_raw:
        res.append(raw.reshape(M, N))
    return res[0] if len(res) == 1 else tuple(res)


def serial_channel(img, psf, K=0.01):
    """serial.cpp:34-39: autoPadToPowerOfTwo -> wienerDeblur_myfft -> crop."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    psf = np.ascontiguousarray(psf, dtype=np.float32)
    out = np.empty_like(img)
    rc = lib().fdr_oracle_serial_channel(_p(img), img.shape[0], img.shape[1], _p(psf), psf.shape[0], psf.shape[1],
                                         np.float32(K), _p(out))
    if rc != 0:
        raise ValueError("fdr_oracle_serial_channel rc=%d" % rc)
    return out


def motion_blur_kernel(size, angle):
    """utils.hpp:15-24 motionBlurKernel."""
    out = np.empty((size, size), dtype=np.float32)
    lib().fdr_oracle_motion_blur_kernel(int(size), float(angle), _p(out))
    return out


def normalize_minmax(a):
    a = np.ascontiguousarray(a, dtype=np.float32).copy()
    lib().fdr_oracle_normalize_minmax(_p(a), a.size, None, None)
    return a


def minmax_scale_shift(smin, smax):
    s = np.zeros(2, dtype=np.float32)
    lib().fdr_oracle_minmax_scale_shift(float(smin), float(smax), _p(s[0:1]), _p(s[1:2]))
    return float(s[0]), float(s[1])


def synth_image(seed, first_index, count):
    out = np.empty(count, dtype=np.float32)
    lib().fdr_oracle_synth_image(seed, first_index, count, _p(out))
    return out


def next_pow2(n):
    return lib().fdr_oracle_next_pow2(int(n))


def optimal_dft_size(n):
    return lib().fdr_oracle_optimal_dft_size(int(n))
