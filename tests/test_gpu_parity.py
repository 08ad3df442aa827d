"""Parity of the HIP path (through the C ABI of libfdr.so) against the CPU oracle.

Tolerances (BASELINE.json north_star: "output within 1e-4 relative error of ./serial"):
  * FDR_MODE_PARITY: the FFT arithmetic is bit-identical to the oracle (0 mismatching values,
    comparing with ==, which treats +0 and -0 as equal); the whole operator is bit-identical too
    because the Wiener quotient and the normalisation follow the oracle's operation order with
    IEEE sqrt / divide.
  * FDR_MODE_FAST: rel-L2 and max-abs <= 1e-4 on the [0,1] output (measured ~1e-6..3e-5).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _rand_c(rng, *shape):
    return (rng.random(shape, dtype=np.float32) - 0.5 + 1j * (rng.random(shape, dtype=np.float32) - 0.5)).astype(np.complex64)


def _assert_same(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    bad = np.count_nonzero(~(a == b))
    assert bad == 0, "%s: %d of %d values differ (max abs %g)" % (what, bad, a.size, np.abs(a - b).max())


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft1d_parity_bit_exact(fdr, oracle, n, inverse):
    rng = np.random.default_rng(n + inverse)
    x = _rand_c(rng, n)
    got = fdr.fft1d(x, inverse, fdr.MODE_PARITY)
    _assert_same(got, oracle.fft_radix2(x, inverse), "fft1d n=%d inv=%d" % (n, inverse))


@pytest.mark.parametrize("n", [8, 64, 1024, 4096, 8192])
def test_fft1d_fast_close(fdr, oracle, n):
    rng = np.random.default_rng(n)
    x = _rand_c(rng, n)
    ref = np.fft.fft(x.astype(np.complex128))
    got = fdr.fft1d(x, False, fdr.MODE_FAST)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 5e-7
    # and it is closer to the truth than the serial recurrence is
    ser = oracle.fft_radix2(x, False)
    assert np.linalg.norm(got - ref) <= np.linalg.norm(ser - ref) * 1.01 + 1e-12


@pytest.mark.parametrize("n", [3, 5, 12, 50, 100])
def test_dft_naive(fdr, oracle, n):
    rng = np.random.default_rng(n)
    x = _rand_c(rng, n)
    got = fdr.fft1d(x, False, fdr.MODE_PARITY)  # non power of two -> naive DFT (fft_serial.cpp:100-101)
    ref = oracle.dft_naive(x, False)
    # device cosf/sinf differ from glibc's by <= 2 ulp: tolerance, not bit parity
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("shape", [(8, 8), (32, 64), (64, 32), (8, 1024), (256, 256), (1024, 8), (512, 2048), (2048, 1024)])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft2d_parity_bit_exact(fdr, oracle, shape, inverse):
    rng = np.random.default_rng(shape[0] * 7 + shape[1] + inverse)
    x = _rand_c(rng, *shape)
    with fdr.Plan(shape[0], shape[1], fdr.MODE_PARITY) as p:
        got = p.fft2d(x, inverse)
    _assert_same(got, oracle.dft2d(x, inverse), "fft2d %s" % (shape,))


@pytest.mark.parametrize("shape", [(1, 1), (1, 8), (2, 2), (4, 16), (16, 4), (2, 1024)])
def test_fft2d_small_dims_simple_path(fdr, oracle, shape):
    rng = np.random.default_rng(11)
    x = _rand_c(rng, *shape)
    with fdr.Plan(shape[0], shape[1], fdr.MODE_PARITY) as p:
        got = p.fft2d(x, False)
    _assert_same(got, oracle.dft2d(x, False), "fft2d small %s" % (shape,))


@pytest.mark.parametrize("shape", [(64, 64), (256, 256), (128, 512)])
def test_simple_path_equals_fast_kernels(fdr, shape):
    """The reference-shaped kernels (row FFT, transpose, row FFT, transpose) and the register/LDS
    kernels must agree bit for bit in parity mode: same butterfly DAG, different storage."""
    rng = np.random.default_rng(5)
    x = _rand_c(rng, *shape)
    with fdr.Plan(shape[0], shape[1], fdr.MODE_PARITY) as p, \
         fdr.Plan(shape[0], shape[1], fdr.MODE_PARITY, flags=fdr.FLAG_SIMPLE_PATH) as q:
        _assert_same(p.fft2d(x), q.fft2d(x), "simple vs fast kernels")


@pytest.mark.parametrize("size,angle", [(50, 0.0), (50, 30.0), (40, 45.0), (15, 10.0), (7, 90.0), (64, 123.4)])
def test_psf_bit_exact(fdr, oracle, size, angle):
    _assert_same(fdr.motionBlurKernel(size, angle), oracle.motion_blur_kernel(size, angle), "motionBlurKernel")


def _image(oracle, rows, cols, seed):
    return oracle.synth_image(seed, 0, rows * cols).reshape(rows, cols)


@pytest.mark.parametrize("shape", [(64, 64), (256, 256), (100, 200), (782 // 4, 1920 // 4), (1024, 1024)])
def test_wiener_parity_bit_exact(fdr, oracle, shape):
    psf = oracle.motion_blur_kernel(50 if min(shape) >= 64 else 15, 30.0)
    img = _image(oracle, shape[0], shape[1], 0x5EED0002)
    ref = oracle.serial_channel(img, psf, 0.01)
    got = fdr.wienerDeblur_myfft(img, psf, 0.01, mode=fdr.MODE_PARITY)
    _assert_same(got, ref, "wiener parity %s" % (shape,))
    assert got.min() >= 0.0 and got.max() <= 1.0


@pytest.mark.parametrize("shape", [(64, 64), (256, 256), (100, 200), (1024, 1024), (330, 640)])
def test_wiener_fast_within_tolerance(fdr, oracle, shape):
    psf = oracle.motion_blur_kernel(50 if min(shape) >= 64 else 15, 30.0)
    img = _image(oracle, shape[0], shape[1], 0x5EED0002)
    ref = oracle.serial_channel(img, psf, 0.01)
    got = fdr.wienerDeblur_myfft(img, psf, 0.01, mode=fdr.MODE_FAST)
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    mx = np.abs(got - ref).max()
    assert rel <= TOL and mx <= TOL, (rel, mx)


def test_wiener_cropped_norm_matches_reference_gpu_semantics(fdr, oracle):
    """FDR_NORM_CROPPED: min/max over the cropped area only (fft/fft_gpu.cu:379-381)."""
    psf = oracle.motion_blur_kernel(15, 30.0)
    img = _image(oracle, 100, 200, 77)
    M, N = 128, 256
    padded = np.zeros((M, N), np.float32); padded[:100, :200] = img
    _, raw = oracle.wiener(padded, psf, 0.01, want_raw=True)
    ref = oracle.normalize_minmax(raw[:100, :200])
    got = fdr.wienerDeblur_myfft(img, psf, 0.01, mode=fdr.MODE_PARITY, norm_area=fdr.NORM_CROPPED)
    _assert_same(got, ref, "cropped-area normalisation")


def test_rgb_entry_points(fdr, oracle):
    psf = oracle.motion_blur_kernel(15, 30.0)
    chans = [_image(oracle, 60, 90, s) for s in (1, 2, 3)]
    ref = [oracle.serial_channel(c, psf, 0.01) for c in chans]
    a = [c.copy() for c in chans]; b = [c.copy() for c in chans]
    fdr.wienerDeblur_RGB_optimized(a, psf, 0.01)
    fdr.wienerDeblur_RGB_naive(b, psf, 0.01)
    for i in range(3):
        _assert_same(a[i], ref[i], "RGB_optimized ch%d" % i)
        _assert_same(b[i], ref[i], "RGB_naive ch%d" % i)


def test_errors(fdr):
    with pytest.raises(fdr.FdrError) as e:
        fdr.Plan(100, 64)
    assert e.value.code == -2
    with pytest.raises(fdr.FdrError):
        fdr.Plan(0, 64)
    with fdr.Plan(64, 64) as p:
        with pytest.raises(fdr.FdrError) as e2:
            p.wiener(np.zeros((64, 64), np.float32))  # no PSF yet
        assert e2.value.code == -4
        with pytest.raises(fdr.FdrError):
            p.set_psf(np.zeros((65, 3), np.float32))  # PSF taller than the plan
        p.set_psf(np.ones((3, 3), np.float32) / 9)
        with pytest.raises(fdr.FdrError):
            p.wiener(np.zeros((65, 64), np.float32))  # image larger than the plan
