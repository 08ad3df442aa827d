"""CPU tests of the oracle (oracle/fdr_oracle.c): committed vectors, an independent float64 model,
and the self-consistency KATs recorded in SURVEY.md section 8a.  No GPU.  Parity unpinned: the
reference ships no expected outputs, so nothing here compares against reference-produced data."""
import os

import numpy as np
import pytest

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))


def _same(a, b):
    return np.count_nonzero(~(np.asarray(a) == np.asarray(b))) == 0


@pytest.mark.parametrize("n", [2, 4, 8, 16, 64, 1024])
def test_fft1d_matches_committed_vectors(oracle, n):
    x = GOLD["fft1d_in_%d" % n]
    assert _same(oracle.fft_radix2(x, False), GOLD["fft1d_fwd_%d" % n])
    assert _same(oracle.fft_radix2(x, True), GOLD["fft1d_inv_%d" % n])


def test_other_committed_vectors(oracle):
    assert _same(oracle.dft_naive(GOLD["naive_in_12"], False), GOLD["naive_fwd_12"])
    for key in ("8x8", "32x64"):
        assert _same(oracle.dft2d(GOLD["fft2d_in_" + key], False), GOLD["fft2d_fwd_" + key])
        assert _same(oracle.dft2d(GOLD["fft2d_in_" + key], True), GOLD["fft2d_inv_" + key])
    assert _same(oracle.twiddle_recurrence(16, False), GOLD["twiddle_fwd_16"])
    assert _same(oracle.twiddle_recurrence(16, True), GOLD["twiddle_inv_16"])
    for size, ang in ((50, 30.0), (40, 45.0), (15, 10.0)):
        assert _same(oracle.motion_blur_kernel(size, ang), GOLD["psf_%d_%d" % (size, int(ang))])
    psf = oracle.motion_blur_kernel(15, 30.0)
    for shape in ((64, 64), (100, 200)):
        img = oracle.synth_image(0x5EED0002, 0, shape[0] * shape[1]).reshape(shape)
        assert _same(oracle.serial_channel(img, psf, 0.01), GOLD["wiener_serial_%dx%d" % shape])


@pytest.mark.parametrize("n,tol", [(2, 1e-7), (64, 5e-7), (1024, 2e-6), (4096, 1.2e-5), (8192, 2e-5)])
def test_fft1d_against_float64(oracle, n, tol):
    """fft/fft_serial.cpp:40-68 is an unscaled forward/inverse DFT; its float recurrence twiddles cost
    accuracy as n grows (SURVEY.md F5 measured 1.1e-6 / 7.4e-6 / 1.2e-5 rel-L2 at 1024 / 4096 / 8192 on
    the reference object code: the restatement must land on the same curve)."""
    rng = np.random.default_rng(n)
    x = (rng.random(n) + 1j * rng.random(n)).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.linalg.norm(oracle.fft_radix2(x) - ref) / np.linalg.norm(ref) < tol
    refi = np.fft.ifft(x.astype(np.complex128)) * n  # unscaled inverse
    assert np.linalg.norm(oracle.fft_radix2(x, True) - refi) / np.linalg.norm(refi) < tol


def test_inverse_twiddles_are_conjugates(oracle):
    for n in (8, 256, 4096):
        f, i = oracle.twiddle_recurrence(n, False), oracle.twiddle_recurrence(n, True)
        assert _same(i, np.conj(f))
    # the "trivial" stage-4 twiddle is NOT exactly -i in the reference (fft_serial.cpp:54-55)
    t = oracle.twiddle_recurrence(4, False)
    assert t[2].real == np.float32(np.cos(np.float32(-np.pi / 2))) and t[2].real != 0.0 and t[2].imag == -1.0


def test_dft2d_against_float64(oracle):
    rng = np.random.default_rng(3)
    x = rng.random((64, 128)).astype(np.float32).astype(np.complex64)
    ref = np.fft.fft2(x.astype(np.complex128))
    assert np.linalg.norm(oracle.dft2d(x) - ref) / np.linalg.norm(ref) < 1e-6
    # non power of two falls back to the O(n^2) DFT (fft_serial.cpp:100-101)
    y = rng.random((6, 10)).astype(np.float32).astype(np.complex64)
    ref = np.fft.fft2(y.astype(np.complex128))
    assert np.linalg.norm(oracle.dft2d(y) - ref) / np.linalg.norm(ref) < 1e-5


def _wiener_float64(img, psf, K):
    M, N = img.shape
    G = np.fft.fft2(img.astype(np.float64))
    hp = np.zeros((M, N)); hp[:psf.shape[0], :psf.shape[1]] = psf
    H = np.fft.fft2(hp)
    r = np.real(np.fft.ifft2(G * np.conj(H) / (np.abs(H) ** 2 + K))) * (M * N)  # unscaled inverse, as the reference
    return (r - r.min()) / (r.max() - r.min())


def test_wiener_against_float64_model(oracle):
    psf = oracle.motion_blur_kernel(15, 30.0)
    img = oracle.synth_image(7, 0, 128 * 256).reshape(128, 256)
    got = oracle.wiener(img, psf, 0.01)
    assert np.abs(got - _wiener_float64(img, psf, 0.01)).max() < 5e-6
    assert got.min() == 0.0 and got.max() == 1.0


def test_serial_wrapper_normalises_over_the_padded_area(oracle):
    """serial.cpp:34-39 pads BEFORE the operator, so min/max span the padded area."""
    psf = oracle.motion_blur_kernel(15, 30.0)
    img = oracle.synth_image(9, 0, 100 * 200).reshape(100, 200)
    padded = np.zeros((128, 256), np.float32); padded[:100, :200] = img
    full = oracle.wiener(padded, psf, 0.01)
    assert _same(oracle.serial_channel(img, psf, 0.01), full[:100, :200])


@pytest.mark.parametrize("size,angle,total,peak,nnz,rows,cols", [
    (50, 0.0, 1.0, 0.02, 50, (25, 25), (0, 49)),
    (50, 30.0, 1.001094, 0.02, 101, (12, 38), (3, 46)),
    (40, 45.0, 1.104150, 0.025, 87, (6, 35), (5, 34)),
])
def test_psf_kats(oracle, size, angle, total, peak, nnz, rows, cols):
    """SURVEY.md section 8a row 7 self-consistency KATs of motionBlurKernel (utils.hpp:15-24)."""
    p = oracle.motion_blur_kernel(size, angle)
    assert abs(p.sum(dtype=np.float64) - total) < 2e-6
    assert abs(float(p.max()) - peak) < 1e-7
    assert int((p != 0).sum()) == nnz
    r = np.nonzero(p.any(1))[0]; c = np.nonzero(p.any(0))[0]
    assert (r[0], r[-1]) == rows and (c[0], c[-1]) == cols


def test_integer_helpers(oracle):
    assert [oracle.next_pow2(n) for n in (1, 2, 3, 782, 1920, 4096, 4097)] == [1, 2, 4, 1024, 2048, 4096, 8192]
    # cv::getOptimalDFTSize: identity on powers of two, 2^a 3^b 5^c otherwise
    assert [oracle.optimal_dft_size(n) for n in (1, 7, 64, 97, 782, 1000)] == [1, 8, 64, 100, 800, 1000]


def test_normalize_semantics(oracle):
    a = np.array([3.0, -1.0, 0.5, 7.0], np.float32)
    n = oracle.normalize_minmax(a)
    assert n.min() == 0.0 and abs(n.max() - 1.0) < 1e-7
    flat = oracle.normalize_minmax(np.full(5, 2.5, np.float32))
    assert np.all(flat == 0.0)  # smax - smin <= DBL_EPSILON -> scale 0, shift 0


def test_synth_image_is_counter_based(oracle):
    a = oracle.synth_image(0x5EED0003, 0, 1000)
    b = oracle.synth_image(0x5EED0003, 400, 100)
    assert _same(a[400:500], b) and a.min() >= 0.0 and a.max() < 1.0
