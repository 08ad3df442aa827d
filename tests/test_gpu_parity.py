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


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft1d_parity_bit_exact(fdr, oracle, n, inverse):
    rng = np.random.default_rng(n + inverse)
    x = _rand_c(rng, n)
    got = fdr.fft1d(x, inverse, fdr.MODE_PARITY)
    _assert_same(got, oracle.fft_radix2(x, inverse), "fft1d n=%d inv=%d" % (n, inverse))


@pytest.mark.parametrize("n", [8, 64, 1024, 4096, 8192, 16384, 32768])
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
    # non power of two -> naive DFT (fft_serial.cpp:100-101); twiddles from the host's cosf / sinf: bit parity
    _assert_same(fdr.fft1d(x, False, fdr.MODE_PARITY), oracle.dft_naive(x, False), "naive DFT fwd n=%d" % n)
    _assert_same(fdr.dft_naive(x, True), oracle.dft_naive(x, True), "naive DFT inv n=%d" % n)


@pytest.mark.parametrize("shape", [(8, 8), (32, 64), (64, 32), (8, 1024), (256, 256), (1024, 8), (512, 2048), (2048, 1024)])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft2d_parity_bit_exact(fdr, oracle, shape, inverse):
    rng = np.random.default_rng(shape[0] * 7 + shape[1] + inverse)
    x = _rand_c(rng, *shape)
    with fdr.Plan(shape[0], shape[1], fdr.MODE_PARITY) as p:
        got = p.fft2d(x, inverse)
    _assert_same(got, oracle.dft2d(x, inverse), "fft2d %s" % (shape,))


@pytest.mark.parametrize("shape", [(8, 8), (64, 32), (256, 1024), (2048, 512)])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft2d_fast_close_to_float64(fdr, shape, inverse):
    """fft_gpu::my_dft2D in the fast mode (double-generated twiddles, FMA butterflies): unscaled, against numpy's
    float64 transform; rel-L2 well below the serial path's own error (1e-6 .. 1e-5)."""
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    with fdr.Plan(shape[0], shape[1], fdr.MODE_FAST) as p:
        got = p.fft2d(x, inverse)
    want = np.fft.ifft2(x.astype(np.complex128)) * (shape[0] * shape[1]) if inverse else np.fft.fft2(x.astype(np.complex128))
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel < 2e-6, rel


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


@pytest.mark.parametrize("shape", [(64, 64), (256, 256), (100, 200), (782 // 4, 1920 // 4), (1024, 1024), (782, 1920)])
def test_wiener_parity_bit_exact(fdr, oracle, shape):
    psf = oracle.motion_blur_kernel(50 if min(shape) >= 64 else 15, 30.0)
    img = _image(oracle, shape[0], shape[1], 0x5EED0002)
    ref = oracle.serial_channel(img, psf, 0.01)
    got = fdr.wienerDeblur_myfft(img, psf, 0.01, mode=fdr.MODE_PARITY)
    _assert_same(got, ref, "wiener parity %s" % (shape,))
    assert got.min() >= 0.0 and got.max() <= 1.0


@pytest.mark.parametrize("shape", [(64, 64), (256, 256), (100, 200), (1024, 1024), (330, 640), (782, 1920)])
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
        with pytest.raises(fdr.FdrError):
            p.wiener_batch(np.zeros((2, 65, 64), np.float32))  # host batch: the same shape check
        assert p.wiener_batch(np.zeros((0, 64, 64), np.float32)).shape == (0, 64, 64)  # empty batch is a no-op
        for bad in ((0, 1), (1, 0), (1, 9), (3, 6), (17, 1)):  # streams * group <= 16, group <= 8
            with pytest.raises(fdr.FdrError):
                p.set_batching(*bad)
        p.set_batching(2, 4)
    with pytest.raises(fdr.FdrError):  # colour epilogue: three planes of one shape, non-empty
        fdr.applyWhiteBalance_u8([np.zeros((0, 4), np.float32)] * 3, [np.zeros((0, 4), np.float32)] * 3)


# ------------------------------------------------------------------------------------------------
# committed vectors (tests/golden/oracle_vectors.npz, generated by tests/golden/make_golden.py)
# ------------------------------------------------------------------------------------------------
import os as _os

_GOLD = np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))


def test_committed_vectors_on_gpu(fdr):
    for n in (2, 4, 8, 16, 64, 1024):
        x = _GOLD["fft1d_in_%d" % n]
        _assert_same(fdr.fft1d(x, False, fdr.MODE_PARITY), _GOLD["fft1d_fwd_%d" % n], "golden fft1d fwd %d" % n)
        _assert_same(fdr.fft1d(x, True, fdr.MODE_PARITY), _GOLD["fft1d_inv_%d" % n], "golden fft1d inv %d" % n)
    for key, shape in (("8x8", (8, 8)), ("32x64", (32, 64))):
        with fdr.Plan(shape[0], shape[1], fdr.MODE_PARITY) as p:
            _assert_same(p.fft2d(_GOLD["fft2d_in_" + key], False), _GOLD["fft2d_fwd_" + key], "golden fft2d fwd " + key)
            _assert_same(p.fft2d(_GOLD["fft2d_in_" + key], True), _GOLD["fft2d_inv_" + key], "golden fft2d inv " + key)
    for size, ang in ((50, 30.0), (40, 45.0), (15, 10.0)):
        _assert_same(fdr.motionBlurKernel(size, ang), _GOLD["psf_%d_%d" % (size, int(ang))], "golden psf")
    psf = _GOLD["psf_15_10"] * 0 + fdr.motionBlurKernel(15, 30.0)
    from oracle import oracle as o
    for shape in ((64, 64), (100, 200)):
        img = o.synth_image(0x5EED0002, 0, shape[0] * shape[1]).reshape(shape)
        got = fdr.wienerDeblur_myfft(img, psf, 0.01, mode=fdr.MODE_PARITY)
        _assert_same(got, _GOLD["wiener_serial_%dx%d" % shape], "golden wiener %s" % (shape,))


# ------------------------------------------------------------------------------------------------
# fast-mode variants, batched mode, large sizes
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("flags_name", ["FLAG_SIMPLE_PATH", "FLAG_FULL_SPECTRUM"])
@pytest.mark.parametrize("shape", [(200, 300), (1024, 1024)])
def test_fast_variants_within_tolerance(fdr, oracle, flags_name, shape):
    psf = oracle.motion_blur_kernel(50, 30.0)
    img = _image(oracle, shape[0], shape[1], 0x5EED0002)
    ref = oracle.serial_channel(img, psf, 0.01)
    with fdr.Plan(fdr.nextPowerOfTwo(shape[0]), fdr.nextPowerOfTwo(shape[1]), fdr.MODE_FAST, flags=getattr(fdr, flags_name)) as p:
        p.set_psf(psf, 0.01)
        got = p.wiener(img)
    assert np.abs(got - ref).max() <= TOL and np.linalg.norm(got - ref) / np.linalg.norm(ref) <= TOL


@pytest.mark.parametrize("shape", [(32, 32), (30, 50), (100, 200), (256, 256), (500, 1000), (1024, 1024), (2000, 2048), (600, 4096), (37, 8100)])
def test_two_sweep_normalisation_equals_raw_plane_passes(fdr, oracle, shape):
    """FDR_OPT_TWO_SWEEP_NORM: passes C1 (inverse rows, min/max only) + C2 (inverse rows again, normalised and cropped on
    store) must give the bits of passes C' (raw plane) + E (normalise): both settings, both normalisation areas, single
    images and grouped launches with a tail."""
    import torch
    rows, cols = shape
    psf = oracle.motion_blur_kernel(15, 30.0)
    B = 5
    host = np.stack([_image(oracle, rows, cols, 900 + i) for i in range(B)])
    M, N = fdr.nextPowerOfTwo(rows), fdr.nextPowerOfTwo(cols)
    d_in = torch.from_numpy(host).cuda()
    s = torch.cuda.current_stream().cuda_stream
    with fdr.Plan(M, N, fdr.MODE_FAST) as p, fdr.Plan(M, N, fdr.MODE_FAST) as q:
        p.set_option(fdr.OPT_TWO_SWEEP_NORM, 1)
        q.set_option(fdr.OPT_TWO_SWEEP_NORM, 0)
        p.set_psf(psf, 0.01)
        q.set_psf(psf, 0.01)
        for area in (fdr.NORM_PADDED, fdr.NORM_CROPPED):
            _assert_same(p.wiener(host[0], norm_area=area), q.wiener(host[0], norm_area=area), "two-sweep vs raw plane, area %d" % area)
            for nstreams, group in ((1, 4), (2, 2), (1, 3), (1, 5)):
                outs = []
                for plan in (p, q):
                    d_o = torch.full_like(d_in, -1.0)
                    plan.set_batching(nstreams, group)
                    plan.wiener_batch_dev(d_in.data_ptr(), rows * cols, B, rows, cols, cols, d_o.data_ptr(), rows * cols, cols, area, stream=s)
                    torch.cuda.synchronize()
                    outs.append(d_o.cpu().numpy())
                _assert_same(outs[0], outs[1], "two-sweep vs raw plane, batch %dx%d, area %d" % (nstreams, group, area))
        ref = oracle.serial_channel(host[0], psf, 0.01)
        assert float(np.max(np.abs(p.wiener(host[0]) - ref))) <= 1e-4


@pytest.mark.parametrize("shape", [(1024, 64), (2000, 100), (4096, 128), (8192, 64), (1024, 1024), (512, 64), (300, 2048), (64, 8192)])
def test_tall_and_wide_shapes_within_tolerance(fdr, oracle, shape):
    """Pass B' with 16 values per thread (columns of 1024 points and more) and the persistent radix-8 kernel (shorter
    columns), rows of 64 .. 8192 points, against the oracle; the full-spectrum variant must agree as well."""
    psf = oracle.motion_blur_kernel(15, 30.0)
    img = _image(oracle, shape[0], shape[1], 0x5EED0002)
    ref = oracle.serial_channel(img, psf, 0.01)
    M, N = fdr.nextPowerOfTwo(shape[0]), fdr.nextPowerOfTwo(shape[1])
    with fdr.Plan(M, N, fdr.MODE_FAST) as p, fdr.Plan(M, N, fdr.MODE_FAST, flags=fdr.FLAG_FULL_SPECTRUM) as q:
        p.set_psf(psf, 0.01)
        q.set_psf(psf, 0.01)
        got, base = p.wiener(img), q.wiener(img)
    assert np.abs(got - ref).max() <= TOL and np.linalg.norm(got - ref) / np.linalg.norm(ref) <= TOL
    assert np.abs(got - base).max() <= TOL


@pytest.mark.parametrize("shape", [(30, 50), (45, 100), (64, 100), (100, 64), (6, 10), (97, 33)])
def test_unpadded_operator_uses_optimal_dft_size_and_naive_dft(fdr, oracle, shape):
    """fft_serial::wienerDeblur_myfft called directly on a channel that is not a power of two (fft/fft_serial.cpp:141-261):
    pad to getOptimalDFTSize (2^a 3^b 5^c), naive DFT along every non-power-of-two dimension (:100-101), crop, normalise
    the cropped plane.  The naive DFT's twiddles are generated on the host with the C library's cosf / sinf (the calls
    the serial path makes), accumulated in its order: bit-identical to the oracle's un-padded entry point."""
    psf = oracle.motion_blur_kernel(5 if min(shape) >= 10 else 3, 30.0)
    img = _image(oracle, shape[0], shape[1], 0xD1F7)
    M, N = fdr.getOptimalDFTSize(shape[0]), fdr.getOptimalDFTSize(shape[1])
    assert (M, N) == (oracle.optimal_dft_size(shape[0]), oracle.optimal_dft_size(shape[1]))
    ref = oracle.wiener(img, psf, 0.01)
    got = fdr.wienerDeblur_myfft_unpadded(img, psf, 0.01)
    _assert_same(got, ref, "un-padded operator %s -> %dx%d" % (shape, M, N))
    # the 2-D transform alone at the padded size
    rng = np.random.default_rng(M * 131 + N)
    x = _rand_c(rng, M, N)
    flags = 0 if (fdr.isPowerOfTwo(M) and fdr.isPowerOfTwo(N)) else fdr.FLAG_ANY_SIZE
    with fdr.Plan(M, N, fdr.MODE_PARITY, flags=flags) as p:
        _assert_same(p.fft2d(x, False), oracle.dft2d(x, False), "fft2d any-size fwd")
        _assert_same(p.fft2d(x, True), oracle.dft2d(x, True), "fft2d any-size inv")


def test_optimal_dft_size(fdr, oracle):
    for n in list(range(1, 200)) + [782, 800, 1000, 1920, 2049, 4095, 4097, 7999, 8000, 8100]:
        assert fdr.getOptimalDFTSize(n) == oracle.optimal_dft_size(n), n


def test_phase_times_and_batch_run(fdr, oracle):
    """fdr_plan_phase_times (the reference Profiler's buckets, fft/fft_gpu.cu:17-57) and fdr_batch_run (the multi-GPU
    batched mode for C callers; here two workers on device 0: devices = {0, 0})."""
    psf = oracle.motion_blur_kernel(15, 30.0)
    imgs = np.stack([_image(oracle, 100, 200, 900 + i) for i in range(5)])
    with fdr.Plan(128, 256, fdr.MODE_PARITY) as p:
        p.set_psf(psf, 0.01)
        one = np.stack([p.wiener(imgs[i]) for i in range(5)])
        ph = p.phase_times()
        assert ph["alloc"] > 0 and ph["h2d"] > 0 and ph["pre"] > 0 and ph["compute"] > 0 and ph["d2h"] > 0 and ph["post"] == 0
        p.phase_times(reset=True)
        assert p.phase_times()["compute"] == 0
        p.wiener_batch(imgs)
        ph2 = p.phase_times()
        assert ph2["h2d"] > 0 and ph2["compute"] > 0 and ph2["d2h"] > 0
    st, outs = fdr.batch_run([0, 0], 128, 256, 5, rows=100, cols=200, mode=fdr.MODE_PARITY, psf=psf, imgs=imgs)
    assert st["images"] == [3, 2] and st["first"] == [0, 3] and st["status"] == [0, 0] and st["images_done"] == 5
    _assert_same(outs, one, "fdr_batch_run (host images, 2 workers) vs one by one")
    assert abs(sum(st["checksum"]) - float(one.astype(np.float64).sum())) < 1e-6 * one.size
    # device-resident synthetic run (config 5 shape, scaled down): 6 images of 256^2 over 3 workers, 2 timed passes
    st2, _ = fdr.batch_run([0, 0, 0], 256, 256, 6, mode=fdr.MODE_FAST, psf_size=15, psf_angle=30.0, seed=0x5EED0005, steps=2, warmup=1)
    assert st2["images"] == [2, 2, 2] and st2["images_done"] == 12 and st2["mpixels_per_s"] > 0
    ref_sum = 0.0
    with fdr.Plan(256, 256, fdr.MODE_FAST) as q:
        q.set_psf(fdr.motionBlurKernel(15, 30.0), 0.01)
        for i in range(6):
            ref_sum += float(q.wiener(oracle.synth_image(0x5EED0005, i * 65536, 65536).reshape(256, 256)).astype(np.float64).sum())
    assert abs(sum(st2["checksum"]) - ref_sum) < 1e-3, (sum(st2["checksum"]), ref_sum)
    # no warm-up and a single timed pass: the very first batch reads the filter the (asynchronous, generated-on-device)
    # PSF preparation writes -- both are ordered on the worker's own stream
    st3, _ = fdr.batch_run([0, 0, 0], 256, 256, 6, mode=fdr.MODE_FAST, psf_size=15, psf_angle=30.0, seed=0x5EED0005, steps=1, warmup=0)
    assert st3["images_done"] == 6 and abs(sum(st3["checksum"]) - ref_sum) < 1e-3, (sum(st3["checksum"]), ref_sum)
    # two host threads on the reference-shaped path at 8192 points: the 64 KiB dynamic-LDS opt-in of its row kernel is made
    # per launch on the launching thread's device (it used to hide behind a process-wide flag set by the first thread)
    st4, _ = fdr.batch_run([0, 0], 64, 8192, 2, mode=fdr.MODE_PARITY, flags=fdr.FLAG_SIMPLE_PATH, psf_size=15, psf_angle=30.0,
                           seed=0x5EED0006, steps=1, warmup=0, nstreams=1, group=1)
    assert st4["status"] == [0, 0] and st4["images_done"] == 2
    ref4 = 0.0
    with fdr.Plan(64, 8192, fdr.MODE_PARITY) as q:
        q.set_psf(fdr.motionBlurKernel(15, 30.0), 0.01)
        for i in range(2):
            ref4 += float(q.wiener(oracle.synth_image(0x5EED0006, i * 64 * 8192, 64 * 8192).reshape(64, 8192)).astype(np.float64).sum())
    assert abs(sum(st4["checksum"]) - ref4) < 1e-3, (sum(st4["checksum"]), ref4)
    # bcast_filter: worker 0 prepares the filter, the others receive its bytes.  Three workers on ONE device: RCCL refuses a
    # repeated ordinal, so this takes the copy path; a single device entry with bcast_filter = 2 goes through RCCL itself
    # (communicator of one, broadcast to itself): library, symbols and call signature are exercised on the one GPU
    st5, _ = fdr.batch_run([0, 0, 0], 256, 256, 6, mode=fdr.MODE_FAST, psf_size=15, psf_angle=30.0, seed=0x5EED0005, steps=1, warmup=0, bcast_filter=True)
    assert st5["filter_path"] == "peer_copy" and st5["images_done"] == 6 and abs(sum(st5["checksum"]) - ref_sum) < 1e-3, st5
    st6, _ = fdr.batch_run([0], 256, 256, 6, mode=fdr.MODE_FAST, psf_size=15, psf_angle=30.0, seed=0x5EED0005, steps=1, warmup=0, bcast_filter=2)
    assert st6["filter_path"] == "rccl_broadcast" and abs(sum(st6["checksum"]) - ref_sum) < 1e-3, st6
    st7, outs7 = fdr.batch_run([0, 0], 128, 256, 5, rows=100, cols=200, mode=fdr.MODE_PARITY, psf=psf, imgs=imgs, bcast_filter=True)
    assert st7["filter_path"] == "peer_copy"
    _assert_same(outs7, one, "fdr_batch_run with the filter handed from worker 0 (host images, parity mode)")
    assert st2["filter_path"] == "local"
    with pytest.raises(fdr.FdrError):
        fdr.batch_run([0, 99], 128, 256, 2, rows=100, cols=200, psf=psf, imgs=imgs[:2])  # device ordinal out of range
    with pytest.raises(fdr.FdrError):
        fdr.batch_run([], 128, 256, 2, psf=psf)


def test_psf_generated_on_device_into_the_plan(fdr, oracle):
    psf = oracle.motion_blur_kernel(50, 30.0)
    img = _image(oracle, 256, 256, 5)
    with fdr.Plan(256, 256, fdr.MODE_PARITY) as p, fdr.Plan(256, 256, fdr.MODE_PARITY) as q:
        p.set_psf(psf, 0.01)
        q.set_psf_motion(50, 30.0, 0.01)
        _assert_same(p.wiener(img), q.wiener(img), "set_psf vs set_psf_motion")


@pytest.mark.parametrize("mode_name", ["MODE_PARITY", "MODE_FAST"])
def test_batched_multistream_equals_one_by_one(fdr, oracle, mode_name):
    import torch
    mode = getattr(fdr, mode_name)
    rows, cols, B = 100, 200, 5
    psf = oracle.motion_blur_kernel(15, 30.0)
    host = np.stack([_image(oracle, rows, cols, 100 + i) for i in range(B)])
    d_in = torch.from_numpy(host).cuda()
    d_out1 = torch.zeros_like(d_in)
    d_out3 = torch.zeros_like(d_in)
    s = torch.cuda.current_stream().cuda_stream
    with fdr.Plan(128, 256, mode) as p:
        p.set_psf(psf, 0.01)
        p.wiener_batch_dev(d_in.data_ptr(), rows * cols, B, rows, cols, cols, d_out1.data_ptr(), rows * cols, cols, stream=s)
        p.set_concurrency(3)
        p.wiener_batch_dev(d_in.data_ptr(), rows * cols, B, rows, cols, cols, d_out3.data_ptr(), rows * cols, cols, stream=s)
        torch.cuda.synchronize()
        for nstreams, group in ((1, 2), (2, 2), (2, 3), (1, 4), (1, 5), (2, 8)):  # several images per pass-B' launch (fast mode)
            d_g = torch.zeros_like(d_in)
            p.set_batching(nstreams, group)
            p.wiener_batch_dev(d_in.data_ptr(), rows * cols, B, rows, cols, cols, d_g.data_ptr(), rows * cols, cols, stream=s)
            torch.cuda.synchronize()
            assert np.count_nonzero(~(d_g.cpu().numpy() == d_out1.cpu().numpy())) == 0, (nstreams, group)
        one = np.stack([p.wiener(host[i]) for i in range(B)])
    _assert_same(d_out1.cpu().numpy(), one, "batched (1 stream) vs one by one")
    _assert_same(d_out3.cpu().numpy(), one, "batched (3 streams) vs one by one")
    if mode == fdr.MODE_PARITY:
        _assert_same(one[2], oracle.serial_channel(host[2], psf, 0.01), "batched vs oracle")


@pytest.mark.parametrize("shape", [(1000, 40), (1024, 20), (2000, 100)])
def test_grouped_long_column_launches_equal_one_by_one(fdr, oracle, shape):
    """Pass B' on columns of 1024 points and more puts the images of one launch on neighbouring workgroups of one XCD
    (they share the tile's slice of W) when the launch has 2 or 4 images and the tile count is a multiple of 8, and
    falls back to a (tiles, images) grid otherwise: every grouping must give the bits of the one-by-one path."""
    import torch
    rows, cols = shape
    B = 7
    M, N = 1 << (rows - 1).bit_length(), 1 << (cols - 1).bit_length()
    psf = oracle.motion_blur_kernel(15, 30.0)
    host = np.stack([_image(oracle, rows, cols, 300 + i) for i in range(B)])
    d_in = torch.from_numpy(host).cuda()
    s = torch.cuda.current_stream().cuda_stream
    with fdr.Plan(M, N, fdr.MODE_FAST) as p:
        p.set_psf(psf, 0.01)
        one = np.stack([p.wiener(host[i]) for i in range(B)])
        for nstreams, group in ((1, 2), (2, 2), (1, 3), (1, 4), (2, 4), (1, 6), (1, 8)):  # 7 = 2+2+2+1 = 3+3+1 = 4+3 = 6+1 = 7
            d_g = torch.zeros_like(d_in)
            p.set_batching(nstreams, group)
            p.wiener_batch_dev(d_in.data_ptr(), rows * cols, B, rows, cols, cols, d_g.data_ptr(), rows * cols, cols, stream=s)
            torch.cuda.synchronize()
            assert np.count_nonzero(~(d_g.cpu().numpy() == one)) == 0, (nstreams, group)
    ref = oracle.serial_channel(host[5], psf, 0.01)
    assert float(np.max(np.abs(one[5] - ref))) <= 1e-4


@pytest.mark.parametrize("shape,mode_name,flags_name", [((1024, 4096), "MODE_FAST", None), ((600, 4096), "MODE_FAST", None),
                                                        ((1024, 4096), "MODE_FAST", "FLAG_FULL_SPECTRUM"), ((2048, 2048), "MODE_FAST", None),
                                                        ((600, 4096), "MODE_PARITY", None)])
def test_overlapping_streams_equal_one_by_one_every_image(fdr, oracle, shape, mode_name, flags_name):
    """Regression (round 2): pass A wrote the second packed spectrum into the LDS buffer the transform's last exchange was
    still being read from -- a race that only showed when a second stream's kernels shared the CUs (up to half of the
    runs had an image with two wrong rows per affected 4-row group; image 0 never).  Every image of a batch, every
    batching, several repetitions, against the one-by-one result -- bit for bit."""
    import torch
    rows, cols = shape
    B, reps = 12, 5
    M, N = fdr.nextPowerOfTwo(rows), fdr.nextPowerOfTwo(cols)
    host = np.stack([_image(oracle, rows, cols, 700 + i) for i in range(B)])
    d_in = torch.from_numpy(host).cuda()
    d_o = torch.empty_like(d_in)
    s = torch.cuda.current_stream().cuda_stream
    flags = getattr(fdr, flags_name) if flags_name else 0
    mode = getattr(fdr, mode_name)
    sweeps = (0, 1) if (mode == fdr.MODE_FAST and not flags) else (0,)
    for two in sweeps:
        with fdr.Plan(M, N, mode, flags=flags) as p:
            p.set_option(fdr.OPT_TWO_SWEEP_NORM, two)
            p.set_psf_motion(15, 30.0, 0.01)
            one = np.stack([p.wiener(host[i]) for i in range(B)])
            for ns, gr in ((2, 1), (3, 1), (2, 2), (3, 2), (2, 4), (2, 8)):
                if mode == fdr.MODE_PARITY and gr > 1:
                    continue
                p.set_batching(ns, gr)
                for rep in range(reps):
                    d_o.fill_(-1.0)
                    p.wiener_batch_dev(d_in.data_ptr(), rows * cols, B, rows, cols, cols, d_o.data_ptr(), rows * cols, cols, stream=s)
                    torch.cuda.synchronize()
                    o = d_o.cpu().numpy()
                    bad = [i for i in range(B) if np.count_nonzero(o[i] != one[i])]
                    assert not bad, "two_sweep %d, %d streams x %d, repetition %d: images %s differ from the one-by-one result" % (two, ns, gr, rep, bad)


@pytest.mark.parametrize("shape", [(100, 200), (512, 512), (1000, 1024)])
def test_batch_graph_replay_equals_plain_launches(fdr, oracle, shape):
    """FDR_OPT_BATCH_GRAPH: the batch call's launches captured as a hipGraph and replayed -- same bits as plain launches;
    the graph is rebuilt when the arguments change (other output buffer, other count, other batching) and follows a new
    PSF (the filter is read at run time, not captured by value)."""
    import torch
    rows, cols = shape
    B = 9
    M, N = fdr.nextPowerOfTwo(rows), fdr.nextPowerOfTwo(cols)
    host = np.stack([_image(oracle, rows, cols, 500 + i) for i in range(B)])
    d_in = torch.from_numpy(host).cuda()
    s = torch.cuda.current_stream().cuda_stream
    with fdr.Plan(M, N, fdr.MODE_FAST) as p, fdr.Plan(M, N, fdr.MODE_FAST) as q:
        p.set_option(fdr.OPT_BATCH_GRAPH, 1)
        for plan in (p, q):
            plan.set_psf_motion(15, 30.0, 0.01)
            plan.set_batching(2, 4)

        def both(count, batching=None):
            outs = []
            for plan in (p, q):
                if batching:
                    plan.set_batching(*batching)
                d_o = torch.full_like(d_in, -1.0)
                for _ in range(3 if plan is p else 1):  # capture, then two replays into the same buffer
                    plan.wiener_batch_dev(d_in.data_ptr(), rows * cols, count, rows, cols, cols, d_o.data_ptr(), rows * cols, cols, stream=s)
                torch.cuda.synchronize()
                outs.append(d_o.cpu().numpy())
            _assert_same(outs[0], outs[1], "graph replay vs plain launches (count %d, batching %s)" % (count, batching))
            return outs[0]

        first = both(B)
        both(B)            # new output buffer: the graph is rebuilt
        both(5)            # other count
        both(B, (3, 2))    # other batching
        for plan in (p, q):
            plan.set_psf_motion(21, 75.0, 0.02)
        other = both(B, (3, 2))   # same arguments apart from the buffer; new filter contents
        assert np.count_nonzero(other[0] != first[0]) > 0
        ref = oracle.serial_channel(host[8], oracle.motion_blur_kernel(21, 75.0), 0.02)
        assert float(np.max(np.abs(other[8] - ref))) <= 1e-4


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("mode_name", ["MODE_PARITY", "MODE_FAST"])
def test_host_batch_pipeline_equals_one_by_one(fdr, oracle, mode_name, pinned):
    """fdr_wiener_batch_f32: H2D / restore / D2H of consecutive images overlap (3 in flight); pageable arrays go
    through pinned staging, fdr_host_alloc arrays by DMA in place.  Same bits as image-by-image calls."""
    rows, cols, B = 100, 200, 7
    psf = oracle.motion_blur_kernel(15, 30.0)
    src = np.stack([_image(oracle, rows, cols, 300 + i) for i in range(B)])
    if pinned:
        imgs = fdr.host_alloc((B, rows, cols)); imgs[...] = src
        out = fdr.host_alloc((B, rows, cols)); out[...] = -1.0
    else:
        imgs, out = src, np.full_like(src, -1.0)
    with fdr.Plan(128, 256, getattr(fdr, mode_name)) as p:
        p.set_psf(psf, 0.01)
        got = p.wiener_batch(imgs, out)
        one = np.stack([p.wiener(src[i]) for i in range(B)])
    assert got is out
    _assert_same(np.array(got), one, "host batch pipeline vs one by one")


def test_white_balance_epilogue_on_device(fdr):
    """fdr_white_balance_u8 (serial.cpp:43-54 / utils.hpp:55-71 on the device) against a float64 numpy model of
    the same formulae; OpenCV's Lab arithmetic is third party and unpinned: +-1 at 8 bit."""
    rng = np.random.default_rng(7)
    rows, cols = 97, 203
    orig = [rng.random((rows, cols), dtype=np.float32) for _ in range(3)]
    rest = [np.clip(o * 0.8 + 0.15 * rng.random((rows, cols), dtype=np.float32), 0, 1).astype(np.float32) for o in orig]
    got = fdr.applyWhiteBalance_u8(orig, rest)

    def to_lab(b, g, r):
        lin = lambda c: np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
        b, g, r = (lin(np.clip(c.astype(np.float64), 0, 1)) for c in (b, g, r))
        X = (0.412453 * r + 0.357580 * g + 0.180423 * b) / 0.950456
        Y = 0.212671 * r + 0.715160 * g + 0.072169 * b
        Z = (0.019334 * r + 0.119193 * g + 0.950227 * b) / 1.088754
        f = lambda t: np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16.0 / 116.0)
        L = np.where(Y > 0.008856, 116.0 * f(Y) - 16.0, 903.3 * Y)
        return L, 500.0 * (f(X) - f(Y)), 200.0 * (f(Y) - f(Z))

    Lo, _, _ = to_lab(*orig)
    L, a, bb = to_lab(*rest)
    L = np.clip(L * (Lo.mean() / (L.mean() + 1e-6)), 0, 100)
    fy = (L + 16.0) / 116.0
    fx, fz = fy + a / 500.0, fy - bb / 200.0
    finv = lambda t: np.where(t > 0.206893, t ** 3, (t - 16.0 / 116.0) / 7.787)
    Y = np.where(L > 7.9996, fy ** 3, L / 903.3)
    X, Z = finv(fx) * 0.950456, finv(fz) * 1.088754
    r = 3.240479 * X - 1.537150 * Y - 0.498535 * Z
    g = -0.969256 * X + 1.875991 * Y + 0.041556 * Z
    b = 0.055648 * X - 0.204043 * Y + 1.057311 * Z
    comp = lambda c: np.where(c <= 0.0031308, 12.92 * c, 1.055 * np.clip(c, 1e-30, None) ** (1 / 2.4) - 0.055)
    want = np.stack([np.clip(np.rint(comp(np.clip(c, 0, 1)) * 255.0), 0, 255) for c in (b, g, r)], axis=-1)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert got.shape == (rows, cols, 3) and d.max() <= 1 and np.count_nonzero(d) < 0.02 * d.size, (int(d.max()), int(np.count_nonzero(d)))


def test_synth_generator_matches_oracle_bits(fdr, oracle):
    import torch
    d = torch.empty(5000, dtype=torch.float32, device="cuda")
    fdr.synth_image_dev(d.data_ptr(), 5000, 0x5EED0003, first_index=123)
    torch.cuda.synchronize()
    _assert_same(d.cpu().numpy(), oracle.synth_image(0x5EED0003, 123, 5000), "synthetic image generator")


@pytest.mark.parametrize("mode_name", ["MODE_PARITY", "MODE_FAST"])
def test_2048_against_oracle(fdr, oracle, mode_name):
    psf = oracle.motion_blur_kernel(50, 30.0)
    img = _image(oracle, 2048, 2048, 0x5EED0005)
    ref = oracle.serial_channel(img, psf, 0.01)
    got = fdr.wienerDeblur_myfft(img, psf, 0.01, mode=getattr(fdr, mode_name))
    if mode_name == "MODE_PARITY":
        _assert_same(got, ref, "2048^2 parity")
    else:
        assert np.abs(got - ref).max() <= TOL


@pytest.mark.parametrize("S", [4096, 8192])
def test_full_size_properties(fdr, S):
    """BASELINE sizes, size-independent properties of the 2-D transform: (a) inverse(forward(x)) = M N x (both
    transforms unscaled, fft_serial.cpp:67), (b) Parseval."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(S)
    x = torch.rand((S, S, 2), generator=g, device="cuda", dtype=torch.float32) - 0.5
    y = x.clone()
    with fdr.Plan(S, S, fdr.MODE_FAST) as p:
        p.fft2d_dev(y.data_ptr(), False)
        torch.cuda.synchronize()
        e_in = float((x.double() ** 2).sum()); e_out = float((y.double() ** 2).sum())
        assert abs(e_out / (e_in * S * S) - 1.0) < 1e-5
        p.fft2d_dev(y.data_ptr(), True)
        torch.cuda.synchronize()
        err = float(((y / float(S * S) - x).abs()).max())
        assert err < 2e-5, err


@pytest.mark.parametrize("S", [4096, 8192])
def test_benchmarked_sizes_against_oracle(fdr, oracle, S):
    """The sizes BENCH is quoted on, compared with the CPU oracle DIRECTLY (about 2 s of oracle time at 4096^2, 9 s at
    8192^2): config 3 = 4096^2 (seed 0x5EED0003), config 4 = a non-power-of-two 8000 x 8100 input (seed 0x5EED0004)
    padded to 8192^2, normalised over the padded area and cropped (serial.cpp:34-39).  Parity mode: 0 differing values;
    fast mode (the bench headline path): max-abs and rel-L2 <= 1e-4."""
    rows, cols = (S, S) if S == 4096 else (8000, 8100)
    img = oracle.synth_image(0x5EED0003 if S == 4096 else 0x5EED0004, 0, rows * cols).reshape(rows, cols)
    psf = oracle.motion_blur_kernel(50, 30.0)
    ref = oracle.serial_channel(img, psf, 0.01)
    assert ref.shape == (rows, cols)
    with fdr.Plan(S, S, fdr.MODE_PARITY) as p:
        p.set_psf(psf, 0.01)
        got_p = p.wiener(img)
    _assert_same(got_p, ref, "%d^2 parity mode vs oracle" % S)
    del got_p
    with fdr.Plan(S, S, fdr.MODE_FAST) as p:
        p.set_psf(psf, 0.01)
        got_f = p.wiener(img)
    mx = float(np.abs(got_f - ref).max())
    rel = float(np.linalg.norm((got_f - ref).astype(np.float64)) / np.linalg.norm(ref.astype(np.float64)))
    assert mx <= TOL and rel <= TOL, (mx, rel)
    assert got_f.min() >= 0.0 and got_f.max() <= 1.0
    if (rows, cols) == (S, S):
        assert ref.min() == 0.0 and abs(float(ref.max()) - 1.0) < 1e-6  # max*scale+shift rounds to 1 - 1ulp at most


@pytest.mark.parametrize("S,B", [(2048, 8), (1024, 8)])
def test_grouped_batch_at_config5_size_against_oracle(fdr, oracle, S, B):
    """BASELINE config 5's device path at its own image size: B x S^2 (seed 0x5EED0005 at 2048^2) through
    fdr_wiener_batch_f32_dev with bench.py's defaults for this size -- 2 internal streams x 4 images per launch, i.e.
    fft_rows4_*_packed / fft_cols_panel_fused16 / normalize launched with blockIdx.y = image -- must equal the
    one-image-at-a-time results bit for bit, and images 0, 3 and 7 must be within 1e-4 of the CPU oracle."""
    import torch
    P = S * S
    seed = 0x5EED0005 if S == 2048 else 0x5EED0002
    psf = oracle.motion_blur_kernel(50, 30.0)
    d_in = torch.empty((B, S, S), dtype=torch.float32, device="cuda")
    fdr.synth_image_dev(d_in.data_ptr(), B * P, seed)
    d_grp = torch.zeros_like(d_in)
    d_one = torch.zeros_like(d_in)
    s = torch.cuda.current_stream().cuda_stream
    with fdr.Plan(S, S, fdr.MODE_FAST) as p:
        p.set_psf(psf, 0.01)
        for i in range(B):
            p.wiener_dev(d_in[i].data_ptr(), S, S, S, d_one[i].data_ptr(), S, stream=s)
        p.set_batching(2, 4)
        p.wiener_batch_dev(d_in.data_ptr(), P, B, S, S, S, d_grp.data_ptr(), P, S, stream=s)
        torch.cuda.synchronize()
    got = d_grp.cpu().numpy()
    _assert_same(got, d_one.cpu().numpy(), "grouped (2 streams x 4 images) vs one by one at %d^2" % S)
    for i in (0, 3, 7):
        img = oracle.synth_image(seed, i * P, P).reshape(S, S)
        _assert_same(img, d_in[i].cpu().numpy(), "device generator, image %d" % i)
        ref = oracle.serial_channel(img, psf, 0.01)
        mx = float(np.abs(got[i] - ref).max())
        rel = float(np.linalg.norm((got[i] - ref).astype(np.float64)) / np.linalg.norm(ref.astype(np.float64)))
        assert mx <= TOL and rel <= TOL, (i, mx, rel)


def test_one_gpu_shard_of_config5_every_image(fdr, oracle):
    """One GPU's shard of BASELINE config 5 at G = 8 -- 64 x 2048^2 (seed 0x5EED0005) -- through fdr_wiener_batch_f32_dev
    with bench.py's batching for this size (2 streams x 4 images per launch), three times over: EVERY image of every pass
    must carry the bits of the one-image-at-a-time path (compared on the device), must span [0, 1], and images 0, 21, 42
    and 63 must be within 1e-4 of the CPU oracle.  (Round 2's LDS race lived in the images nobody compared.)"""
    import torch
    S, B = 2048, 64
    P = S * S
    psf = oracle.motion_blur_kernel(50, 30.0)
    d_in = torch.empty((B, S, S), dtype=torch.float32, device="cuda")
    fdr.synth_image_dev(d_in.data_ptr(), B * P, 0x5EED0005)
    d_grp = torch.zeros_like(d_in)
    d_one = torch.zeros_like(d_in)
    s = torch.cuda.current_stream().cuda_stream
    with fdr.Plan(S, S, fdr.MODE_FAST) as p:
        p.set_psf(psf, 0.01)
        for i in range(B):
            p.wiener_dev(d_in[i].data_ptr(), S, S, S, d_one[i].data_ptr(), S, stream=s)
        p.set_batching(2, 4)
        for rep in range(3):
            d_grp.zero_()
            p.wiener_batch_dev(d_in.data_ptr(), P, B, S, S, S, d_grp.data_ptr(), P, S, stream=s)
            torch.cuda.synchronize()
            bad = [i for i in range(B) if not bool(torch.equal(d_grp[i], d_one[i]))]
            assert not bad, ("pass %d: images that differ from the one-by-one path" % rep, bad)
    assert bool(((d_grp.amin(dim=(1, 2)) == 0.0) & (d_grp.amax(dim=(1, 2)) > 1.0 - 1e-6)).all().item())
    for i in (0, 21, 42, 63):
        img = oracle.synth_image(0x5EED0005, i * P, P).reshape(S, S)
        ref = oracle.serial_channel(img, psf, 0.01)
        got = d_grp[i].cpu().numpy()
        mx = float(np.abs(got - ref).max())
        rel = float(np.linalg.norm((got - ref).astype(np.float64)) / np.linalg.norm(ref.astype(np.float64)))
        assert mx <= TOL and rel <= TOL, (i, mx, rel)


@pytest.mark.parametrize("shape", [(100, 9000), (9000, 100), (16384, 16), (40, 20000)])
def test_lengths_above_8192_follow_the_serial_path(fdr, oracle, shape):
    """fft_serial::transform_row_inplace takes ANY length (fft/fft_serial.cpp:90-108: radix-2 for every power of two), so a
    9000-pixel-wide picture pads to 16384 and works through ./serial; here the same through the library: 8192-point blocks
    on chip plus radix-2 stages in global memory (csrc/fdr_aux.hip, long_gather_kernel), the reference's sequence rows /
    transpose / rows / transpose.  Parity mode: bit-identical to the oracle, 2-D transform and whole operator; fast mode:
    within 1e-4."""
    rows, cols = shape
    M, N = fdr.nextPowerOfTwo(rows), fdr.nextPowerOfTwo(cols)
    assert max(M, N) > 8192
    rng = np.random.default_rng(rows * 7 + cols)
    # the 2-D transform itself on a small-by-long complex field (both directions)
    Ms, Ns = (8, N) if N > 8192 else (M, 8)
    x = _rand_c(rng, Ms, Ns)
    with fdr.Plan(Ms, Ns, fdr.MODE_PARITY) as p:
        _assert_same(p.fft2d(x, False), oracle.dft2d(x, False), "fft2d %dx%d fwd" % (Ms, Ns))
        _assert_same(p.fft2d(x, True), oracle.dft2d(x, True), "fft2d %dx%d inv" % (Ms, Ns))
    # the operator as the drivers call it: pad to powers of two, restore, crop, normalise over the padded area
    psf = oracle.motion_blur_kernel(15, 30.0)
    img = _image(oracle, rows, cols, 4242 + rows)
    ref = oracle.serial_channel(img, psf, 0.01)
    got_p = fdr.wienerDeblur_myfft(img, psf, 0.01, mode=fdr.MODE_PARITY)
    _assert_same(got_p, ref, "operator %dx%d -> %dx%d, parity mode" % (rows, cols, M, N))
    got_f = fdr.wienerDeblur_myfft(img, psf, 0.01, mode=fdr.MODE_FAST)
    mx = float(np.abs(got_f - ref).max())
    rel = float(np.linalg.norm((got_f - ref).astype(np.float64)) / np.linalg.norm(ref.astype(np.float64)))
    # 16384 points: inside the stated 1e-4.  32768 points: the serial recurrence `w *= wlen` itself has drifted by then
    # (its error grows with the length; measured here 1.9e-4 max-abs between the accurate twiddles and the
    # recurrence), so only the parity mode -- compared for equality above -- can be "within 1e-4 of ./serial" at that length;
    # the fast mode is held to 5e-4 there and the bound is stated in DESIGN.md.
    tol = TOL if max(M, N) <= 16384 else 5e-4
    assert mx <= tol and rel <= TOL, (shape, mx, rel, tol)


def test_config5_partition_over_eight_workers_every_image(fdr, oracle):
    """BASELINE config 5's OWN partition -- 512 x 2048^2 over G = 8, 64 images per worker by calculate_distribution
    (fft/fft_mpi.cpp:89-100) -- through fdr_batch_run with eight device entries.  The test box has one GPU, so all eight
    workers (host threads with their own plans and streams) share device 0; host images go in (the copy path, pipelined
    per worker) and ALL 512 results must carry the bits of the one-image-at-a-time path.  Also the device-resident
    synthetic run of the same partition (what bench / a C caller would time): per-worker image counts, the start line
    (every worker's timed region begins after the slowest one's set-up) and the checksum."""
    import torch
    S, B, G = 2048, 512, 8
    P = S * S
    d_in = torch.empty((B, S, S), dtype=torch.float32, device="cuda")
    fdr.synth_image_dev(d_in.data_ptr(), B * P, 0x5EED0005)
    d_one = torch.empty_like(d_in)
    s = torch.cuda.current_stream().cuda_stream
    with fdr.Plan(S, S, fdr.MODE_FAST) as p:
        p.set_psf_motion(50, 30.0, 0.01, stream=s)
        for i in range(B):
            p.wiener_dev(d_in[i].data_ptr(), S, S, S, d_one[i].data_ptr(), S, stream=s)
    torch.cuda.synchronize()
    imgs = d_in.cpu().numpy()
    del d_in
    st, outs = fdr.batch_run([0] * G, S, S, B, mode=fdr.MODE_FAST, psf_size=50, psf_angle=30.0, imgs=imgs)
    assert st["images"] == [64] * G and st["first"] == [64 * g for g in range(G)] and st["status"] == [0] * G, st
    assert st["images_done"] == B and st["filter_path"] == "local"
    bad = []
    for lo in range(0, B, 64):  # compared on the device, a worker's shard at a time
        got = torch.from_numpy(outs[lo:lo + 64]).cuda()
        bad += [lo + i for i in range(64) if not bool(torch.equal(got[i], d_one[lo + i]))]
        del got
    assert not bad, ("images that differ from the one-by-one path", bad[:16], len(bad))
    ref_sum = [float(d_one[64 * g:64 * (g + 1)].to(torch.float64).sum().item()) for g in range(G)]
    for g in range(G):
        assert abs(st["checksum"][g] - ref_sum[g]) <= 1e-9 * abs(ref_sum[g]), (g, st["checksum"][g], ref_sum[g])
    del imgs, outs
    # the same partition device resident and synthetic, as a C caller of the batched mode runs it
    st2, _ = fdr.batch_run([0] * G, S, S, B, mode=fdr.MODE_FAST, psf_size=50, psf_angle=30.0, seed=0x5EED0005, steps=2, warmup=1)
    assert st2["images"] == [64] * G and st2["status"] == [0] * G and st2["images_done"] == 2 * B and st2["mpixels_per_s"] > 0
    for g in range(G):
        assert abs(st2["checksum"][g] - ref_sum[g]) <= 1e-9 * abs(ref_sum[g]), (g, st2["checksum"][g], ref_sum[g])
    # start line: no worker's timed region is longer than the common wall time (they all start together)
    assert max(st2["elapsed_ms"]) <= st2["wall_ms"] * 1.0001 + 1e-3, st2


def test_filter_block_export_import_between_plans(fdr, oracle):
    """fdr_plan_export_filter_dev / fdr_plan_import_filter_dev (what bench.py --bcast-filter moves with dist.broadcast):
    a plan that IMPORTS another plan's filter block restores the same bits as the plan that built it from the PSF, in both
    modes; size and state errors are reported."""
    import torch
    psf = oracle.motion_blur_kernel(15, 30.0)
    img = _image(oracle, 200, 300, 0x1234)
    for mode in (fdr.MODE_FAST, fdr.MODE_PARITY):
        with fdr.Plan(256, 512, mode) as a, fdr.Plan(256, 512, mode) as b:
            a.set_psf(psf, 0.01)
            n = a.filter_bytes()
            assert n == b.filter_bytes() and n > 0
            blk = torch.empty(n, dtype=torch.uint8, device="cuda")
            with pytest.raises(fdr.FdrError):
                b.export_filter_dev(blk.data_ptr(), n)  # no PSF on b yet
            with pytest.raises(fdr.FdrError):
                b.wiener(img)
            a.export_filter_dev(blk.data_ptr(), n)
            with pytest.raises(fdr.FdrError):
                b.import_filter_dev(blk.data_ptr(), n - 8, 0.01)
            b.import_filter_dev(blk.data_ptr(), n, 0.01)
            torch.cuda.synchronize()
            _assert_same(b.wiener(img), a.wiener(img), "imported filter block, mode %d" % mode)
            if mode == fdr.MODE_PARITY:
                _assert_same(b.wiener(img), oracle.serial_channel(img, psf, 0.01), "imported filter block vs oracle")


def test_rccl_single_rank_smoke():
    """The collectives of the batched mode under torch.distributed's "nccl" backend (= RCCL) on the one GPU of the box: a
    one-rank process group in a child process (tests/_nccl_worker.py)."""
    import json
    import subprocess
    import sys
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    env = dict(_os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + _os.getpid() % 300), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, _os.path.join(root, "tests", "_nccl_worker.py")], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] is True and out["backend"] == "nccl", out


def test_rccl_refusal_falls_back_to_gloo_by_consensus():
    """The batched mode has no data-path collective, so a rendezvous problem must not cost a multi-GPU run: bench.py asked for
    "nccl" with TWO ranks on the ONE GPU of the test box -- which RCCL refuses ("invalid usage") -- must notice it on every
    rank (proof all-reduce on the RCCL group), agree over the gloo group, and finish with a valid line that says so."""
    import json
    import subprocess
    import sys
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    env = dict(_os.environ, MASTER_PORT="29577", FDR_RCCL_PROBE_TIMEOUT_S="60")
    r = subprocess.run([sys.executable, _os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "nccl", "--one-device", "--size", "512",
                        "--batch", "4", "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-psf-recompute", "--no-parity-leg",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["check"]["ranks_ok"] == 2 and d["check"]["images_done"] == d["check"]["images_expected"]
    assert d["config"]["collectives"].startswith("gloo (rccl unavailable"), d["config"]["collectives"]
    assert "RCCL group unavailable" in r.stderr


def test_cat_picture_through_both_clis(fdr, oracle, tmp_path):
    """BASELINE config 1's named input: `./serial input/cat_blurred.png 50 30` (782 x 1920 -> 1024 x 2048).  This is the
    picture whose minimum lies in the PADDING (normalising over the cropped area instead would move the
    result by 1.6e-2), so it catches a normalise-area mistake in the drivers.  Planes of tools/cli/serial and of
    tools/cli/gpu (parity mode) == oracle.serial_channel; tools/cli/gpu in its default fast mode within 1e-4."""
    import subprocess
    from PIL import Image
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", _os.path.join(root, "tools", "cli"), "-s", "serial", "gpu"])
    png = _os.path.join(root, "tests", "golden", "cat_blurred.png")
    rgb = np.asarray(Image.open(png).convert("RGB"), dtype=np.float32) / 255.0
    h, w = rgb.shape[:2]
    assert (h, w) == (782, 1920)
    psf = oracle.motion_blur_kernel(50, 30.0)
    ref = [oracle.serial_channel(np.ascontiguousarray(rgb[:, :, ch]), psf, 0.01) for ch in (2, 1, 0)]  # B, G, R
    # the property that makes this picture the interesting one: the padded-area minimum is below the cropped-area one
    padded = np.zeros((1024, 2048), np.float32); padded[:h, :w] = rgb[:, :, 2]
    _, raw = oracle.wiener(padded, psf, 0.01, want_raw=True)
    assert raw.min() < raw[:h, :w].min()
    runs = (("serial", []), ("gpu", ["--mode", "parity"]), ("gpu", []))
    for exe, extra in runs:
        out_raw = str(tmp_path / ("cat_%s_%d.f32" % (exe, len(extra))))
        r = subprocess.run([_os.path.join(root, "tools", "cli", exe), png, "50", "30", "--raw-out", out_raw] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        planes = np.fromfile(out_raw, dtype=np.float32).reshape(3, h, w)
        for k in range(3):
            if exe == "serial" or extra:
                _assert_same(planes[k], ref[k], "%s %s plane %d vs oracle (cat)" % (exe, extra, k))
            else:
                assert np.abs(planes[k] - ref[k]).max() <= TOL, (exe, k, float(np.abs(planes[k] - ref[k]).max()))


def test_drop_in_cli(fdr, tmp_path):
    """tools/cli/gpu (C++ over the C ABI: include/fft/fft.hpp + include/utils.hpp) on the reference's
    own input picture: same restored planes as the Python binding of the same library, and the
    printed lines of the reference driver (gpu.cpp:104-113)."""
    import subprocess
    from PIL import Image
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", _os.path.join(root, "tools", "cli"), "-s"])
    png = _os.path.join(root, "tests", "golden", "car_blurred.png")
    out_png, out_raw = str(tmp_path / "car.png"), str(tmp_path / "car.f32")
    r = subprocess.run([_os.path.join(root, "tools", "cli", "gpu"), png, "40", "45", "--mode", "parity", "--out", out_png,
                        "--raw-out", out_raw], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Deblurring 3 channels took(gpu[optimize]):" in r.stdout and "Deblurring 3 channels took(gpu):" in r.stdout
    assert r.stdout.count("[Speedup]") == 2 and "=== FAST (Reuse Memory) Profiling (3 Channels) ===" in r.stdout
    assert "Deblurring 3 channels took(serial):" in r.stdout and "=== Accumulated Time ===" in r.stdout
    # the reference Profiler's buckets (fft/fft_gpu.cu:45-56) come from device events: uploads and downloads are real
    import re as _re
    for label in ("2. H2D Copy", "4. GPU Compute", "5. D2H Copy", "3. Pre-process", "1. Allocation"):
        vals = [float(v) for v in _re.findall(_re.escape("[" + label + "]") + r"\s*Time: ([0-9.eE+-]+) ms", r.stdout)]
        assert len(vals) == 3, (label, vals)  # warm-up + optimized + naive
        if label == "1. Allocation":  # the timed _optimized call REUSES a plan of the per-thread cache (here even the warm-up
            assert vals[1] == 0 and vals[2] > 0, vals  # does: the serial leg created the parity-mode plan); _naive never does
        else:
            assert all(v > 0 for v in vals), (label, vals)
    rgb = np.asarray(Image.open(png).convert("RGB"), dtype=np.float32) / 255.0  # (330, 640, 3)
    h, w = rgb.shape[:2]
    planes = np.fromfile(out_raw, dtype=np.float32).reshape(3, h, w)  # B, G, R
    psf = fdr.motionBlurKernel(40, 45.0)
    for k, ch in enumerate((2, 1, 0)):
        want = fdr.wienerDeblur_myfft(np.ascontiguousarray(rgb[:, :, ch]), psf, 0.01, mode=fdr.MODE_PARITY)
        _assert_same(planes[k], want, "CLI plane %d" % k)
    res = np.asarray(Image.open(out_png))
    assert res.shape == (h, w, 3) and res.dtype == np.uint8 and 20 < res.mean() < 235
    # the colour epilogue (Lab white balance, 8 bit) ran on the device; the host restatement of the same OpenCV
    # formulae (--host-epilogue) must agree within 1 at 8 bit (powf / cbrtf differ in the last place)
    out_host = str(tmp_path / "car_host.png")
    r2 = subprocess.run([_os.path.join(root, "tools", "cli", "gpu"), png, "40", "45", "--mode", "parity", "--out", out_host,
                         "--host-epilogue"], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    # the reference's (commented-out) areChannelsEqual check, fast mode against the serial-equivalent parity mode
    r3 = subprocess.run([_os.path.join(root, "tools", "cli", "gpu"), png, "40", "45", "--mode", "fast", "--verify"],
                        capture_output=True, text=True, timeout=300)
    assert r3.returncode == 0 and "[Success] fast mode matches the serial-equivalent parity mode" in r3.stdout, r3.stdout + r3.stderr
    assert "serial and GPU implementations" not in r3.stdout
    d = np.abs(res.astype(np.int16) - np.asarray(Image.open(out_host)).astype(np.int16))
    assert d.max() <= 1 and np.count_nonzero(d) < 0.02 * d.size, (int(d.max()), int(np.count_nonzero(d)))
    # usage / unreadable image behave as the reference driver (return -1 -> exit status 255)
    assert subprocess.run([_os.path.join(root, "tools", "cli", "gpu")], capture_output=True).returncode == 255
    bad = subprocess.run([_os.path.join(root, "tools", "cli", "gpu"), "/nonexistent.png", "40", "45"], capture_output=True, text=True)
    assert bad.returncode == 255 and "Cannot read image" in bad.stdout


def test_serial_style_cli_gives_the_serial_pixels(fdr, oracle, tmp_path):
    """tools/cli/serial (the reference's serial driver contract, BASELINE config 1, on the GPU parity mode): its
    restored planes are bit-identical to the CPU restatement of ./serial wrapped as serial.cpp:34-39 does (pad to
    powers of two, normalise over the padded area, crop), and it prints the reference driver's lines."""
    import subprocess
    from PIL import Image
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", _os.path.join(root, "tools", "cli"), "-s", "serial"])
    png = _os.path.join(root, "tests", "golden", "car_blurred.png")
    out_png, out_raw = str(tmp_path / "car_serial.png"), str(tmp_path / "car_serial.f32")
    r = subprocess.run([_os.path.join(root, "tools", "cli", "serial"), png, "40", "45", "--out", out_png, "--raw-out", out_raw],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Deblurring 3 channels took(serial):" in r.stdout and "Total program time:" in r.stdout
    # the accumulated phase block of fft/fft_serial.cpp:249-258, printed by the third channel's call
    assert r.stdout.count("=== Accumulated Time ===") == 1 and "this round total:" in r.stdout
    import re as _re
    for name in ("Pre-process", "FFT Image", "FFT PSF", "Wiener Filter", "IFFT", "Post-process"):
        m = _re.search(r"Serial: " + name + r" total: ([0-9.eE+-]+) ms", r.stdout)
        assert m, name
        if name != "Wiener Filter":  # fused into the forward column pass here
            assert float(m.group(1)) > 0, name
    assert r.stdout.index("=== Accumulated Time ===") < r.stdout.index("Deblurring 3 channels took(serial):")
    rgb = np.asarray(Image.open(png).convert("RGB"), dtype=np.float32) / 255.0
    h, w = rgb.shape[:2]
    planes = np.fromfile(out_raw, dtype=np.float32).reshape(3, h, w)  # B, G, R
    psf = oracle.motion_blur_kernel(40, 45.0)
    for k, ch in enumerate((2, 1, 0)):
        _assert_same(planes[k], oracle.serial_channel(np.ascontiguousarray(rgb[:, :, ch]), psf, 0.01), "serial CLI plane %d vs oracle" % k)
    res = np.asarray(Image.open(out_png))
    assert res.shape == (h, w, 3) and res.dtype == np.uint8
    assert subprocess.run([_os.path.join(root, "tools", "cli", "serial")], capture_output=True).returncode == 255
    bad = subprocess.run([_os.path.join(root, "tools", "cli", "serial"), "/nonexistent.png", "40", "45"], capture_output=True, text=True)
    assert bad.returncode == 255 and "Cannot read image" in bad.stdout


def test_reference_drivers_built_unchanged_run_on_the_drop_in_surface(fdr, oracle, tmp_path):
    """oracle/_ref/{serial,gpu}_{swap,bind}: the reference's OWN serial.cpp / gpu.cpp, compiled unchanged in the build
    container (oracle/Makefile `ref_mains`; the GPU box has no reference checkout and uses the built files) against
    include/ -- `swap`: with include/utils.hpp + include/fft/fft.hpp; `bind`: with the reference's utils.hpp + fft/fft.hpp and
    the binding file tools/cli/fft_hip.cpp, so motionBlurKernel is the reference's code over cv::warpAffine of
    include/fdr_cv.hpp (the device kernel).  The picture serial.cpp hands to imshow (written to FDR_IMSHOW_DIR by the
    shim) must equal what tools/cli/serial writes for the same command line within 1 at 8 bit (host Lab formulae against
    the device epilogue; the restored planes underneath are the oracle-identical parity-mode ones)."""
    import subprocess
    from PIL import Image
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    ref_dir = _os.path.join(root, "oracle", "_ref")
    absent = [n for n in ("serial_swap", "serial_bind", "gpu_swap", "gpu_bind") if not _os.path.exists(_os.path.join(ref_dir, n))]
    if absent:
        # The binaries are build products of the container that HAS the reference (git-ignored, pushed to the GPU box with
        # the tree).  FDR_REQUIRE_REF_MAINS=1 (set by tools/collect_profiles.sh; for the driver's round-end run too) turns
        # their absence into a failure, so that rows a16 / b's strongest evidence cannot drop out silently; without it the
        # skip reason is printed into the pytest summary (-rs is in pytest.ini's addopts).
        why = "oracle/_ref/{%s} not built -- they need /root/reference at build time (`make -C oracle ref_mains`)" % ",".join(absent)
        if _os.environ.get("FDR_REQUIRE_REF_MAINS") == "1":
            pytest.fail(why + "; FDR_REQUIRE_REF_MAINS=1 requires them")
        pytest.skip(why)
    subprocess.check_call(["make", "-C", _os.path.join(root, "tools", "cli"), "-s", "serial"])
    png = _os.path.join(root, "tests", "golden", "car_blurred.png")
    ours = str(tmp_path / "ours.png")
    r = subprocess.run([_os.path.join(root, "tools", "cli", "serial"), png, "40", "45", "--out", ours], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    want = np.asarray(Image.open(ours)).astype(np.int16)
    for exe in ("serial_swap", "serial_bind"):
        d = tmp_path / exe
        d.mkdir()
        env = dict(_os.environ, FDR_IMSHOW_DIR=str(d))
        r = subprocess.run([_os.path.join(ref_dir, exe), png, "40", "45"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (exe, r.stdout, r.stderr)
        assert "Deblurring 3 channels took(serial):" in r.stdout and "Total program time:" in r.stdout, r.stdout
        if exe == "serial_swap":  # include/fft/fft.hpp prints the accumulated phase block of fft/fft_serial.cpp:249-258
            assert r.stdout.count("=== Accumulated Time ===") == 1
        got = np.asarray(Image.open(str(d / "Deblurred_Color_Image.png"))).astype(np.int16)
        assert got.shape == want.shape
        diff = np.abs(got - want)
        assert diff.max() <= 1 and np.count_nonzero(diff) < 0.02 * diff.size, (exe, int(diff.max()), int(np.count_nonzero(diff)))
        assert subprocess.run([_os.path.join(ref_dir, exe)], capture_output=True).returncode == 255          # serial.cpp:12-15
        bad = subprocess.run([_os.path.join(ref_dir, exe), "/nonexistent.png", "40", "45"], capture_output=True, text=True)
        assert bad.returncode == 255 and "Cannot read image" in bad.stdout                                 # serial.cpp:23
    for exe in ("gpu_swap", "gpu_bind"):
        r = subprocess.run([_os.path.join(ref_dir, exe), png, "40", "45"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (exe, r.stdout, r.stderr)
        for line in ("Deblurring 3 channels took(serial):", "Deblurring 3 channels took(gpu[optimize]):", "Deblurring 3 channels took(gpu):"):
            assert line in r.stdout, (exe, line, r.stdout)                                                  # gpu.cpp:91,104,112
        assert r.stdout.count("[Speedup]") == 2
        assert subprocess.run([_os.path.join(ref_dir, exe)], capture_output=True).returncode == 255


def test_warp_affine_on_device_is_the_psf_generator(fdr, oracle):
    """cv::warpAffine of the drop-in surface (fdr_warp_affine_f32): applied to the line kernel of utils.hpp:17-19 with
    cv::getRotationMatrix2D's matrix (:20) it must give the bits of motionBlurKernel / the oracle; identity and integer
    translation are exact copies with a zero border."""
    for size, angle in ((50, 30.0), (40, 45.0), (15, 10.0), (7, 90.0), (64, 123.4)):
        kernel = np.zeros((size, size), np.float32)
        kernel[size // 2, :] = np.float32(1.0 / size)
        M = fdr.getRotationMatrix2D((size // 2, size // 2), angle, 1.0)
        _assert_same(fdr.warpAffine(kernel, M, (size, size)), oracle.motion_blur_kernel(size, angle), "warpAffine(line kernel) %d/%g" % (size, angle))
    img = _image(oracle, 37, 53, 0x77)
    _assert_same(fdr.warpAffine(img, [[1, 0, 0], [0, 1, 0]], (53, 37)), img, "identity warp")
    sh = fdr.warpAffine(img, [[1, 0, 5], [0, 1, 3]], (60, 45))  # dst(x, y) = src(x - 5, y - 3)
    want = np.zeros((45, 60), np.float32)
    want[3:40, 5:58] = img
    _assert_same(sh, want, "integer translation with zero border")


def test_cpp_shim_surface(fdr, oracle, tmp_path):
    """tools/cli/shim_test.cpp calls every name of the drop-in C++ surface (namespace fft_gpu of fft/fft.hpp:31-45 and
    the utils.hpp helpers) the way a caller of the reference would; its dumps must equal what the Python binding of
    the same library (and, in parity mode, the CPU oracle) gives for the same inputs."""
    import subprocess
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", _os.path.join(root, "tools", "cli"), "-s", "shim_test"])
    r = subprocess.run([_os.path.join(root, "tools", "cli", "shim_test"), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "shim ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    rd = lambda name, *shape: np.fromfile(str(tmp_path / name), dtype=np.float32).reshape(shape)
    cplx = lambda a: a[..., 0::2] + 1j * a[..., 1::2]
    psf, img = rd("psf.f32", 15, 15), rd("img.f32", 100, 200)
    _assert_same(psf, oracle.motion_blur_kernel(15, 30.0), "motionBlurKernel")
    _assert_same(rd("wiener_parity.f32", 100, 200), oracle.serial_channel(img, psf, 0.01), "fft_serial::wienerDeblur_myfft on the padded channel vs oracle")
    # called on the un-padded channel the operator pads to getOptimalDFTSize (100 x 200 stays) and runs the naive DFT
    _assert_same(rd("wiener_unpadded.f32", 100, 200), oracle.wiener(img, psf, 0.01), "wienerDeblur_myfft un-padded (naive DFT) vs oracle")
    _assert_same(rd("wiener_fast.f32", 100, 200), fdr.wienerDeblur_myfft(img, psf, 0.01, mode=fdr.MODE_FAST), "wienerDeblur_myfft (fast)")
    _assert_same(rd("rgb1.f32", 100, 200), fdr.wienerDeblur_myfft(img * np.float32(0.75), psf, 0.01, mode=fdr.MODE_FAST), "wienerDeblur_RGB_* channel 1")
    x = cplx(rd("fft1d_in.f32", 128)).astype(np.complex64)
    _assert_same(cplx(rd("fft1d_fwd.f32", 128)).astype(np.complex64), oracle.fft_radix2(x, False), "fft_radix2_kernel")
    _assert_same(cplx(rd("fft1d_inv.f32", 128)).astype(np.complex64), oracle.fft_radix2(x, True), "transform_row_kernel (inverse)")
    z = cplx(rd("dft_in.f32", 24)).astype(np.complex64)
    got = cplx(rd("dft_fwd.f32", 24))
    assert np.abs(got - np.fft.fft(z.astype(np.complex128))).max() < 1e-5, "dft_naive_kernel"
    c2 = cplx(rd("fft2d_in.f32", 32, 128)).astype(np.complex64)
    _assert_same(cplx(rd("fft2d_fwd.f32", 32, 128)).astype(np.complex64), oracle.dft2d(c2, False), "my_dft2D_forward")
    _assert_same(cplx(rd("fft2d_inv.f32", 32, 128)).astype(np.complex64), oracle.dft2d(c2, True), "my_dft2D(inverse)")
    rt = cplx(rd("fft2d_rt.f32", 32, 128)) / (32 * 64)
    assert np.abs(rt - c2).max() < 1e-5, "my_dft2D round trip"


@pytest.mark.parametrize("shape", [(8, 8), (16, 32), (32, 16), (8, 64), (64, 8), (32, 32), (5, 7), (33, 17), (1, 100), (100, 1)])
@pytest.mark.parametrize("mode_name", ["MODE_PARITY", "MODE_FAST"])
def test_small_and_ragged_shapes(fdr, oracle, shape, mode_name):
    """Edge shapes: below the register-kernel threshold (simple path), exactly at the half-spectrum threshold
    (N = 32), non powers of two, single rows / columns."""
    psf = oracle.motion_blur_kernel(3, 30.0)
    if shape[0] < 3 or shape[1] < 3:
        psf = np.ones((1, 1), np.float32)
    img = _image(oracle, shape[0], shape[1], 0xABC)
    ref = oracle.serial_channel(img, psf, 0.01)
    got = fdr.wienerDeblur_myfft(img, psf, 0.01, mode=getattr(fdr, mode_name))
    if mode_name == "MODE_PARITY":
        _assert_same(got, ref, "small shape %s" % (shape,))
    else:
        assert np.abs(got - ref).max() <= TOL, np.abs(got - ref).max()


def test_constant_image_normalises_to_zero(fdr, oracle):
    """smax - smin <= DBL_EPSILON -> scale 0 (cv::normalize); a 1x1 delta PSF keeps a constant image constant"""
    img = np.zeros((16, 16), np.float32)
    psf = np.ones((1, 1), np.float32)
    for mode in (fdr.MODE_PARITY, fdr.MODE_FAST):
        got = fdr.wienerDeblur_myfft(img, psf, 0.01, mode=mode)
        _assert_same(got, oracle.serial_channel(img, psf, 0.01), "constant image")
        assert np.all(got == 0.0)


@pytest.mark.parametrize("rows,cols,world,extra", [(1024, 2048, 2, []), (782, 1920, 2, []), (100, 200, 3, ["--cropped"]), (64, 64, 1, []),
                                                   (60, 9000, 2, []), (9000, 60, 2, [])])  # (the last two: 16384-point rows / columns, round 4)
def test_single_image_slab_mode_equals_single_gpu(rows, cols, world, extra):
    """SURVEY 8f-3 / fft/fft_mpi.cpp:170-307: ONE image as row slabs over `world` ranks, the 2-D transform as local row
    passes + all-to-all transposes (..._amd/slab.py over the fdr_slab_* kernels).  Rehearsal on one GPU (every rank on
    cuda:0, backend gloo, exchanges staged through the host); the gathered picture must equal the single-GPU parity-mode
    result bit for bit (and therefore the oracle).  Unmeasured on multi-GPU hardware."""
    import json, subprocess, sys
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    port = 29700 + (_os.getpid() + rows) % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), _os.path.join(root, "tests", "_slab_worker.py"), str(rows), str(cols), "50" if min(rows, cols) >= 64 and rows > 100 else "15",
           "--one-device", "--backend", "gloo"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["world"] == world and r["shape"] == [rows, cols] and sum(r["slab_rows"]) == rows
    assert r["mismatch_vs_single_gpu"] == 0 and r["mismatch_vs_oracle"] == 0, r
