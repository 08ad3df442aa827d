// shim_test.cpp -- exercises every name of the drop-in C++ surface (include/fft/fft.hpp: namespace fft_gpu as declared at
// fft/fft.hpp:31-45 of the reference; include/utils.hpp: utils.hpp:9-71) the way a caller of the reference would, and
// dumps the results as raw float32 for tests/test_gpu_parity.py::test_cpp_shim_surface to compare with the Python
// binding of the same library.  usage: shim_test <out-dir>
#include "utils.hpp"
#include "fft/fft.hpp"
#include <cstdio>
#include <string>
#include <thread>

static void dump(const std::string& path, const float* p, size_t n) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(p, sizeof(float), n, f) != n) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
    std::fclose(f);
}
static void dump(const std::string& path, const Mat& m) {
    std::vector<float> v;
    const int cn = m.channels();
    for (int r = 0; r < m.rows; ++r) v.insert(v.end(), m.ptr<float>(r), m.ptr<float>(r) + (size_t)m.cols * cn);
    dump(path, v.data(), v.size());
}
static float lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); }

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: shim_test <out-dir>\n"); return -1; }
    const std::string out = std::string(argv[1]) + "/";
    unsigned seed = 12345u;

    // utils.hpp: nextPowerOfTwo / getNextPowerOf2 / isPowerOfTwo / autoPadToPowerOfTwo / motionBlurKernel / getElapsedMs
    if (nextPowerOfTwo(782) != 1024 || getNextPowerOf2(1920) != 2048 || !isPowerOfTwo(4096) || isPowerOfTwo(782)) return 3;
    auto t0 = high_resolution_clock::now();
    Mat psf = motionBlurKernel(15, 30.0);
    if (getElapsedMs(t0, high_resolution_clock::now()) < 0.0) return 4;
    dump(out + "psf.f32", psf);
    Mat img(100, 200, CV_32F);
    for (int r = 0; r < img.rows; ++r) for (int c = 0; c < img.cols; ++c) img.ptr<float>(r)[c] = lcg(seed);
    dump(out + "img.f32", img);
    Mat padded = autoPadToPowerOfTwo(img);
    if (padded.rows != 128 || padded.cols != 256 || padded.ptr<float>(127)[255] != 0.0f || padded.ptr<float>(99)[199] != img.ptr<float>(99)[199]) return 5;

    // fft_gpu::wienerDeblur_myfft (fft/fft.hpp:44) with the semantics of fft/fft_serial.cpp:141-261.
    // (1) called directly on the un-padded 100 x 200 channel: getOptimalDFTSize leaves 100 = 2^2 5^2 and 200 = 2^3 5^2
    //     alone, both dimensions go through the naive DFT (:100-101), the crop is a no-op, normalised as a whole
    fft_gpu::Options par; par.mode = FDR_MODE_PARITY;
    dump(out + "wiener_unpadded.f32", fft_gpu::wienerDeblur_myfft(img, psf, 0.01f, par));
    // (2) the way serial.cpp:34-39 calls it: pad to powers of two first, crop afterwards -- through the fft_serial names
    {
        Mat restored = fft_serial::wienerDeblur_myfft(padded, psf, 0.01f);
        if (restored.rows != 128 || restored.cols != 256) return 8;
        dump(out + "wiener_parity.f32", restored(Rect(0, 0, img.cols, img.rows)).clone());
    }
    // (3) fast arithmetic through the process-wide default of the reference-signature overload
    fft_gpu::set_mode(FDR_MODE_FAST);
    {
        Mat restored = fft_gpu::wienerDeblur_myfft(padded, psf, 0.01f);
        dump(out + "wiener_fast.f32", restored(Rect(0, 0, img.cols, img.rows)).clone());
    }

    // fft_gpu::wienerDeblur_RGB_optimized / _naive (fft/fft.hpp:32-33): three channels replaced in place
    std::vector<Mat> ch;
    for (int k = 0; k < 3; ++k) { Mat c = img.clone(); for (int r = 0; r < c.rows; ++r) for (int x = 0; x < c.cols; ++x) c.ptr<float>(r)[x] *= 0.5f + 0.25f * k; ch.push_back(c); }
    std::vector<Mat> a = ch, b = ch;
    fft_gpu::wienerDeblur_RGB_optimized(a, psf, 0.01f);
    fft_gpu::wienerDeblur_RGB_naive(b, psf, 0.01f);
    for (int k = 0; k < 3; ++k) {
        if (a[k].rows != 100 || a[k].cols != 200) return 6;
        for (int r = 0; r < 100; ++r) for (int x = 0; x < 200; ++x) if (a[k].ptr<float>(r)[x] != b[k].ptr<float>(r)[x]) return 7;
    }
    dump(out + "rgb1.f32", a[1]);

    // fft_gpu::fft_radix2_kernel / transform_row_kernel / dft_naive_kernel (fft/fft.hpp:35-39): interleaved complex by pointer
    std::vector<float> x(2 * 64), y, z(2 * 12);
    for (auto& v : x) v = lcg(seed) - 0.5f;
    for (auto& v : z) v = lcg(seed) - 0.5f;
    dump(out + "fft1d_in.f32", x.data(), x.size());
    dump(out + "dft_in.f32", z.data(), z.size());
    y = x;
    fft_gpu::fft_radix2_kernel(y.data(), 64, false);
    dump(out + "fft1d_fwd.f32", y.data(), y.size());
    y = x;
    fft_gpu::transform_row_kernel(y.data(), 64, true);
    dump(out + "fft1d_inv.f32", y.data(), y.size());
    fft_gpu::dft_naive_kernel(z.data(), 12, false);
    dump(out + "dft_fwd.f32", z.data(), z.size());

    // fft_gpu::my_dft2D / my_dft2D_forward / my_dft2D_inverse (fft/fft.hpp:40-42) on a CV_32FC2 Mat
    Mat c2(32, 64, CV_32FC2);
    for (int r = 0; r < 32; ++r) for (int i = 0; i < 128; ++i) c2.ptr<float>(r)[i] = lcg(seed) - 0.5f;
    dump(out + "fft2d_in.f32", c2);
    Mat f2 = c2.clone();
    fft_gpu::my_dft2D_forward(f2);
    dump(out + "fft2d_fwd.f32", f2);
    fft_gpu::my_dft2D_inverse(f2);  // unscaled round trip: 32*64 times the input
    dump(out + "fft2d_rt.f32", f2);
    Mat g2 = c2.clone();
    fft_gpu::my_dft2D(g2, true);
    dump(out + "fft2d_inv.f32", g2);
    // the plan cache behind these entry points (include/fft/fft.hpp): bounded, released on demand and at thread exit, and
    // with capacity 0 nothing is retained between calls (the reference's allocate-and-free behaviour, fft/fft_gpu.cu:389-393)
    {
        if (fft_gpu::plan_cache().entries.empty()) return 9;  // the calls above left plans behind
        fft_gpu::release_cached_plans();
        if (!fft_gpu::plan_cache().entries.empty()) return 10;
        Mat r1 = fft_gpu::wienerDeblur_myfft(padded, psf, 0.01f);  // rebuilt from nothing: same pixels
        fft_gpu::set_plan_cache_capacity(0);
        if (!fft_gpu::plan_cache().entries.empty()) return 11;
        Mat r2 = fft_gpu::wienerDeblur_myfft(padded, psf, 0.01f);
        if (!fft_gpu::plan_cache().entries.empty()) return 12;  // nothing kept with capacity 0
        for (int r = 0; r < r1.rows; ++r) for (int x = 0; x < r1.cols; ++x) if (r1.ptr<float>(r)[x] != r2.ptr<float>(r)[x]) return 13;
        fft_gpu::set_plan_cache_capacity(2);
        int rc_thread = 0;
        std::thread th([&] {  // a thread that restores and ends: its cache object's destructor frees its plans
            Mat r3 = fft_gpu::wienerDeblur_myfft(padded, psf, 0.01f);
            for (int r = 0; r < r1.rows; ++r) for (int x = 0; x < r1.cols; ++x) if (r1.ptr<float>(r)[x] != r3.ptr<float>(r)[x]) rc_thread = 14;
            if (fft_gpu::plan_cache().entries.size() != 1) rc_thread = 15;
        });
        th.join();
        if (rc_thread) return rc_thread;
        for (int k = 0; k < 4; ++k) {  // capacity 2: never more than two plans whatever sizes come by
            Mat im(16 << k, 32, CV_32F);
            for (int r = 0; r < im.rows; ++r) for (int c = 0; c < im.cols; ++c) im.ptr<float>(r)[c] = lcg(seed);
            (void)fft_gpu::wienerDeblur_myfft(im, psf, 0.01f);
            if (fft_gpu::plan_cache().entries.size() > 2) return 16;
        }
        fft_gpu::set_plan_cache_capacity(4);
    }
    std::printf("shim ok\n");
    return 0;
}
