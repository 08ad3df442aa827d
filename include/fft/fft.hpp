// fft/fft.hpp -- drop-in counterpart of the reference's fft/fft.hpp for the GPU namespace
// (reference fft/fft.hpp:31-45).  Same names, same argument meaning, same error behaviour
// (print "Error: file:line, msg" and exit(1), fft/fft_gpu.cu:59-66); everything is a thin call into
// the C ABI of libfdr.so (include/fdr.h).  Link with -lfdr.
//
//   fft_gpu::wienerDeblur_RGB_optimized / _naive   fft/fft_gpu.cu:279-394 / :400-512
//   fft_gpu::fft_radix2_kernel / transform_row_kernel / dft_naive_kernel / my_dft2D / wienerDeblur_myfft
//                                                  declared at fft/fft.hpp:35-44, empty or missing in the reference
// Differences that are deliberate: the PSF spectrum is built once per call, not once per
// channel; results follow the SERIAL path's semantics (normalise over the padded area, then crop; no 1/(MN)
// factor before the min-max normalisation); fft_gpu::set_mode / set_norm_area select the alternatives.
#pragma once
#include "../utils.hpp"
#include <iostream>
#include <vector>

namespace fft_gpu {

inline int& mode_ref() { static int m = FDR_MODE_FAST; return m; }
inline int& norm_ref() { static int n = FDR_NORM_PADDED; return n; }
// FDR_MODE_FAST (default) or FDR_MODE_PARITY (bit-identical FFT arithmetic to fft_serial)
inline void set_mode(int mode) { mode_ref() = mode; }
// FDR_NORM_PADDED (default, ./serial semantics) or FDR_NORM_CROPPED (reference ./gpu semantics, fft_gpu.cu:379-381)
inline void set_norm_area(int area) { norm_ref() = area; }

// The reference's Profiler buckets (fft/fft_gpu.cu:17-57), filled from host timers around the C ABI calls.
struct Profiler {
    double t_alloc = 0, t_h2d = 0, t_pre = 0, t_compute = 0, t_d2h = 0, t_post = 0;
    void print(const std::string& title) const {
        std::cout << "=== " << title << " Profiling (3 Channels) ===" << std::endl;
        std::cout << "[1. Allocation]  Time: " << t_alloc << " ms (plan: twiddles + workspaces)" << std::endl;
        std::cout << "[2. H2D Copy]    Time: " << t_h2d << " ms (folded into [4]: the C ABI copies in and out)" << std::endl;
        std::cout << "[3. Pre-process] Time: " << t_pre << " ms (Padding + PSF FFT)" << std::endl;
        std::cout << "[4. GPU Compute] Time: " << t_compute << " ms (H2D + FFT + Filter + IFFT + normalize + D2H)" << std::endl;
        std::cout << "[5. D2H Copy]    Time: " << t_d2h << " ms (folded into [4])" << std::endl;
        std::cout << "[6. Post-process]Time: " << t_post << " ms (Mat wrap)" << std::endl;
        std::cout << "--------------------------------------------" << std::endl;
        std::cout << "Total (Sum)      Time: " << (t_alloc + t_h2d + t_pre + t_compute + t_d2h + t_post) << " ms" << std::endl;
        std::cout << "============================================" << std::endl;
    }
};

inline Mat run_channel(fdr_plan* plan, const Mat& img) {
    Mat src = img.isContinuous() ? img : img.clone();
    Mat out(img.rows, img.cols, CV_32F);
    FDR_CHECK(fdr_wiener_f32(plan, src.ptr<float>(0), img.rows, img.cols, img.cols, out.ptr<float>(0), img.cols, norm_ref()));
    return out;
}

// Version A (fft/fft_gpu.cu:279-394): one plan, one PSF spectrum, all channels; replaces every element of `channels`.
inline void wienerDeblur_RGB_optimized(std::vector<Mat>& channels, const Mat& psf, float K) {
    if (channels.empty()) return;
    Profiler p;
    const int imgRows = channels[0].rows, imgCols = channels[0].cols;
    auto t0 = high_resolution_clock::now();
    fdr_plan* plan = nullptr;
    FDR_CHECK(fdr_plan_create(0, nextPowerOfTwo(imgRows), nextPowerOfTwo(imgCols), mode_ref(), 0, &plan));
    auto t1 = high_resolution_clock::now();
    p.t_alloc = getElapsedMs(t0, t1);
    Mat psfc = psf.isContinuous() ? psf : psf.clone();
    FDR_CHECK(fdr_set_psf(plan, psfc.ptr<float>(0), psf.rows, psf.cols, psf.cols, K));
    auto t2 = high_resolution_clock::now();
    p.t_pre = getElapsedMs(t1, t2);
    {   // all channels through the host batch pipeline: upload, restoration and download of consecutive channels overlap
        // (what the stream + pinned-buffer set-up of fft/fft_gpu.cu:304-350 is after)
        auto a = high_resolution_clock::now();
        bool same = true;
        for (const Mat& c : channels) same = same && c.rows == imgRows && c.cols == imgCols && c.type() == CV_32F;
        if (same) {
            std::vector<Mat> src, out;
            std::vector<const float*> ins;
            std::vector<float*> outs;
            for (const Mat& c : channels) {
                src.push_back(c.isContinuous() ? c : c.clone());
                out.push_back(Mat(imgRows, imgCols, CV_32F));
            }
            for (size_t i = 0; i < channels.size(); ++i) { ins.push_back(src[i].ptr<float>(0)); outs.push_back(out[i].ptr<float>(0)); }
            FDR_CHECK(fdr_wiener_batch_ptrs_f32(plan, ins.data(), outs.data(), (int)channels.size(), imgRows, imgCols, imgCols, imgCols, norm_ref()));
            for (size_t i = 0; i < channels.size(); ++i) channels[i] = out[i];
        } else {
            for (size_t i = 0; i < channels.size(); ++i) channels[i] = run_channel(plan, channels[i]);
        }
        p.t_compute += getElapsedMs(a, high_resolution_clock::now());
    }
    p.print("FAST (Reuse Memory)");
    fdr_plan_destroy(plan);
}

// Version B (fft/fft_gpu.cu:400-512): every channel allocates, builds the PSF spectrum and frees.
inline void wienerDeblur_RGB_naive(std::vector<Mat>& channels, const Mat& psf, float K) {
    Profiler p;
    for (size_t i = 0; i < channels.size(); ++i) {
        auto t0 = high_resolution_clock::now();
        fdr_plan* plan = nullptr;
        FDR_CHECK(fdr_plan_create(0, nextPowerOfTwo(channels[i].rows), nextPowerOfTwo(channels[i].cols), mode_ref(), 0, &plan));
        auto t1 = high_resolution_clock::now();
        Mat psfc = psf.isContinuous() ? psf : psf.clone();
        FDR_CHECK(fdr_set_psf(plan, psfc.ptr<float>(0), psf.rows, psf.cols, psf.cols, K));
        auto t2 = high_resolution_clock::now();
        channels[i] = run_channel(plan, channels[i]);
        auto t3 = high_resolution_clock::now();
        fdr_plan_destroy(plan);
        p.t_alloc += getElapsedMs(t0, t1); p.t_pre += getElapsedMs(t1, t2); p.t_compute += getElapsedMs(t2, t3);
    }
    p.print("SLOW (Naive Allocation)");
}

// fft/fft.hpp:44 -- one channel, pads to powers of two on the device
inline Mat wienerDeblur_myfft(const Mat& img, const Mat& psf, float K) {
    std::vector<Mat> one(1, img);
    fdr_plan* plan = nullptr;
    FDR_CHECK(fdr_plan_create(0, nextPowerOfTwo(img.rows), nextPowerOfTwo(img.cols), mode_ref(), 0, &plan));
    Mat psfc = psf.isContinuous() ? psf : psf.clone();
    FDR_CHECK(fdr_set_psf(plan, psfc.ptr<float>(0), psf.rows, psf.cols, psf.cols, K));
    Mat out = run_channel(plan, img);
    fdr_plan_destroy(plan);
    return out;
}

// fft/fft.hpp:35-39: n interleaved complex values by host pointer, unscaled
inline void fft_radix2_kernel(float* data, int n, bool inverse) { FDR_CHECK(fdr_fft1d_c2c(data, n, inverse ? 1 : 0, FDR_MODE_PARITY)); }
inline void dft_naive_kernel(float* data, int n, bool inverse) { FDR_CHECK(fdr_dft_naive_c2c(data, n, inverse ? 1 : 0)); }
inline void transform_row_kernel(float* rowPtr, int N, bool inverse) { FDR_CHECK(fdr_fft1d_c2c(rowPtr, N, inverse ? 1 : 0, FDR_MODE_PARITY)); }

// fft/fft.hpp:40-42: in-place unscaled 2-D transform of a CV_32FC2 Mat (rows, transpose, rows, transpose)
inline void my_dft2D(Mat& complexMat, bool inverse) {
    if (complexMat.type() != CV_32FC2) { std::fprintf(stderr, "Error: %s:%d, my_dft2D needs CV_32FC2\n", __FILE__, __LINE__); std::exit(1); }
    const int M = complexMat.rows, N = complexMat.cols;
    Mat c = complexMat.isContinuous() ? complexMat : complexMat.clone();
    if (isPowerOfTwo(M) && isPowerOfTwo(N) && M <= 8192 && N <= 8192) {
        fdr_plan* plan = nullptr;
        FDR_CHECK(fdr_plan_create(0, M, N, FDR_MODE_PARITY, 0, &plan));
        FDR_CHECK(fdr_fft2d_c2c(plan, c.ptr<float>(0), inverse ? 1 : 0));
        fdr_plan_destroy(plan);
    } else {  // arbitrary sizes: row by row (naive DFT for non powers of two), as fft_serial.cpp:113-139
        for (int r = 0; r < M; ++r) transform_row_kernel(c.ptr<float>(r), N, inverse);
        std::vector<float> col(2 * (size_t)M);
        for (int x = 0; x < N; ++x) {
            for (int r = 0; r < M; ++r) { col[2 * r] = c.ptr<float>(r)[2 * x]; col[2 * r + 1] = c.ptr<float>(r)[2 * x + 1]; }
            transform_row_kernel(col.data(), M, inverse);
            for (int r = 0; r < M; ++r) { c.ptr<float>(r)[2 * x] = col[2 * r]; c.ptr<float>(r)[2 * x + 1] = col[2 * r + 1]; }
        }
    }
    if (c.data != complexMat.data)
        for (int r = 0; r < M; ++r) std::memcpy(complexMat.ptr<float>(r), c.ptr<float>(r), sizeof(float) * 2 * (size_t)N);
}
inline void my_dft2D_forward(Mat& complexMat) { my_dft2D(complexMat, false); }
inline void my_dft2D_inverse(Mat& complexMat) { my_dft2D(complexMat, true); }

}  // namespace fft_gpu

// fft/fft.hpp:9-18 of the reference: the serial back-end's names.  Here they run on the GPU in the PARITY mode, whose
// FFT arithmetic is bit-identical to fft/fft_serial.cpp (per-stage twiddles replayed from its float recurrence, no FMA;
// tests/test_gpu_parity.py holds the proof against the CPU restatement), so a serial.cpp-style caller gets the pixels
// ./serial would give without a CPU path in this library.
#include <complex>
namespace fft_serial {

inline void fft_radix2_inplace(std::vector<std::complex<float>>& a, bool inverse) {  // fft/fft_serial.cpp:40-68
    if (a.empty()) return;
    FDR_CHECK(fdr_fft1d_c2c(reinterpret_cast<float*>(a.data()), (int)a.size(), inverse ? 1 : 0, FDR_MODE_PARITY));
}
inline void dft_naive_inplace(std::vector<std::complex<float>>& a, bool inverse) {   // fft/fft_serial.cpp:71-87
    if (a.empty()) return;
    FDR_CHECK(fdr_dft_naive_c2c(reinterpret_cast<float*>(a.data()), (int)a.size(), inverse ? 1 : 0));
}
inline void transform_row_inplace(cv::Vec2f* rowPtr, int N, bool inverse) {          // fft/fft_serial.cpp:90-108
    FDR_CHECK(fdr_fft1d_c2c(reinterpret_cast<float*>(rowPtr), N, inverse ? 1 : 0, FDR_MODE_PARITY));
}
inline void my_dft2D(Mat& complexMat, bool inverse) { fft_gpu::my_dft2D(complexMat, inverse); }  // :113-139 (parity plan)
inline void my_dft2D_forward(Mat& complexMat) { my_dft2D(complexMat, false); }
inline void my_dft2D_inverse(Mat& complexMat) { my_dft2D(complexMat, true); }
// fft/fft_serial.cpp:141-261: img is the (already padded) channel; the result has img's size, normalised over all of it
inline Mat wienerDeblur_myfft(const Mat& img, const Mat& psf, float K) {
    const int mode = fft_gpu::mode_ref(), area = fft_gpu::norm_ref();
    fft_gpu::set_mode(FDR_MODE_PARITY);
    fft_gpu::set_norm_area(FDR_NORM_PADDED);
    Mat out = fft_gpu::wienerDeblur_myfft(img, psf, K);
    fft_gpu::set_mode(mode);
    fft_gpu::set_norm_area(area);
    return out;
}

}  // namespace fft_serial
