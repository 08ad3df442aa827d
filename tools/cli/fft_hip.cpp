// fft/fft_hip.cpp -- the binding a maintainer of the reference adds in place of fft/fft_gpu.cu:
// namespace fft_gpu (declared at fft/fft.hpp:31-45 of the reference) and namespace fft_serial (fft/fft.hpp:9-18) defined
// over the C ABI of libfdr.so (include/fdr.h).  It is compiled against the REFERENCE's own headers, unchanged:
//     g++ -std=c++17 -O2 -I<reference> -I<this repo>/include <reference>/gpu.cpp tools/cli/fft_hip.cpp -lfdr -lz
// (`-I<reference>` first, so "fft/fft.hpp" / "utils.hpp" are the reference's; <opencv2/opencv.hpp> is OpenCV's where it
// is installed, else this repository's include/opencv2/opencv.hpp).  oracle/Makefile builds the reference's serial.cpp
// and gpu.cpp this way into oracle/_ref/ when /root/reference is present; tests/ run them on the GPU box.
// There is no CPU code path: fft_serial:: runs on the GPU in the parity mode (bit-identical FFT arithmetic).
#include "fft/fft.hpp"  // the reference's header
#include "utils.hpp"    // the reference's helpers (nextPowerOfTwo, utils.hpp:27-31)
#include <fdr.h>
#include <cstdio>
#include <cstdlib>

// the reference's CHECK_CUDA convention, fft/fft_gpu.cu:59-66
#define FDR_CHECK(call)                                                                       \
    do {                                                                                      \
        if ((call) != FDR_OK) {                                                               \
            std::fprintf(stderr, "Error: %s:%d, %s\n", __FILE__, __LINE__, fdr_last_error()); \
            std::exit(1);                                                                     \
        }                                                                                     \
    } while (0)

namespace {

Mat restore(fdr_plan* plan, const Mat& ch, int norm_area) {
    Mat src = ch.isContinuous() ? ch : ch.clone();  // as fft/fft_gpu.cu:347-348
    Mat out(ch.rows, ch.cols, CV_32F);
    FDR_CHECK(fdr_wiener_f32(plan, src.ptr<float>(), ch.rows, ch.cols, ch.cols, out.ptr<float>(), ch.cols, norm_area));
    return out;
}

// the operator as fft/fft_serial.cpp:141-261 defines it: pad to getOptimalDFTSize (:153-154; a non-power-of-two dimension
// goes through the naive DFT, :100-101), restore, crop to the input, normalise the cropped plane (:243-246)
Mat operator_as_serial(const Mat& img, const Mat& psf, float K, int mode) {
    const int M = fdr_optimal_dft_size(img.rows), N = fdr_optimal_dft_size(img.cols);
    const unsigned flags = (fdr_is_pow2(M) && fdr_is_pow2(N)) ? 0u : FDR_FLAG_ANY_SIZE;
    fdr_plan* plan = nullptr;
    FDR_CHECK(fdr_plan_create(0, M, N, mode, flags, &plan));
    Mat psfc = psf.isContinuous() ? psf : psf.clone();
    FDR_CHECK(fdr_set_psf(plan, psfc.ptr<float>(), psf.rows, psf.cols, psf.cols, K));
    Mat out = restore(plan, img, FDR_NORM_CROPPED);
    fdr_plan_destroy(plan);
    return out;
}

}  // namespace

namespace fft_gpu {

void wienerDeblur_RGB_optimized(vector<Mat>& channels, const Mat& psf, float K) {  // fft/fft_gpu.cu:279-394
    if (channels.empty()) return;
    fdr_plan* plan = nullptr;
    FDR_CHECK(fdr_plan_create(0, nextPowerOfTwo(channels[0].rows), nextPowerOfTwo(channels[0].cols), FDR_MODE_FAST, 0, &plan));  // replaces :304-322
    Mat psfc = psf.isContinuous() ? psf : psf.clone();
    FDR_CHECK(fdr_set_psf(plan, psfc.ptr<float>(), psf.rows, psf.cols, psf.cols, K));
    for (size_t i = 0; i < channels.size(); ++i) channels[i] = restore(plan, channels[i], FDR_NORM_PADDED);  // :325-385, ./serial semantics
    fdr_plan_destroy(plan);  // replaces :389-393
}

void wienerDeblur_RGB_naive(vector<Mat>& channels, const Mat& psf, float K) {  // fft/fft_gpu.cu:400-512
    for (size_t i = 0; i < channels.size(); ++i) {
        vector<Mat> one(1, channels[i]);
        wienerDeblur_RGB_optimized(one, psf, K);  // allocation inside the loop
        channels[i] = one[0];
    }
}

void fft_radix2_kernel(float* d, int n, bool inv) { FDR_CHECK(fdr_fft1d_c2c(d, n, inv ? 1 : 0, FDR_MODE_PARITY)); }     // fft/fft.hpp:35
void dft_naive_kernel(float* d, int n, bool inv) { FDR_CHECK(fdr_dft_naive_c2c(d, n, inv ? 1 : 0)); }                   // fft/fft.hpp:37
void transform_row_kernel(float* d, int n, bool inv) { FDR_CHECK(fdr_fft1d_c2c(d, n, inv ? 1 : 0, FDR_MODE_PARITY)); }  // fft/fft.hpp:39

void my_dft2D(Mat& m, bool inverse) {  // fft/fft.hpp:40
    if (m.type() != CV_32FC2 || !m.isContinuous()) { std::fprintf(stderr, "Error: %s:%d, my_dft2D needs a continuous CV_32FC2 Mat\n", __FILE__, __LINE__); std::exit(1); }
    const unsigned flags = (fdr_is_pow2(m.rows) && fdr_is_pow2(m.cols)) ? 0u : FDR_FLAG_ANY_SIZE;
    fdr_plan* plan = nullptr;
    FDR_CHECK(fdr_plan_create(0, m.rows, m.cols, FDR_MODE_PARITY, flags, &plan));
    FDR_CHECK(fdr_fft2d_c2c(plan, m.ptr<float>(), inverse ? 1 : 0));
    fdr_plan_destroy(plan);
}

Mat wienerDeblur_myfft(const Mat& img, const Mat& psf, float K) { return operator_as_serial(img, psf, K, FDR_MODE_FAST); }  // fft/fft.hpp:44

}  // namespace fft_gpu

namespace fft_serial {

void fft_radix2_inplace(vector<complex<float>>& a, bool inverse) {  // fft/fft_serial.cpp:40-68
    if (!a.empty()) FDR_CHECK(fdr_fft1d_c2c(reinterpret_cast<float*>(a.data()), (int)a.size(), inverse ? 1 : 0, FDR_MODE_PARITY));
}
void dft_naive_inplace(vector<complex<float>>& a, bool inverse) {  // fft/fft_serial.cpp:71-87
    if (!a.empty()) FDR_CHECK(fdr_dft_naive_c2c(reinterpret_cast<float*>(a.data()), (int)a.size(), inverse ? 1 : 0));
}
void transform_row_inplace(Vec2f* rowPtr, int N, bool inverse) {  // fft/fft_serial.cpp:90-108
    FDR_CHECK(fdr_fft1d_c2c(reinterpret_cast<float*>(rowPtr), N, inverse ? 1 : 0, FDR_MODE_PARITY));
}
void my_dft2D(Mat& complexMat, bool inverse) { fft_gpu::my_dft2D(complexMat, inverse); }  // fft/fft_serial.cpp:113-139
Mat wienerDeblur_myfft(const Mat& img, const Mat& psf, float K) { return operator_as_serial(img, psf, K, FDR_MODE_PARITY); }  // :141-261

}  // namespace fft_serial
