// utils.hpp -- drop-in counterpart of the reference's utils.hpp (same names, same argument meaning):
//   getElapsedMs            reference utils.hpp:9-12
//   motionBlurKernel        reference utils.hpp:15-24   -> PSF kernel on the GPU (fdr_psf_motion)
//   nextPowerOfTwo / getNextPowerOf2   :27-37           -> fdr_next_pow2
//   autoPadToPowerOfTwo     :40-47
//   isPowerOfTwo            :50-52                      -> fdr_is_pow2
//   applyWhiteBalance       :55-71
// Built on cv::Mat when FDR_WITH_OPENCV is defined, else on the bundled container (fdr_mat.hpp).
#pragma once
#include "fdr.h"
#include "fdr_mat.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace cv;
using namespace std;
using namespace std::chrono;

inline double getElapsedMs(high_resolution_clock::time_point start, high_resolution_clock::time_point end) {
    return duration<double, std::milli>(end - start).count();
}

// the reference's CHECK_CUDA convention (fft/fft_gpu.cu:59-66): print "Error: file:line, msg" and exit(1)
#define FDR_CHECK(call)                                                                          \
    do {                                                                                         \
        const int fdr_rc_ = (call);                                                              \
        if (fdr_rc_ != FDR_OK) {                                                                 \
            std::fprintf(stderr, "Error: %s:%d, %s\n", __FILE__, __LINE__, fdr_last_error());    \
            std::exit(1);                                                                        \
        }                                                                                        \
    } while (0)

inline Mat motionBlurKernel(int size, double angle) {
    Mat rotated(size, size, CV_32F);
    FDR_CHECK(fdr_psf_motion(size, angle, rotated.ptr<float>(0)));
    return rotated;
}

inline int nextPowerOfTwo(int n) { return fdr_next_pow2(n); }
inline int getNextPowerOf2(int n) { return fdr_next_pow2(n); }
inline bool isPowerOfTwo(int n) { return fdr_is_pow2(n) != 0; }

inline Mat autoPadToPowerOfTwo(const Mat& src) {
    const int newRows = nextPowerOfTwo(src.rows), newCols = nextPowerOfTwo(src.cols);
    Mat padded = Mat::zeros(newRows, newCols, CV_32F);
    for (int r = 0; r < src.rows; ++r) std::memcpy(padded.ptr<float>(r), src.ptr<float>(r), sizeof(float) * (size_t)src.cols);
    return padded;
}

inline Mat applyWhiteBalance(const Mat& img_Lab, const Mat& img_orig_Lab) {
    vector<Mat> orig_channels, deblur_channels;
    split(img_orig_Lab, orig_channels);
    split(img_Lab, deblur_channels);
    const double avgL_orig = mean(orig_channels[0])[0];
    const double avgL_deblur = mean(deblur_channels[0])[0];
    const double gain = avgL_orig / (avgL_deblur + 1e-6);
    deblur_channels[0] = deblur_channels[0] * gain;
    cv::min(deblur_channels[0], 100.0f, deblur_channels[0]);
    cv::max(deblur_channels[0], 0.0f, deblur_channels[0]);
    Mat corrected_Lab;
    merge(deblur_channels, corrected_Lab);
    return corrected_Lab;
}
