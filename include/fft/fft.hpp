// fft/fft.hpp -- drop-in counterpart of the reference's fft/fft.hpp for the GPU namespace (reference fft/fft.hpp:31-45)
// and the serial namespace (:9-18).  Same names, same argument meaning, same error behaviour (print
// "Error: file:line, msg" and exit(1), fft/fft_gpu.cu:59-66); everything is a thin call into the C ABI of libfdr.so
// (include/fdr.h).  Link with -lfdr.
//
//   fft_gpu::wienerDeblur_RGB_optimized / _naive   fft/fft_gpu.cu:279-394 / :400-512
//   fft_gpu::fft_radix2_kernel / transform_row_kernel / dft_naive_kernel / my_dft2D / wienerDeblur_myfft
//                                                  declared at fft/fft.hpp:35-44, empty or missing in the reference
//   fft_serial::*                                  fft/fft_serial.cpp:40-261, run on the GPU in the parity mode
// Differences that are deliberate: the PSF spectrum is built once per call, not once per
// channel; wienerDeblur_RGB_* follow the SERIAL DRIVER's semantics by default (normalise over the padded area, then
// crop: serial.cpp:34-39) because ./serial is the parity target -- Options::norm_area = FDR_NORM_CROPPED gives the
// reference GPU order (fft/fft_gpu.cu:367-368,379-381).  Every entry point takes its mode / area / device as an
// argument (Options); the argument-free overloads of the reference's signatures use process-wide defaults.
#pragma once
#include "../utils.hpp"
#include <atomic>
#include <complex>
#include <iostream>
#include <map>
#include <string>
#include <vector>

namespace fft_gpu {

struct Options {
    int mode = FDR_MODE_FAST;         // FDR_MODE_FAST or FDR_MODE_PARITY (bit-identical FFT arithmetic to fft_serial)
    int norm_area = FDR_NORM_PADDED;  // FDR_NORM_PADDED (./serial semantics) or FDR_NORM_CROPPED (reference ./gpu, fft_gpu.cu:379-381)
    int device = 0;
};
// process-wide defaults of the reference-signature overloads (the drivers' --mode / --norm flags); atomics: reading
// them from several threads is safe, and no entry point ever changes them behind the caller's back
inline std::atomic<int>& default_mode() { static std::atomic<int> m{FDR_MODE_FAST}; return m; }
inline std::atomic<int>& default_norm() { static std::atomic<int> n{FDR_NORM_PADDED}; return n; }
inline void set_mode(int mode) { default_mode().store(mode); }
inline void set_norm_area(int area) { default_norm().store(area); }
inline Options defaults() { Options o; o.mode = default_mode().load(); o.norm_area = default_norm().load(); return o; }

// The reference's Profiler buckets (fft/fft_gpu.cu:17-57).  alloc / h2d / pre / compute / d2h come from
// fdr_plan_phase_times (hipEvent pairs on the streams the work ran on); post is the host time of wrapping the results.
struct Profiler {
    double t_alloc = 0, t_h2d = 0, t_pre = 0, t_compute = 0, t_d2h = 0, t_post = 0;
    void add(fdr_plan* plan) {
        float ms[FDR_N_PHASES];
        FDR_CHECK(fdr_plan_phase_times(plan, ms, 1));
        t_alloc += ms[FDR_PHASE_ALLOC]; t_h2d += ms[FDR_PHASE_H2D]; t_pre += ms[FDR_PHASE_PRE];
        t_compute += ms[FDR_PHASE_COMPUTE]; t_d2h += ms[FDR_PHASE_D2H]; t_post += ms[FDR_PHASE_POST];
    }
    void print(const std::string& title) const {
        std::cout << "=== " << title << " Profiling (3 Channels) ===" << std::endl;
        std::cout << "[1. Allocation]  Time: " << t_alloc << " ms (plan: twiddles + workspaces)" << std::endl;
        std::cout << "[2. H2D Copy]    Time: " << t_h2d << " ms (Raw Img + PSF)" << std::endl;
        std::cout << "[3. Pre-process] Time: " << t_pre << " ms (Padding + PSF FFT)" << std::endl;
        std::cout << "[4. GPU Compute] Time: " << t_compute << " ms (FFT + Filter + IFFT + Normalize)" << std::endl;
        std::cout << "[5. D2H Copy]    Time: " << t_d2h << " ms (Result Transfer)" << std::endl;
        std::cout << "[6. Post-process]Time: " << t_post << " ms (CPU Copy)" << std::endl;
        std::cout << "--------------------------------------------" << std::endl;
        std::cout << "Total (Sum)      Time: " << (t_alloc + t_h2d + t_pre + t_compute + t_d2h + t_post) << " ms" << std::endl;
        std::cout << "============================================" << std::endl;
    }
};

// Plans kept BETWEEN calls of wienerDeblur_RGB_optimized and of the per-channel operator wienerDeblur_myfft ("Reuse Memory"
// taken one step further than fft/fft_gpu.cu:304-322, which still allocates once per call): a driver that warms up and then
// times the entry point (gpu.cpp:96-105), loops over the channels (serial.cpp:34-39) or calls once per picture of one size
// pays for the twiddle tables, the workspaces and their hipFree once per thread (measured on 782 x 1920: the timed
// _optimized call 17.6 -> 1.1 ms).  wienerDeblur_RGB_naive keeps allocating per channel, as its name says.  Per thread (a plan serves
// one host thread at a time), at most plan_cache_capacity() plans (default 4), least recently used evicted.  The reference
// frees everything per call (fft/fft_gpu.cu:389-393); what this header retains instead is bounded and released:
//   * when the owning thread ends (the cache is a thread_local OBJECT; its destructor destroys every plan -- on the main
//     thread that happens at exit() before any atexit handler, i.e. while the HIP runtime is still up; once the process is
//     past that point fdr_plan_destroy only frees host memory, see fdr.h),
//   * on demand: fft_gpu::release_cached_plans() (this thread's), fft_gpu::set_plan_cache_capacity(n) (0 = keep nothing
//     between calls: the reference's behaviour).
struct PlanCache {
    struct Entry { int device, M, N, mode; unsigned flags; fdr_plan* plan; };
    std::vector<Entry> entries;
    PlanCache() = default;
    PlanCache(const PlanCache&) = delete;
    PlanCache& operator=(const PlanCache&) = delete;
    ~PlanCache() { clear(); }
    void clear() {
        for (Entry& e : entries) fdr_plan_destroy(e.plan);
        entries.clear();
    }
    static std::atomic<int>& capacity() { static std::atomic<int> c{4}; return c; }
    // A plan handed out with capacity 0 is not retained: the caller destroys it (see PlanLease).
    fdr_plan* get(int device, int M, int N, int mode, bool* created, unsigned flags = 0u) {
        for (size_t i = 0; i < entries.size(); ++i)
            if (entries[i].device == device && entries[i].M == M && entries[i].N == N && entries[i].mode == mode && entries[i].flags == flags) {
                const Entry e = entries[i];
                entries.erase(entries.begin() + (long)i);
                entries.push_back(e);  // most recently used last
                *created = false;
                return e.plan;
            }
        const int cap = capacity().load();
        while (!entries.empty() && (int)entries.size() >= (cap > 0 ? cap : 1)) { fdr_plan_destroy(entries.front().plan); entries.erase(entries.begin()); }
        fdr_plan* plan = nullptr;
        FDR_CHECK(fdr_plan_create(device, M, N, mode, flags, &plan));
        entries.push_back(Entry{device, M, N, mode, flags, plan});
        *created = true;
        return plan;
    }
    // end of an entry point: with capacity 0 nothing stays allocated between calls
    void settle() { if (capacity().load() <= 0) clear(); }
};
inline PlanCache& plan_cache() { static thread_local PlanCache c; return c; }
// Destroys the calling thread's cached plans now (device workspaces, filter, staging buffers, streams).
inline void release_cached_plans() { plan_cache().clear(); }
// Plans kept per thread between calls (default 4); 0 = allocate and free inside every call, as fft/fft_gpu.cu:304-322,389-393.
inline void set_plan_cache_capacity(int n) { PlanCache::capacity().store(n < 0 ? 0 : n); if (n <= 0) plan_cache().clear(); }
inline int plan_cache_capacity() { return PlanCache::capacity().load(); }
struct PlanCacheSettle { ~PlanCacheSettle() { plan_cache().settle(); } };

inline Mat run_channel(fdr_plan* plan, const Mat& img, int norm_area) {
    Mat src = img.isContinuous() ? img : img.clone();
    Mat out(img.rows, img.cols, CV_32F);
    FDR_CHECK(fdr_wiener_f32(plan, src.ptr<float>(0), img.rows, img.cols, img.cols, out.ptr<float>(0), img.cols, norm_area));
    return out;
}

// Version A (fft/fft_gpu.cu:279-394): one plan, one PSF spectrum, all channels; replaces every element of `channels`.
inline void wienerDeblur_RGB_optimized(std::vector<Mat>& channels, const Mat& psf, float K, const Options& o) {
    if (channels.empty()) return;
    Profiler p;
    const int imgRows = channels[0].rows, imgCols = channels[0].cols;
    bool created = false;
    PlanCacheSettle settle_;
    fdr_plan* plan = plan_cache().get(o.device, nextPowerOfTwo(imgRows), nextPowerOfTwo(imgCols), o.mode, &created);
    if (!created) { float discard[FDR_N_PHASES]; FDR_CHECK(fdr_plan_phase_times(plan, discard, 1)); }  // this call's phases only ([1. Allocation] = 0: reused)
    Mat psfc = psf.isContinuous() ? psf : psf.clone();
    FDR_CHECK(fdr_set_psf(plan, psfc.ptr<float>(0), psf.rows, psf.cols, psf.cols, K));
    // all channels through the host batch pipeline: upload, restoration and download of consecutive channels overlap
    // (what the stream + pinned-buffer set-up of fft/fft_gpu.cu:304-350 is after)
    bool same = true;
    for (const Mat& c : channels) same = same && c.rows == imgRows && c.cols == imgCols && c.type() == CV_32F;
    if (same) {
        std::vector<Mat> src, out;
        std::vector<const float*> ins;
        std::vector<float*> outs;
        auto a = high_resolution_clock::now();
        for (const Mat& c : channels) {
            src.push_back(c.isContinuous() ? c : c.clone());
            out.push_back(Mat(imgRows, imgCols, CV_32F));
        }
        p.t_post += getElapsedMs(a, high_resolution_clock::now());
        for (size_t i = 0; i < channels.size(); ++i) { ins.push_back(src[i].ptr<float>(0)); outs.push_back(out[i].ptr<float>(0)); }
        FDR_CHECK(fdr_wiener_batch_ptrs_f32(plan, ins.data(), outs.data(), (int)channels.size(), imgRows, imgCols, imgCols, imgCols, o.norm_area));
        for (size_t i = 0; i < channels.size(); ++i) channels[i] = out[i];
    } else {
        for (size_t i = 0; i < channels.size(); ++i) channels[i] = run_channel(plan, channels[i], o.norm_area);
    }
    p.add(plan);
    p.print("FAST (Reuse Memory)");
}
inline void wienerDeblur_RGB_optimized(std::vector<Mat>& channels, const Mat& psf, float K) {
    wienerDeblur_RGB_optimized(channels, psf, K, defaults());
}

// Version B (fft/fft_gpu.cu:400-512): every channel allocates, builds the PSF spectrum and frees.
inline void wienerDeblur_RGB_naive(std::vector<Mat>& channels, const Mat& psf, float K, const Options& o) {
    Profiler p;
    for (size_t i = 0; i < channels.size(); ++i) {
        fdr_plan* plan = nullptr;
        FDR_CHECK(fdr_plan_create(o.device, nextPowerOfTwo(channels[i].rows), nextPowerOfTwo(channels[i].cols), o.mode, 0, &plan));
        Mat psfc = psf.isContinuous() ? psf : psf.clone();
        FDR_CHECK(fdr_set_psf(plan, psfc.ptr<float>(0), psf.rows, psf.cols, psf.cols, K));
        channels[i] = run_channel(plan, channels[i], o.norm_area);
        p.add(plan);
        fdr_plan_destroy(plan);
    }
    p.print("SLOW (Naive Allocation)");
}
inline void wienerDeblur_RGB_naive(std::vector<Mat>& channels, const Mat& psf, float K) {
    wienerDeblur_RGB_naive(channels, psf, K, defaults());
}

// The operator exactly as fft_serial::wienerDeblur_myfft defines it (fft/fft_serial.cpp:141-261; the fft_gpu
// declaration at fft/fft.hpp:44 has no body in the reference): pad to getOptimalDFTSize (2^a 3^b 5^c, :153-154 -- a
// non-power-of-two dimension is transformed by the naive DFT, :100-101), restore, crop to img's size, normalise over
// the cropped plane (:243-246).  For the pre-padded channels the drivers pass (serial.cpp:36) pad and crop are no-ops.
inline Mat wienerDeblur_myfft(const Mat& img, const Mat& psf, float K, const Options& o) {
    const int M = fdr_optimal_dft_size(img.rows), N = fdr_optimal_dft_size(img.cols);
    const unsigned flags = (isPowerOfTwo(M) && isPowerOfTwo(N)) ? 0u : FDR_FLAG_ANY_SIZE;
    bool created = false;
    PlanCacheSettle settle_;
    fdr_plan* plan = plan_cache().get(o.device, M, N, o.mode, &created, flags);
    Mat psfc = psf.isContinuous() ? psf : psf.clone();
    FDR_CHECK(fdr_set_psf(plan, psfc.ptr<float>(0), psf.rows, psf.cols, psf.cols, K));
    return run_channel(plan, img, FDR_NORM_CROPPED);
}
inline Mat wienerDeblur_myfft(const Mat& img, const Mat& psf, float K) { return wienerDeblur_myfft(img, psf, K, defaults()); }

// fft/fft.hpp:35-39: n interleaved complex values by host pointer, unscaled
inline void fft_radix2_kernel(float* data, int n, bool inverse) { FDR_CHECK(fdr_fft1d_c2c(data, n, inverse ? 1 : 0, FDR_MODE_PARITY)); }
inline void dft_naive_kernel(float* data, int n, bool inverse) { FDR_CHECK(fdr_dft_naive_c2c(data, n, inverse ? 1 : 0)); }
inline void transform_row_kernel(float* rowPtr, int N, bool inverse) { FDR_CHECK(fdr_fft1d_c2c(rowPtr, N, inverse ? 1 : 0, FDR_MODE_PARITY)); }

// fft/fft.hpp:40-42: in-place unscaled 2-D transform of a CV_32FC2 Mat (rows, transpose, rows, transpose); any size
// up to 32768 for powers of two (above 8192: 8192-point blocks + radix-2 stages in global memory), non-powers of two up to
// 4096 (naive DFT along that dimension, as fft_serial.cpp:100-101)
inline void my_dft2D(Mat& complexMat, bool inverse) {
    if (complexMat.type() != CV_32FC2) { std::fprintf(stderr, "Error: %s:%d, my_dft2D needs CV_32FC2\n", __FILE__, __LINE__); std::exit(1); }
    const int M = complexMat.rows, N = complexMat.cols;
    Mat c = complexMat.isContinuous() ? complexMat : complexMat.clone();
    fdr_plan* plan = nullptr;
    const unsigned flags = (isPowerOfTwo(M) && isPowerOfTwo(N)) ? 0u : FDR_FLAG_ANY_SIZE;
    FDR_CHECK(fdr_plan_create(0, M, N, FDR_MODE_PARITY, flags, &plan));
    FDR_CHECK(fdr_fft2d_c2c(plan, c.ptr<float>(0), inverse ? 1 : 0));
    fdr_plan_destroy(plan);
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

// The accumulated phase timers of fft/fft_serial.cpp:13-35: "Serial: ..." names (:158-236), printed once, when the
// call count reaches CHANNELS = 3 (:249-258; never reset afterwards, as in the reference).  The figures are DEVICE
// times (hipEvent pairs): uploads -> "Pre-process" (the padding happens on load inside the first pass); the PSF
// spectrum -> "FFT PSF"; passes A + B (forward rows, forward columns with the Wiener quotient fused into their
// epilogue) -> "FFT Image"; "Wiener Filter" stays 0 because the quotient has no pass of its own here; passes C + D
// (inverse rows, inverse columns + real part + min/max) -> "IFFT"; normalise + crop + download -> "Post-process".
struct PhaseAccum {
    int callCount = 0;
    std::map<std::string, double> accum;
};
inline PhaseAccum& phase_accum() { static PhaseAccum a; return a; }

// fft/fft_serial.cpp:141-261: pads to getOptimalDFTSize, crops to img's size, normalises over the cropped plane
inline Mat wienerDeblur_myfft(const Mat& img, const Mat& psf, float K) {
    PhaseAccum& acc = phase_accum();
    if (acc.callCount == 0) acc.accum.clear();
    acc.callCount++;
    const int M = fdr_optimal_dft_size(img.rows), N = fdr_optimal_dft_size(img.cols);
    const unsigned flags = (isPowerOfTwo(M) && isPowerOfTwo(N)) ? 0u : FDR_FLAG_ANY_SIZE;
    bool created = false;
    fft_gpu::PlanCacheSettle settle_;
    fdr_plan* plan = fft_gpu::plan_cache().get(0, M, N, FDR_MODE_PARITY, &created, flags);  // kept between the channels of a driver's loop
    float ph[FDR_N_PHASES] = {0}, ms[FDR_MAX_PASSES] = {0};
    if (!created) FDR_CHECK(fdr_plan_phase_times(plan, ph, 1));  // this call's phases only
    Mat psfc = psf.isContinuous() ? psf : psf.clone();
    FDR_CHECK(fdr_set_psf(plan, psfc.ptr<float>(0), psf.rows, psf.cols, psf.cols, K));
    FDR_CHECK(fdr_plan_profile(plan, 1));
    Mat out = fft_gpu::run_channel(plan, img, FDR_NORM_CROPPED);
    const char* names[FDR_MAX_PASSES] = {nullptr};
    int n = 0, launches[FDR_MAX_PASSES] = {0};
    FDR_CHECK(fdr_plan_phase_times(plan, ph, 1));
    FDR_CHECK(fdr_plan_pass_times(plan, &n, ms, names, launches));
    FDR_CHECK(fdr_plan_profile(plan, 0));
    double fwd = 0, inv = 0, post = 0;
    for (int i = 0; i < n; ++i) {
        const std::string nm = names[i] ? names[i] : "";
        const double t = (double)ms[i] * launches[i];
        if (nm.rfind("A ", 0) == 0 || nm.rfind("B ", 0) == 0 || nm.rfind("simple", 0) == 0) fwd += t;
        else if (nm.rfind("C ", 0) == 0 || nm.rfind("D ", 0) == 0) inv += t;
        else post += t;  // E normalize + crop
    }
    acc.accum["Serial: Pre-process"] += ph[FDR_PHASE_H2D];
    acc.accum["Serial: FFT Image"] += fwd;
    acc.accum["Serial: FFT PSF"] += ph[FDR_PHASE_PRE];
    acc.accum["Serial: Wiener Filter"] += 0.0;
    acc.accum["Serial: IFFT"] += inv;
    acc.accum["Serial: Post-process"] += post + ph[FDR_PHASE_D2H];
    if (acc.callCount == 3) {
        std::cout << "=== Accumulated Time ===" << std::endl;
        float this_round_total = 0;
        for (auto& p : acc.accum) {
            std::cout << p.first << " total: " << p.second << " ms" << std::endl;
            this_round_total += (float)p.second;
        }
        std::cout << "this round total: " << this_round_total << " ms" << std::endl;
        std::cout << "=========================" << std::endl;
    }
    return out;
}

}  // namespace fft_serial
