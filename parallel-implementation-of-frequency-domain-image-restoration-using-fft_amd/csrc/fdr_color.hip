// fdr_color.hip -- the colour epilogue of the reference drivers on the device (serial.cpp:43-54, gpu.cpp:123-137):
// merge the three restored planes, BGR -> Lab, applyWhiteBalance (utils.hpp:55-71: L of the restored image scaled so
// that its mean matches the blurred input's, clamped to [0, 100]), Lab -> BGR, convertTo(CV_8U, 255).
// Two streaming passes: (1) the two L means (only L is needed for the gain) as per-workgroup double partial sums,
// (2) every workgroup folds the partials, then converts its pixels.  The Lab formulae are OpenCV's float path for
// images in [0,1] (sRGB companding, D65) as restated in tools/cli/fdr_image_io.hpp -- third-party arithmetic, version
// unpinned by the reference; the tests compare against that host restatement and accept +-1 at 8 bit.
#include "fdr_kernels.hpp"

namespace fdr {

namespace {

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float srgb_to_lin(float c) { return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float lin_to_srgb(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f; }
__device__ __forceinline__ float lab_f(float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.f / 116.f; }

struct Lab { float L, a, b; };
__device__ __forceinline__ Lab bgr_to_lab(float bb, float gg, float rr) {
    const float Xn = 0.950456f, Zn = 1.088754f;
    const float b = srgb_to_lin(clamp01(bb)), g = srgb_to_lin(clamp01(gg)), r = srgb_to_lin(clamp01(rr));
    const float X = (0.412453f * r + 0.357580f * g + 0.180423f * b) / Xn, Y = 0.212671f * r + 0.715160f * g + 0.072169f * b,
                Z = (0.019334f * r + 0.119193f * g + 0.950227f * b) / Zn;
    const float fx = lab_f(X), fy = lab_f(Y), fz = lab_f(Z);
    Lab o;
    o.L = Y > 0.008856f ? 116.f * fy - 16.f : 903.3f * Y;
    o.a = 500.f * (fx - fy);
    o.b = 200.f * (fy - fz);
    return o;
}
__device__ __forceinline__ float lab_finv(float t) { return t > 0.206893f ? t * t * t : (t - 16.f / 116.f) / 7.787f; }
__device__ __forceinline__ void lab_to_bgr(const Lab& p, float& bb, float& gg, float& rr) {
    const float Xn = 0.950456f, Zn = 1.088754f;
    const float fy = (p.L + 16.f) / 116.f, fx = fy + p.a / 500.f, fz = fy - p.b / 200.f;
    const float Y = p.L > 7.9996f ? fy * fy * fy : p.L / 903.3f;
    const float X = lab_finv(fx) * Xn, Z = lab_finv(fz) * Zn;
    const float r = 3.240479f * X - 1.537150f * Y - 0.498535f * Z, g = -0.969256f * X + 1.875991f * Y + 0.041556f * Z,
                b = 0.055648f * X - 0.204043f * Y + 1.057311f * Z;
    bb = lin_to_srgb(clamp01(b)); gg = lin_to_srgb(clamp01(g)); rr = lin_to_srgb(clamp01(r));
}

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    __syncthreads();
    return s;
}

constexpr int kThreads = 256;

// pass 1: partial sums of L over the original (blurred) planes and over the restored planes
__global__ __launch_bounds__(kThreads) void lab_mean_kernel(ColorArgs a, double2* __restrict__ part) {
    __shared__ double red[kThreads / 64];
    double so = 0.0, sr = 0.0;
    const long long n = (long long)a.rows * a.cols;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const int y = (int)(i / a.cols), x = (int)(i - (long long)y * a.cols);
        const size_t o = (size_t)y * a.stride + x;
        so += (double)bgr_to_lab(a.orig[0][o], a.orig[1][o], a.orig[2][o]).L;
        sr += (double)bgr_to_lab(a.rest[0][o], a.rest[1][o], a.rest[2][o]).L;
    }
    so = block_sum(so, red);
    sr = block_sum(sr, red);
    if (threadIdx.x == 0) part[blockIdx.x] = make_double2(so, sr);
}

// pass 2: gain from the partials (fixed order: deterministic), white balance, back to BGR, 8 bit
__global__ __launch_bounds__(kThreads) void lab_apply_kernel(ColorArgs a, const double2* __restrict__ part, int n_part) {
    __shared__ double red[kThreads / 64];
    double so = 0.0, sr = 0.0;
    for (int i = threadIdx.x; i < n_part; i += kThreads) { so += part[i].x; sr += part[i].y; }
    so = block_sum(so, red);
    sr = block_sum(sr, red);
    const double n = (double)a.rows * (double)a.cols;
    const float gain = (float)((so / n) / (sr / n + 1e-6));  // utils.hpp:62-64 (Mat * double scales by the float of it)
    const long long npx = (long long)a.rows * a.cols;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < npx; i += (long long)gridDim.x * kThreads) {
        const int y = (int)(i / a.cols), x = (int)(i - (long long)y * a.cols);
        const size_t o = (size_t)y * a.stride + x;
        Lab p = bgr_to_lab(a.rest[0][o], a.rest[1][o], a.rest[2][o]);
        p.L = fmaxf(fminf(p.L * gain, 100.0f), 0.0f);  // utils.hpp:66-68
        float b, g, r;
        lab_to_bgr(p, b, g, r);
        unsigned char* q = a.out + (size_t)y * a.out_stride + 3 * (size_t)x;
        // convertTo(CV_8U, 255.0): double product, round half to even, saturate
        const int qb = __double2int_rn((double)b * 255.0), qg = __double2int_rn((double)g * 255.0), qr = __double2int_rn((double)r * 255.0);
        q[0] = (unsigned char)min(max(qb, 0), 255);
        q[1] = (unsigned char)min(max(qg, 0), 255);
        q[2] = (unsigned char)min(max(qr, 0), 255);
    }
}

}  // namespace

int color_partials(int rows, int cols) {
    long long n = ((long long)rows * cols + kThreads - 1) / kThreads;
    return (int)(n > 1024 ? 1024 : (n < 1 ? 1 : n));
}

hipError_t launch_color_epilogue(const ColorArgs& a, double2* part, hipStream_t s) {
    const int nb = color_partials(a.rows, a.cols);
    hipLaunchKernelGGL(lab_mean_kernel, dim3(nb), dim3(kThreads), 0, s, a, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lab_apply_kernel, dim3(nb * 4 > 4096 ? 4096 : nb * 4), dim3(kThreads), 0, s, a, part, nb);
    return hipGetLastError();
}

}  // namespace fdr
