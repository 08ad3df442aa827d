// seam_bench -- what a grid-wide seam costs on this chip, three ways (the question behind "one cooperative launch for a
// single small image"): a restoration of one image is FOUR phases separated by three all-to-all seams (every column tile
// needs every row group, every row group needs every tile, the normalisation needs every row group's extremes).
//   (a) launches : 4 dependent trivial kernels on one stream                       -- what libfdr does
//   (b) counter  : ONE plain launch, 3 hand-rolled grid barriers (monotonic counter, agent-scope release / acquire,
//                  bounded spin), grid = one workgroup per CU so that every workgroup is resident
//   (c) coop     : the same through hipLaunchCooperativeKernel + cooperative_groups::this_grid().sync()
// Each "phase" does the same token amount of work (one 16-byte load + store per thread), so the figures are the seams'.
//   usage: seam_bench [workgroups=256] [threads=256] [iterations=200]
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)
namespace cg = cooperative_groups;

__device__ __forceinline__ void phase(float4* __restrict__ buf, int p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = buf[i];
    v.x += (float)p;
    buf[i] = v;
}

__global__ void phase_kernel(float4* buf, int p) { phase(buf, p); }

// arrive on a monotonic counter, then poll it (relaxed, agent scope) until every workgroup of this generation has arrived;
// bounded: gives up after ~2 ms and sets *timeout (the grid is sized to be resident, so this only trips on a mistake)
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, unsigned* timeout) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);  // (device scope)
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 200000u) { *timeout = 1u; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

__global__ void fused_counter_kernel(float4* buf, unsigned* counter, unsigned base, unsigned* timeout) {
    for (int p = 0; p < 4; ++p) {
        phase(buf, p);
        if (p < 3) grid_barrier(counter, base + (unsigned)(p + 1) * gridDim.x, timeout);
    }
}

__global__ void fused_coop_kernel(float4* buf) {
    cg::grid_group g = cg::this_grid();
    for (int p = 0; p < 4; ++p) {
        phase(buf, p);
        if (p < 3) g.sync();
    }
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? std::atoi(argv[1]) : 256, threads = argc > 2 ? std::atoi(argv[2]) : 256, iters = argc > 3 ? std::atoi(argv[3]) : 200;
    float4* buf = nullptr;
    unsigned *counter = nullptr, *timeout = nullptr;
    CK(hipMalloc((void**)&buf, (size_t)wgs * threads * sizeof(float4)));
    CK(hipMemset(buf, 0, (size_t)wgs * threads * sizeof(float4)));
    CK(hipMalloc((void**)&counter, 64));
    CK(hipMalloc((void**)&timeout, 64));
    CK(hipMemset(counter, 0, 64));
    CK(hipMemset(timeout, 0, 64));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto median_us = [&](auto&& body) {
        std::vector<double> t;
        for (int r = 0; r < 7; ++r) {
            CK(hipStreamSynchronize(st));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < iters; ++i) body(i);
            CK(hipStreamSynchronize(st));
            t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters);
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2];
    };
    const double a = median_us([&](int) {
        for (int p = 0; p < 4; ++p) hipLaunchKernelGGL(phase_kernel, dim3(wgs), dim3(threads), 0, st, buf, p);
    });
    unsigned gen = 0;  // the counter is monotonic: generation `gen` starts at gen * 3 * wgs
    const double b = median_us([&](int) {
        hipLaunchKernelGGL(fused_counter_kernel, dim3(wgs), dim3(threads), 0, st, buf, counter, gen * 3u * (unsigned)wgs, timeout);
        ++gen;
    });
    unsigned h_timeout = 0;
    CK(hipMemcpy(&h_timeout, timeout, sizeof(unsigned), hipMemcpyDeviceToHost));
    double c = -1.0;
    int coop = 0;
    CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0));
    if (coop) {
        void* args[] = {&buf};
        c = median_us([&](int) { CK(hipLaunchCooperativeKernel((const void*)fused_coop_kernel, dim3(wgs), dim3(threads), args, 0, st)); });
    }
    const double e = median_us([&](int) { hipLaunchKernelGGL(phase_kernel, dim3(wgs), dim3(threads), 0, st, buf, 0); });
    std::printf("seam_bench: %d workgroups x %d threads, %d iterations, median of 7 (us per 4-phase image, back to back on one stream)\n", wgs, threads, iters);
    std::printf("  (a) 4 dependent launches                      : %7.2f us   (one launch of the same kernel: %.2f us)\n", a, e);
    std::printf("  (b) 1 launch, 3 counter barriers (plain)       : %7.2f us   %s\n", b, h_timeout ? "BARRIER TIMED OUT" : "");
    if (c >= 0) std::printf("  (c) 1 cooperative launch, 3 grid.sync()        : %7.2f us\n", c);
    else std::printf("  (c) cooperative launch not supported on this device\n");
    return h_timeout ? 3 : 0;
}
