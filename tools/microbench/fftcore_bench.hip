// Compute-floor microbenchmark for the register/LDS FFT core (no HBM traffic in the loop):
// each thread group loads its transform(s) once, runs the core REPS times, stores once.
// Prints microseconds per transform per CU-equivalent so pass timings can be decomposed.
#include "../../parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd/csrc/fdr_fft_core.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace fdr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int LOGL, int B, int NBUF, class Pol, int MINW, int LOGV = 3>
__global__ __launch_bounds__((Steps<LOGL, LOGV>::T), MINW) void core_loop(float2* data, const float2* __restrict__ tw, int reps) {
    using St = Steps<LOGL, LOGV>;
    using Core = FftCore<LOGL, B, NBUF, Pol, LOGV>;
    constexpr int V = St::V;
    __shared__ float2 lds[NBUF * St::BUF];
    const int tid = threadIdx.x;
    float2 v[B][V];
    typename Core::Bases bases;
    Core::init_bases(bases, tw, tid);
    float2* base = data + (size_t)blockIdx.x * B * St::L;
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int s = 0; s < V; ++s) v[b][s] = base[b * St::L + tid + s * St::T];
    for (int r = 0; r < reps; ++r) {
        Core::template run<0, true>(v, lds, tw, bases, tid);
        if (NBUF == 2 && (Core::SLOTS & 1)) __syncthreads();  // keep buffer parity hazard-free across reps
        if (NBUF == 2) __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int s = 0; s < V; ++s) base[b * St::L + tid + s * St::T] = v[b][s];
}

template <typename F>
float time_ms(F f, int iters = 5) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <int LOGL, int B, int NBUF, class Pol, int MINW, int LOGV = 3>
void run(const char* name, float2* data, float2* tw, int wgs_per_cu) {
    const int reps = 64;
    const int grid = 256 * wgs_per_cu;
    float ms = time_ms([&] { hipLaunchKernelGGL((core_loop<LOGL, B, NBUF, Pol, MINW, LOGV>), dim3(grid), dim3(Steps<LOGL, LOGV>::T), 0, 0, data, tw, reps); });
    CK(hipGetLastError());
    double ffts_per_cu = (double)wgs_per_cu * B * reps;
    printf("%-40s L=%5d B=%d NBUF=%d wg/cu=%d : %8.3f ms total, %7.3f us per transform per CU, => %6.1f us for 4096 transforms/256 CUs x16\n", name,
           1 << LOGL, B, NBUF, wgs_per_cu, ms, ms * 1e3 / ffts_per_cu, ms * 1e3 / ffts_per_cu * 16);
}

int main() {
    float2 *data, *tw;
    size_t n = (size_t)256 * 8 * 4 * 8192;
    CK(hipMalloc(&data, n * 8)); CK(hipMemset(data, 0, n * 8));
    std::vector<float2> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = make_float2(1.f, 0.f);
    CK(hipMalloc(&tw, 8192 * 8)); CK(hipMemcpy(tw, h.data(), 8192 * 8, hipMemcpyHostToDevice));
    run<12, 4, 2, PolicyFast, 2, 4>("16 values/thread, B=4 NBUF=2, 256 thr, 2 wg/CU", data, tw, 2);
    run<12, 4, 2, PolicyFast, 2, 4>("16 values/thread, B=4 NBUF=2, 256 thr, 1 wg/CU", data, tw, 1);
    run<12, 2, 2, PolicyFast, 2, 4>("16 values/thread, B=2 NBUF=2, 256 thr, 2 wg/CU", data, tw, 2);
    run<13, 4, 2, PolicyFast, 2, 4>("16 values/thread, L=8192 B=4, 512 thr, 1 wg/CU", data, tw, 1);
    run<11, 4, 2, PolicyFast, 1, 4>("16 values/thread, L=2048 B=4, 128 thr, 4 wg/CU", data, tw, 4);
    run<12, 2, 2, PolicyFast, 1>("B=2 NBUF=2, ONE workgroup per CU", data, tw, 1);
    run<12, 4, 2, PolicyFast, 1>("B=4 NBUF=2, ONE workgroup per CU", data, tw, 1);
    run<12, 1, 1, PolicyFast, 1>("row-like B=1 NBUF=1", data, tw, 4);
    run<12, 1, 1, PolicyParity, 1>("row-like B=1 NBUF=1 parity", data, tw, 4);
    run<12, 1, 2, PolicyFast, 1>("row-like B=1 NBUF=2", data, tw, 2);
    run<12, 1, 1, PolicyFast, 1>("row-like B=1 NBUF=1", data, tw, 1);
    run<12, 2, 2, PolicyFast, 1>("B=2 NBUF=2", data, tw, 2);
    run<12, 4, 2, PolicyFast, 4>("col-like B=4 NBUF=2 (128 vgpr)", data, tw, 2);
    run<12, 4, 2, PolicyParity, 4>("col-like B=4 NBUF=2 parity (128 vgpr)", data, tw, 2);
    run<12, 4, 2, PolicyFast, 1>("col-like B=4 NBUF=2 (free vgpr)", data, tw, 1);
    run<12, 4, 1, PolicyFast, 4>("col-like B=4 NBUF=1 (128 vgpr)", data, tw, 2);
    run<10, 4, 2, PolicyFast, 1>("L=1024 B=4 NBUF=2", data, tw, 8);
    run<11, 4, 2, PolicyFast, 1>("L=2048 B=4 NBUF=2", data, tw, 4);
    run<13, 1, 1, PolicyFast, 1>("L=8192 B=1 NBUF=1", data, tw, 2);
    return 0;
}
