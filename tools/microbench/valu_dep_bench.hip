// valu_dep_bench -- what a DEPENDENT vector instruction costs on gfx950, by distance to its producer and by waves per SIMD.
//
// The passes' butterflies are chains  t = fma(vy, wr, u); a = fma(vx, w, t); b = fma(2, u, -a)  and hipcc, short of
// registers, recycles ONE temporary: the consumer sits one or two instructions behind its producer (ISA of
// fft_cols_panel_fused16_kernel<12>).  rocprofv3 attributes 55 % of pass B''s wave-cycles to SQ_WAIT_INST_ANY (issue
// stall) with every VMEM / LDS FIFO-full and instruction-cache counter near zero (profiles/r04_issue_counters_*.txt), which
// leaves the dependency itself.  This bench measures it: D independent chains of one instruction kind issued round robin
// (so a consumer is D instructions behind its producer), inline asm so that the order is the order written; cycles per
// instruction per wave from s_memtime, at 1, 2 and 4 waves per SIMD (block = 256 x waves threads, one block per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND> __device__ __forceinline__ void op(v2f& x, v2f a, v2f b) {
    if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x.x) : "v"(a.x), "v"(b.x));
    else if constexpr (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
}

template <int KIND, int D>
__global__ __launch_bounds__(1024) void dep_kernel(float* out, long long* cycles, int iters) {
    v2f x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = v2f{1.0f + threadIdx.x * 1e-9f + i, 0.5f};
    const v2f a = {0.999999f, 1.000001f}, b = {1e-7f, -1e-7f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48 / D; ++r)
#pragma unroll
            for (int i = 0; i < D; ++i) op<KIND>(x[i], a, b);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int D>
static double run(int waves_per_simd, int iters) {
    const int threads = 256 * waves_per_simd, blocks = 256;
    float* out; long long* cyc;
    CK(hipMalloc(&out, sizeof(float) * threads * blocks));
    CK(hipMalloc(&cyc, sizeof(long long) * blocks * (threads / 64)));
    hipLaunchKernelGGL((dep_kernel<KIND, D>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
    hipLaunchKernelGGL((dep_kernel<KIND, D>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(blocks * (threads / 64));
    CK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    double sum = 0;
    for (long long c : h) sum += (double)c;
    CK(hipFree(out)); CK(hipFree(cyc));
    const double n_inst = (double)iters * (48 / D) * D;
    return sum / h.size() / n_inst;  // s_memtime ticks (100 MHz constant clock on gfx9: reported as is) per instruction per wave
}

template <int KIND>
static void table(const char* name, int iters) {
    std::printf("%-14s distance:      1       2       3       4       6       8      12   (ticks of the cycle counter per instruction, per wave)\n", name);
    for (int w : {1, 2, 4}) {
        std::printf("  %d wave(s)/SIMD      %7.3f %7.3f %7.3f %7.3f %7.3f %7.3f %7.3f\n", w, run<KIND, 1>(w, iters), run<KIND, 2>(w, iters), run<KIND, 3>(w, iters),
                    run<KIND, 4>(w, iters), run<KIND, 6>(w, iters), run<KIND, 8>(w, iters), run<KIND, 12>(w, iters));
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
    // calibrate the cycle counter against wall time: a known count of independent v_fma_f32 is not needed -- report the
    // counter's frequency from hipDeviceAttributeWallClockRate / ClockRate so that ticks can be read as shader cycles
    int wall_khz = 0, clk_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    std::printf("wall clock rate %d kHz, shader clock rate %d kHz (s_memtime / readcyclecounter ticks: see MI355X_MICROARCH.md)\n", wall_khz, clk_khz);
    table<0>("v_pk_fma_f32", iters);
    table<1>("v_fma_f32", iters);
    table<2>("v_pk_add_f32", iters);
    table<3>("v_pk_mul_f32", iters);
    return 0;
}
