// Access-pattern microbenchmark for the column pass design (DESIGN.md §"column pass").
// Not part of the product path: it answers "how narrow may a column tile be before
// HBM/L2 efficiency collapses on gfx950", with and without an XCD-aware tile mapping.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void copy_f4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}

// tile id for block b: natural or XCD-grouped so that the G tiles that share one
// 128-byte line (G = 16/C, C = columns per tile of 8-byte elements) sit on one XCD.
__device__ inline int tile_of_block(int b, int ntiles, int G, int swz) {
    if (!swz || G <= 1) return b;
    // blocks b, b+8, b+16.. share an XCD. group of 8*G blocks -> 8 lines, G tiles each
    int grp = b / (8 * G), r = b % (8 * G);
    int xcd = r % 8, j = r / 8;
    return grp * 8 * G + xcd * G + j;
}

// C columns (C*8 bytes) per tile, each lane moves 16 bytes (C>=2) or 8 bytes (C==1)
template <int C, int MODE>  // MODE 0 = read only, 1 = read-modify-write, 2 = write only
__global__ void coltile(float2* __restrict__ a, float* __restrict__ sink, int M, int N, int swz) {
    constexpr int LPR = (C >= 2) ? C / 2 : 1;       // lanes per row
    constexpr int G = (16 / C) > 0 ? (16 / C) : 1;
    int tile = tile_of_block(blockIdx.x, N / C, G, swz);
    int lane_c = threadIdx.x % LPR, lane_r = threadIdx.x / LPR;
    int rows_per_iter = blockDim.x / LPR;
    float acc = 0.f;
    for (int r = lane_r; r < M; r += rows_per_iter) {
        size_t idx = (size_t)r * N + (size_t)tile * C + lane_c * 2;
        if (C >= 2) {
            float4* p = reinterpret_cast<float4*>(a + idx);
            if (MODE == 0) { float4 v = *p; acc += v.x + v.y + v.z + v.w; }
            else if (MODE == 1) { float4 v = *p; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; *p = v; }
            else { *p = make_float4(r, tile, 1.f, 2.f); }
        } else {
            float2* p = a + (size_t)r * N + tile;
            if (MODE == 0) { float2 v = *p; acc += v.x + v.y; }
            else if (MODE == 1) { float2 v = *p; v.x += 1.f; v.y += 1.f; *p = v; }
            else { *p = make_float2(r, tile); }
        }
    }
    if (MODE == 0) sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// row WG writing to panel-major layout: element (m, n) -> (n/C)*M*C + m*C + n%C ; R rows per WG
template <int C>
__global__ void row_to_panel(float2* __restrict__ out, int M, int N, int R) {
    int m0 = blockIdx.x * R;
    for (int rr = 0; rr < R; ++rr) {
        int m = m0 + rr;
        for (int n = threadIdx.x * 2; n < N; n += blockDim.x * 2) {
            size_t idx = (size_t)(n / C) * M * C + (size_t)m * C + (n % C);
            if (C >= 2) *reinterpret_cast<float4*>(out + idx) = make_float4(m, n, 1.f, 2.f);
        }
    }
}

template <typename F>
float time_ms(F f, int iters = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <int C>
void run_col(float2* a, float* sink, int M, int N, int threads) {
    size_t bytes = (size_t)M * N * 8;
    for (int swz = 0; swz < 2; ++swz) {
        float r = time_ms([&] { coltile<C, 0><<<N / C, threads>>>(a, sink, M, N, swz); });
        float rw = time_ms([&] { coltile<C, 1><<<N / C, threads>>>(a, sink, M, N, swz); });
        float w = time_ms([&] { coltile<C, 2><<<N / C, threads>>>(a, sink, M, N, swz); });
        printf("coltile M=%d N=%d C=%2d (%3d B) thr=%d swz=%d : read %7.1f GB/s  rmw %7.1f GB/s (r+w)  write %7.1f GB/s\n",
               M, N, C, C * 8, threads, swz, bytes / r * 1e-6, 2.0 * bytes / rw * 1e-6, bytes / w * 1e-6);
        CK(hipGetLastError());
    }
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d\n", prop.name, prop.multiProcessorCount);
    for (int S : {4096, 8192}) {
        int M = S, N = S;
        size_t n = (size_t)M * N;
        float2 *a, *b; float* sink;
        CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&sink, 64 << 20));
        CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
        float t = time_ms([&] { copy_f4<<<2048, 256>>>((float4*)a, (float4*)b, n / 2); });
        printf("copy_f4 %zu MiB : %.1f GB/s (r+w)\n", n * 8 >> 20, 2.0 * n * 8 / t * 1e-6);
        run_col<1>(a, sink, M, N, 256);
        run_col<2>(a, sink, M, N, 256);
        run_col<4>(a, sink, M, N, 256);
        run_col<8>(a, sink, M, N, 256);
        run_col<16>(a, sink, M, N, 256);
        run_col<32>(a, sink, M, N, 256);
        run_col<4>(a, sink, M, N, 1024);
        run_col<8>(a, sink, M, N, 1024);
        for (int R : {1, 4}) {
            float t2 = time_ms([&] { row_to_panel<2><<<M / R, 256>>>(a, M, N, R); });
            float t4 = time_ms([&] { row_to_panel<4><<<M / R, 256>>>(a, M, N, R); });
            float t8 = time_ms([&] { row_to_panel<8><<<M / R, 256>>>(a, M, N, R); });
            float t16 = time_ms([&] { row_to_panel<16><<<M / R, 256>>>(a, M, N, R); });
            printf("row_to_panel S=%d R=%d : C=2 %7.1f  C=4 %7.1f  C=8 %7.1f  C=16 %7.1f GB/s (write)\n", S, R,
                   n * 8 / t2 * 1e-6, n * 8 / t4 * 1e-6, n * 8 / t8 * 1e-6, n * 8 / t16 * 1e-6);
        }
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(sink));
    }
    return 0;
}
