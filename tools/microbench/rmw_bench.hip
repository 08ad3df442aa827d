// rmw_bench -- what the memory system gives pass B''s TRAFFIC (4 images x 64 MiB read and written in place + one shared
// 64 MiB filter read) when nothing but the traffic is left, in three shapes:
//   (a) elementwise : grid-stride, 16 bytes per lane per step, load a, load w, store a*w            (a copy's shape)
//   (b) tile        : one 256-thread workgroup per 128 KB tile, all 128 KB loaded, then w, then all stored (pass B''s shape)
//   (c) tile-out    : as (b) but the result goes to a second buffer (not in place)
//   (d) copy        : 4 x 64 MiB -> other buffer, no filter                                              (reference point)
//   (e) / (f) / (g) : read-only over all images / over one image; copy of one image (does the 256 MiB Infinity Cache show?)
//   (h) / (i)       : the inverse row passes' gather as 8-byte loads / as 32-byte rows of a panel, at 7 / 4 / 2 workgroups per CU
//   (j) / (k)       : pass A's input rows as one dword / one float4 per lane and load (6.7 vs 7.1 TB/s: not what bounds pass A)
//   usage: rmw_bench [images=4] [MiB per image=64] [iterations=20]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

__global__ void ew_kernel(float4* __restrict__ a, const float4* __restrict__ w, size_t n_img, int images) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_img; i += stride) {
        const float4 f = w[i];
        for (int k = 0; k < images; ++k) {
            float4 v = a[(size_t)k * n_img + i];
            v.x *= f.x; v.y *= f.y; v.z *= f.z; v.w *= f.w;
            a[(size_t)k * n_img + i] = v;
        }
    }
}
__global__ void read_kernel(const float4* __restrict__ a, float* __restrict__ sink, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345e-30f) sink[0] = acc;
}
// pass A's input shape: a workgroup reads 4 rows of 4096 floats, one DWORD per lane and load (64 loads per lane), and writes nothing
__global__ __launch_bounds__(256, 2) void rows_dword_kernel(const float* __restrict__ a, float* __restrict__ sink, int rows_total) {
    const size_t r0 = (size_t)blockIdx.x * 4;
    float acc = 0.f;
    float v[4][16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) v[r][q] = __builtin_nontemporal_load(a + (r0 + r) * 4096 + threadIdx.x + 256 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[r][q];
    if (acc == 1.2345e-30f) sink[0] = acc;
    (void)rows_total;
}
// the same bytes as one float4 per lane and load (16 loads per lane): lane t takes columns 4 (t & 63) + 256 q' .. of row t >> 6
__global__ __launch_bounds__(256, 2) void rows_f4_kernel(const float4* __restrict__ a, float* __restrict__ sink, int rows_total) {
    const size_t r0 = (size_t)blockIdx.x * 4;
    float acc = 0.f;
    float4 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = a[(r0 + (threadIdx.x >> 6)) * 1024 + (threadIdx.x & 63) + 64 * q];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
    if (acc == 1.2345e-30f) sink[0] = acc;
    (void)rows_total;
}
__global__ void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}
// tile = 8192 float4 (128 KB); workgroup b -> (image, tile) with the images of a tile neighbours on one XCD
template <bool INPLACE>
__global__ __launch_bounds__(256, 2) void tile_kernel(float4* __restrict__ a, float4* __restrict__ out, const float4* __restrict__ w, size_t n_img, int img_shift) {
    const int b = blockIdx.x, j = b >> 3;
    const int img = j & ((1 << img_shift) - 1);
    const size_t tl = (size_t)(((j >> img_shift) << 3) | (b & 7));
    const float4* src = a + (size_t)img * n_img + tl * 8192;
    const float4* ws = w + tl * 8192;
    float4* dst = (INPLACE ? a : out) + (size_t)img * n_img + tl * 8192;
    float4 v[32];
#pragma unroll
    for (int s = 0; s < 16; ++s) { v[2 * s] = src[s * 512 + threadIdx.x * 2]; v[2 * s + 1] = src[s * 512 + threadIdx.x * 2 + 1]; }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float4 f0 = ws[s * 512 + threadIdx.x * 2], f1 = ws[s * 512 + threadIdx.x * 2 + 1];
        v[2 * s].x *= f0.x; v[2 * s].y *= f0.y; v[2 * s].z *= f0.z; v[2 * s].w *= f0.w;
        v[2 * s + 1].x *= f1.x; v[2 * s + 1].y *= f1.y; v[2 * s + 1].z *= f1.z; v[2 * s + 1].w *= f1.w;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) { dst[s * 512 + threadIdx.x * 2] = v[2 * s]; dst[s * 512 + threadIdx.x * 2 + 1] = v[2 * s + 1]; }
}

// the inverse row passes' READ pattern: a 4-row group takes 128 bytes (4 rows x 4 columns) from each of 512 panels, PS elements
// apart.  gather8: as the product kernels, lane -> (panel, column), four 8-byte loads (the rows).  gather32: lane -> (panel, row),
// one 32-byte row of the panel per lane (4 lanes = one 128-byte line).
constexpr int kPS = 4 * 4096 + 16;
template <int LDSKB>
__global__ __launch_bounds__(256, 4) void gather8_kernel(const float2* __restrict__ a, float* __restrict__ sink, size_t img_elems) {
    __shared__ float pad[LDSKB * 256];  // LDSKB KB of LDS per workgroup: 37 -> four workgroups per CU, as the product kernels
    if (threadIdx.x == 1023) pad[blockIdx.x & 255] = 1.f;
    const float2* img = a + (size_t)blockIdx.y * img_elems;
    const unsigned t = threadIdx.x, r0 = blockIdx.x * 4;
    float2 y[4][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float2* p = img + (size_t)((t >> 2) + 64 * i) * kPS + (t & 3u) + r0 * 4u;
        y[0][i] = p[0]; y[1][i] = p[4]; y[2][i] = p[8]; y[3][i] = p[12];
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += y[0][i].x + y[1][i].y + y[2][i].x + y[3][i].y;
    if (acc == 1.2345e-30f) sink[0] = acc;
}
template <int LDSKB>
__global__ __launch_bounds__(256, 4) void gather32_kernel(const float2* __restrict__ a, float* __restrict__ sink, size_t img_elems) {
    __shared__ float pad[LDSKB * 256];
    if (threadIdx.x == 1023) pad[blockIdx.x & 255] = 1.f;
    const float2* img = a + (size_t)blockIdx.y * img_elems;
    const unsigned t = threadIdx.x, r0 = blockIdx.x * 4;
    float4 y[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4* p = reinterpret_cast<const float4*>(img + (size_t)((t >> 2) + 64 * i) * kPS + (r0 + (t & 3u)) * 4u);
        y[i][0] = p[0]; y[i][1] = p[1];
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += y[i][0].x + y[i][0].w + y[i][1].y + y[i][1].z;
    if (acc == 1.2345e-30f) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int images = argc > 1 ? std::atoi(argv[1]) : 4, mib = argc > 2 ? std::atoi(argv[2]) : 64, iters = argc > 3 ? std::atoi(argv[3]) : 20;
    const size_t n_img = (size_t)mib * 1024 * 1024 / 16;
    int shift = images == 1 ? 0 : images == 2 ? 1 : images == 4 ? 2 : 3;
    float4 *a, *b, *w;
    CK(hipMalloc((void**)&a, n_img * 16 * images)); CK(hipMalloc((void**)&b, n_img * 16 * images)); CK(hipMalloc((void**)&w, n_img * 16));
    CK(hipMemset(a, 0, n_img * 16 * images)); CK(hipMemset(b, 0, n_img * 16 * images)); CK(hipMemset(w, 0, n_img * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double bytes, auto&& launch) {
        std::vector<float> t;
        for (int r = 0; r < 5; ++r) {
            launch();
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) launch();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / iters);
        }
        std::sort(t.begin(), t.end());
        std::printf("  %-44s %8.2f us  %7.0f GB/s\n", name, t[2] * 1e3, bytes / (t[2] * 1e-3) / 1e9);
    };
    const double rmw_bytes = (double)n_img * 16 * (2 * images + 1);
    const int ntiles = (int)(n_img / 8192);
    std::printf("rmw_bench: %d images x %d MiB, filter %d MiB\n", images, mib, mib);
    for (int wg : {2048, 4096, 8192, 16384})
        timeit((std::string("(a) elementwise in place, grid ") + std::to_string(wg)).c_str(), rmw_bytes, [&] { hipLaunchKernelGGL(ew_kernel, dim3(wg), dim3(256), 0, 0, a, w, n_img, images); });
    timeit("(b) tile, in place", rmw_bytes, [&] { hipLaunchKernelGGL(tile_kernel<true>, dim3(ntiles * images), dim3(256), 0, 0, a, b, w, n_img, shift); });
    timeit("(c) tile, to a second buffer", rmw_bytes, [&] { hipLaunchKernelGGL(tile_kernel<false>, dim3(ntiles * images), dim3(256), 0, 0, a, b, w, n_img, shift); });
    timeit("(d) copy", (double)n_img * 16 * images * 2, [&] { hipLaunchKernelGGL(copy_kernel, dim3(8192), dim3(256), 0, 0, a, b, n_img * images); });
    timeit("(e) read only", (double)n_img * 16 * images, [&] { hipLaunchKernelGGL(read_kernel, dim3(8192), dim3(256), 0, 0, a, (float*)b, n_img * images); });
    timeit("(f) read only, ONE image (stays in the Infinity Cache?)", (double)n_img * 16, [&] { hipLaunchKernelGGL(read_kernel, dim3(8192), dim3(256), 0, 0, a, (float*)b, n_img); });
    timeit("(g) copy, ONE image", (double)n_img * 16 * 2, [&] { hipLaunchKernelGGL(copy_kernel, dim3(8192), dim3(256), 0, 0, a, b, n_img); });
    if (mib == 64 && (size_t)512 * kPS * 8 <= n_img * 16 + (size_t)images * 0) {
        // (an image of 512 panels x kPS float2 = 64.06 MiB: the images overlap by 64 KiB in this buffer, which only reads care about)
    }
    if (mib == 64) {
        const int rows = 4096 * images;
        timeit("(j) pass A's input: 4 rows per workgroup, one dword per lane and load", (double)rows * 4096 * 4, [&] { hipLaunchKernelGGL(rows_dword_kernel, dim3(rows / 4), dim3(256), 0, 0, (const float*)a, (float*)b, rows); });
        timeit("(k) the same rows, one float4 per lane and load", (double)rows * 4096 * 4, [&] { hipLaunchKernelGGL(rows_f4_kernel, dim3(rows / 4), dim3(256), 0, 0, (const float4*)a, (float*)b, rows); });
    }
    {
        const size_t img_elems = (size_t)512 * kPS;  // float2 elements per image
        float2* g; CK(hipMalloc((void**)&g, img_elems * 8 * images)); CK(hipMemset(g, 0, img_elems * 8 * images));
        const double bytes = (double)1024 * 512 * 128 * images;
        timeit("(h) row-group gather, 8 B per lane x 4 rows, 7 workgroups per CU", bytes, [&] { hipLaunchKernelGGL(gather8_kernel<1>, dim3(1024, images), dim3(256), 0, 0, g, (float*)b, img_elems); });
        timeit("(i) row-group gather, 32 B per lane,          7 workgroups per CU", bytes, [&] { hipLaunchKernelGGL(gather32_kernel<1>, dim3(1024, images), dim3(256), 0, 0, g, (float*)b, img_elems); });
        timeit("(h4) 8 B per lane x 4 rows, FOUR workgroups per CU (pass C1 / C2)", bytes, [&] { hipLaunchKernelGGL(gather8_kernel<37>, dim3(1024, images), dim3(256), 0, 0, g, (float*)b, img_elems); });
        timeit("(i4) 32 B per lane,         FOUR workgroups per CU", bytes, [&] { hipLaunchKernelGGL(gather32_kernel<37>, dim3(1024, images), dim3(256), 0, 0, g, (float*)b, img_elems); });
        timeit("(h2) 8 B per lane x 4 rows, TWO workgroups per CU", bytes, [&] { hipLaunchKernelGGL(gather8_kernel<74>, dim3(1024, images), dim3(256), 0, 0, g, (float*)b, img_elems); });
        timeit("(i2) 32 B per lane,         TWO workgroups per CU", bytes, [&] { hipLaunchKernelGGL(gather32_kernel<74>, dim3(1024, images), dim3(256), 0, 0, g, (float*)b, img_elems); });
        CK(hipFree(g));
    }
    return 0;
}
