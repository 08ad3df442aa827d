// rmw_bench -- what the memory system gives pass B''s TRAFFIC (4 images x 64 MiB read and written in place + one shared
// 64 MiB filter read) when nothing but the traffic is left, in three shapes:
//   (a) elementwise : grid-stride, 16 bytes per lane per step, load a, load w, store a*w            (a copy's shape)
//   (b) tile        : one 256-thread workgroup per 128 KB tile, all 128 KB loaded, then w, then all stored (pass B''s shape)
//   (c) tile-out    : as (b) but the result goes to a second buffer (not in place)
//   (d) copy        : 4 x 64 MiB -> other buffer, no filter                                              (reference point)
//   (e) / (f) / (g) : read-only over all images / over one image; copy of one image (does the 256 MiB Infinity Cache show?)
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
    return 0;
}
