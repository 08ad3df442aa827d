// fdr_aux.hip -- auxiliary kernels: PSF generation, normalisation, synthetic input, the
// Wiener filter construction, the O(n^2) DFT, and the reference-shaped "simple path"
// (pad -> row FFT -> transpose -> row FFT -> transpose, fft/fft_gpu.cu:214-240) that serves as an
// on-device cross-check and as the fallback for dimensions below 8.
#include "fdr_fft_core.hpp"
#include "fdr_kernels.hpp"

namespace fdr {

// ---- preprocess_kernel equivalent (fft/fft_gpu.cu:85-103): real -> complex with zero padding ----
__global__ void pad_real_to_complex_kernel(const float* __restrict__ src, int rows, int cols, int stride,
                                           float2* __restrict__ dst, int M, int N) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x < N && y < M) {
        float p = 0.f;
        if (x < cols && y < rows) p = src[(size_t)y * stride + x];
        dst[(size_t)y * N + x] = make_float2(p, 0.f);
    }
}

hipError_t launch_pad_real_to_complex(const float* src, int rows, int cols, int stride, float2* dst, int M, int N,
                                      hipStream_t s) {
    const dim3 block(64, 4), grid((N + 63) / 64, (M + 3) / 4);
    hipLaunchKernelGGL(pad_real_to_complex_kernel, grid, block, 0, s, src, rows, cols, stride, dst, M, N);
    return hipGetLastError();
}

// ---- reference-shaped row FFT: whole row in LDS, explicit bit reversal, one radix-2 stage per
// barrier, flat butterfly index k (the shape of fft/fft_gpu.cu:108-148) but with the per-stage
// table so that parity mode reproduces fft/fft_serial.cpp:53-66 bit for bit.
template <class Pol>
__global__ void simple_rows_kernel(float2* __restrict__ data, int rows, int L, int logl, const float2* __restrict__ tw) {
    extern __shared__ float2 s_data[];  // tw: table of the requested direction (both modes)
    const int row = blockIdx.x;
    if (row >= rows) return;
    float2* p = data + (size_t)row * L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const int rev = logl ? (int)(__brev((unsigned)i) >> (32 - logl)) : 0;
        s_data[rev] = p[i];
    }
    __syncthreads();
    for (int len = 2; len <= L; len <<= 1) {
        const int half = len >> 1;
        for (int k = threadIdx.x; k < (L >> 1); k += blockDim.x) {
            const int off = k & (half - 1);
            const int ui = ((k - off) << 1) + off, vi = ui + half;
            float2 u = s_data[ui], v = s_data[vi];
            Pol::bfly(u, v, tw[(half - 1) + off]);
            s_data[ui] = u;
            s_data[vi] = v;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < L; i += blockDim.x) p[i] = s_data[i];
}

hipError_t launch_simple_rows(float2* data, int rows, int L, int logl, const float2* tw, int mode, hipStream_t s) {
    const size_t smem = (size_t)L * sizeof(float2);
    int threads = L / 2;
    if (threads < 64) threads = 64;
    if (threads > 1024) threads = 1024;
    if (smem > 48 * 1024) {
        // opt in to a 64 KiB dynamic LDS row (the reference never does).  A function attribute belongs to the
        // CURRENT device's copy of the code object, and fdr_batch_run drives one host thread per device: set it on every such
        // launch (a host-side table write, no device work) instead of remembering a process-wide "done" flag
        hipError_t e = mode == 0 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&simple_rows_kernel<PolicyParity>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)
                                 : hipFuncSetAttribute(reinterpret_cast<const void*>(&simple_rows_kernel<PolicyFast>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return e;
    }
    if (mode == 0)
        hipLaunchKernelGGL(simple_rows_kernel<PolicyParity>, dim3(rows), dim3(threads), smem, s, data, rows, L, logl, tw);
    else
        hipLaunchKernelGGL(simple_rows_kernel<PolicyFast>, dim3(rows), dim3(threads), smem, s, data, rows, L, logl, tw);
    return hipGetLastError();
}

// ---- transforms longer than one LDS row (L > 8192, a power of two): fft_serial::fft_radix2_inplace (fft/fft_serial.cpp:40-68)
// takes any power of two.  Its stages len = 2 .. L0 act inside aligned blocks of L0 = 8192 positions of the bit-reversed
// array, and block B of that array is the L0-point transform of the subsequence x[j S + bitrev(B)], S = L / L0: so the
// subsequences are gathered into blocks (long_gather_kernel), every block runs through the ordinary L0-point row kernels as a
// row of its own, and the remaining log2 S stages are plain butterflies over the whole row in global memory
// (long_stage_kernel), with the SAME per-stage twiddle table and butterfly as every other stage -- parity mode stays
// bit-identical to the serial recurrence.  Two extra passes over the data per transform plus one per stage above L0: the
// serial path "only gets slow" beyond 8192 points, and so does this one.
__global__ void long_gather_kernel(const float2* __restrict__ src, float2* __restrict__ dst, size_t rows, int L, int logs) {
    const int S = 1 << logs, L0 = L >> logs;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over rows x L destination elements
    if (idx >= rows * (size_t)L) return;
    const size_t row = idx / (size_t)L;
    const int pos = (int)(idx - row * (size_t)L);
    const int B = pos / L0, j = pos - B * L0;
    const int h = logs ? (int)(__brev((unsigned)B) >> (32 - logs)) : 0;
    dst[idx] = src[row * (size_t)L + (size_t)j * S + h];
}

template <class Pol>
__global__ void long_stage_kernel(const float2* src, float2* dst, size_t rows, int L, int half,  // (src may be dst: own pair only)
                                  const float2* __restrict__ tw) {  // tw: table of the requested direction, all stages
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over rows x L/2 butterflies
    const size_t per_row = (size_t)(L >> 1);
    if (idx >= rows * per_row) return;
    const size_t row = idx / per_row;
    const int k = (int)(idx - row * per_row);
    const int off = k & (half - 1);
    const size_t ui = row * (size_t)L + (size_t)(((k - off) << 1) + off), vi = ui + (size_t)half;
    float2 u = src[ui], v = src[vi];
    Pol::bfly(u, v, tw[(half - 1) + off]);
    dst[ui] = u;
    dst[vi] = v;
}

hipError_t launch_long_gather(const float2* src, float2* dst, size_t rows, int L, int logs, hipStream_t s) {
    const size_t n = rows * (size_t)L;
    hipLaunchKernelGGL(long_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, L, logs);
    return hipGetLastError();
}

hipError_t launch_long_stage(const float2* src, float2* dst, size_t rows, int L, int half, const float2* tw, int mode, hipStream_t s) {
    const size_t n = rows * (size_t)(L >> 1);
    if (mode == 0)
        hipLaunchKernelGGL(long_stage_kernel<PolicyParity>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, L, half, tw);
    else
        hipLaunchKernelGGL(long_stage_kernel<PolicyFast>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, L, half, tw);
    return hipGetLastError();
}

// ---- tile transpose through LDS (fft/fft_gpu.cu:153-164), 64-lane friendly 32x32 tile, +1 pad ----
__global__ void transpose_kernel(const float2* __restrict__ src, float2* __restrict__ dst, int rows, int cols) {
    __shared__ float2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    int x = blockIdx.x * 32 + tx;
    for (int j = ty; j < 32; j += 8) {
        const int y = blockIdx.y * 32 + j;
        if (x < cols && y < rows) tile[j][tx] = src[(size_t)y * cols + x];
    }
    __syncthreads();
    x = blockIdx.y * 32 + tx;
    for (int j = ty; j < 32; j += 8) {
        const int y = blockIdx.x * 32 + j;
        if (x < rows && y < cols) dst[(size_t)y * rows + x] = tile[tx][j];
    }
}

hipError_t launch_transpose(const float2* src, float2* dst, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, src, dst, rows, cols);
    return hipGetLastError();
}

// ---- Wiener quotient, pointwise (simple path); parity: fft/fft_serial.cpp:186-224 op order ----
__global__ void wiener_pointwise_kernel(float2* __restrict__ g, const float2* __restrict__ filt, size_t count, float K,
                                        int mode) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float2 G = g[i], h = filt[i];
    float2 o;
    if (mode == 0) {
        const float hr2 = h.x * h.x, hi2 = h.y * h.y;
        const float mag = sqrtf(hr2 + hi2);
        const float mag2 = mag * mag;
        const float denom = mag2 + K;
        const float chi = -h.y;
        const float p0 = G.x * h.x, p1 = G.y * chi, p2 = G.x * chi, p3 = G.y * h.x;
        const float nr = p0 - p1, ni = p2 + p3;
        o.x = denom != 0.0f ? nr / denom : 0.0f;
        o.y = denom != 0.0f ? ni / denom : 0.0f;
    } else {
        o.x = __builtin_fmaf(G.x, h.x, -(G.y * h.y));
        o.y = __builtin_fmaf(G.x, h.y, G.y * h.x);
    }
    g[i] = o;
}

hipError_t launch_wiener_pointwise(float2* g, const float2* filt, size_t count, float K, int mode, hipStream_t s) {
    hipLaunchKernelGGL(wiener_pointwise_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, g, filt, count, K,
                       mode);
    return hipGetLastError();
}

// ---- fast mode: W = conj(H) / (|H|^2 + K), evaluated in double, rounded once ----
__global__ void make_filter_fast_kernel(const float2* __restrict__ H, float2* __restrict__ W, size_t count, float K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    W[i] = wiener_filter_fast(H[i], K);
}

hipError_t launch_make_filter_fast(const float2* H, float2* W, size_t count, float K, hipStream_t s) {
    hipLaunchKernelGGL(make_filter_fast_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, H, W, count, K);
    return hipGetLastError();
}

// ---- simple path: real plane + min/max (postprocess_kernel, fft/fft_gpu.cu:187-201, unscaled) ----
__global__ void real_minmax_kernel(const float2* __restrict__ src, float* __restrict__ dst, int M, int N, int mm_rows,
                                   int mm_cols, float2* __restrict__ mm_part) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    if (x < N && y < M) {
        const float r = src[(size_t)y * N + x].x;
        dst[(size_t)y * N + x] = r;
        if (y < mm_rows && x < mm_cols) { mn = r; mx = r; }
    }
    block_minmax_store(mn, mx, mm_part);
}

hipError_t launch_real_minmax(const float2* src, float* dst, int M, int N, int mm_rows, int mm_cols, float2* mm_part,
                              int* n_part, hipStream_t s) {
    const dim3 grid((N + 255) / 256, M);
    *n_part = (int)(grid.x * grid.y);
    hipLaunchKernelGGL(real_minmax_kernel, grid, dim3(256), 0, s, src, dst, M, N, mm_rows, mm_cols, mm_part);
    return hipGetLastError();
}

// ---- final min/max over the per-workgroup partials: one workgroup, fixed order => deterministic ----
__global__ void reduce_minmax_kernel(const float2* __restrict__ part, int n, float* __restrict__ mm) {
    __shared__ float2 red[16];
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float2 p = part[i];
        mn = fminf(mn, p.x);
        mx = fmaxf(mx, p.y);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_float2(mn, mx);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            mn = fminf(mn, red[w].x);
            mx = fmaxf(mx, red[w].y);
        }
        mm[0] = mn;
        mm[1] = mx;
    }
}

hipError_t launch_reduce_minmax(const float2* mm_part, int n_part, float* mm, hipStream_t s) {
    hipLaunchKernelGGL(reduce_minmax_kernel, dim3(1), dim3(1024), 0, s, mm_part, n_part, mm);
    return hipGetLastError();
}

// ---- cv::normalize(src, dst, 0, 1, NORM_MINMAX) (fft/fft_serial.cpp:246) + crop (serial.cpp:38) ----
// scale/shift exactly as OpenCV 4.x derives them for CV_32F: double min/max, scale rounded to
// float, shift = (float)dmin - (float)(smin*scale); applied as a float multiply then a float add.
// Every workgroup first folds the (few thousand) per-workgroup min/max partials itself -- a fixed
// order, so the result is deterministic -- which saves a separate reduce launch (~4.4 us).

template <bool VEC4>
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ raw, int N, const float2* __restrict__ part,
                                                        int n_part, const float* __restrict__ mm, float* __restrict__ out,
                                                        int rows, int cols, int out_stride, const NormBatch nb) {
    if (nb.nimg > 1) {  // blockIdx.y = image
        const int i = blockIdx.y;
        // (direct member accesses: through pick_image's array reference this by-value kernel argument went to scratch
        // memory -- 208 bytes per lane and 150 instead of 95 us per 4-image launch at 4096^2)
#define FDR_PICK8(arr) (i < 4 ? (i == 0 ? arr[0] : i == 1 ? arr[1] : i == 2 ? arr[2] : arr[3]) : (i == 4 ? arr[4] : i == 5 ? arr[5] : i == 6 ? arr[6] : arr[7]))
        static_assert(kMaxGroup == 8, "select chain written for 8 entries");
        raw = FDR_PICK8(nb.raw);
        part = FDR_PICK8(nb.part);
        out = FDR_PICK8(nb.out);
#undef FDR_PICK8
    }
    __shared__ float2 red[4];
    float mn, mx;
    if (part != nullptr) {
        mn = __builtin_inff(); mx = -__builtin_inff();
        for (int i = threadIdx.x; i < n_part; i += 256) {
            const float2 p = part[i];
            mn = fminf(mn, p.x);
            mx = fmaxf(mx, p.y);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, off));
            mx = fmaxf(mx, __shfl_xor(mx, off));
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_float2(mn, mx);
        __syncthreads();
        mn = fminf(fminf(red[0].x, red[1].x), fminf(red[2].x, red[3].x));
        mx = fmaxf(fmaxf(red[0].y, red[1].y), fmaxf(red[2].y, red[3].y));
    } else {
        mn = mm[0]; mx = mm[1];
    }
    float fscale, fshift;
    minmax_to_scale_shift(mn, mx, fscale, fshift);

    constexpr int W = VEC4 ? 1024 : 256;               // elements per workgroup per segment
    const int segs_per_row = (cols + W - 1) / W;
    const long long nseg = (long long)rows * segs_per_row;
    for (long long sgi = blockIdx.x; sgi < nseg; sgi += gridDim.x) {
        const int y = (int)(sgi / segs_per_row);
        const int x = (int)(sgi % segs_per_row) * W + threadIdx.x * (VEC4 ? 4 : 1);
        if (VEC4) {
            if (x < cols) {  // cols % 4 == 0
                typedef float nf4 __attribute__((ext_vector_type(4)));
                const nf4 vv = __builtin_nontemporal_load(reinterpret_cast<const nf4*>(raw + (size_t)y * N + x));  // last use
                const float4 v = make_float4(vv.x, vv.y, vv.z, vv.w);
                float4 o;
                o.x = v.x * fscale; o.y = v.y * fscale; o.z = v.z * fscale; o.w = v.w * fscale;
                o.x = o.x + fshift; o.y = o.y + fshift; o.z = o.z + fshift; o.w = o.w + fshift;
                nf4 oo; oo.x = o.x; oo.y = o.y; oo.z = o.z; oo.w = o.w;
                __builtin_nontemporal_store(oo, reinterpret_cast<nf4*>(out + (size_t)y * out_stride + x));  // written once, read by the caller
            }
        } else {
            if (x < cols) {
                const float p = raw[(size_t)y * N + x] * fscale;
                out[(size_t)y * out_stride + x] = p + fshift;
            }
        }
    }
}

hipError_t launch_normalize(const float* raw, int N, const float2* mm_part, int n_part, const float* mm, float* out,
                            int rows, int cols, int out_stride, hipStream_t s, const NormBatch* batch) {
    NormBatch nb{};
    if (batch) nb = *batch;
    const int ny = nb.nimg > 1 ? nb.nimg : 1;
    if (rows <= 0 || cols <= 0) return hipSuccess;
    bool vec4 = (cols % 4 == 0) && (out_stride % 4 == 0) && (N % 4 == 0) &&
                ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && ((reinterpret_cast<uintptr_t>(raw) & 15) == 0);
    for (int k = 0; k < nb.nimg; ++k)
        vec4 = vec4 && ((reinterpret_cast<uintptr_t>(nb.out[k]) & 15) == 0) && ((reinterpret_cast<uintptr_t>(nb.raw[k]) & 15) == 0);
    const int W = vec4 ? 1024 : 256;
    long long nseg = (long long)rows * ((cols + W - 1) / W);
    int grid = nseg > 2048 ? 2048 : (int)nseg;
    if (vec4)
        hipLaunchKernelGGL(normalize_kernel<true>, dim3(grid, ny), dim3(256), 0, s, raw, N, mm_part, n_part, mm, out, rows, cols, out_stride, nb);
    else
        hipLaunchKernelGGL(normalize_kernel<false>, dim3(grid, ny), dim3(256), 0, s, raw, N, mm_part, n_part, mm, out, rows, cols, out_stride, nb);
    return hipGetLastError();
}

// ---- the same normalisation from a PANEL-major real plane (parity operator since round 4): a workgroup takes 16 rows x 64
// columns, reads 16 panels x (16 rows x 16 bytes = 256 contiguous bytes), turns the block through LDS and writes 16 rows x
// 256 contiguous bytes.  scale / shift and the two roundings exactly as normalize_kernel.
template <bool VEC4>
__global__ __launch_bounds__(256) void normalize_panels_kernel(const float* __restrict__ raw, int M, const float2* __restrict__ part, int n_part,
                                                               const float* __restrict__ mm, float* __restrict__ out, int rows, int cols,
                                                               int out_stride) {
    __shared__ float tile[32][132];  // 32 rows x 128 columns (+4: the transposing accesses fall on distinct banks)
    __shared__ float2 red[4];
    float mn, mx;
    if (part != nullptr) {
        mn = __builtin_inff(); mx = -__builtin_inff();
        for (int i = threadIdx.x; i < n_part; i += 256) {
            const float2 p = part[i];
            mn = fminf(mn, p.x);
            mx = fmaxf(mx, p.y);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, off));
            mx = fmaxf(mx, __shfl_xor(mx, off));
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_float2(mn, mx);
        __syncthreads();
        mn = fminf(fminf(red[0].x, red[1].x), fminf(red[2].x, red[3].x));
        mx = fmaxf(fmaxf(red[0].y, red[1].y), fmaxf(red[2].y, red[3].y));
    } else {
        mn = mm[0]; mx = mm[1];
    }
    float fscale, fshift;
    minmax_to_scale_shift(mn, mx, fscale, fshift);
    typedef float nf4 __attribute__((ext_vector_type(4)));
    const int cblocks = (cols + 127) / 128, rblocks = (rows + 31) / 32;
    for (long long bi = blockIdx.x; bi < (long long)cblocks * rblocks; bi += gridDim.x) {
        const int rb = (int)(bi / cblocks) * 32, cb = (int)(bi % cblocks) * 128;
        // read: 32 panels x 32 rows of 16 bytes; a wave takes two panels x 32 rows = two runs of 512 contiguous bytes
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = (int)threadIdx.x + 256 * k;  // 0 .. 1023
            const int pi = e >> 5, ri = e & 31;
            const int m = rb + ri, c0 = cb + pi * 4;
            nf4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < rows && c0 < cols) v = __builtin_nontemporal_load(reinterpret_cast<const nf4*>(raw + ((size_t)(c0 >> 2) * (size_t)M + (size_t)m) * 4));  // last use
            *reinterpret_cast<float4*>(&tile[ri][pi * 4]) = make_float4(v.x, v.y, v.z, v.w);
        }
        __syncthreads();
        // write: a row of the block is 128 columns = 512 contiguous bytes = 32 lanes x 16 bytes
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = (int)threadIdx.x + 256 * k;
            const int ri = e >> 5, ci = (e & 31) * 4;
            const int m = rb + ri, c = cb + ci;
            if (m < rows && c < cols) {
                const float4 t = *reinterpret_cast<const float4*>(&tile[ri][ci]);
                float4 o;
                o.x = t.x * fscale; o.y = t.y * fscale; o.z = t.z * fscale; o.w = t.w * fscale;
                o.x = o.x + fshift; o.y = o.y + fshift; o.z = o.z + fshift; o.w = o.w + fshift;
                float* dst = out + (size_t)m * out_stride + c;
                if (VEC4) {  // cols % 4 == 0, rows of the output 16-byte aligned
                    nf4 oo; oo.x = o.x; oo.y = o.y; oo.z = o.z; oo.w = o.w;
                    __builtin_nontemporal_store(oo, reinterpret_cast<nf4*>(dst));  // written once, read by the caller
                } else {
                    dst[0] = o.x;
                    if (c + 1 < cols) dst[1] = o.y;
                    if (c + 2 < cols) dst[2] = o.z;
                    if (c + 3 < cols) dst[3] = o.w;
                }
            }
        }
        __syncthreads();
    }
}

hipError_t launch_normalize_panels(const float* raw, int M, int N, const float2* mm_part, int n_part, const float* mm, float* out,
                                   int rows, int cols, int out_stride, hipStream_t s) {
    (void)N;
    if (rows <= 0 || cols <= 0) return hipSuccess;
    const long long nb = (long long)((cols + 127) / 128) * ((rows + 31) / 32);
    const int grid = nb > 4096 ? 4096 : (int)nb;
    const bool vec4 = (cols % 4 == 0) && (out_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (vec4)
        hipLaunchKernelGGL(normalize_panels_kernel<true>, dim3(grid), dim3(256), 0, s, raw, M, mm_part, n_part, mm, out, rows, cols, out_stride);
    else
        hipLaunchKernelGGL(normalize_panels_kernel<false>, dim3(grid), dim3(256), 0, s, raw, M, mm_part, n_part, mm, out, rows, cols, out_stride);
    return hipGetLastError();
}

// ---- utils.hpp:15-24 motionBlurKernel on the device ----
// The source kernel (row size/2 set to 1/size) is analytic; the inverse affine map is prepared on
// the host in double exactly as cv::getRotationMatrix2D + cv::warpAffine do, and each destination
// pixel replays WarpAffineInvoker's 10-bit fixed-point coordinates and remapBilinear's 32x32
// float weights with BORDER_CONSTANT 0.
struct PsfMap { double m[6]; };

__device__ __forceinline__ int cv_round_dev(double v) {
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return __double2int_rn(v);
}

__global__ void psf_motion_kernel(int size, PsfMap map, float* __restrict__ out) {
    const float line = (float)(1.0 / (double)size);
    const int cy = size / 2;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < size * size; idx += gridDim.x * blockDim.x) {
        const int y = idx / size, x = idx % size;
        const int X0 = cv_round_dev((map.m[1] * y + map.m[2]) * 1024.0) + 16;
        const int Y0 = cv_round_dev((map.m[4] * y + map.m[5]) * 1024.0) + 16;
        const int adelta = cv_round_dev(map.m[0] * x * 1024.0), bdelta = cv_round_dev(map.m[3] * x * 1024.0);
        const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
        const int sx = X >> 5, sy = Y >> 5, ax = X & 31, ay = Y & 31;
        const float fx = ax * (1.f / 32.f), fy = ay * (1.f / 32.f);
        const float vx0 = 1.f - fx, vx1 = fx, vy0 = 1.f - fy, vy1 = fy;
        const float w0 = vy0 * vx0, w1 = vy0 * vx1, w2 = vy1 * vx0, w3 = vy1 * vx1;
        const bool x0in = sx >= 0 && sx < size, x1in = sx + 1 >= 0 && sx + 1 < size;
        const float s00 = (sy == cy && x0in) ? line : 0.f, s01 = (sy == cy && x1in) ? line : 0.f;
        const float s10 = (sy + 1 == cy && x0in) ? line : 0.f, s11 = (sy + 1 == cy && x1in) ? line : 0.f;
        const float t0 = s00 * w0, t1 = s01 * w1, t2 = s10 * w2, t3 = s11 * w3;
        float acc = t0 + t1;
        acc = acc + t2;
        acc = acc + t3;
        out[idx] = acc;
    }
}

hipError_t launch_psf_motion(int size, double angle_deg, float* d_out, hipStream_t s) {
    const double PI = 3.1415926535897932384626433832795;
    const double a = angle_deg * PI / 180.0;
    const double alpha = cos(a), beta = sin(a);
    const double cx = (double)(float)(size / 2), cy = (double)(float)(size / 2);
    double M[6] = {alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy};
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    PsfMap map;
    for (int i = 0; i < 6; ++i) map.m[i] = M[i];
    int blocks = (size * size + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(psf_motion_kernel, dim3(blocks), dim3(256), 0, s, size, map, d_out);
    return hipGetLastError();
}

// ---- cv::warpAffine(src, dst, M, dsize) with its defaults (INTER_LINEAR, BORDER_CONSTANT 0) for a single-channel float
// image, as utils.hpp:22 calls it: the same fixed-point replay as psf_motion_kernel, the source read from memory.
// `map` is the INVERTED matrix (dst -> src), prepared on the host in double as cv::warpAffine does.
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__global__ void warp_affine_kernel(const float* __restrict__ src, int srows, int scols, int sstride, PsfMap map, float* __restrict__ dst,
                                   int drows, int dcols, int dstride) {
    const long long total = (long long)drows * dcols;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(idx / dcols), x = (int)(idx % dcols);
        const int X0 = cv_round_dev((map.m[1] * y + map.m[2]) * 1024.0) + 16;
        const int Y0 = cv_round_dev((map.m[4] * y + map.m[5]) * 1024.0) + 16;
        const int adelta = cv_round_dev(map.m[0] * x * 1024.0), bdelta = cv_round_dev(map.m[3] * x * 1024.0);
        const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
        const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5), ax = X & 31, ay = Y & 31;
        const float fx = ax * (1.f / 32.f), fy = ay * (1.f / 32.f);
        const float vx0 = 1.f - fx, vx1 = fx, vy0 = 1.f - fy, vy1 = fy;
        const float w0 = vy0 * vx0, w1 = vy0 * vx1, w2 = vy1 * vx0, w3 = vy1 * vx1;
        const bool x0in = sx >= 0 && sx < scols, x1in = sx + 1 >= 0 && sx + 1 < scols;
        const bool y0in = sy >= 0 && sy < srows, y1in = sy + 1 >= 0 && sy + 1 < srows;
        const float s00 = (y0in && x0in) ? src[(size_t)sy * sstride + sx] : 0.f;
        const float s01 = (y0in && x1in) ? src[(size_t)sy * sstride + sx + 1] : 0.f;
        const float s10 = (y1in && x0in) ? src[(size_t)(sy + 1) * sstride + sx] : 0.f;
        const float s11 = (y1in && x1in) ? src[(size_t)(sy + 1) * sstride + sx + 1] : 0.f;
        const float t0 = s00 * w0, t1 = s01 * w1, t2 = s10 * w2, t3 = s11 * w3;
        float acc = t0 + t1;
        acc = acc + t2;
        acc = acc + t3;
        dst[(size_t)y * dstride + x] = acc;
    }
}

hipError_t launch_warp_affine(const float* src, int srows, int scols, int sstride, const double fwd[6], float* dst, int drows, int dcols,
                              int dstride, hipStream_t s) {
    if (drows <= 0 || dcols <= 0) return hipSuccess;
    double M[6];
    for (int i = 0; i < 6; ++i) M[i] = fwd[i];
    double D = M[0] * M[4] - M[1] * M[3];  // invertAffineTransform inside cv::warpAffine
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    PsfMap map;
    for (int i = 0; i < 6; ++i) map.m[i] = M[i];
    long long blocks = ((long long)drows * dcols + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(warp_affine_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, srows, scols, sstride, map, dst, drows, dcols, dstride);
    return hipGetLastError();
}

// ---- counter-based synthetic image: top 24 bits of splitmix64(seed + first + i) / 2^24 ----
__global__ void synth_kernel(uint64_t seed, uint64_t first, size_t count, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = seed + first + i;
        x += 0x9E3779B97F4A7C15ULL;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
        x = x ^ (x >> 31);
        out[i] = (float)(x >> 40) * (1.0f / 16777216.0f);
    }
}

hipError_t launch_synth(uint64_t seed, uint64_t first, size_t count, float* d_out, hipStream_t s) {
    if (count == 0) return hipSuccess;
    size_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, s, seed, first, count, d_out);
    return hipGetLastError();
}

// ---- fft_serial::dft_naive_inplace (fft/fft_serial.cpp:71-87): one thread per output k, terms accumulated in the
// reference's order t = 0..n-1, every product and sum rounded separately (-ffp-contract=off).
// Table form (bit parity): table[t * n + k] = (cosf(ang), sinf(ang)) with ang = (float)(2.0f*CV_PI*k*t/n*sign) evaluated
// left to right in double -- generated on the HOST with the C library's cosf / sinf, the functions the serial path
// itself calls (the device's own cosf / sinf differ from them by up to 2 ulp).  Forward table only: the inverse angle
// is the exact negation, cosf is even and sinf odd.  Batched over rows (blockIdx.y); src and dst must differ.
__global__ void dft_naive_rows_kernel(const float2* __restrict__ src, float2* __restrict__ dst, int n, const float2* __restrict__ table,
                                      int inverse) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float2* __restrict__ row = src + (size_t)blockIdx.y * n;
    float sr = 0.f, si = 0.f;
    for (int t = 0; t < n; ++t) {
        const float2 w = table[(size_t)t * n + k];
        const float wr = w.x, wi = inverse ? -w.y : w.y;
        const float2 a = row[t];
        const float pr = a.x * wr - a.y * wi, pi = a.x * wi + a.y * wr;
        sr += pr;
        si += pi;
    }
    dst[(size_t)blockIdx.y * n + k] = make_float2(sr, si);
}

// the same with the angle's cosine and sine evaluated on the device (lengths whose n x n table would be too large):
// within 2 ulp per twiddle of the table form
__global__ void dft_naive_kernel(const float2* __restrict__ src, float2* __restrict__ dst, int n, int inverse) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double sign = inverse ? 1.0 : -1.0;
    float sr = 0.f, si = 0.f;
    for (int t = 0; t < n; ++t) {
        const float ang = (float)(2.0 * 3.1415926535897932384626433832795 * (double)k * (double)t / (double)n * sign);
        const float wr = cosf(ang), wi = sinf(ang);
        const float2 a = src[t];
        const float pr = a.x * wr - a.y * wi, pi = a.x * wi + a.y * wr;
        sr += pr;
        si += pi;
    }
    dst[k] = make_float2(sr, si);
}

hipError_t launch_dft_naive(const float2* src, float2* dst, int n, int inverse, hipStream_t s) {
    hipLaunchKernelGGL(dft_naive_kernel, dim3((n + 127) / 128), dim3(128), 0, s, src, dst, n, inverse);
    return hipGetLastError();
}

hipError_t launch_dft_naive_rows(const float2* src, float2* dst, int rows, int n, const float2* table, int inverse, hipStream_t s) {
    if (rows <= 0 || n <= 0) return hipSuccess;
    hipLaunchKernelGGL(dft_naive_rows_kernel, dim3((n + 127) / 128, rows), dim3(128), 0, s, src, dst, n, table, inverse);
    return hipGetLastError();
}

// ---- building blocks of the single-image multi-GPU mode (slab decomposition; the reference's
// fft/fft_mpi.cpp:170-307): pack the column blocks of a row slab for the all-to-all, transpose what came back ----
struct SlabParts { int parts; int counts[16]; int displs[16]; };

// dst = [block 0 | block 1 | ...], block p = src[:, displs[p] : displs[p] + counts[p]] stored row-major (rows x counts[p]);
// the send buffer of fft/fft_mpi.cpp:118-135 (displs are the prefix sums of counts, so block p starts at rows * displs[p])
template <class E>
__global__ void slab_pack_kernel(const E* __restrict__ src, int rows, int ld, SlabParts sp, E* __restrict__ dst) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= ld || r >= rows) return;
    int p = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k)
        if (k < sp.parts && c >= sp.displs[k]) p = k;
    dst[(size_t)rows * sp.displs[p] + (size_t)r * sp.counts[p] + (c - sp.displs[p])] = src[(size_t)r * ld + c];
}

template <class E>
__global__ void transpose_any_kernel(const E* __restrict__ src, E* __restrict__ dst, int rows, int cols) {
    __shared__ E tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    int x = blockIdx.x * 32 + tx;
    for (int j = ty; j < 32; j += 8) {
        const int y = blockIdx.y * 32 + j;
        if (x < cols && y < rows) tile[j][tx] = src[(size_t)y * cols + x];
    }
    __syncthreads();
    x = blockIdx.y * 32 + tx;
    for (int j = ty; j < 32; j += 8) {
        const int y = blockIdx.x * 32 + j;
        if (x < rows && y < cols) dst[(size_t)y * rows + x] = tile[tx][j];
    }
}

__global__ void real_part_kernel(const float2* __restrict__ src, float* __restrict__ dst, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = src[i].x;
}

// (min, max) partials of a real rows x ld plane over the counted window [0, mm_rows) x [0, mm_cols)
__global__ void minmax_real_kernel(const float* __restrict__ src, int rows, int ld, int mm_rows, int mm_cols, float2* __restrict__ part) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    if (x < ld && y < rows && y < mm_rows && x < mm_cols) { mn = mx = src[(size_t)y * ld + x]; }
    block_minmax_store(mn, mx, part);
}

hipError_t launch_slab_pack(const void* src, int rows, int ld, int parts, const int* counts, int elem_size, void* dst, hipStream_t s) {
    if (parts < 1 || parts > 16 || (elem_size != 4 && elem_size != 8)) return hipErrorInvalidValue;
    SlabParts sp{};
    sp.parts = parts;
    int d = 0;
    for (int k = 0; k < parts; ++k) {
        if (counts[k] < 0) return hipErrorInvalidValue;  // (a negative block could still sum to ld)
        sp.counts[k] = counts[k]; sp.displs[k] = d; d += counts[k];
    }
    if (d != ld) return hipErrorInvalidValue;
    if (rows <= 0 || ld <= 0) return hipSuccess;
    const dim3 grid((ld + 255) / 256, rows), block(256);
    if (elem_size == 8) hipLaunchKernelGGL(slab_pack_kernel<float2>, grid, block, 0, s, (const float2*)src, rows, ld, sp, (float2*)dst);
    else hipLaunchKernelGGL(slab_pack_kernel<float>, grid, block, 0, s, (const float*)src, rows, ld, sp, (float*)dst);
    return hipGetLastError();
}

hipError_t launch_transpose_any(const void* src, void* dst, int rows, int cols, int elem_size, hipStream_t s) {
    if (elem_size != 4 && elem_size != 8) return hipErrorInvalidValue;
    if (rows <= 0 || cols <= 0) return hipSuccess;
    const dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(256);
    if (elem_size == 8) hipLaunchKernelGGL(transpose_any_kernel<float2>, grid, block, 0, s, (const float2*)src, (float2*)dst, rows, cols);
    else hipLaunchKernelGGL(transpose_any_kernel<float>, grid, block, 0, s, (const float*)src, (float*)dst, rows, cols);
    return hipGetLastError();
}

hipError_t launch_real_part(const float2* src, float* dst, size_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(real_part_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, src, dst, count);
    return hipGetLastError();
}

hipError_t launch_minmax_real(const float* src, int rows, int ld, int mm_rows, int mm_cols, float2* part, int* n_part, hipStream_t s) {
    const dim3 grid((ld + 255) / 256, rows);
    *n_part = (int)(grid.x * grid.y);
    hipLaunchKernelGGL(minmax_real_kernel, grid, dim3(256), 0, s, src, rows, ld, mm_rows, mm_cols, part);
    return hipGetLastError();
}

}  // namespace fdr
