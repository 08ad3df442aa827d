// fdr_cols.hip -- column passes without a transpose: a thread group keeps B = 4 adjacent columns
// (32 contiguous bytes per row) entirely in registers, 8 values per column per thread, and uses
// LDS only as the inter-step exchange stage (two alternating buffers, one barrier per column).
// This replaces transpose_kernel_opt + fft_row_optimized_kernel + transpose_kernel_opt of the
// reference (fft/fft_gpu.cu:153-164, 214-240) and, in fast mode, also wiener_kernel (:169-181).
//
// A 128-byte line of the row-major M x N array is shared by 16/(4*G) workgroups; the tile map
// below places those workgroups on the same XCD (blocks b, b+8, b+16.. share an L2).  This only
// affects speed, and it is not enough: with >= 128 KB of lines in flight per CU the sharing cannot
// be served from a 4 MiB L2, which is why the fast mode moved to the panel-major layout of
// fdr_panel.hip.  These kernels remain the parity-mode column passes (reference pass order) and the
// column half of fdr_fft2d_c2c in both modes.
#include "fdr_fft_core.hpp"
#include "fdr_kernels.hpp"
#include <type_traits>

namespace fdr {

template <int LOGM>
struct ColGeom {
    static constexpr int T = Steps<LOGM>::T;
    static constexpr int B = 4;                                      // columns per thread group
    static constexpr int G = T >= 512 ? 1 : (T >= 256 ? 2 : 4);      // thread groups per workgroup
    static constexpr int THREADS = T * G;
    static constexpr int COLS = B * G;                               // columns per workgroup
    static constexpr int SHARE = 16 / COLS;                          // workgroups per 128-byte line
    // two 512-thread workgroups per CU (4 waves/SIMD) need <= 128 VGPRs; LDS (2 x 74 KB) allows it
    static constexpr int WAVES_PER_SIMD = THREADS >= 512 ? 4 : 1;
};

// Addressing discipline for the column tiles: every access is  uniform_base[thread_off]  with a
// wave-uniform 64-bit base (row block q, kept in SGPRs) and ONE unsigned 32-bit per-thread element
// offset ((tid + T u) * N + col0), so the 16 wide loads of a tile share a single address VGPR
// (global_load_dwordx4 v, v_off, s[base]) instead of 16 64-bit pointers.  M*N*8 < 4 GiB here.
// fft/fft_serial.cpp:186-224 in the reference's operation order, every op rounded separately:
// mag = sqrt(Hr^2 + Hi^2); denom = mag*mag + K; num = G * conj(H); out = num / denom.
__device__ __forceinline__ float2 wiener_parity(float2 g, float2 h, float K) {
    const float hr2 = h.x * h.x, hi2 = h.y * h.y;
    const float mag = sqrtf(hr2 + hi2);
    const float mag2 = mag * mag;
    const float denom = mag2 + K;
    const float chi = -h.y;
    const float p0 = g.x * h.x, p1 = g.y * chi, p2 = g.x * chi, p3 = g.y * h.x;
    const float nr = p0 - p1, ni = p2 + p3;
    float2 o;
    o.x = denom != 0.0f ? nr / denom : 0.0f;  // cv::divide yields 0 on a zero divisor
    o.y = denom != 0.0f ? ni / denom : 0.0f;
    return o;
}
// PANEL = 1 (parity operator since round 4): the array is panel-major -- a thread group's four columns are ONE contiguous
// M x 32-byte block, whole 128-byte lines per quad of lanes -- instead of 32 bytes of every row of the row-major array.
template <int LOGM, class Pol, int KIND, int PANEL>
__global__ __launch_bounds__(ColGeom<LOGM>::THREADS, ColGeom<LOGM>::WAVES_PER_SIMD) void fft_cols_kernel(const ColArgs a, const float2* __restrict__ tw_fwd,
                                                                          const float2* __restrict__ tw_inv) {
    using St = Steps<LOGM>;
    using Geo = ColGeom<LOGM>;
    constexpr int B = Geo::B, G = Geo::G, T = St::T;
    using Core = FftCore<LOGM, B, 2, Pol>;
    __shared__ float2 lds[G * 2 * St::BUF];

    const int N = a.N;
    const int ntiles = (N + Geo::COLS - 1) / Geo::COLS;
    const int tile = PANEL ? (int)blockIdx.x : col_tile_of_block(blockIdx.x, ntiles, Geo::SHARE);
    const int g = threadIdx.x >> St::LOGT, tid = threadIdx.x & (T - 1);
    const int col0 = (tile * G + g) * B;
    const bool active = col0 < N;  // N is a multiple of 4 on this path
    // element offset of (row m, this group's first column): row-major m N + col0; panel-major panel * pstride + 4 m
    // (one base per thread group + an unsigned 32-bit element offset per row: M N < 2^32 elements)
    const size_t pbase = PANEL ? (size_t)(col0 >> 2) * a.pstride : (size_t)col0;
    const unsigned rstep = PANEL ? 4u : (unsigned)N;
    float2* __restrict__ gdata = a.data + pbase;
    float2* grp_lds = lds + g * 2 * St::BUF;

    constexpr bool kInverseOnly = (KIND == COL_INV || KIND == COL_INV_REAL);
    // parity: direction-specific table; fast: forward table, conjugated in registers when INV
    const float2* __restrict__ tw0 = (kInverseOnly && !Pol::kHoist) ? tw_inv : tw_fwd;
    typename Core::Bases bases;
    Core::init_bases(bases, tw_fwd, tid);

    float2 v[B][8];
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHO0; ++q) {
            const int m = Core::in_index(tid, u, q);
            const int s = u * Core::RHO0 + q;
            if (active) load4(gdata + (unsigned)m * rstep, v[0][s], v[1][s], v[2][s], v[3][s]);
            else v[0][s] = v[1][s] = v[2][s] = v[3][s] = make_float2(0.f, 0.f);
        }

    Core::template run<0, kInverseOnly>(v, grp_lds, tw0, bases, tid);

    if (KIND == COL_FWD || KIND == COL_INV) {
        if (active) {
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) {
                    const int s = u * Core::RHOL + q;
                    store4(gdata + (unsigned)Core::out_index(tid, u, q) * rstep, v[0][s], v[1][s], v[2][s], v[3][s]);
                }
        }
    } else if (KIND == COL_FWD_WIENER) {
        // The quotient needs IEEE square roots and divisions -- a dozen temporaries each -- beside the 64 registers of the tile,
        // and at 128 registers per lane hipcc spilled 15-39 of them (PMC: +23 % HBM writes, +12 % reads in this pass).  The
        // exchange buffers are idle by now: the second half of the tile (slots 4..7) waits THERE while the first half is
        // divided, 64 KB of LDS traffic per tile instead of scratch memory.  Each thread reads back only what it wrote.
        constexpr int HALF = 4;
        static_assert(2 * St::BUF >= 4 * HALF * T, "the parked half tile fits the group's exchange buffers");
        __syncthreads();  // every wave has read its last exchange values
#pragma unroll
        for (int s = HALF; s < 8; ++s)
#pragma unroll
            for (int b = 0; b < B; ++b) grp_lds[((s - HALF) * B + b) * T + tid] = v[b][s];
        auto quotient_rows = [&](int s0) {
#pragma unroll
            for (int s = s0; s < s0 + HALF; ++s) {
                const int u = s / Core::RHOL, q = s % Core::RHOL;
                const unsigned off = (unsigned)Core::out_index(tid, u, q) * rstep;
                float2 h0, h1, h2, h3;
                if (active) {
                    load4(a.filt + pbase + off, h0, h1, h2, h3);
                    store4(gdata + off, wiener_parity(v[0][s], h0, a.K), wiener_parity(v[1][s], h1, a.K),
                           wiener_parity(v[2][s], h2, a.K), wiener_parity(v[3][s], h3, a.K));
                }
                asm volatile("" ::: "memory");  // (one row of quotients at a time)
            }
        };
        quotient_rows(0);
#pragma unroll
        for (int s = HALF; s < 8; ++s)
#pragma unroll
            for (int b = 0; b < B; ++b) v[b][s] = grp_lds[((s - HALF) * B + b) * T + tid];
        quotient_rows(HALF);
    } else if (KIND == COL_INV_REAL) {
        float mn = __builtin_inff(), mx = -__builtin_inff();
        if (active) {
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) {
                    const int s = u * Core::RHOL + q;
                    const int m = Core::out_index(tid, u, q);
                    const float4 r = make_float4(v[0][s].x, v[1][s].x, v[2][s].x, v[3][s].x);
                    // (real plane: row-major, or panel-major with panels exactly 4 M floats apart -- it fits the M x N plane)
                    if (PANEL) *reinterpret_cast<float4*>(a.dst_real + ((size_t)(col0 >> 2) * (size_t)St::L + (size_t)m) * 4) = r;
                    else *reinterpret_cast<float4*>(a.dst_real + (size_t)m * N + col0) = r;
                    if (m < a.mm_rows) {
                        if (col0 + 0 < a.mm_cols) { mn = fminf(mn, r.x); mx = fmaxf(mx, r.x); }
                        if (col0 + 1 < a.mm_cols) { mn = fminf(mn, r.y); mx = fmaxf(mx, r.y); }
                        if (col0 + 2 < a.mm_cols) { mn = fminf(mn, r.z); mx = fmaxf(mx, r.z); }
                        if (col0 + 3 < a.mm_cols) { mn = fminf(mn, r.w); mx = fmaxf(mx, r.w); }
                    }
                }
        }
        block_minmax_store(mn, mx, a.mm_part);
    }
}

template <int LOGM, class Pol, int KIND>
static hipError_t launch_cols_one(const ColArgs& a, const float2* twf, const float2* twi, hipStream_t s) {
    using Geo = ColGeom<LOGM>;
    const int ntiles = (a.N + Geo::COLS - 1) / Geo::COLS;
    if constexpr (KIND == COL_FWD_WIENER || KIND == COL_INV_REAL) {  // the operator's own passes: panel-major only since round 4
        if (!a.panel_c) return hipErrorInvalidValue;
        hipLaunchKernelGGL((fft_cols_kernel<LOGM, Pol, KIND, 1>), dim3(ntiles), dim3(Geo::THREADS), 0, s, a, twf, twi);
        return hipGetLastError();
    } else if constexpr (std::is_same<Pol, PolicyParity>::value && KIND == COL_FWD) {  // (the PSF spectrum of the parity operator)
        if (a.panel_c) {
            hipLaunchKernelGGL((fft_cols_kernel<LOGM, Pol, KIND, 1>), dim3(ntiles), dim3(Geo::THREADS), 0, s, a, twf, twi);
            return hipGetLastError();
        }
    }
    if (a.panel_c) return hipErrorInvalidValue;
    hipLaunchKernelGGL((fft_cols_kernel<LOGM, Pol, KIND, 0>), dim3(ntiles), dim3(Geo::THREADS), 0, s, a, twf, twi);
    return hipGetLastError();
}

template <int LOGM>
static hipError_t launch_cols_kind(int mode, ColKind kind, const ColArgs& a, const float2* twf, const float2* twi,
                                   hipStream_t s) {
    if (mode == 0) {
        switch (kind) {
            case COL_FWD: return launch_cols_one<LOGM, PolicyParity, COL_FWD>(a, twf, twi, s);
            case COL_INV: return launch_cols_one<LOGM, PolicyParity, COL_INV>(a, twf, twi, s);
            case COL_FWD_WIENER: return launch_cols_one<LOGM, PolicyParity, COL_FWD_WIENER>(a, twf, twi, s);
            case COL_INV_REAL: return launch_cols_one<LOGM, PolicyParity, COL_INV_REAL>(a, twf, twi, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (kind) {
        case COL_FWD: return launch_cols_one<LOGM, PolicyFast, COL_FWD>(a, twf, twi, s);
        case COL_INV: return launch_cols_one<LOGM, PolicyFast, COL_INV>(a, twf, twi, s);
        default: return hipErrorInvalidValue;
    }
}

template <int LOGM>
static int cols_partials(int N) { return (N + ColGeom<LOGM>::COLS - 1) / ColGeom<LOGM>::COLS; }

int cols_minmax_partials(int logm, int N) {
    switch (logm) {
        case 3: return cols_partials<3>(N);
        case 4: return cols_partials<4>(N);
        case 5: return cols_partials<5>(N);
        case 6: return cols_partials<6>(N);
        case 7: return cols_partials<7>(N);
        case 8: return cols_partials<8>(N);
        case 9: return cols_partials<9>(N);
        case 10: return cols_partials<10>(N);
        case 11: return cols_partials<11>(N);
        case 12: return cols_partials<12>(N);
        case 13: return cols_partials<13>(N);
        default: return 0;
    }
}

hipError_t launch_cols(int logm, int mode, ColKind kind, const ColArgs& a, const float2* twf, const float2* twi,
                       hipStream_t s) {
    switch (logm) {
        case 3: return launch_cols_kind<3>(mode, kind, a, twf, twi, s);
        case 4: return launch_cols_kind<4>(mode, kind, a, twf, twi, s);
        case 5: return launch_cols_kind<5>(mode, kind, a, twf, twi, s);
        case 6: return launch_cols_kind<6>(mode, kind, a, twf, twi, s);
        case 7: return launch_cols_kind<7>(mode, kind, a, twf, twi, s);
        case 8: return launch_cols_kind<8>(mode, kind, a, twf, twi, s);
        case 9: return launch_cols_kind<9>(mode, kind, a, twf, twi, s);
        case 10: return launch_cols_kind<10>(mode, kind, a, twf, twi, s);
        case 11: return launch_cols_kind<11>(mode, kind, a, twf, twi, s);
        case 12: return launch_cols_kind<12>(mode, kind, a, twf, twi, s);
        case 13: return launch_cols_kind<13>(mode, kind, a, twf, twi, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fdr
