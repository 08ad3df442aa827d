// fdr_rows.hip -- row passes: one contiguous row of L complex values per thread group.
//   pass A  : real image (zero-padded on load) -> row FFT -> complex      (replaces preprocess_kernel
//             + fft_row_optimized_kernel of the reference, fft/fft_gpu.cu:85-103,108-148)
//   pass C  : complex -> row (I)FFT -> complex
//   pass C' : complex -> row IFFT -> real plane + running min/max          (fuses postprocess_kernel,
//             fft/fft_gpu.cu:187-201, and the min/max half of cv::normalize)
// Loads and stores are natural order and coalesced: thread t of a group touches elements
// t + (L/rho) q, so each of the rho wave-level accesses is one contiguous sweep of the row.
#include "fdr_fft_core.hpp"
#include "fdr_kernels.hpp"
#include <type_traits>

namespace fdr {

template <int LOGL>
struct RowGeom {
    static constexpr int T = Steps<LOGL>::T;
    static constexpr int G = T >= 256 ? 1 : 256 / T;  // rows per workgroup
    static constexpr int THREADS = T * G;
};

// INV matters only to the fast policy (it conjugates the hoisted forward twiddles); the parity
// policy receives a direction-specific table and is instantiated with INV = false only.
// PANEL = 1 (parity operator since round 4): the complex side(s) are panel-major, so that the column passes work on
// contiguous tiles.  A row then touches 32 bytes (its four columns) of a 128-byte line that it shares with three neighbouring
// rows; the workgroups of such a 4-row group are placed on one XCD back to back (col_tile_of_block) so that the four
// partial lines meet in that XCD's L2 and HBM sees whole lines.  (A kernel that gives a thread group FOUR rows -- whole lines,
// one 32-byte row of a panel per lane and a 4 x 4 transpose inside the quad of lanes, as the fast mode's inverse row passes
// do -- was built, passed the tests and measured SLOWER at every size: A 61 -> 84, C 63 -> 76 us per 4096^2 image; four
// parity transforms per thread do not fit 128 registers without spills at 4096 points and more.  The same with 16 values per
// thread and a radix-16 step of the recurrence tables -- 256 threads per 4-row group at 4096 points, no spills once the rows
// of a stage are pinned one after the other -- passed every test too and was no better: A 66 -> 67, C 66 -> 83 us at 4096^2,
// 20 -> 25 / 21 -> 29 us at 2048^2, 235 -> 290 / 381 -> 386 us at 8192^2.  The row passes are not bound by the size of their
// pieces; the one-row kernels stay.)
template <int LOGL, class Pol, int IN, int OUT, bool INV, int PANEL>
__global__ __launch_bounds__(RowGeom<LOGL>::THREADS) void fft_rows_kernel(const RowArgs a, const float2* __restrict__ tw) {
    using St = Steps<LOGL>;
    using Core = FftCore<LOGL, 1, 1, Pol>;
    constexpr int L = St::L, T = St::T, G = RowGeom<LOGL>::G;
    __shared__ float2 lds[G * St::BUF];

    const int g = threadIdx.x >> St::LOGT, tid = threadIdx.x & (T - 1);
    constexpr int kShare = G >= 4 ? 1 : 4 / G;  // workgroups per 4-row group
    const int blk = PANEL ? col_tile_of_block((int)blockIdx.x, (int)gridDim.x, kShare) : (int)blockIdx.x;
    const int row = blk * G + g;
    const bool active = row < a.M;

    typename Core::Bases bases;
    Core::init_bases(bases, tw, tid);

    float2 v[1][8];
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHO0; ++q) {
            const int n = Core::in_index(tid, u, q);
            float2 x = make_float2(0.f, 0.f);
            if (IN == ROW_IN_REAL) {
                if (active && row < a.src_rows && n < a.src_cols) x.x = a.src_real[(size_t)row * a.src_stride + n];
            } else {
                // (panel-major: unsigned 32-bit element offsets -- (N/4) panels of 4 M + 16 elements stay below 2^32 -- so that an
                // access is base + one 32-bit register, not a 64-bit multiply-add per element: 10 % of this kernel's VALU work)
                if (active) x = PANEL ? a.src_c[(unsigned)(n >> 2) * (unsigned)a.pstride + (unsigned)row * 4u + (unsigned)(n & 3)] : a.src_c[(size_t)row * L + n];
            }
            v[0][u * Core::RHO0 + q] = x;
        }

    Core::template run<0, INV>(v, lds + g * St::BUF, tw, bases, tid);

    if (OUT == ROW_OUT_COMPLEX) {
        if (active) {
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) {
                    const int n = Core::out_index(tid, u, q);
                    if (PANEL) a.dst_c[(unsigned)(n >> 2) * (unsigned)a.pstride + (unsigned)row * 4u + (unsigned)(n & 3)] = v[0][u * Core::RHOL + q];
                    else a.dst_c[(size_t)row * L + n] = v[0][u * Core::RHOL + q];
                }
        }
    } else {
        float mn = __builtin_inff(), mx = -__builtin_inff();
        if (active) {
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) {
                    const int n = Core::out_index(tid, u, q);
                    const float r = v[0][u * Core::RHOL + q].x;
                    a.dst_real[(size_t)row * L + n] = r;
                    if (row < a.mm_rows && n < a.mm_cols) {
                        mn = fminf(mn, r);
                        mx = fmaxf(mx, r);
                    }
                }
        }
        block_minmax_store(mn, mx, a.mm_part);
    }
}

template <int LOGL, class Pol, bool INV>
static hipError_t launch_rows_io(RowIn in, RowOut out, const RowArgs& a, const float2* tw, hipStream_t s) {
    constexpr int G = RowGeom<LOGL>::G, THREADS = RowGeom<LOGL>::THREADS;
    const dim3 grid((a.M + G - 1) / G), block(THREADS);
    if constexpr (std::is_same<Pol, PolicyParity>::value && LOGL >= 2) {  // (panel-major: the parity operator only; rows of >= 4 columns)
        if (a.panel_c) {
            if (in == ROW_IN_REAL && out == ROW_OUT_COMPLEX)
                hipLaunchKernelGGL((fft_rows_kernel<LOGL, Pol, ROW_IN_REAL, ROW_OUT_COMPLEX, INV, 1>), grid, block, 0, s, a, tw);
            else if (in == ROW_IN_COMPLEX && out == ROW_OUT_COMPLEX)
                hipLaunchKernelGGL((fft_rows_kernel<LOGL, Pol, ROW_IN_COMPLEX, ROW_OUT_COMPLEX, INV, 1>), grid, block, 0, s, a, tw);
            else
                return hipErrorInvalidValue;
            return hipGetLastError();
        }
    }
    if (a.panel_c) return hipErrorInvalidValue;
    if (in == ROW_IN_REAL && out == ROW_OUT_COMPLEX)
        hipLaunchKernelGGL((fft_rows_kernel<LOGL, Pol, ROW_IN_REAL, ROW_OUT_COMPLEX, INV, 0>), grid, block, 0, s, a, tw);
    else if (in == ROW_IN_COMPLEX && out == ROW_OUT_COMPLEX)
        hipLaunchKernelGGL((fft_rows_kernel<LOGL, Pol, ROW_IN_COMPLEX, ROW_OUT_COMPLEX, INV, 0>), grid, block, 0, s, a, tw);
    else if (in == ROW_IN_COMPLEX && out == ROW_OUT_REAL_MINMAX)
        hipLaunchKernelGGL((fft_rows_kernel<LOGL, Pol, ROW_IN_COMPLEX, ROW_OUT_REAL_MINMAX, INV, 0>), grid, block, 0, s, a, tw);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// tw: parity -> the table of the requested direction; fast -> always the forward table
template <int LOGL>
static hipError_t launch_rows_mode(int mode, RowIn in, RowOut out, bool inverse, const RowArgs& a, const float2* tw,
                                   hipStream_t s) {
    if (mode == 0) return launch_rows_io<LOGL, PolicyParity, false>(in, out, a, tw, s);
    // fast mode: real input is only ever transformed forward, real output only ever inverse
    if (inverse) {
        if (in == ROW_IN_REAL) return hipErrorInvalidValue;
        return launch_rows_io<LOGL, PolicyFast, true>(in, out, a, tw, s);
    }
    if (out == ROW_OUT_REAL_MINMAX) return hipErrorInvalidValue;
    return launch_rows_io<LOGL, PolicyFast, false>(in, out, a, tw, s);
}

template <int LOGL>
static int rows_partials(int M) { return (M + RowGeom<LOGL>::G - 1) / RowGeom<LOGL>::G; }

int rows_minmax_partials(int logl, int M) {
    switch (logl) {
        case 3: return rows_partials<3>(M);
        case 4: return rows_partials<4>(M);
        case 5: return rows_partials<5>(M);
        case 6: return rows_partials<6>(M);
        case 7: return rows_partials<7>(M);
        case 8: return rows_partials<8>(M);
        case 9: return rows_partials<9>(M);
        case 10: return rows_partials<10>(M);
        case 11: return rows_partials<11>(M);
        case 12: return rows_partials<12>(M);
        case 13: return rows_partials<13>(M);
        default: return 0;
    }
}

hipError_t launch_rows(int logl, int mode, RowIn in, RowOut out, bool inverse, const RowArgs& a, const float2* tw,
                       hipStream_t s) {
    switch (logl) {
        case 3: return launch_rows_mode<3>(mode, in, out, inverse, a, tw, s);
        case 4: return launch_rows_mode<4>(mode, in, out, inverse, a, tw, s);
        case 5: return launch_rows_mode<5>(mode, in, out, inverse, a, tw, s);
        case 6: return launch_rows_mode<6>(mode, in, out, inverse, a, tw, s);
        case 7: return launch_rows_mode<7>(mode, in, out, inverse, a, tw, s);
        case 8: return launch_rows_mode<8>(mode, in, out, inverse, a, tw, s);
        case 9: return launch_rows_mode<9>(mode, in, out, inverse, a, tw, s);
        case 10: return launch_rows_mode<10>(mode, in, out, inverse, a, tw, s);
        case 11: return launch_rows_mode<11>(mode, in, out, inverse, a, tw, s);
        case 12: return launch_rows_mode<12>(mode, in, out, inverse, a, tw, s);
        case 13: return launch_rows_mode<13>(mode, in, out, inverse, a, tw, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fdr
