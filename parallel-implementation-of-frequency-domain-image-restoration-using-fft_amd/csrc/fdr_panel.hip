// fdr_panel.hip -- fast-mode passes on a PANEL-MAJOR intermediate spectrum.
//
// Why: a column pass that keeps 4 adjacent columns of a row-major M x N array touches 32 of the
// 128 bytes of every line; four workgroups share each line, and with >= 128 KB of lines in flight
// per CU the sharing cannot be served from the 4 MiB XCD L2 -- measured on MI355X: pass B' with
// the transform compiled out still took 191 us at 4096^2 (2.1 TB/s algorithmic, each line moved
// ~4x).  The fix is a layout in which every pass moves whole lines:
//
//     panel_index(m, n) = (n >> 2) * PS + m * 4 + (n & 3)     (M rows, N columns, float2 elements)
//
// i.e. panels of 4 columns, each panel a contiguous M x 4 array; the panel stride PS = 4 M + 16 is
// deliberately not a power of two, so the 128-byte lines a row workgroup scatters over all panels
// do not all fall on the same memory channel.  A column tile (4 columns, all
// rows) is then ONE contiguous 32*M-byte chunk, and a row workgroup that owns 4 consecutive rows
// writes / reads complete 128-byte lines (4 rows x 4 columns) in every panel.
//
//   pass A  : 4 real rows  (zero-padded on load) -> row FFTs -> panel layout
//   pass B' : one panel: column FFTs . W . column IFFTs, in place, persistent + register
//             double-buffered (next panel's spectrum / filter stream in behind the butterflies)
//   pass C' : 4 rows gathered from the panels -> row IFFTs -> real plane + min/max partial
// The layout is private to a plan (never visible through the C ABI); W is stored the same way.
#include "fdr_fft_core.hpp"
#include <type_traits>
#include "fdr_kernels.hpp"

namespace fdr {

// ---------------------------------------------------------------------------------------------
// rows, 4 at a time
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Two-for-one row transforms (fast mode only; rounding differs from the serial path at the 1e-7
// level, far inside the 1e-4 budget).  The image rows are real, and after the inverse column pass
// every row spectrum is Hermitian, so two rows share one complex transform:
//   forward : z = x_a + i x_b  ->  Z = FFT(z);  X_a[n] = (Z[n] + conj Z[N-n]) / 2,
//                                               X_b[n] = (Z[n] - conj Z[N-n]) / (2i)
//             (Z[N-n] lives in another thread: one natural-order LDS round trip)
//   inverse : Z = Y_a + i Y_b  ->  z = IFFT(Z);  row a = Re z, row b = Im z   (no fix-up at all)
// A 4-row group therefore costs 2 complex transforms instead of 4.
// ---------------------------------------------------------------------------------------------
#ifndef FDR_SWAP0
#define FDR_SWAP0 1  // v_permlane32_swap exchange behind a radix-2 first step (8192-point transforms on the 16-value core)
#endif
// The one-group row kernels (a workgroup lives for one 4-row group) hold 8 values per thread (radix-8 steps) -- except for
// rows of FDR_ROWS_ONE_V16_MIN .. FDR_ROWS_ONE_V16_MAX points (log2), which take 16 (radix-16 steps: two LDS exchanges
// instead of three per 4096-point transform, half the threads).  Measured on MI355X (passbench, 24 x 4096^2, 4 images per
// launch, us per image): A 25.7 -> 25.2, C1 18.7 -> 17.2, C2 23.9 -> 23.1; at 2048 / 1024 points the 16-value form is slower
// (fewer waves per transform: 2048^2 C2 6.8 -> 7.3; one 1024^2 image 26.0 -> 30.1 us), so it starts at 4096.
// The inverse kernels of that form own ONE exchange buffer per thread group (FDR_ROWS_INV_NBUF1_MIN: the two packed pairs
// hand their mirrored halves over one after the other, the exchanges take a second barrier): 37 KB instead of 74 KB of LDS,
// four 256-thread workgroups per CU instead of two -- C1 17.2 -> 14.7 us per 4096^2 image (0.45 -> 0.57 of the roofline).
#ifndef FDR_ROWS_ONE_V16_MIN
#define FDR_ROWS_ONE_V16_MIN 12
#endif
#ifndef FDR_ROWS_ONE_V16_MAX
#define FDR_ROWS_ONE_V16_MAX 13
#endif
#ifndef FDR_ROWS_INV_NBUF1_MIN
#define FDR_ROWS_INV_NBUF1_MIN 10  // (wherever the inverse kernels hold 16 values per thread)
#endif
// (The FORWARD kernel with one buffer -- pairs separated and stored one after the other, a lane pair writing the 64-byte half
// of a line that its rows 2b, 2b+1 make up, four workgroups per CU -- was measured too: pass A 25.5 -> 28.5 us per 4096^2 image;
// half-line stores cost more than the occupancy gains.  The same with the lines of both pairs held in registers (2 x 4 items
// of 32 bytes per lane) so that the stores stay whole lines: 25.5 -> 25.2 us -- pass A is not bound by its occupancy.  Not kept.
// Non-temporal stores of the spectrum (it is read again by pass B', but four images do not stay in any cache): 24.0 -> 27.0 us.)
// The inverse kernels have their own lower bound, FDR_ROWS_INV_V16_MIN = 2048 points: with ONE exchange buffer the 16-value form
// keeps four workgroups per CU there too -- 2048^2 in launches of 4: C2 6.65 -> 5.94, C1 5.27 -> 5.08 us per image, the two-stream
// batch 20.89 -> 20.47 us per image (config 5's size; the forward kernel with 16 values at 2048 points: 7.9 -> 8.2 us, not
// taken); at 1024 points the 16-value inverse kernels are slower (C2 1.83 -> 2.05 us per image in launches of 8).
#ifndef FDR_ROWS_INV_V16_MIN
#define FDR_ROWS_INV_V16_MIN 11
#endif
template <int LOGL, bool INV = false>
struct Rows4PackGeom {
    static constexpr int LOGV = (LOGL <= FDR_ROWS_ONE_V16_MAX && LOGL >= (INV ? FDR_ROWS_INV_V16_MIN : FDR_ROWS_ONE_V16_MIN)) ? 4 : 3;
    using St = Steps<LOGL, LOGV>;
    static constexpr int T = St::T;
    static constexpr int G = T >= 256 ? 1 : 256 / T;
    static constexpr int THREADS = T * G;
    // inverse kernels with ONE exchange buffer: as many workgroups per CU as the LDS admits, registers capped to match
    static constexpr int INV_LDS = G * St::BUF * 8;
    static constexpr int INV_WG_PER_CU = (INV && LOGL >= FDR_ROWS_INV_NBUF1_MIN && LOGV == 4) ? ((160 * 1024) / INV_LDS > 4 ? 4 : (160 * 1024) / INV_LDS) : 1;
    static constexpr int INV_WAVES_PER_SIMD = INV_WG_PER_CU * THREADS / 256 > 0 ? (INV_WG_PER_CU * THREADS / 256 > 8 ? 8 : INV_WG_PER_CU * THREADS / 256) : 1;
};

// Reads every register of a prefetched set through an empty asm: the compiler places the wait for those loads HERE (with
// the exact vmcnt for this point of the program) and treats them as landed afterwards.  Used at the bottom of the
// persistent loops, right behind the stores of the group just finished: the prefetch is older than those stores, so the
// wait is vmcnt(#stores) and the stores keep draining; left to the first use (copies at the loop top, where the state
// of the first iteration merges in) the compiler emits vmcnt(0..1) and the stores drain before the next transform.
template <int R, int C>
__device__ __forceinline__ void landed_f(const float (&d)[R][C]) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < C; c += 4) asm volatile("" ::"v"(d[r][c]), "v"(d[r][c + 1]), "v"(d[r][c + 2]), "v"(d[r][c + 3]));
}
template <int R, int C>
__device__ __forceinline__ void landed_f2(const float2 (&d)[R][C]) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < C; c += 2) asm volatile("" ::"v"(d[r][c].x), "v"(d[r][c].y), "v"(d[r][c + 1].x), "v"(d[r][c + 1].y));
}


// HALF: keep only the non-redundant half of each Hermitian row spectrum -- columns 0 .. N/2-1 in panels
// 0 .. N/8-1.  X[m,0] and X[m,N/2] are real for a real row, so the Nyquist column rides in the imaginary
// part of column 0: stored(m, 0) = X[m,0] + i X[m,N/2]  ("packed column", undone in passes B' and C').
template <int LOGL, bool HALF>
__global__ __launch_bounds__(Rows4PackGeom<LOGL>::THREADS) void fft_rows4_fwd_packed_kernel(const RowArgs a0,
                                                                                          const float2* __restrict__ tw_fwd) {
    RowArgs a = a0;
    if (a0.batch.nimg > 1) {  // blockIdx.y = image
        a.src_real = pick_image(a0.batch.src_real, blockIdx.y);
        a.dst_c = pick_image(a0.batch.spec, blockIdx.y);
    }
    using Geo = Rows4PackGeom<LOGL>;
    using St = typename Geo::St;
    constexpr int G = Geo::G, T = St::T, L = St::L;
    using Core = FftCore<LOGL, 2, 2, PolicyFast, Geo::LOGV, (St::lr(0) == 1 && T >= 64 && FDR_SWAP0)>;
    __shared__ float2 lds[G * 2 * St::BUF];
    const int g = threadIdx.x >> St::LOGT, tid = Core::thread_index(threadIdx.x & (T - 1));
    float2* grp_lds = lds + g * 2 * St::BUF;
    const int M = a.M;
    const int r0 = (blockIdx.x * G + g) * 4;
    const bool active = r0 < M;
    const int rr = active ? r0 : 0;

    typename Core::Bases bases;
    Core::init_bases(bases, tw_fwd, tid);

    float2 z[2][Core::V];
    // common case first: all four rows and all L columns inside the image -> 32 unpredicated loads
    // from four wave-uniform row bases with one 32-bit per-thread offset
    const bool interior = (rr + 3 < a.src_rows) && (a.src_cols >= L);
    if (interior) {
        const float* row0 = a.src_real + (size_t)rr * a.src_stride;
        const float* row1 = row0 + a.src_stride;
        const float* row2 = row1 + a.src_stride;
        const float* row3 = row2 + a.src_stride;
#pragma unroll
        for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
            for (int q = 0; q < Core::RHO0; ++q) {
                const int s = u * Core::RHO0 + q;
                const unsigned n = (unsigned)Core::in_index(tid, u, q);
                // the image is read exactly once: keep it out of the caches that hold the intermediates
                z[0][s] = make_float2(__builtin_nontemporal_load(row0 + n), __builtin_nontemporal_load(row1 + n));
                z[1][s] = make_float2(__builtin_nontemporal_load(row2 + n), __builtin_nontemporal_load(row3 + n));
            }
    } else {
#pragma unroll
        for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
            for (int q = 0; q < Core::RHO0; ++q) {
                const int s = u * Core::RHO0 + q;
                const int n = Core::in_index(tid, u, q);
                float x[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    x[b] = 0.f;
                    if (rr + b < a.src_rows && n < a.src_cols) x[b] = a.src_real[(size_t)(rr + b) * a.src_stride + n];
                }
                z[0][s] = make_float2(x[0], x[1]);
                z[1][s] = make_float2(x[2], x[3]);
            }
    }

    Core::template run<0, false>(z, grp_lds, tw_fwd, bases, tid);

    // Separate the two real rows of each packed transform and store all four spectra panel-major.
    constexpr int SEQ1 = Core::SLOTS;
    if constexpr (T >= 4) {
        // Both packed spectra go to LDS in natural order; then the threads re-partition the work so
        // that a quad of lanes owns one 128-byte line (4 rows x 4 columns of a panel): lane j of the
        // quad builds row j's four columns (32 contiguous bytes).  Every store instruction of a wave
        // then covers 16 complete lines instead of 64 scattered 8-byte pieces.
        // The buffer of pair b = 1 is the one the transform's LAST exchange read from: a barrier has to separate those
        // reads from this write (pair b = 0 goes to the other buffer, which the last barrier of the transform already
        // protects).  Without it a wave that runs ahead overwrites values a slower wave is still picking up -- rows 2, 3
        // of the group came out wrong for a few lanes' columns whenever a second stream's kernels shared the CUs.
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float2* buf = grp_lds + ((SEQ1 + b) & 1) * St::BUF;
#ifndef FDR_DEBUG_OMIT_SEPARATION_BARRIER  // (the race fuzzer's own check: with the barrier left out it must find the race)
            if (b == 1) __syncthreads();
#endif
            FDR_JITTER(2001 + b);
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) buf[Core::out_index(tid, u, q)] = z[b][u * Core::RHOL + q];
        }
        __syncthreads();
        FDR_JITTER(2003);
        const int j = tid & 3;                                    // row inside the 4-row group
        const float2* buf = grp_lds + ((SEQ1 + (j >> 1)) & 1) * St::BUF;  // packed pair holding row j
        const bool odd = (j & 1) != 0;                            // row b of the pair (else row a)
#pragma unroll
        for (int i = 0; i < (HALF ? L / 8 : L / 4) / (T / 4); ++i) {
            const int c = (tid >> 2) + (T / 4) * i;  // panel
            const int n0 = c * 4;
            float2 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float2 zn = buf[n0 + k];
                const float2 zm = buf[(L - n0 - k) & (L - 1)];
                o[k] = odd ? make_float2(0.5f * (zn.y + zm.y), 0.5f * (zm.x - zn.x))
                           : make_float2(0.5f * (zn.x + zm.x), 0.5f * (zn.y - zm.y));
            }
            if (HALF && n0 == 0) {  // packed column: (X[0], X[N/2]), both real: Re/Im of Z[0] and Z[N/2]
                const float2 z0 = buf[0], zq = buf[L / 2];
                o[0] = odd ? make_float2(z0.y, zq.y) : make_float2(z0.x, zq.x);
            }
            if (active) store4(a.dst_c + (size_t)c * a.pstride + (size_t)(r0 + j) * 4, o[0], o[1], o[2], o[3]);
        }
        } else {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float2* buf = grp_lds + ((SEQ1 + b) & 1) * St::BUF;
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) buf[Core::out_index(tid, u, q)] = z[b][u * Core::RHOL + q];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) {
                    const int s = u * Core::RHOL + q;
                    const int n = Core::out_index(tid, u, q);
                    const float2 zn = z[b][s];
                    const float2 zm = buf[(L - n) & (L - 1)];
                    const float2 xa = make_float2(0.5f * (zn.x + zm.x), 0.5f * (zn.y - zm.y));
                    const float2 xb = make_float2(0.5f * (zn.y + zm.y), 0.5f * (zm.x - zn.x));
                    if (active) {
                        float2* p = a.dst_c + (size_t)(n >> 2) * a.pstride + (size_t)(r0 + 2 * b) * 4 + (n & 3);
                        p[0] = xa;
                        p[4] = xb;
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent form of pass A (rows of 2048 points and more: one thread group = one workgroup).  A launch of the kernel
// above runs in lockstep -- every workgroup loads, then every workgroup transforms, then every workgroup stores -- so
// HBM idles while the CUs compute and the CUs idle while HBM streams (measured: 18 us per round of 512 workgroups at
// 4096^2 = 10.9 us of memory time + 7 us of transform, nothing overlapped).  Here a workgroup walks over its row groups
// and requests the NEXT group's four image rows (4 V floats per thread) before it transforms the current one; the
// stores of the current group drain behind the next group's transform.  The prefetch is unconditional (clamped
// addresses, collapsed onto one element when there is no next group; zero padding is applied when the values are
// packed): a conditional load would make the compiler wait for it right away (lesson 1).
//   LOGV = 3: 8 values per thread, T = L/8 threads, two workgroups per CU at 4096 points (2 x 74 KB of LDS)
//   LOGV = 4: 16 values per thread, T = L/16 threads: 8192-point rows as ONE 512-thread workgroup per CU with a
//             256-register budget (the 8-value form needs 1024 threads at 128 registers and cannot hold a prefetch)
// ---------------------------------------------------------------------------------------------
#ifndef FDR_ROWS12_LOGV
#define FDR_ROWS12_LOGV 3  // values per thread (log2) of the persistent row passes for rows of 4096 points (A/B builds)
#endif
// Shortest rows (log2) that take the persistent form.  Measured on MI355X (passbench, 24 x 4096^2 / 6 x 8192^2 / 32 x
// 2048^2, us per image, one-group-per-workgroup kernel vs persistent):  8192: A 144.6 -> 121.7, C' 155.2 -> 143.1;
// 4096: A 34.7 -> 36.2, C' 31.2 -> 34.1;  2048 (4 images per launch): A 8.1 -> 9.0, C' 6.9 -> 8.3.  With the transform
// compiled out the 4096^2 row passes take 28.5 / 23.8 us: only 6-7 us of transform are exposed there, less than the
// persistent form's own cost (dummy prefetch, barrier, fewer independent workgroups), so it is used for 8192-point rows
// only, where the alternative is a 1024-thread workgroup alone on its CU.
#ifndef FDR_ROWS_PERS_MIN_LOG
#define FDR_ROWS_PERS_MIN_LOG 13
#endif
// The INVERSE passes (C', C1, C2) stopped using it late in round 3: at 8192 points the one-group kernel with 16 values per
// thread, ONE exchange buffer (two 512-thread workgroups per CU instead of one persistent workgroup) and whole-row loads
// (FDR_ROWS_LOAD32) beats the persistent kernel -- C1 76.3 -> 65.2, C2 106.3 -> 100.3 us per 8192^2 image, the two-stream batch
// 426.9 -> 413.7 us per image.  The persistent inverse kernel stays for A/B builds (-DFDR_ROWS_INV_PERS_MIN_LOG=13).
#ifndef FDR_ROWS_INV_PERS_MIN_LOG
#define FDR_ROWS_INV_PERS_MIN_LOG 14
#endif
template <int LOGL, int LOGV>
struct RowsPersGeom {
    using St = Steps<LOGL, LOGV>;
    static constexpr int T = St::T;
    static_assert(T >= 256, "one thread group per workgroup");
    static constexpr int THREADS = T;
    static constexpr int LDS_BYTES = 2 * St::BUF * 8;
    static constexpr int BY_LDS = (160 * 1024) / LDS_BYTES;
    static constexpr int BY_REGS = (LOGV == 3 ? 4 : 2) * 256 / THREADS;  // 128 registers (V = 8) or 256 (V = 16) per lane
    static constexpr int WG_PER_CU = BY_LDS < BY_REGS ? (BY_LDS < 1 ? 1 : BY_LDS) : (BY_REGS < 1 ? 1 : BY_REGS);
    static constexpr int WAVES_PER_SIMD = WG_PER_CU * THREADS / 256;
};

template <int LOGL, int LOGV, bool HALF, bool INTERIOR>
__global__ __launch_bounds__((RowsPersGeom<LOGL, LOGV>::THREADS), (RowsPersGeom<LOGL, LOGV>::WAVES_PER_SIMD)) void fft_rows4_fwd_pers_kernel(
    const RowArgs a, const float2* __restrict__ tw_fwd, const int ngroups, const int total) {
    using St = Steps<LOGL, LOGV>;
    constexpr int T = St::T, L = St::L, V = St::V;
    using Core = FftCore<LOGL, 2, 2, PolicyFast, LOGV, (St::lr(0) == 1 && FDR_SWAP0)>;  // 8192 points: wave-local first exchange
    __shared__ float2 lds[2 * St::BUF];
    const int tid = Core::thread_index(threadIdx.x);
    const int nimg = a.batch.nimg > 1 ? a.batch.nimg : 1;

    typename Core::Bases bases;
    Core::init_bases(bases, tw_fwd, tid);

    // group gi = (image, 4-row group); advanced by scalar add / subtract (no division on the vector unit)
    int gi = blockIdx.x;
    if (gi >= total) return;
    int img = 0, grp = gi;
    while (grp >= ngroups) { grp -= ngroups; ++img; }
    auto src_of = [&](int im) -> const float* { return nimg > 1 ? pick_image(a.batch.src_real, im) : a.src_real; };
    auto dst_of = [&](int im) -> float2* { return nimg > 1 ? pick_image(a.batch.spec, im) : a.dst_c; };

    // four image rows of group `g` of image `im`: unconditional loads from clamped coordinates; scale = 0 collapses
    // every address onto element 0 of the image (a prefetch with nothing to fetch: conditional loads would make the
    // compiler wait for them on the spot)
    float x[4][V];
    auto request = [&](const float* __restrict__ src, int g, unsigned scale) __attribute__((always_inline)) {
        const int r0 = g * 4;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            int r = r0 + b;
            if (!INTERIOR) r = r < a.src_rows ? r : a.src_rows - 1;
            const float* __restrict__ row = src + (size_t)r * (size_t)a.src_stride * scale;
#pragma unroll
            for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHO0; ++q) {
                    const int s = u * Core::RHO0 + q;
                    unsigned n = (unsigned)Core::in_index(tid, u, q);
                    if (!INTERIOR) n = n < (unsigned)a.src_cols ? n : (unsigned)a.src_cols - 1u;
                    x[b][s] = __builtin_nontemporal_load(row + n * scale);  // the image is read exactly once
                }
        }
    };
    // z[0] = rows 0 + i 1, z[1] = rows 2 + i 3 of group g (zero padding applied here)
    float2 z[2][V];
    auto pack = [&](int g) __attribute__((always_inline)) {
        const int r0 = g * 4;
#pragma unroll
        for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
            for (int q = 0; q < Core::RHO0; ++q) {
                const int s = u * Core::RHO0 + q;
                float v0 = x[0][s], v1 = x[1][s], v2 = x[2][s], v3 = x[3][s];
                if (!INTERIOR) {
                    const bool cok = Core::in_index(tid, u, q) < a.src_cols;
                    v0 = (cok && r0 + 0 < a.src_rows) ? v0 : 0.f;
                    v1 = (cok && r0 + 1 < a.src_rows) ? v1 : 0.f;
                    v2 = (cok && r0 + 2 < a.src_rows) ? v2 : 0.f;
                    v3 = (cok && r0 + 3 < a.src_rows) ? v3 : 0.f;
                }
                z[0][s] = make_float2(v0, v1);
                z[1][s] = make_float2(v2, v3);
            }
    };
    // transform the packed pair in z, separate the two real rows of each transform and store all four spectra
    // panel-major: both packed spectra go to LDS in natural order, then a quad of lanes owns one 128-byte line (4 rows x
    // 4 columns of a panel): lane j of the quad builds row j's four columns (see fft_rows4_fwd_packed_kernel)
    auto body = [&](int im, int g) __attribute__((always_inline)) {
        {
            int tr = tid;  // opaque copy: the exchange addresses are recomputed per group, not carried across the loop
            asm volatile("" : "+v"(tr));
            Core::template run<0, false>(z, lds, tw_fwd, bases, tr);
        }
        constexpr int SEQ1 = Core::SLOTS;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float2* buf = lds + ((SEQ1 + b) & 1) * St::BUF;
            if (b == 1) __syncthreads();  // pair 1's buffer was read by the transform's last exchange (see fft_rows4_fwd_packed_kernel)
            FDR_JITTER(2011 + b);
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) buf[Core::out_index(tid, u, q)] = z[b][u * Core::RHOL + q];
        }
        __syncthreads();
        FDR_JITTER(2013);
        float2* __restrict__ dst = dst_of(im);
        const int r0 = g * 4;
        // (opaque copy of the thread index: the LDS and panel addresses below are loop invariant, and hoisted out of
        // the group loop they would occupy ~25 registers for the whole kernel -- recomputing them costs a few adds)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int j = tq & 3;                                            // row inside the 4-row group
        const float2* buf = lds + ((SEQ1 + (j >> 1)) & 1) * St::BUF;     // packed pair holding row j
        const bool odd = (j & 1) != 0;                                   // row b of the pair (else row a)
        constexpr int NIT = (HALF ? L / 8 : L / 4) / (T / 4);            // panels per lane
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int c = (tq >> 2) + (T / 4) * i;  // panel
            const int n0 = c * 4;
            float2 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float2 zn = buf[n0 + k];
                const float2 zm = buf[(L - n0 - k) & (L - 1)];
                o[k] = odd ? make_float2(0.5f * (zn.y + zm.y), 0.5f * (zm.x - zn.x))
                           : make_float2(0.5f * (zn.x + zm.x), 0.5f * (zn.y - zm.y));
            }
            if (HALF && n0 == 0) {  // packed column: (X[0], X[N/2]), both real: Re/Im of Z[0] and Z[N/2]
                const float2 z0 = buf[0], zq = buf[L / 2];
                o[0] = odd ? make_float2(z0.y, zq.y) : make_float2(z0.x, zq.x);
            }
            store4(dst + (size_t)c * a.pstride + (size_t)(r0 + j) * 4, o[0], o[1], o[2], o[3]);
        }
    };

    // Loop shape: the wait for a prefetch sits at the BOTTOM of the loop (pack), behind the stores of the group just
    // finished.  vmcnt counts loads and stores in issue order and the prefetch is older than those stores, so the wait
    // there is vmcnt(#stores) and the stores keep draining behind the next transform; with the wait at the loop top the
    // compiler has to merge it with the first iteration's state (no stores yet) and emits vmcnt(0).
    request(src_of(img), grp, 1u);
    landed_f(x);  // (also here: the loop top must see landed values on both of its entries, or it waits again)
    pack(grp);
    while (true) {
        const bool more = gi + (int)gridDim.x < total;
        int nimg_i = img, ngrp = grp;
        if (more) {
            ngrp += (int)gridDim.x;
            while (ngrp >= ngroups) { ngrp -= ngroups; ++nimg_i; }
        }
        request(src_of(nimg_i), more ? ngrp : 0, more ? 1u : 0u);
        body(img, grp);
        if (!more) break;
        landed_f(x);      // wait for the prefetch here, behind this group's stores
        __syncthreads();  // the separation's reads are done before the next transform's first exchange writes
        pack(ngrp);
        gi += (int)gridDim.x; img = nimg_i; grp = ngrp;
    }
}

// Four Hermitian row spectra (rows rr .. rr+3) rebuilt from the panel-major (half) spectrum and packed two rows per
// complex transform, in two steps so that the loads of the NEXT row group can be issued (into y) before the values
// are touched:  rows4_load_raw -> y[row][slot] (first-step operand order, stored column of slot s),
//               rows4_pack     -> z[0] = Y_a + i Y_b (rows rr, rr+1), z[1] = rows rr+2, rr+3.
//
// Instruction count matters here (a VALU instruction costs 4 cycles per wave, and these kernels run 2 waves per SIMD):
// which half of the spectrum a slot lies in is a compile-time property of its q (n = tid + u T + q 2^LOGR0, and
// q >= RHO0/2  <=>  n >= L/2), so the mirrored slots (n > L/2: stored column L-n, conjugated) need no per-lane
// selects; only lane tid = 0 differs (n = 0: DC, n = L/2: Nyquist -- both live in the packed column 0) and is
// patched separately.  Addresses: one uniform base per slot + two per-lane 32-bit offsets (direct / mirrored).
template <int LOGL, bool HALF, class Core>
__device__ __forceinline__ void rows4_load_raw(const RowArgs& a, int rr, int tid, float2 (&y)[4][Core::V], unsigned scale = 1u) {
    constexpr int L = Steps<LOGL>::L;
    if constexpr (!HALF || LOGL < 5) {  // (half-spectrum plans need N >= 32; smaller instantiations are never launched)
#pragma unroll
        for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
            for (int q = 0; q < Core::RHO0; ++q) {
                const int s = u * Core::RHO0 + q;
                const int n = Core::in_index(tid, u, q);
                const float2* p = a.src_c + ((size_t)(n >> 2) * a.pstride + (size_t)rr * 4 + (n & 3)) * scale;
                y[0][s] = p[0]; y[1][s] = p[4]; y[2][s] = p[8]; y[3][s] = p[12];
            }
    } else {
        static_assert(Core::RHO0 >= 2 && Core::LOGR0 >= 2, "n = t + q Q with Q a multiple of 4");
        const unsigned ps = (unsigned)a.pstride * scale;  // scale = 0: every address collapses onto rows 0..3 of panel 0 (a prefetch with nothing to fetch)
        rr = (int)((unsigned)rr * scale);
        constexpr unsigned PMID = (unsigned)(L / 8);  // panel of column L/2 (one past the stored panels)
        unsigned off_d[Core::NU0], off_m[Core::NU0];
#pragma unroll
        for (int u = 0; u < Core::NU0; ++u) {
            const unsigned t = (unsigned)(tid + u * Core::T);        // n = t + q Q,  Q = 2^LOGR0 (a multiple of 4)
            const unsigned ta = t >> 2, tb = t & 3u;
            // direct half: stored column t + qQ -> panel qQ/4 + ta, column tb
            off_d[u] = ta * ps + tb * scale + (unsigned)rr * 4u;
            // mirrored half: stored column (RHO0 - q) Q - t -> panel (RHO0-q)Q/4 - ta - (tb != 0), column (4 - tb) & 3,
            // relative to q = RHO0/2 (panel L/8)
            const unsigned pm = PMID - ta - (tb != 0u ? 1u : 0u);     // panel of the mirrored column at q = RHO0/2
            off_m[u] = pm * ps + ((4u - tb) & 3u) * scale + (unsigned)rr * 4u;
        }
        auto load_slot = [&](int u, int q) __attribute__((always_inline)) {
            const int s = u * Core::RHO0 + q;
#ifdef FDR_DEBUG_SKIP_MEM  // timing-only builds
            y[0][s] = y[1][s] = y[2][s] = y[3][s] = make_float2((float)(off_d[u] + q), (float)(off_m[u]));
#else
            const float2* p;
            if (q < Core::RHO0 / 2) {
                p = a.src_c + (size_t)((q << Core::LOGR0) >> 2) * ps + off_d[u];
            } else if (q == Core::RHO0 / 2) {
                // n = L/2 (lane t = 0 of the u = 0 slot): the Nyquist value rides in column 0 of panel 0
                const unsigned off_n = (u == 0 && tid == 0) ? (unsigned)rr * 4u : off_m[u];
                p = a.src_c + off_n;
            } else {  // (RHO0 - q) Q = L/2 - (q - RHO0/2) Q: uniform step back from the q = RHO0/2 panel
                p = a.src_c - (size_t)(((q - Core::RHO0 / 2) << Core::LOGR0) >> 2) * ps + off_m[u];
            }
            y[0][s] = p[0]; y[1][s] = p[4]; y[2][s] = p[8]; y[3][s] = p[12];
#endif
        };
        // Issue order: every stored line is read twice by the workgroup, once for a direct slot and once for the mirrored
        // slot that covers the same block of panels -- direct (u, q) and mirrored (NU0-1-u, RHO0-1-q).  Requested back to
        // back the second touch finds the line in (or on its way into) L1 / L2; in slot order the two are half a tile of
        // loads apart and the second one goes out to the fabric again (measured at 8192^2: 372 MB read for 268 MB stored).
#pragma unroll
        for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
            for (int q = 0; q < Core::RHO0 / 2; ++q) {
                load_slot(u, q);
                load_slot(Core::NU0 - 1 - u, Core::RHO0 - 1 - q);
#ifndef FDR_DEBUG_SKIP_MEM
                asm volatile("" ::: "memory");
#endif
            }
    }
}
template <int LOGL, bool HALF, class Core>
__device__ __forceinline__ void rows4_pack(int tid, const float2 (&y)[4][Core::V], float2 (&z)[2][Core::V]) {
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHO0; ++q) {
            const int s = u * Core::RHO0 + q;
            const float2 y0 = y[0][s], y1 = y[1][s], y2 = y[2][s], y3 = y[3][s];
            if (HALF && LOGL >= 5 && q >= Core::RHO0 / 2) {  // mirrored column: conjugate, then Y_a + i Y_b
                z[0][s] = make_float2(y0.x + y1.y, y1.x - y0.y);
                z[1][s] = make_float2(y2.x + y3.y, y3.x - y2.y);
            } else {
                z[0][s] = make_float2(y0.x - y1.y, y0.y + y1.x);  // Y_a + i Y_b
                z[1][s] = make_float2(y2.x - y3.y, y2.y + y3.x);
            }
        }
    if (HALF && LOGL >= 5 && tid == 0) {  // n = 0 (DC) and n = L/2 (Nyquist): real values packed as (DC, Nyquist) in column 0
        constexpr int SN = Core::RHO0 / 2;  // slot of n = L/2 (u = 0)
        z[0][0] = make_float2(y[0][0].x, y[1][0].x);
        z[1][0] = make_float2(y[2][0].x, y[3][0].x);
        z[0][SN] = make_float2(y[0][SN].y, y[1][SN].y);
        z[1][SN] = make_float2(y[2][SN].y, y[3][SN].y);
    }
}

// ---------------------------------------------------------------------------------------------
// The same rebuild WITHOUT the second touch of memory (half-spectrum plans, rows of 32 points and more).  The packed
// input of the inverse transform is Z[n] = Y_a[n] + i Y_b[n]; for the upper half, Z[N-n] = conj(Y_a[n]) + i conj(Y_b[n])
// -- a function of the SAME two stored values.  So a thread loads only its direct slots (stored columns n < N/2: half
// the gathers of rows4_load_raw, every stored line requested once), forms Z[n] for itself and Z[N-n] for whichever
// thread owns index N-n, and hands the latter over through LDS in natural order (the exchange buffers are idle at that
// point): 4 V / 8 writes + reads per thread and two barriers, against V/2 x 4 eight-byte gathers that went out to the
// fabric a second time (fabric reads of pass C' measured at 1.10x / 1.45x the stored bytes at 4096^2 / 8192^2).
//   y[row][j], j = u (RHO0/2) + q : stored column in_index(tid, u, q), q < RHO0/2
// ---------------------------------------------------------------------------------------------
// How the four rows reach the registers (round 3).  The transform wants lane t to hold COLUMN t & 3 of the four rows of its
// panel; loaded that way every lane issues four 8-byte loads and a 128-byte line (4 rows x 4 columns of a panel) is requested
// in sixteen pieces -- `tools/microbench/rmw_bench` (h) / (i): that gather reads a 4096^2 half spectrum at 4.7 TB/s, the same
// lines requested as ONE 32-byte row of the panel per lane (a quad of lanes = the whole line) at 6.7 TB/s, at any occupancy.
// So lane t loads ROW t & 3 -- four columns, two 16-byte loads -- and the quad transposes its 4 x 4 block in registers: two
// rounds of `v_cndmask_b32_dpp` (quad_perm [1,0,3,2], then [2,3,0,1]), 16 VALU instructions per panel and lane, no LDS.
// Measured (passbench, A/B builds on one box): with the transforms compiled out C1 14.5 -> 11.0 and C2 29.9 -> 24.1 us per
// 4096^2 image; with them the passes alone do not move (C1 13.3 -> 13.8: four waves per SIMD keep the VALU 60-70 % busy and the
// transposes are VALU work) but the two-stream batch does, 87.1 -> 86.2 us per image (twice: +1.1 / +1.6 %), because the
// other stream's passes get the memory system sooner.  Rows of 2048 points and fewer: 1-2 % slower, so FDR_ROWS_LOAD32 applies
// to 4096- and 8192-point rows (FDR_ROWS_LOAD32_LOG .. _MAX); at 8192 points it is what lets the one-group kernel fit 128
// registers (the serialised transposes bound the live set; the 8-byte gathers spilled 13-22 registers there).
#ifndef FDR_ROWS_LOAD32
#define FDR_ROWS_LOAD32 1
#endif
#ifndef FDR_ROWS_LOAD32_LOG
#define FDR_ROWS_LOAD32_LOG 12
#endif
#ifndef FDR_ROWS_LOAD32_LOG_MAX
#define FDR_ROWS_LOAD32_LOG_MAX 13
#endif
template <int CTRL>
__device__ __forceinline__ float2 quad_swap(float2 v) {  // the value the lane CTRL points at holds (a permutation inside every quad)
#if defined(__HIP_DEVICE_COMPILE__)
    const int x = __builtin_amdgcn_mov_dpp(__float_as_int(v.x), CTRL, 0xF, 0xF, false);  // (every lane is written: no `old` value)
    const int y = __builtin_amdgcn_mov_dpp(__float_as_int(v.y), CTRL, 0xF, 0xF, false);
    return make_float2(__int_as_float(x), __int_as_float(y));
#else
    return v;
#endif
}
// r[c] = (row l, column c) on lane l of the quad  ->  q[r] = (row r, column l)
#ifndef FDR_QUAD_ASM
#define FDR_QUAD_ASM 1
#endif
// d = (lane in MASK) ? keep : (value of `from` on the lane quad_perm points at), both halves of two float2: four
// v_cndmask_b32_dpp (select and cross-lane read in ONE instruction; hipcc emits v_mov_b32_dpp + v_cndmask_b32 for the C form
// below, twice the VALU work in a pass that is short of VALU issue slots).  s_nop 1: a DPP operand written by the
// preceding VALU instruction needs two wait states, and the hazard recogniser does not look into inline asm.
#define FDR_QUAD_SEL(MASK, PERM, d0, d1, from0, from1, keep0, keep1)                                                        \
    asm("s_nop 1\n\ts_mov_b32 vcc_lo, " MASK "\n\ts_mov_b32 vcc_hi, " MASK "\n\t"                                          \
        "v_cndmask_b32_dpp %0, %4, %8, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                              \
        "v_cndmask_b32_dpp %1, %5, %9, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                              \
        "v_cndmask_b32_dpp %2, %6, %10, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_cndmask_b32_dpp %3, %7, %11, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf"                                  \
        : "=&v"(d0.x), "=&v"(d0.y), "=&v"(d1.x), "=&v"(d1.y)                                                               \
        : "v"(from0.x), "v"(from0.y), "v"(from1.x), "v"(from1.y), "v"(keep0.x), "v"(keep0.y), "v"(keep1.x), "v"(keep1.y)   \
        : "vcc")
__device__ __forceinline__ void quad_transpose(int lane, const float2 (&r)[4], float2& q0, float2& q1, float2& q2, float2& q3) {
#if FDR_QUAD_ASM && defined(__HIP_DEVICE_COMPILE__)
    (void)lane;  // (the masks below are the physical lane's low bits, which the logical thread index keeps)
    float2 a00, a01, a10, a11;
    FDR_QUAD_SEL("0x55555555", "[1,0,3,2]", a00, a10, r[1], r[3], r[0], r[2]);  // even lanes keep columns 0 / 2, odd lanes take the
    FDR_QUAD_SEL("0xaaaaaaaa", "[1,0,3,2]", a01, a11, r[0], r[2], r[1], r[3]);  // neighbour's 1 / 3 (and the other way round)
    FDR_QUAD_SEL("0x33333333", "[2,3,0,1]", q0, q1, a10, a11, a00, a01);
    FDR_QUAD_SEL("0xcccccccc", "[2,3,0,1]", q2, q3, a00, a01, a10, a11);
#else
    constexpr int X1 = 0xB1, X2 = 0x4E;  // quad_perm [1,0,3,2] (lane ^ 1), [2,3,0,1] (lane ^ 2)
    const bool odd = (lane & 1) != 0, hi = (lane & 2) != 0;
    const float2 s0 = quad_swap<X1>(r[0]), s1 = quad_swap<X1>(r[1]), s2 = quad_swap<X1>(r[2]), s3 = quad_swap<X1>(r[3]);
    const float2 a00 = odd ? s1 : r[0], a01 = odd ? r[1] : s0;  // column (lane & 1) of rows 2 j, 2 j + 1 (j = lane / 2) ...
    const float2 a10 = odd ? s3 : r[2], a11 = odd ? r[3] : s2;  // ... and column 2 + (lane & 1)
    const float2 t00 = quad_swap<X2>(a00), t01 = quad_swap<X2>(a01), t10 = quad_swap<X2>(a10), t11 = quad_swap<X2>(a11);
    q0 = hi ? t10 : a00; q1 = hi ? t11 : a01;
    q2 = hi ? a10 : t00; q3 = hi ? a11 : t01;
#endif
}

template <int LOGL, class Core>
__device__ __forceinline__ void rows4_load_direct(const RowArgs& a, int rr, int tid, float2 (&y)[4][Core::V / 2], unsigned scale = 1u) {
    static_assert(Core::RHO0 >= 2 && Core::LOGR0 >= 2, "n = t + q Q with Q a multiple of 4");
    constexpr int HQ = Core::RHO0 / 2;
    constexpr bool kRowLoads = FDR_ROWS_LOAD32 && LOGL >= FDR_ROWS_LOAD32_LOG && LOGL <= FDR_ROWS_LOAD32_LOG_MAX;
    const unsigned ps = (unsigned)a.pstride * scale;  // scale = 0: every address collapses onto rows 0..3 of panel 0
    rr = (int)((unsigned)rr * scale);
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u) {
        const unsigned t = (unsigned)(tid + u * Core::T);  // stored column t + q Q -> panel q Q / 4 + t / 4, column t & 3
        // kRowLoads: row rr + (t & 3) of the panel, its 4 columns; else column t & 3 of rows rr .. rr + 3
        const unsigned off = kRowLoads ? (t >> 2) * ps + ((unsigned)rr + (t & 3u) * scale) * 4u : (t >> 2) * ps + (t & 3u) * scale + (unsigned)rr * 4u;
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int j = u * HQ + q;
#ifdef FDR_DEBUG_SKIP_MEM  // timing-only builds
            y[0][j] = y[1][j] = y[2][j] = y[3][j] = make_float2((float)(off + q), 1.0f);
#else
            const float2* p = a.src_c + (size_t)((q << Core::LOGR0) >> 2) * ps + off;
            if constexpr (kRowLoads) {
                const float4 lo = reinterpret_cast<const float4*>(p)[0], hi = reinterpret_cast<const float4*>(p)[1];
                y[0][j] = make_float2(lo.x, lo.y); y[1][j] = make_float2(lo.z, lo.w);
                y[2][j] = make_float2(hi.x, hi.y); y[3][j] = make_float2(hi.z, hi.w);
            } else {
                y[0][j] = p[0]; y[1][j] = p[4]; y[2][j] = p[8]; y[3][j] = p[12];
            }
#endif
        }
    }
#ifndef FDR_DEBUG_SKIP_MEM
    if constexpr (kRowLoads) {  // (every load of the group is requested before the first transpose)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < Core::NU0 * HQ; ++j) {
            // one panel after the other: left alone hipcc interleaves all the transposes and their temporaries push a
            // 128-register kernel over the edge (this asm makes panel j's inputs depend on panel j-1's results)
            if (j > 0)
                asm volatile("" : "+v"(y[0][j].x), "+v"(y[0][j].y), "+v"(y[1][j].x), "+v"(y[1][j].y), "+v"(y[2][j].x), "+v"(y[2][j].y),
                             "+v"(y[3][j].x), "+v"(y[3][j].y), "+v"(y[0][j - 1].x), "+v"(y[1][j - 1].y), "+v"(y[2][j - 1].x), "+v"(y[3][j - 1].y));
            const float2 r[4] = {y[0][j], y[1][j], y[2][j], y[3][j]};
            quad_transpose(tid, r, y[0][j], y[1][j], y[2][j], y[3][j]);
        }
    }
#endif
}
// z[0] = Y_a + i Y_b of rows 0, 1, z[1] of rows 2, 3; grp_lds: the thread group's two exchange buffers.  Barriers inside
// (every thread of the workgroup must come here); returns with both buffers free again.
// ONE_BUF: the thread group owns a single exchange buffer (more workgroups per CU): the two packed pairs hand their mirrored
// halves over one after the other (two more barriers), same values.
template <int LOGL, class Core, bool ONE_BUF = false>
__device__ __forceinline__ void rows4_pack_mirror(int tid, const float2 (&y)[4][Core::V / 2], float2 (&z)[2][Core::V], float2* grp_lds) {
    using St = typename Core::St;
    constexpr int L = St::L, HQ = Core::RHO0 / 2;
    if constexpr (ONE_BUF) {
        float2* m = grp_lds;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            FDR_JITTER(3011 + p);
#pragma unroll
            for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
                for (int q = 0; q < HQ; ++q) {
                    const int j = u * HQ + q, s = u * Core::RHO0 + q;
                    const int n = Core::in_index(tid, u, q);
                    const float2 ya = y[2 * p][j], yb = y[2 * p + 1][j];
                    z[p][s] = make_float2(ya.x - yb.y, ya.y + yb.x);  // Y_a + i Y_b
                    const int k = (L - n) & (L - 1);
                    if (!(u == 0 && q == 0) || tid != 0) m[k] = make_float2(ya.x + yb.y, yb.x - ya.y);
                }
            __syncthreads();
            FDR_JITTER(3013 + p);
#pragma unroll
            for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
                for (int q = HQ; q < Core::RHO0; ++q) z[p][u * Core::RHO0 + q] = m[Core::in_index(tid, u, q)];
            if (tid == 0) {
                z[p][0] = make_float2(y[2 * p][0].x, y[2 * p + 1][0].x);
                z[p][HQ] = make_float2(y[2 * p][0].y, y[2 * p + 1][0].y);
            }
            __syncthreads();  // the next pair's hand-over (or the transform's first exchange) overwrites the buffer
        }
        return;
    }
    float2* m0 = grp_lds;
    float2* m1 = grp_lds + St::BUF;
    FDR_JITTER(3001);
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int j = u * HQ + q, s = u * Core::RHO0 + q;
            const int n = Core::in_index(tid, u, q);
            const float2 y0 = y[0][j], y1 = y[1][j], y2 = y[2][j], y3 = y[3][j];
            z[0][s] = make_float2(y0.x - y1.y, y0.y + y1.x);  // Y_a + i Y_b
            z[1][s] = make_float2(y2.x - y3.y, y2.y + y3.x);
            // conj(Y_a) + i conj(Y_b) belongs to index N - n (n = 0 has no mirror: it is the packed DC / Nyquist column)
            const int k = (L - n) & (L - 1);
            if (!(u == 0 && q == 0) || tid != 0) {
                m0[k] = make_float2(y0.x + y1.y, y1.x - y0.y);
                m1[k] = make_float2(y2.x + y3.y, y3.x - y2.y);
            }
        }
    __syncthreads();
    FDR_JITTER(3002);
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
        for (int q = HQ; q < Core::RHO0; ++q) {
            const int s = u * Core::RHO0 + q;
            const int k = Core::in_index(tid, u, q);
            z[0][s] = m0[k];
            z[1][s] = m1[k];
        }
    if (tid == 0) {  // n = 0 (DC) and n = L/2 (Nyquist): real values packed as (DC, Nyquist) in stored column 0
        constexpr int SN = HQ;  // slot of n = L/2 (u = 0, q = RHO0/2); its LDS cell was never written
        z[0][0] = make_float2(y[0][0].x, y[1][0].x);
        z[1][0] = make_float2(y[2][0].x, y[3][0].x);
        z[0][SN] = make_float2(y[0][0].y, y[1][0].y);
        z[1][SN] = make_float2(y[2][0].y, y[3][0].y);
    }
    __syncthreads();  // the transform's first exchange may overwrite either buffer
}

// Epilogue of the inverse row passes for one 4-row group (z = two packed transforms: rows r0, r0+1 and r0+2, r0+3).
//   OUT 0 (pass C') : real plane + running min/max
//   OUT 1 (pass C1) : running min/max only -- nothing is stored
//   OUT 2 (pass C2) : value * fscale + fshift (two roundings, as normalize_kernel) to the cropped result, non-temporal
template <class Core, int OUT, int V>
__device__ __forceinline__ void rows4_inv_epilogue(const RowArgs& a, const int r0, const int tq, const float2 (&z)[2][V], const float fscale,
                                                   const float fshift, float& mn, float& mx) {
    constexpr int T = Core::T, L = T * V;
    if constexpr (OUT == 0) {
        // four row bases + the lane's column: the stores need no per-element 64-bit address arithmetic
        float* o0 = a.dst_real + (size_t)r0 * L + tq;
        float* o1 = o0 + L;
        float* o2 = o1 + L;
        float* o3 = o2 + L;
#pragma unroll
        for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
            for (int q = 0; q < Core::RHOL; ++q) {
                const int s = u * Core::RHOL + q;
                const int c = u * T + (q << Core::LOGOUT);
                o0[c] = z[0][s].x; o1[c] = z[0][s].y; o2[c] = z[1][s].x; o3[c] = z[1][s].y;
            }
    }
    if constexpr (OUT == 2) {
        float* o0 = a.out + (size_t)r0 * a.out_stride + tq;
        float* o1 = o0 + a.out_stride;
        float* o2 = o1 + a.out_stride;
        float* o3 = o2 + a.out_stride;
        if (r0 + 3 < a.out_rows && a.out_cols >= L) {  // nothing cropped in this group
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) {
                    const int s = u * Core::RHOL + q;
                    const int c = u * T + (q << Core::LOGOUT);
                    const float p0 = z[0][s].x * fscale, p1 = z[0][s].y * fscale, p2 = z[1][s].x * fscale, p3 = z[1][s].y * fscale;
                    __builtin_nontemporal_store(p0 + fshift, o0 + c);
                    __builtin_nontemporal_store(p1 + fshift, o1 + c);
                    __builtin_nontemporal_store(p2 + fshift, o2 + c);
                    __builtin_nontemporal_store(p3 + fshift, o3 + c);
                }
        } else {
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) {
                    const int s = u * Core::RHOL + q;
                    const int c = u * T + (q << Core::LOGOUT);
                    if (tq + c < a.out_cols) {
                        const float p0 = z[0][s].x * fscale, p1 = z[0][s].y * fscale, p2 = z[1][s].x * fscale, p3 = z[1][s].y * fscale;
                        if (r0 + 0 < a.out_rows) o0[c] = p0 + fshift;
                        if (r0 + 1 < a.out_rows) o1[c] = p1 + fshift;
                        if (r0 + 2 < a.out_rows) o2[c] = p2 + fshift;
                        if (r0 + 3 < a.out_rows) o3[c] = p3 + fshift;
                    }
                }
        }
    } else {
        if (r0 + 3 < a.mm_rows && a.mm_cols >= L) {  // whole group counted (always, with FDR_NORM_PADDED)
#pragma unroll
            for (int s = 0; s < V; ++s) {
                mn = fdr_min3(fdr_min3(mn, z[0][s].x, z[0][s].y), z[1][s].x, z[1][s].y);
                mx = fdr_max3(fdr_max3(mx, z[0][s].x, z[0][s].y), z[1][s].x, z[1][s].y);
            }
        } else {
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) {
                    const int s = u * Core::RHOL + q;
                    const int n = Core::out_index(tq, u, q);
                    const float r[4] = {z[0][s].x, z[0][s].y, z[1][s].x, z[1][s].y};
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (r0 + b < a.mm_rows && n < a.mm_cols) {
                            mn = fminf(mn, r[b]);
                            mx = fmaxf(mx, r[b]);
                        }
                }
        }
    }
}

// HALF: the row spectra hold columns 0 .. N/2-1 only, column 0 packed as Y[m,0] + i Y[m,N/2] (see the forward
// kernel); the upper half is rebuilt as the conjugate of the mirrored column (rows4_pack_mirror).
template <int LOGL, bool HALF, int OUT = 0>
__global__ __launch_bounds__((Rows4PackGeom<LOGL, true>::THREADS), (HALF ? Rows4PackGeom<LOGL, true>::INV_WAVES_PER_SIMD : 1)) void fft_rows4_inv_packed_kernel(const RowArgs a0,
                                                                                          const float2* __restrict__ tw_fwd) {
    RowArgs a = a0;
    if (a0.batch.nimg > 1) {  // blockIdx.y = image
        a.src_c = pick_image(a0.batch.spec, blockIdx.y);
        if constexpr (OUT == 0) a.dst_real = pick_image(a0.batch.raw, blockIdx.y);
        if constexpr (OUT == 2) a.out = pick_image(a0.batch.out, blockIdx.y);
        a.mm_part = pick_image(a0.batch.mm_part, blockIdx.y);
    }
    float fscale = 0.f, fshift = 0.f;
    using Geo = Rows4PackGeom<LOGL, true>;
    using St = typename Geo::St;
    constexpr int G = Geo::G, T = St::T;
    constexpr int NBUF = (HALF && LOGL >= FDR_ROWS_INV_NBUF1_MIN && Geo::LOGV == 4) ? 1 : 2;  // 1: one exchange buffer per thread group (more workgroups per CU)
    using Core = FftCore<LOGL, 2, NBUF, PolicyFast, Geo::LOGV, (St::lr(0) == 1 && T >= 64 && FDR_SWAP0)>;
    __shared__ float2 lds[G * NBUF * St::BUF];
    const int g = threadIdx.x >> St::LOGT, tid = Core::thread_index(threadIdx.x & (T - 1));
    const int M = a.M;
    const int r0 = (blockIdx.x * G + g) * 4;
    const bool active = r0 < M;
    const int rr = active ? r0 : 0;

    typename Core::Bases bases;
    Core::init_bases(bases, tw_fwd, tid);

    float2 z[2][Core::V];
    if constexpr (HALF && LOGL >= 5) {  // direct half from memory, mirrored half through LDS
        float2 y[4][Core::V / 2];
        rows4_load_direct<LOGL, Core>(a, rr, tid, y);
        // pass C2: the partials are folded BEHIND the group's own loads (a workgroup lives for one group here: a fold in
        // front of them adds its full memory latency to every workgroup -- measured +6 us per 4096^2 image)
        if constexpr (OUT == 2) block_fold_partials(a.mm_part, a.n_part, fscale, fshift);
        rows4_pack_mirror<LOGL, Core, NBUF == 1>(tid, y, z, lds + g * NBUF * St::BUF);
    } else {
        float2 y[4][Core::V];
        rows4_load_raw<LOGL, HALF, Core>(a, rr, tid, y);
        if constexpr (OUT == 2) block_fold_partials(a.mm_part, a.n_part, fscale, fshift);
        rows4_pack<LOGL, HALF, Core>(tid, y, z);
    }

    Core::template run<0, true>(z, lds + g * NBUF * St::BUF, tw_fwd, bases, tid);

    float mn = __builtin_inff(), mx = -__builtin_inff();
    if (active) rows4_inv_epilogue<Core, OUT, Core::V>(a, r0, tid, z, fscale, fshift, mn, mx);
    if constexpr (OUT != 2) block_minmax_store(mn, mx, a.mm_part, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Persistent form of pass C' (see fft_rows4_fwd_pers_kernel): the raw half spectrum of the NEXT 4-row group is
// requested (rows4_load_raw into y) before the current group is transformed and stored.  grid (workgroups, images):
// a workgroup stays inside one image, so it writes ONE (min, max) partial for all its groups.
// ---------------------------------------------------------------------------------------------
template <int LOGL, int LOGV, bool HALF, int OUT = 0>
__global__ __launch_bounds__((RowsPersGeom<LOGL, LOGV>::THREADS), (RowsPersGeom<LOGL, LOGV>::WAVES_PER_SIMD)) void fft_rows4_inv_pers_kernel(
    const RowArgs a0, const float2* __restrict__ tw_fwd, const int ngroups) {
    RowArgs a = a0;
    if (a0.batch.nimg > 1) {  // blockIdx.y = image
        a.src_c = pick_image(a0.batch.spec, blockIdx.y);
        if constexpr (OUT == 0) a.dst_real = pick_image(a0.batch.raw, blockIdx.y);
        if constexpr (OUT == 2) a.out = pick_image(a0.batch.out, blockIdx.y);
        a.mm_part = pick_image(a0.batch.mm_part, blockIdx.y);
    }
    float fscale = 0.f, fshift = 0.f;
    if constexpr (OUT == 2) block_fold_partials(a.mm_part, a.n_part, fscale, fshift);
    using St = Steps<LOGL, LOGV>;
    constexpr int V = St::V;
    using Core = FftCore<LOGL, 2, 2, PolicyFast, LOGV, (St::lr(0) == 1 && FDR_SWAP0)>;  // 8192 points: wave-local first exchange
    __shared__ float2 lds[2 * St::BUF];
    const int tid = Core::thread_index(threadIdx.x);

    typename Core::Bases bases;
    Core::init_bases(bases, tw_fwd, tid);

    float mn = __builtin_inff(), mx = -__builtin_inff();
    constexpr bool MIRROR = HALF;  // direct half from memory, mirrored half through LDS (rows4_pack_mirror)
    float2 y[4][MIRROR ? V / 2 : V], z[2][V];
    auto request = [&](int g, unsigned scale) __attribute__((always_inline)) {
        int tl = tid;  // opaque copy: the per-lane panel offsets are recomputed per group instead of living in ~16 registers
        asm volatile("" : "+v"(tl));
        if constexpr (MIRROR) rows4_load_direct<LOGL, Core>(a, g * 4, tl, y, scale);
        else rows4_load_raw<LOGL, HALF, Core>(a, g * 4, tl, y, scale);
    };
    auto pack = [&]() __attribute__((always_inline)) {
        if constexpr (MIRROR) rows4_pack_mirror<LOGL, Core>(tid, y, z, lds);
        else rows4_pack<LOGL, HALF, Core>(tid, y, z);
    };
    auto body = [&](int g) __attribute__((always_inline)) {
        {
            int tr = tid;  // opaque copy: the exchange addresses are recomputed per group, not carried (and spilled) across the loop
            asm volatile("" : "+v"(tr));
            Core::template run<0, true>(z, lds, tw_fwd, bases, tr);
        }
        const int r0 = g * 4;
        int tq = tid;  // opaque copy: keeps the store addresses from being hoisted out of the group loop
        asm volatile("" : "+v"(tq));
        rows4_inv_epilogue<Core, OUT, V>(a, r0, tq, z, fscale, fshift, mn, mx);
    };
    // loop shape as in fft_rows4_fwd_pers_kernel: the prefetch is waited for at the bottom (pack), behind the stores
    int grp = blockIdx.x;  // (the grid never exceeds the number of groups)
    request(grp, 1u);
    landed_f2(y);
    pack();
    while (true) {
        const int gn = grp + (int)gridDim.x;
        const bool more = gn < ngroups;
        if constexpr (LOGV == 3) {
            // 128-register budget: the three hoisted twiddle bases do not survive the loop in registers; fetched again
            // here (L1 hits), BEFORE the prefetch is issued -- a reload behind it (a spill slot is vector memory too)
            // could only be waited for together with the whole prefetch
            int tb = tid;
            asm volatile("" : "+v"(tb));
            Core::init_bases(bases, tw_fwd, tb);
        }
        request(more ? gn : 0, more ? 1u : 0u);
        body(grp);
        if (!more) break;
        landed_f2(y);     // wait for the prefetch here, behind this group's stores
        __syncthreads();  // keeps the two transforms' LDS traffic apart
        pack();
        grp = gn;
    }
    if constexpr (OUT != 2) block_minmax_store(mn, mx, a.mm_part, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Row passes for ONE small image (single-image calls, rows of 256 .. 2048 points, at most 2048 rows; BASELINE config 2).
// A lone 1024^2 image is 256 four-row groups: the kernels above give each group to ONE thread group that runs its two
// packed transforms one after the other (128 workgroups of two groups at 1024 points), and with nothing else in flight the
// launch lasts as long as that dependent chain.  Here the two packed pairs of a group go to two thread groups of one
// workgroup (B = 1 transform each, twice the waves on half the chain, one workgroup per group: 256 of them at 1024^2); the
// step plan, policy and pack / separate formulas are those of the packed kernels, so the bits are the same.
// ---------------------------------------------------------------------------------------------
#ifndef FDR_ROWS_SPLIT
#define FDR_ROWS_SPLIT 1
#endif
// (INV: the inverse kernel follows the batched inverse kernels' step plan -- 16 values per thread from FDR_ROWS_INV_V16_MIN points
// on -- so that ONE image restored alone and the same image inside a batch come out with identical bits.)
template <int LOGL, bool INV = false>
struct RowsSplitGeom {
    static constexpr int LOGV = Rows4PackGeom<LOGL, INV>::LOGV;
    using St = Steps<LOGL, LOGV>;
    static constexpr int T = St::T;
    static constexpr int THREADS = 2 * T;
    static constexpr bool SWAP = St::lr(0) == 1 && T >= 64 && FDR_SWAP0;
};
static inline bool rows4_use_split(int logl, int M, int nimg, int half) {
    return FDR_ROWS_SPLIT && nimg <= 1 && half && logl >= 8 && logl <= 11 && (M & 3) == 0 && M > 0 && M <= 2048;
}

template <int LOGL>
__global__ __launch_bounds__(RowsSplitGeom<LOGL>::THREADS) void fft_rows4_fwd_split_kernel(const RowArgs a, const float2* __restrict__ tw_fwd) {
    using Geo = RowsSplitGeom<LOGL>;
    using St = typename Geo::St;
    constexpr int T = St::T, L = St::L;
    using Core = FftCore<LOGL, 1, 2, PolicyFast, 3, Geo::SWAP>;
    __shared__ float2 lds[2 * 2 * St::BUF];
    const int p = (int)(threadIdx.x >> St::LOGT);  // packed pair of the group: rows 2p, 2p + 1
    const int tid = Core::thread_index((int)(threadIdx.x & (T - 1)));
    float2* grp_lds = lds + p * 2 * St::BUF;
    const int r0 = (int)blockIdx.x * 4;
    const int ra = r0 + 2 * p, rb = ra + 1;

    typename Core::Bases bases;
    Core::init_bases(bases, tw_fwd, tid);

    float2 z[1][8];
    const float* __restrict__ rowa = a.src_real + (size_t)(ra < a.src_rows ? ra : 0) * a.src_stride;
    const float* __restrict__ rowb = a.src_real + (size_t)(rb < a.src_rows ? rb : 0) * a.src_stride;
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHO0; ++q) {
            const int n = Core::in_index(tid, u, q);
            float xa = 0.f, xb = 0.f;
            if (n < a.src_cols) {
                if (ra < a.src_rows) xa = __builtin_nontemporal_load(rowa + n);
                if (rb < a.src_rows) xb = __builtin_nontemporal_load(rowb + n);
            }
            z[0][u * Core::RHO0 + q] = make_float2(xa, xb);
        }

    Core::template run<0, false>(z, grp_lds, tw_fwd, bases, tid);

    // the pair's packed spectrum in natural order into the buffer the last exchange did NOT use (free: its last readers
    // passed that exchange's barrier), then the whole workgroup separates: a quad of lanes owns one 128-byte line
    constexpr int SEQ1 = Core::SLOTS;
    float2* mine = grp_lds + (SEQ1 & 1) * St::BUF;
    FDR_JITTER(2031);
#pragma unroll
    for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHOL; ++q) mine[Core::out_index(tid, u, q)] = z[0][u * Core::RHOL + q];
    __syncthreads();
    FDR_JITTER(2032);
    const int w = (int)threadIdx.x;
    const int j = w & 3;                                                        // row inside the 4-row group
    const float2* buf = lds + (j >> 1) * 2 * St::BUF + (SEQ1 & 1) * St::BUF;    // packed pair holding row j
    const bool odd = (j & 1) != 0;                                              // row b of the pair (else row a)
    constexpr int NIT = (L / 8) / (2 * T / 4);                                  // panels per lane (half spectrum)
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int c = (w >> 2) + (2 * T / 4) * i;  // panel
        const int n0 = c * 4;
        float2 o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float2 zn = buf[n0 + k];
            const float2 zm = buf[(L - n0 - k) & (L - 1)];
            o[k] = odd ? make_float2(0.5f * (zn.y + zm.y), 0.5f * (zm.x - zn.x))
                       : make_float2(0.5f * (zn.x + zm.x), 0.5f * (zn.y - zm.y));
        }
        if (n0 == 0) {  // packed column: (X[0], X[N/2]), both real: Re/Im of Z[0] and Z[N/2]
            const float2 z0 = buf[0], zq = buf[L / 2];
            o[0] = odd ? make_float2(z0.y, zq.y) : make_float2(z0.x, zq.x);
        }
        store4(a.dst_c + (size_t)c * a.pstride + (size_t)(r0 + j) * 4, o[0], o[1], o[2], o[3]);
    }
}

// OUT as in rows4_inv_epilogue: 0 raw real plane + min/max, 1 min/max only, 2 normalised and cropped
template <int LOGL, int OUT>
__global__ __launch_bounds__((RowsSplitGeom<LOGL, true>::THREADS)) void fft_rows4_inv_split_kernel(const RowArgs a, const float2* __restrict__ tw_fwd) {
    using Geo = RowsSplitGeom<LOGL, true>;
    using St = typename Geo::St;
    constexpr int T = St::T, L = St::L, V = St::V;
    using Core = FftCore<LOGL, 1, 2, PolicyFast, Geo::LOGV, Geo::SWAP>;
    static_assert(Core::RHO0 >= 2 && Core::LOGR0 >= 2, "n = t + q Q with Q a multiple of 4");
    constexpr int HQ = Core::RHO0 / 2;
    __shared__ float2 lds[2 * 2 * St::BUF];
    const int p = (int)(threadIdx.x >> St::LOGT);
    const int tid = Core::thread_index((int)(threadIdx.x & (T - 1)));
    float2* grp_lds = lds + p * 2 * St::BUF;
    const int r0 = (int)blockIdx.x * 4;
    const int ra = r0 + 2 * p, rb = ra + 1;

    typename Core::Bases bases;
    Core::init_bases(bases, tw_fwd, tid);

    // direct half of the pair's two Hermitian row spectra (stored columns n < L/2), see rows4_load_direct
    float2 ya[V / 2], yb[V / 2];
    const unsigned ps = (unsigned)a.pstride;
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u) {
        const unsigned t = (unsigned)(tid + u * Core::T);
        const unsigned off_d = (t >> 2) * ps + (t & 3u) + (unsigned)ra * 4u;
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const float2* ptr = a.src_c + (size_t)((q << Core::LOGR0) >> 2) * ps + off_d;
            ya[u * HQ + q] = ptr[0];
            yb[u * HQ + q] = ptr[4];
        }
    }
    float fscale = 0.f, fshift = 0.f;
    if constexpr (OUT == 2) block_fold_partials(a.mm_part, a.n_part, fscale, fshift);  // behind the group's own loads

    // Z[n] = Y_a[n] + i Y_b[n] for the direct half; conj(Y_a) + i conj(Y_b) belongs to index L - n: handed over in LDS
    float2 z[1][V];
    float2* m = grp_lds;
    FDR_JITTER(3021);
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int jj = u * HQ + q, s = u * Core::RHO0 + q;
            const int n = Core::in_index(tid, u, q);
            const float2 y0 = ya[jj], y1 = yb[jj];
            z[0][s] = make_float2(y0.x - y1.y, y0.y + y1.x);
            const int k = (L - n) & (L - 1);
            if (!(u == 0 && q == 0) || tid != 0) m[k] = make_float2(y0.x + y1.y, y1.x - y0.y);
        }
    __syncthreads();
    FDR_JITTER(3022);
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
        for (int q = HQ; q < Core::RHO0; ++q) z[0][u * Core::RHO0 + q] = m[Core::in_index(tid, u, q)];
    if (tid == 0) {  // n = 0 (DC) and n = L/2 (Nyquist): real values packed as (DC, Nyquist) in stored column 0
        z[0][0] = make_float2(ya[0].x, yb[0].x);
        z[0][HQ] = make_float2(ya[0].y, yb[0].y);
    }
    __syncthreads();  // the transform's first exchange may overwrite the buffer

    Core::template run<0, true>(z, grp_lds, tw_fwd, bases, tid);

    float mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHOL; ++q) {
            const int n = Core::out_index(tid, u, q);
            const float va = z[0][u * Core::RHOL + q].x, vb = z[0][u * Core::RHOL + q].y;  // rows ra, rb
            if constexpr (OUT == 0) {
                a.dst_real[(size_t)ra * L + n] = va;
                a.dst_real[(size_t)rb * L + n] = vb;
            }
            if constexpr (OUT == 2) {
                const float pa = va * fscale, pb = vb * fscale;
                if (n < a.out_cols) {
                    if (ra < a.out_rows) __builtin_nontemporal_store(pa + fshift, a.out + (size_t)ra * a.out_stride + n);
                    if (rb < a.out_rows) __builtin_nontemporal_store(pb + fshift, a.out + (size_t)rb * a.out_stride + n);
                }
            } else {
                if (n < a.mm_cols) {
                    if (ra < a.mm_rows) { mn = fminf(mn, va); mx = fmaxf(mx, va); }
                    if (rb < a.mm_rows) { mn = fminf(mn, vb); mx = fmaxf(mx, vb); }
                }
            }
        }
    if constexpr (OUT != 2) block_minmax_store(mn, mx, a.mm_part, (int)blockIdx.x);
}

// persistent pass C' is used when a workgroup gets more than one group (else there is nothing to overlap); always for
// 8192-point rows (see launch_rows4_t)
template <int LOGL>
static int rows4_inv_pers_grid(int M, int num_cu, int nimg);
template <int LOGL>
static bool rows4_inv_use_pers(int M, int num_cu, int nimg) {
    if constexpr (LOGL >= 11) {
        if ((M & 3) != 0) return false;
        return LOGL >= 13 || rows4_inv_pers_grid<LOGL>(M, num_cu, nimg) < (M + 3) / 4;
    } else {
        return false;
    }
}
// persistent pass C': workgroups per image (the number of min/max partials the pass writes for one image)
template <int LOGL>
static int rows4_inv_pers_grid(int M, int num_cu, int nimg) {
    if constexpr (LOGL >= 11) {
        constexpr int LOGV = LOGL >= 13 ? 4 : (LOGL == 12 ? FDR_ROWS12_LOGV : 3);
        using PG = RowsPersGeom<LOGL, LOGV>;
        const int groups = (M + 3) / 4;
        int g = (num_cu > 0 ? num_cu : 256) * PG::WG_PER_CU / (nimg > 1 ? nimg : 1);
        if (g < 1) g = 1;
        return g > groups ? groups : g;
    } else {
        return 0;
    }
}

#ifndef FDR_ROWS_PERSISTENT
#define FDR_ROWS_PERSISTENT 1
#endif
template <int LOGL, int OUT>
static hipError_t launch_rows4_inv_t(const RowArgs& a, const float2* tw, hipStream_t s, int groups, int nimg, dim3 grid, dim3 block) {
    if constexpr (LOGL >= FDR_ROWS_INV_PERS_MIN_LOG && FDR_ROWS_PERSISTENT) {
        constexpr int LOGV = LOGL >= 13 ? 4 : (LOGL == 12 ? FDR_ROWS12_LOGV : 3);
        using PG = RowsPersGeom<LOGL, LOGV>;
        if (rows4_inv_use_pers<LOGL>(a.M, a.num_cu, nimg)) {
            const dim3 pgrid(rows4_inv_pers_grid<LOGL>(a.M, a.num_cu, nimg), nimg), pblock(PG::THREADS);
            if (a.half) hipLaunchKernelGGL((fft_rows4_inv_pers_kernel<LOGL, LOGV, true, OUT>), pgrid, pblock, 0, s, a, tw, groups);
            else if constexpr (OUT == 0) hipLaunchKernelGGL((fft_rows4_inv_pers_kernel<LOGL, LOGV, false, 0>), pgrid, pblock, 0, s, a, tw, groups);
            return hipGetLastError();
        }
    }
    if (a.half) hipLaunchKernelGGL((fft_rows4_inv_packed_kernel<LOGL, true, OUT>), grid, block, 0, s, a, tw);
    else if constexpr (OUT == 0) hipLaunchKernelGGL((fft_rows4_inv_packed_kernel<LOGL, false, 0>), grid, block, 0, s, a, tw);
    return hipGetLastError();
}

template <int LOGL>
static hipError_t launch_rows4_t(RowIn in, RowOut out, const RowArgs& a, const float2* tw, hipStream_t s) {
    using Geo = Rows4PackGeom<LOGL>;
    const int groups = (a.M + 3) / 4;
    const int nimg = a.batch.nimg > 1 ? a.batch.nimg : 1;
    const dim3 grid((groups + Geo::G - 1) / Geo::G, nimg), block(Geo::THREADS);
    using IGeo = Rows4PackGeom<LOGL, true>;  // the inverse kernels' own thread-group shape
    const dim3 igrid((groups + IGeo::G - 1) / IGeo::G, nimg), iblock(IGeo::THREADS);
    if constexpr (LOGL >= 8 && LOGL <= 11) {
        if (rows4_use_split(LOGL, a.M, nimg, a.half)) {  // one small image: two thread groups per 4-row group (see above)
            using SG = RowsSplitGeom<LOGL>;
            const dim3 sgrid(a.M / 4), sblock(SG::THREADS), siblock(RowsSplitGeom<LOGL, true>::THREADS);
            if (in == ROW_IN_REAL && out == ROW_OUT_COMPLEX) {
                if (a.src_rows <= 0 || a.src_cols <= 0) return hipErrorInvalidValue;
                hipLaunchKernelGGL((fft_rows4_fwd_split_kernel<LOGL>), sgrid, sblock, 0, s, a, tw);
            } else if (in == ROW_IN_COMPLEX && out == ROW_OUT_REAL_MINMAX) {
                hipLaunchKernelGGL((fft_rows4_inv_split_kernel<LOGL, 0>), sgrid, siblock, 0, s, a, tw);
            } else if (in == ROW_IN_COMPLEX && out == ROW_OUT_MINMAX_ONLY) {
                hipLaunchKernelGGL((fft_rows4_inv_split_kernel<LOGL, 1>), sgrid, siblock, 0, s, a, tw);
            } else if (in == ROW_IN_COMPLEX && out == ROW_OUT_NORMALIZED) {
                hipLaunchKernelGGL((fft_rows4_inv_split_kernel<LOGL, 2>), sgrid, siblock, 0, s, a, tw);
            } else {
                return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
    }
    if (in == ROW_IN_REAL && out == ROW_OUT_COMPLEX) {
        if constexpr (LOGL >= FDR_ROWS_PERS_MIN_LOG && FDR_ROWS_PERSISTENT) {
            constexpr int LOGV = LOGL >= 13 ? 4 : (LOGL == 12 ? FDR_ROWS12_LOGV : 3);
            using PG = RowsPersGeom<LOGL, LOGV>;
            const int total = groups * nimg;
            int g = (a.num_cu > 0 ? a.num_cu : 256) * PG::WG_PER_CU;
            // persistent, prefetching form -- when a workgroup gets more than one group (else there is nothing to overlap
            // and the one-group-per-workgroup kernel has less to do); always for 8192-point rows, which the 8-value
            // kernel can only run as 1024-thread workgroups
            if ((a.M & 3) == 0 && a.src_rows > 0 && a.src_cols > 0 && (total > g || LOGL >= 13)) {
                if (g > total) g = total;
                const bool interior = a.src_rows >= a.M && a.src_cols >= (1 << LOGL);
                const dim3 pgrid(g), pblock(PG::THREADS);
                if (a.half) {
                    if (interior) hipLaunchKernelGGL((fft_rows4_fwd_pers_kernel<LOGL, LOGV, true, true>), pgrid, pblock, 0, s, a, tw, groups, total);
                    else hipLaunchKernelGGL((fft_rows4_fwd_pers_kernel<LOGL, LOGV, true, false>), pgrid, pblock, 0, s, a, tw, groups, total);
                } else {
                    if (interior) hipLaunchKernelGGL((fft_rows4_fwd_pers_kernel<LOGL, LOGV, false, true>), pgrid, pblock, 0, s, a, tw, groups, total);
                    else hipLaunchKernelGGL((fft_rows4_fwd_pers_kernel<LOGL, LOGV, false, false>), pgrid, pblock, 0, s, a, tw, groups, total);
                }
                return hipGetLastError();
            }
        }
        if (a.half) hipLaunchKernelGGL((fft_rows4_fwd_packed_kernel<LOGL, true>), grid, block, 0, s, a, tw);
        else hipLaunchKernelGGL((fft_rows4_fwd_packed_kernel<LOGL, false>), grid, block, 0, s, a, tw);
    } else if (in == ROW_IN_COMPLEX && out == ROW_OUT_REAL_MINMAX) {
        return launch_rows4_inv_t<LOGL, 0>(a, tw, s, groups, nimg, igrid, iblock);
    } else if (in == ROW_IN_COMPLEX && out == ROW_OUT_MINMAX_ONLY) {
        if (!a.half) return hipErrorInvalidValue;  // two-sweep normalisation: half-spectrum path only
        return launch_rows4_inv_t<LOGL, 1>(a, tw, s, groups, nimg, igrid, iblock);
    } else if (in == ROW_IN_COMPLEX && out == ROW_OUT_NORMALIZED) {
        if (!a.half) return hipErrorInvalidValue;
        return launch_rows4_inv_t<LOGL, 2>(a, tw, s, groups, nimg, igrid, iblock);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int LOGL>
static int rows4_partials_t(int M, int num_cu, int nimg, int half) {
    if (rows4_use_split(LOGL, M, nimg, half)) return M / 4;  // one workgroup, one partial per 4-row group
    if constexpr (LOGL >= FDR_ROWS_INV_PERS_MIN_LOG && FDR_ROWS_PERSISTENT) {
        if (rows4_inv_use_pers<LOGL>(M, num_cu, nimg)) return rows4_inv_pers_grid<LOGL>(M, num_cu, nimg);
    }
    return ((M + 3) / 4 + Rows4PackGeom<LOGL, true>::G - 1) / Rows4PackGeom<LOGL, true>::G;
}

#define FDR_DISPATCH_LOG(var, expr)                                                     \
    switch (var) {                                                                      \
        case 3: { constexpr int LG = 3; return expr; }                                  \
        case 4: { constexpr int LG = 4; return expr; }                                  \
        case 5: { constexpr int LG = 5; return expr; }                                  \
        case 6: { constexpr int LG = 6; return expr; }                                  \
        case 7: { constexpr int LG = 7; return expr; }                                  \
        case 8: { constexpr int LG = 8; return expr; }                                  \
        case 9: { constexpr int LG = 9; return expr; }                                  \
        case 10: { constexpr int LG = 10; return expr; }                                \
        case 11: { constexpr int LG = 11; return expr; }                                \
        case 12: { constexpr int LG = 12; return expr; }                                \
        case 13: { constexpr int LG = 13; return expr; }                                \
        default: break;                                                                 \
    }

hipError_t launch_rows4(int logl, RowIn in, RowOut out, const RowArgs& a, const float2* tw_fwd, hipStream_t s) {
    FDR_DISPATCH_LOG(logl, launch_rows4_t<LG>(in, out, a, tw_fwd, s));
    return hipErrorInvalidValue;
}

int rows4_minmax_partials(int logl, int M, int num_cu, int nimg, int half) {
    FDR_DISPATCH_LOG(logl, rows4_partials_t<LG>(M, num_cu, nimg, half));
    return 0;
}


// ---------------------------------------------------------------------------------------------
// columns of one panel (contiguous M x 4 chunk)
// ---------------------------------------------------------------------------------------------
template <int LOGM>
struct PanelGeom {
    static constexpr int T = Steps<LOGM>::T;
    static constexpr int G = T >= 512 ? 1 : (T >= 256 ? 2 : 4);  // panels per workgroup
    static constexpr int THREADS = T * G;
    // persistent pipelined kernel: two register sets, one workgroup per CU for 512 threads
    static constexpr int PIPE_WAVES_PER_SIMD = THREADS >= 1024 ? 4 : (THREADS >= 512 ? 2 : 1);
    static constexpr int PIPE_WG_PER_CU = THREADS >= 512 ? 1 : 512 / THREADS;
    static constexpr int WAVES_PER_SIMD = THREADS >= 512 ? 4 : 1;
};

// element offsets inside a panel: row m -> m*4 ; the uniform part (q) stays in SGPRs
template <class Core>
__device__ __forceinline__ void panel_store_out(float2* __restrict__ pbase, int tid, const float2 (&d)[4][8]) {
#pragma unroll
    for (int u = 0; u < Core::NUL; ++u) {
        const unsigned toff = (unsigned)(tid + u * Core::T) * 4u;
#pragma unroll
        for (int q = 0; q < Core::RHOL; ++q) {
            const int s = u * Core::RHOL + q;
            store4(pbase + ((size_t)(q << Core::LOGOUT) * 4) + toff, d[0][s], d[1][s], d[2][s], d[3][s]);
        }
    }
}

// forward column FFT of every panel, in place, for the fast path's PSF preparation: only the first
// `nvalid` rows of a panel hold data -- the row pass before it transformed just the row groups the PSF reaches, everything
// below is taken as zero without being read -- and the spectrum leaves as W = conj(H) / (|H|^2 + K) directly; the packed
// DC / Nyquist column (column 0 of panel 0, half spectrum) leaves as its filter slots (packed_column_filter_slot).
// Against the separate row pass over all M rows + column pass + make_filter pass this drops 20 of 24 bytes per pixel.
template <int LOGM>
__global__ __launch_bounds__(PanelGeom<LOGM>::THREADS, PanelGeom<LOGM>::WAVES_PER_SIMD) void fft_cols_panel_fwd_filter_kernel(
    float2* __restrict__ data, const float2* __restrict__ tw_fwd, const size_t pstride, const int npanels, const int nvalid, const float K,
    const int packed0) {
    using St = Steps<LOGM>;
    using Geo = PanelGeom<LOGM>;
    constexpr int G = Geo::G, T = St::T;
    using Core = FftCore<LOGM, 4, 2, PolicyFast>;
    __shared__ float2 lds[G * 2 * St::BUF];
    const int g = threadIdx.x >> St::LOGT, tid = threadIdx.x & (T - 1);
    const int p = blockIdx.x * G + g;
    const bool active = p < npanels;
    float2* pbase = data + (size_t)(active ? p : 0) * pstride;
    typename Core::Bases bases;
    Core::init_bases(bases, tw_fwd, tid);
    float2 v[4][8];
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHO0; ++q) {
            const int s = u * Core::RHO0 + q;
            const int m = Core::in_index(tid, u, q);
            if (m < nvalid) load4(pbase + (size_t)m * 4, v[0][s], v[1][s], v[2][s], v[3][s]);
            else v[0][s] = v[1][s] = v[2][s] = v[3][s] = make_float2(0.f, 0.f);
        }
    Core::template run<0, false>(v, lds + g * 2 * St::BUF, tw_fwd, bases, tid);
    const bool raw0 = packed0 && p == 0;  // uniform per thread group
    if (packed0 && blockIdx.x == 0) {     // uniform per workgroup: the packed column's slots need C[k] and C[M - k]
        float2* buf = lds + g * 2 * St::BUF;
        __syncthreads();  // the transform's last exchange has been read by every wave
        FDR_JITTER(4021);
        if (raw0) {
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) buf[Core::out_index(tid, u, q)] = v[0][u * Core::RHOL + q];
        }
        __syncthreads();
        FDR_JITTER(4022);
        if (raw0) {
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) {
                    const int s = u * Core::RHOL + q, k = Core::out_index(tid, u, q);
                    v[0][s] = packed_column_filter_slot(v[0][s], buf[(St::L - k) & (St::L - 1)], k, St::L, K);
                }
        }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (!raw0) v[0][s] = wiener_filter_fast(v[0][s], K);
        v[1][s] = wiener_filter_fast(v[1][s], K);
        v[2][s] = wiener_filter_fast(v[2][s], K);
        v[3][s] = wiener_filter_fast(v[3][s], K);
    }
    if (active) panel_store_out<Core>(pbase, tid, v);
}

// One panel of pass B' on register set `cur` (spectrum, first-step order) with the filter in `flt`
// (last-step order): forward, multiply, then -- `flt` now free -- queue the NEXT panel's spectrum
// into it, inverse, store, and queue the next panel's filter into `cur`.
// Packed column (half-spectrum mode, column 0 of panel 0): the column carries c[m] = X[m,0] + i X[m,N/2] with both
// parts real, so its transform is C = F0 + i FN with F0, FN Hermitian.  Separate them with the mirrored value
// C[M-k] (one LDS round trip), filter each with its own W, and re-pack Z0 + i ZN; the inverse transform then
// returns the two filtered real columns in the real and imaginary parts.  The filter slot of this column holds
//   S[k] = W0[k] (0 < k < M/2),  S[k] = WN[M-k] (M/2 < k < M),  S[0] = (W0[0], WN[0]),  S[M/2] = (W0[M/2], WN[M/2])
// (W0 = W[.,0], WN = W[.,N/2]; both Hermitian, their values at 0 and M/2 real), built by the PSF column pass (packed_column_filter_slot).
template <int LOGM, class Core, int SEQ>
__device__ __forceinline__ void packed_column_filter(float2 (&cur)[4][8], const float2 (&flt)[4][8], float2* grp_lds, int tid,
                                                     bool apply) {
    using St = Steps<LOGM>;
    constexpr int M = St::L;
    float2* bufc = grp_lds + (SEQ & 1) * St::BUF;
    float2* bufs = grp_lds + ((SEQ + 1) & 1) * St::BUF;
    __syncthreads();  // the other buffer was read by the last exchange of the forward transform
    FDR_JITTER(4001);
#pragma unroll
    for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHOL; ++q) {
            const int k = Core::out_index(tid, u, q);
            bufc[k] = cur[0][u * Core::RHOL + q];
            bufs[k] = flt[0][u * Core::RHOL + q];
        }
    __syncthreads();
    FDR_JITTER(4002);
#pragma unroll
    for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHOL; ++q) {
            const int s = u * Core::RHOL + q;
            const int k = Core::out_index(tid, u, q);
            const int km = (M - k) & (M - 1);
            const float2 c = cur[0][s], cm = bufc[km], sl = flt[0][s], sm = bufs[km];
            const float2 f0 = make_float2(0.5f * (c.x + cm.x), 0.5f * (c.y - cm.y));
            const float2 fn = make_float2(0.5f * (c.y + cm.y), 0.5f * (cm.x - c.x));
            float2 w0, wn;
            if (k == 0 || k == M / 2) { w0 = make_float2(sl.x, 0.f); wn = make_float2(sl.y, 0.f); }
            else if (k < M / 2) { w0 = sl; wn = sm; }
            else { w0 = make_float2(sm.x, -sm.y); wn = make_float2(sl.x, -sl.y); }
            const float2 z0 = cmul_fma(f0, w0), zn = cmul_fma(fn, wn);
            // thread groups of this workgroup that hold other panels only came along for the barriers
            cur[0][s] = apply ? make_float2(z0.x - zn.y, z0.y + zn.x) : cmul_fma(c, sl);
        }
    __syncthreads();  // both buffers were just read: the next exchange may overwrite either
}

// Tile addressing of the persistent kernel: a wave-uniform tile base (SGPRs) plus ONE 32-bit per-lane element offset
// `loff` = (thread group's panel inside the tile) * pstride + tid * 4, so every load / store is
// `global_* v, v_off, s[base:base+1]` and no 64-bit per-lane address lives in VGPRs.  `scale` (0 or 1, uniform)
// collapses a prefetch onto the first 32 bytes of `ubase` when there is no next tile: the loads stay UNCONDITIONAL
// -- a conditional prefetch makes PHIs of (loaded, old) values whose copies hipcc places right behind the loads,
// i.e. it waits for the prefetch before the transform it was meant to hide behind (seen in the ISA as
// `vmcnt(11) .. vmcnt(1)` directly after the 16 loads).
// Pins a wave-uniform GLOBAL address in an SGPR pair.  Without it hipcc re-associates (uniform base + constant) +
// lane offset into (base + lane offset) + constant: one 64-bit VGPR address per load, kept alive for the stores of the
// same tile -- 30+ registers that end up spilled in the 128-data-register kernels.  The pointer keeps its address
// space through the asm (a generic pointer would turn every access into a flat_load).
typedef float nfloat4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) char gchar;
typedef __attribute__((address_space(1))) nfloat4 g_nfloat4;
__device__ __forceinline__ gchar* uniform_gptr(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (gchar*)(((unsigned long long)hi << 32) | lo);
}
// 32 bytes at (uniform base) + (32-bit lane byte offset): global_load_dwordx4 v, v_off, s[base:base+1] {offset:16}.
// (Round 3: the tile loads as non-temporal loads -- to keep the shared filter W in L2 longer -- measured 35.4 -> 35.6 us per
// 4096^2 image, FDR_WPIECE 2 instead of 4: 36.5 us; neither kept.)
// (HIP's float4, field by field: with native vector types the two halves reach the register arrays as <2 x float>
// stores, which SROA does not promote -- the arrays then live in scratch memory.)
#define FDR_GLOAD32(ub, lane_bytes, a, b, c, d)                                                       \
    do {                                                                                              \
        const float4* p_ = reinterpret_cast<const float4*>((const char*)(ub) + (lane_bytes));         \
        const float4 x0_ = p_[0], x1_ = p_[1];                                                        \
        a = make_float2(x0_.x, x0_.y); b = make_float2(x0_.z, x0_.w);                                 \
        c = make_float2(x1_.x, x1_.y); d = make_float2(x1_.z, x1_.w);                                 \
    } while (0)
__device__ __forceinline__ void gstore32(gchar* ub, unsigned lane_bytes, float2 a, float2 b, float2 c, float2 d) {
    float4* p = reinterpret_cast<float4*>((char*)ub + lane_bytes);
    p[0] = make_float4(a.x, a.y, b.x, b.y);
    p[1] = make_float4(c.x, c.y, d.x, d.y);
}

template <class Core, bool OUT_ORDER>
__device__ __forceinline__ void tile_load(const float2* __restrict__ ubase, unsigned loff, unsigned scale, float2 (&d)[4][Core::V]) {
    constexpr int NU = OUT_ORDER ? Core::NUL : Core::NU0, RHO = OUT_ORDER ? Core::RHOL : Core::RHO0;
    constexpr int LOGQ = OUT_ORDER ? Core::LOGOUT : Core::LOGR0;
    // byte offsets in 32 bits: (uniform 64-bit base) + zext(32-bit lane offset) is the form hipcc turns into
    // `global_load_dwordx4 v, v_off, s[base:base+1]`, i.e. ONE address VGPR for the whole tile
    const unsigned lo = loff * scale * 8u;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int q = 0; q < RHO; ++q) {
            const int s = u * RHO + q;
            const unsigned uoff = (unsigned)(((q << LOGQ) + u * Core::T) * 4) * scale;  // uniform, elements
#ifdef FDR_DEBUG_SKIP_MEM  // timing-only builds: pass B' without its HBM traffic
            (void)ubase; (void)uoff;
            d[0][s] = d[1][s] = d[2][s] = d[3][s] = make_float2(__uint_as_float(lo), 1.0f);
#else
            const gchar* ub = uniform_gptr(ubase + uoff);
            FDR_GLOAD32(ub, lo, d[0][s], d[1][s], d[2][s], d[3][s]);
            // keep the two 16-byte halves of a row together in the instruction stream: left alone the scheduler issues the
            // 16 first halves of a tile, then the 16 second halves, and with every wave of an XCD doing the same (8 MB of
            // lines requested before the first second half) part of the lines has left the 4 MiB L2 again by then
            // (measured: +9 % fabric reads in the 4096-point column pass)
            asm volatile("" ::: "memory");
#endif
        }
}
template <class Core>
__device__ __forceinline__ void tile_store(float2* __restrict__ ubase, unsigned loff, const float2 (&d)[4][Core::V]) {
    const unsigned lo = loff * 8u;
#pragma unroll
    for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHOL; ++q) {
            const int s = u * Core::RHOL + q;
            const unsigned uoff = (unsigned)(((q << Core::LOGOUT) + u * Core::T) * 4);
#ifdef FDR_DEBUG_SKIP_MEM
            if (d[0][s].x != 1.2345e-30f) continue;
#endif
            gstore32(uniform_gptr(ubase + uoff), lo, d[0][s], d[1][s], d[2][s], d[3][s]);
        }
}

// Reads every register of a prefetched set through an empty asm, so the compiler places the wait for those loads HERE
// and treats them as landed afterwards.  Used right before the tile's stores are issued: vmcnt counts loads and
// stores in issue order, so a wait for the spectrum prefetch placed after the stores (where the values are first
// used) would also wait for the stores to drain -- ~5 us per tile that the next forward transform should hide.
__device__ __forceinline__ void landed(const float2 (&d)[4][8]) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int s = 0; s < 8; s += 4)
            asm volatile("" ::"v"(d[b][s].x), "v"(d[b][s].y), "v"(d[b][s + 1].x), "v"(d[b][s + 1].y), "v"(d[b][s + 2].x),
                         "v"(d[b][s + 2].y), "v"(d[b][s + 3].x), "v"(d[b][s + 3].y));
}

struct PanelTile {
    float2* data;        // uniform: image base + first panel of the tile
    const float2* filt;  // uniform: filter, same panel
    unsigned loff;       // per lane: (group's panel in the tile) * pstride + tid * 4   [float2 elements]
    int tl;              // tile index inside its image
    bool ok;             // this thread group's panel exists (else it reads the tile's first panel and stores nothing)
};

template <int LOGM, class Core>
__device__ __forceinline__ void panel_tile(float2 (&cur)[4][8], float2 (&flt)[4][8], const PanelTile& c, const PanelTile& n,
                                           unsigned nscale, float2* grp_lds, const typename Core::Bases& bases,
                                           const float2* __restrict__ tw_fwd, int tid, bool packed_tile, bool packed_group) {
    Core::template run<0, false>(cur, grp_lds, tw_fwd, bases, tid);
    // column 0 of panel 0 in half-spectrum mode: uniform branch per workgroup (barriers inside); thread groups of
    // the same workgroup that hold other panels go through the same barriers and keep the plain product
    if (packed_tile) {
        packed_column_filter<LOGM, Core, Core::SLOTS>(cur, flt, grp_lds, tid, packed_group);
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) cur[0][s] = cmul_fma(cur[0][s], flt[0][s]);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        cur[1][s] = cmul_fma(cur[1][s], flt[1][s]);
        cur[2][s] = cmul_fma(cur[2][s], flt[2][s]);
        cur[3][s] = cmul_fma(cur[3][s], flt[3][s]);
    }
    tile_load<Core, false>(n.data, n.loff, nscale, flt);  // next spectrum streams in behind the inverse transform
    constexpr int SEQ1 = Core::SLOTS;
    Core::permute_out_to_in(cur);  // (a renaming of registers when the first and the last radix differ)
    Core::template run<SEQ1, true>(cur, grp_lds, tw_fwd, bases, tid);
    landed(flt);
    if (c.ok) tile_store<Core>(c.data, c.loff, cur);
    tile_load<Core, true>(n.filt, n.loff, nscale, cur);   // next filter streams in behind the next forward transform
}

// The tile sequence of one launch runs over the panels of up to 4 images (PanelBatch): global tile
// t = image * ntiles + tile.  With several images per launch the un-overlapped prologue (first spectrum) and
// epilogue (last inverse + store) of the persistent workgroups amortise over more tiles, and small images fill the chip.
template <int LOGM>
__global__ __launch_bounds__(PanelGeom<LOGM>::THREADS, PanelGeom<LOGM>::PIPE_WAVES_PER_SIMD) void fft_cols_panel_fused_kernel(
    const PanelBatch pb, const float2* __restrict__ filt, const float2* __restrict__ tw_fwd, const unsigned pstride,
    const int npanels, const int ntiles, const int packed0) {
    using St = Steps<LOGM>;
    using Geo = PanelGeom<LOGM>;
    constexpr int G = Geo::G, T = St::T;
    using Core = FftCore<LOGM, 4, 2, PolicyFast>;
    __shared__ float2 lds[G * 2 * St::BUF];
    // one thread group per workgroup (M >= 4096): everything about a tile except tid is wave-uniform
    const int g = G == 1 ? 0 : (int)(threadIdx.x >> St::LOGT);
    const int tid = threadIdx.x & (T - 1);
    float2* grp_lds = lds + g * 2 * St::BUF;
    const int total = ntiles * pb.nimg;
    int t = blockIdx.x;
    if (t >= total) return;  // uniform over the workgroup

    typename Core::Bases bases;
    Core::init_bases(bases, tw_fwd, tid);

    // (image, tile) advance by scalar add / subtract: an integer division per tile would run on the VALU and drag
    // every tile address into VGPRs
    auto tile_of = [&](int img, int tl) {
        PanelTile r;
        r.tl = tl;
        r.ok = tl * G + g < npanels;
        r.loff = (r.ok ? (unsigned)g : 0u) * pstride + (unsigned)tid * 4u;
        const size_t tbase = (size_t)(tl * G) * pstride;
        r.data = pick_image(pb.data, img) + tbase;
        r.filt = filt + tbase;
        return r;
    };
    int img = 0, tl = t;
    while (tl >= ntiles) { tl -= ntiles; ++img; }
    auto advance = [&](int& im, int& tt) {
        tt += (int)gridDim.x;
        while (tt >= ntiles) { tt -= ntiles; ++im; }
    };

    float2 P[4][8], Q[4][8];
    PanelTile c = tile_of(img, tl);
    tile_load<Core, false>(c.data, c.loff, 1u, P);
    tile_load<Core, true>(c.filt, c.loff, 1u, Q);
    while (true) {
        int tn = t + gridDim.x;
        bool more = tn < total;
        int nimg = img, ntl = tl;
        if (more) advance(nimg, ntl);
        PanelTile n = tile_of(nimg, ntl);
        if (!more) n.data = const_cast<float2*>(n.filt);  // dummy prefetch source: read-only memory
        panel_tile<LOGM, Core>(P, Q, c, n, more ? 1u : 0u, grp_lds, bases, tw_fwd, tid, packed0 && c.tl == 0, g == 0);
        if (!more) break;
        t = tn; c = n; img = nimg; tl = ntl;
        tn = t + gridDim.x;
        more = tn < total;
        if (more) advance(nimg, ntl);
        n = tile_of(nimg, ntl);
        if (!more) n.data = const_cast<float2*>(n.filt);
        panel_tile<LOGM, Core>(Q, P, c, n, more ? 1u : 0u, grp_lds, bases, tw_fwd, tid, packed0 && c.tl == 0, g == 0);
        if (!more) break;
        t = tn; c = n; img = nimg; tl = ntl;
    }
}

// ---------------------------------------------------------------------------------------------
// Pass B' with 16 values per thread (radix-16 steps): a 4096-point column takes 256 threads, so a 4-column tile is
// ONE 256-thread workgroup holding 128 data registers per lane, and two such workgroups share a CU (2 x 74 KB of LDS,
// 256 VGPRs each): the hardware overlaps one tile's loads / stores with the other tile's transforms, which the
// single persistent workgroup of the radix-8 kernel has to arrange by hand (and only half manages: DESIGN.md 5).
// 8192-point columns: 512 threads, one workgroup per CU, no spills (the radix-8 kernel needs 1024 threads at 128
// VGPRs there).  One tile per workgroup; the tile sequence runs over the images of the launch.
// ---------------------------------------------------------------------------------------------
#ifndef FDR_WPIECE
#define FDR_WPIECE 4
#endif
#ifndef FDR_SHARE_W
#define FDR_SHARE_W 1
#endif
#ifndef FDR_COLS12_PACKED
#define FDR_COLS12_PACKED 1
#endif
#ifndef FDR_PARK_BASES
#define FDR_PARK_BASES 1  // 8192-point column pass: twiddle bases parked in LDS across the filter phase (A/B builds: 0)
#endif

// Phase stamps of pass B' (timing-only debug builds, -DFDR_DEBUG_STAMPS; read back by tools/microbench/passbench): the
// shader-clock counter of wave 0 of every workgroup at start / tile landed / forward transform done / filter applied /
// inverse transform done / stores issued / stores retired.  The waits the "landed" and "retired" stamps need are part of
// such a build only.
#ifdef FDR_DEBUG_STAMPS
// (the record itself -- fdr_dbg_stamps, 32 entries per workgroup -- lives in fdr_fft_core.hpp: the core stamps its steps too)
#define FDR_STAMP(i) do { if (threadIdx.x == 0) fdr_dbg_stamps[((blockIdx.x + gridDim.x * blockIdx.y) & 8191) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#define FDR_STAMP_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// every value of the tile is final here and nothing that uses it starts earlier: makes the stamp a real phase boundary
// (VALU work and the waits in front of it would otherwise drift across the stamp's store)
#define FDR_STAMP_PIN(v) do { _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_) _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) asm volatile("" : "+v"(v[b_][s_].x), "+v"(v[b_][s_].y)); } while (0)
extern "C" int fdr_debug_read_stamps(unsigned long long* out, size_t count) {
    if (count > 8192 * 32) count = 8192 * 32;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fdr_dbg_stamps), count * sizeof(unsigned long long));
}
#else
#define FDR_STAMP(i) ((void)0)
#define FDR_STAMP_WAIT_VM() ((void)0)
#define FDR_STAMP_PIN(v) ((void)0)
#endif

template <int LOGM>
struct Panel16Geom {
    static constexpr int T = Steps<LOGM, 4>::T;
    // one panel per workgroup at every size: with short columns (64 or 128 threads per transform) grouping several
    // panels into a 256-thread workgroup was measured slower (2048^2 x 4 images: 45.1 vs 39.7 us) and leaves CUs idle
    static constexpr int G = 1;
    static constexpr int THREADS = T * G;
};

template <int LOGM>
__global__ __launch_bounds__(Panel16Geom<LOGM>::THREADS, 2) void fft_cols_panel_fused16_kernel(
    const PanelBatch pb, const float2* __restrict__ filt, const float2* __restrict__ tw_fwd, const unsigned pstride,
    const int npanels, const int ntiles, const int packed0, const int img_shift) {
    using St = Steps<LOGM, 4>;
    constexpr int G = Panel16Geom<LOGM>::G, T = St::T, M = St::L, V = 16;
    using Core = FftCore<LOGM, 4, 2, typename std::conditional<(LOGM == 12 && !FDR_COLS12_PACKED), PolicyFastScalar, PolicyFast>::type, 4,
                         (St::lr(0) == 1 && T >= 64 && FDR_SWAP0)>;  // see PolicyFastScalar; 8192 points: wave-local first exchange
    __shared__ float2 lds[G * 2 * St::BUF];
    const int g = G == 1 ? 0 : (int)(threadIdx.x >> St::LOGT);
    const int tid = Core::thread_index(threadIdx.x & (T - 1));
    float2* grp_lds = lds + g * 2 * St::BUF;
    // Two mappings of workgroups to (tile, image), both free of integer divisions (which would run on the VALU and drag
    // every tile address into VGPRs).  img_shift < 0: grid (ntiles, images).  Otherwise (2, 4 or 8 images, tiles a multiple
    // of 8): a flat grid in which the workgroups that share a tile -- and so its slice of the filter W -- are neighbours
    // on the SAME XCD (workgroup b lands on XCD b % 8), so W crosses the fabric once per tile, not once per image.
    int img, tl;
    if (img_shift < 0) { img = blockIdx.y; tl = blockIdx.x; }
    else {
        const int b = blockIdx.x, j = b >> 3;
        img = j & ((1 << img_shift) - 1);
        tl = ((j >> img_shift) << 3) | (b & 7);
    }
    const bool active = tl * G + g < npanels;
    const size_t tbase = (size_t)(tl * G) * pstride;
    float2* __restrict__ data = pick_image(pb.data, img) + tbase;
    const float2* __restrict__ tfilt = filt + tbase;
    const unsigned loff = (active ? (unsigned)g : 0u) * pstride + (unsigned)tid * 4u;

    typename Core::Bases bases;
    Core::init_bases(bases, tw_fwd, tid);
    // 8192 points: the tile fills 128 of the 256 registers a lane has and the filter phase needs the rest, so the hoisted
    // twiddle bases (one float2 per radix-16 step) did not survive it -- hipcc spilled them in the forward transform and
    // reloaded them from scratch in the inverse (3 x 8 bytes per lane: 28 bytes of scratch, ~50 MB of HBM traffic per
    // two-image launch, and three exposed memory round trips).  They are parked in the 12 KB of LDS the exchange buffers
    // leave instead and picked up again for the inverse: an LDS read where a scratch load was.
    constexpr bool kParkBases = (LOGM == 13) && FDR_PARK_BASES;
    __shared__ float2 parked[kParkBases ? (St::S - 1) * T : 1];
    if constexpr (kParkBases) {
#pragma unroll
        for (int j = 1; j < St::S; ++j) parked[(j - 1) * T + tid] = bases.b[j][0];  // (the logical index: one slot per thread)
    }

    // (Delaying the workgroup that landed in the odd wave slots by half a load phase, so that the two workgroups of
    // a CU alternate between memory and LDS phases, was measured: no gain up to 5 us of delay, slower beyond.  Round 3, the
    // same across CUs: every other TILE's first-round workgroup started 7 .. 40 us late -- 4096^2 35.1 / 34.0 / 34.4 / 36.5 us,
    // 8192^2 192.7 / 186.3 / 193.9 / 201.5 us per image at 0 / 7 / 14 / 20 us: the CUs are not in lockstep to begin with.)
    // (Round 3, built, verified against the tests, measured and not kept -- the code is in the history, commit "Experiments on
    // pass B'", numbers in DESIGN.md section 5:
    // (1) a PERSISTENT form, one 256-thread workgroup per CU at one wave per SIMD with the whole filter tile and then the next
    //     tile prefetched into 128 AGPRs by inline-asm `global_load_dwordx4 a[..]` and hand-placed vmcnt waits -- no spills,
    //     nothing waited for, but 41.3 against 32.7 us per 4096^2 image: alone on its SIMD a wave exposes every LDS round trip and
    //     barrier of the two transforms, 20 us per tile against the 16.5 us two co-resident workgroups take per tile between them;
    // (2) column-pipelined transforms (a transform's LDS round trip behind the next transform's butterflies): +13 % VALU
    //     instructions, 33.9 against 32.7 us;
    // (3) the first filter piece requested before the forward transform's LAST exchange: the transform's register peak is
    //     there, 19 spilled registers, 34.2 against 32.7 us (8192 points: 176.9 against 182.4 us with 23 spilled registers);
    // (4) the second half of the filter tile by `global_load_lds_dwordx4` into the exchange buffers, idle between the
    //     transforms, so that all of W is in flight at once (one round trip instead of two, two more barriers): 33.4 against
    //     32.7 us, 8192 points 186.6 against 179.2 us.
    // Without its filter (-DFDR_DEBUG_SKIP_W) the pass takes 28.6 us, and `tools/microbench/rmw_bench` moves the pass's
    // traffic alone -- four 64 MiB images in place plus one shared 64 MiB filter -- in 94-106 us, 24-26.5 us per image.)
    float2 v[4][V];
    FDR_STAMP(0);
    tile_load<Core, false>(data, loff, 1u, v);
    FDR_STAMP_WAIT_VM();
    FDR_STAMP_PIN(v);
    FDR_STAMP(1);
    Core::template run<0, false>(v, grp_lds, tw_fwd, bases, tid);
    FDR_STAMP_PIN(v);
    FDR_STAMP(2);

    const bool packed_tile = packed0 && tl == 0;  // uniform per workgroup
    constexpr int SEQ = Core::SLOTS;
    if (packed_tile) {  // column 0 of panel 0 (packed DC + i Nyquist) finished on its own; see the lean kernel
        float2* bufc = grp_lds + (SEQ & 1) * St::BUF;
        float2* bufs = grp_lds + ((SEQ + 1) & 1) * St::BUF;
        __syncthreads();
        FDR_JITTER(4011);
        // (once per image, on one workgroup: kept cheap in REGISTERS, not in time -- the filter values go to LDS two at
        // a time behind compiler barriers and every slot is finished before the next one starts, so this path adds
        // nothing to the pressure of the common one)
#pragma unroll
        for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
            for (int q = 0; q < Core::RHOL; ++q) bufc[Core::out_index(tid, u, q)] = v[0][u * Core::RHOL + q];
#pragma unroll
        for (int s = 0; s < V; s += 2) {
            const int k0 = Core::out_index(tid, s / Core::RHOL, s % Core::RHOL), k1 = Core::out_index(tid, (s + 1) / Core::RHOL, (s + 1) % Core::RHOL);
            const float2 f0 = tfilt[loff - (unsigned)tid * 4u + (unsigned)k0 * 4u], f1 = tfilt[loff - (unsigned)tid * 4u + (unsigned)k1 * 4u];
            bufs[k0] = f0; bufs[k1] = f1;
            asm volatile("" ::: "memory");
        }
        __syncthreads();
        FDR_JITTER(4012);
        if (g == 0) {
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) {
                    const int s = u * Core::RHOL + q;
                    const int k = Core::out_index(tid, u, q);
                    const int km = (M - k) & (M - 1);
                    const float2 c = v[0][s], cm = bufc[km], sm = bufs[km], sl_s = bufs[k];
                    const float2 f0 = make_float2(0.5f * (c.x + cm.x), 0.5f * (c.y - cm.y));
                    const float2 fn = make_float2(0.5f * (c.y + cm.y), 0.5f * (cm.x - c.x));
                    float2 a0, an;
                    if (k == 0 || k == M / 2) { a0 = make_float2(sl_s.x, 0.f); an = make_float2(sl_s.y, 0.f); }
                    else if (k < M / 2) { a0 = sl_s; an = sm; }
                    else { a0 = make_float2(sm.x, -sm.y); an = make_float2(sl_s.x, -sl_s.y); }
                    const float2 z0 = cmul_fma(f0, a0), zn = cmul_fma(fn, an);
                    v[0][s] = make_float2(z0.x - zn.y, z0.y + zn.x);
                    asm volatile("" ::: "memory");
                }
        }
        __syncthreads();  // both buffers were read above
    }
    {
        const bool col0_done = packed_tile && g == 0;
        // W in pieces of PC slots, the next piece requested before the current one is used.  (Measured: requesting the
        // first pieces before the forward transform costs 27 spilled registers and 5 us; the compiler barriers keep
        // hipcc from hoisting all 32 loads to the top.)
        constexpr int PC = FDR_WPIECE;  // slots per piece: 8 VGPRs per slot, two pieces in flight
        auto wload = [&](int h, float2 (&w)[PC][4]) {
#pragma unroll
            for (int i = 0; i < PC; ++i) {
                const int s = PC * h + i, u = s / Core::RHOL, q = s % Core::RHOL;
                const unsigned uoff = (unsigned)(((q << Core::LOGOUT) + u * Core::T) * 4);
#ifdef FDR_DEBUG_SKIP_W  // timing-only builds: pass B' without its filter traffic
                (void)uoff;
                w[i][0] = w[i][1] = w[i][2] = w[i][3] = make_float2(1.0f, __uint_as_float(loff) * 0.f);
#else
                const gchar* ub = uniform_gptr(tfilt + uoff);
                FDR_GLOAD32(ub, loff * 8u, w[i][0], w[i][1], w[i][2], w[i][3]);
#endif
            }
        };
        auto wmul = [&](int h, const float2 (&w)[PC][4]) {
#pragma unroll
            for (int i = 0; i < PC; ++i) {
                const int s = PC * h + i;
                v[0][s] = cmul_fma(v[0][s], col0_done ? make_float2(1.f, 0.f) : w[i][0]);
                v[1][s] = cmul_fma(v[1][s], w[i][1]);
                v[2][s] = cmul_fma(v[2][s], w[i][2]);
                v[3][s] = cmul_fma(v[3][s], w[i][3]);
            }
        };
        float2 wa[PC][4], wb[PC][4];
        wload(0, wa);
#pragma unroll
        for (int h = 0; h < V / PC; h += 2) {
            asm volatile("" ::: "memory");
            wload(h + 1, wb);
            wmul(h, wa);
            asm volatile("" ::: "memory");
            if (h + 2 < V / PC) wload(h + 2, wa);
            wmul(h + 1, wb);
        }
    }
    FDR_STAMP_PIN(v);
    FDR_STAMP(3);
    {
        // opaque copy of the thread index: the inverse transform's LDS addresses equal the forward transform's, and as
        // common subexpressions they would stay alive across the filter phase, where register pressure peaks
        int ti = tid;
        asm volatile("" : "+v"(ti));
        Core::permute_out_to_in(v);  // (a renaming of registers when the first and the last radix differ)
        if constexpr (kParkBases) {
            typename Core::Bases inv_bases;
#pragma unroll
            for (int j = 1; j < St::S; ++j) inv_bases.b[j][0] = parked[(j - 1) * T + ti];
            Core::template run<SEQ, true>(v, grp_lds, tw_fwd, inv_bases, ti);
        } else {
            Core::template run<SEQ, true>(v, grp_lds, tw_fwd, bases, ti);
        }
    }
    FDR_STAMP_PIN(v);
    FDR_STAMP(4);
    if (active) tile_store<Core>(data, loff, v);
    FDR_STAMP(5);
    FDR_STAMP_WAIT_VM();
    FDR_STAMP(6);
}

// ---------------------------------------------------------------------------------------------
// Pass B' for ONE small image (the single-image call of BASELINE config 2: M <= 2048).  The tile kernels above give a
// 4-column tile to one thread group (64 threads at 1024 points): with a single image in flight that is 128 one-wave
// workgroups, each running eight 1024-point transforms back to back on one SIMD -- 10 us of dependent VALU and memory
// latency on a chip that is 95 % idle (measured 14.2 us event-timed at 1024^2, 9.8 with the transforms compiled out).
// Here the four columns of a panel go to four thread groups of one workgroup (B = 1 transform per group, same FftCore
// step plan and policy as the tile kernel of that length, so the bits are the same), the filter is requested together with
// the spectrum (16 or 8 values per lane leave the registers for it), and nothing is staged.  The thread groups are
// INTERLEAVED over the lanes -- column = lane & 3, logical thread = lane >> 2 -- so that for every (u, q) slot the 64 lanes
// of a wave touch 16 rows x 4 columns = 512 contiguous bytes: dense 8-byte-per-lane accesses (with column = lane / T each
// wave read 8 of every 32 bytes, which held the 2048-row case at the tile kernel's time).  The groups' exchange buffers are
// 16 dwords apart modulo the 64 banks, so the four 8-lane runs of a half wave fall on disjoint banks in the contiguous
// phases of an exchange.
// ---------------------------------------------------------------------------------------------
template <int LOGM>
struct PanelSplitGeom {
    static constexpr int LOGV = LOGM >= 10 ? 4 : 3;  // as the tile kernels: radix-16 steps from 1024 points on, radix-8 below
    using St = Steps<LOGM, LOGV>;
    static constexpr int T = St::T;
    static constexpr int THREADS = 4 * T;
};

// (Round 3: the same kernel as the BATCHED pass B' at 4096 points -- 1024 threads per tile, one exchange buffer per column
// (NBUF = 1, 148 KB: one workgroup per CU), W requested up front, grid over (tile, image) as the tile kernel -- measured 43.3
// against 34.6 us per image: with one tile in flight per CU nothing overlaps its memory phases.  The tile kernel stays.)
template <int LOGM, int NBUF = 2>
__global__ __launch_bounds__(PanelSplitGeom<LOGM>::THREADS) void fft_cols_panel_split_kernel(
    const PanelBatch pb, const float2* __restrict__ filt, const float2* __restrict__ tw_fwd, const unsigned pstride, const int packed0,
    const int img_shift) {
    using Geo = PanelSplitGeom<LOGM>;
    using St = typename Geo::St;
    constexpr int M = St::L, V = St::V;
    using Core = FftCore<LOGM, 1, NBUF, PolicyFast, Geo::LOGV, false>;
    constexpr int GRP = NBUF * St::BUF + ((24 - (NBUF * St::BUF) % 32) & 31);  // float2 elements per group, = 24 (mod 32): 48 dwords (mod 64)
    static_assert(GRP % 32 == 24, "group stride");
    __shared__ float2 lds[4 * GRP];
    const int c = (int)(threadIdx.x & 3);   // column of the panel
    const int tid = (int)(threadIdx.x >> 2);  // logical thread of that column's transform
    float2* grp_lds = lds + c * GRP;
    int img = 0, tl = (int)blockIdx.x;
    if (img_shift >= 0) {  // flat grid: the workgroups that share a tile (and its slice of W) are neighbours on one XCD
        const int b = (int)blockIdx.x, jj = b >> 3;
        img = jj & ((1 << img_shift) - 1);
        tl = ((jj >> img_shift) << 3) | (b & 7);
    } else {
        img = (int)blockIdx.y;
    }
    // element (m, c) of the panel lies at panel[4 m + c]: a wave-uniform base per (u, q) slot (SGPRs) plus ONE 32-bit lane
    // offset for every access of the kernel -- no 64-bit per-lane address lives in VGPRs (see uniform_gptr)
    float2* __restrict__ panel = pick_image(pb.data, img) + (size_t)tl * pstride;
    const float2* __restrict__ wpanel = filt + (size_t)tl * pstride;
    const float2* __restrict__ wcol = wpanel + c;
    const unsigned lane_off = (unsigned)threadIdx.x * 8u;  // (4 tid + c) float2 elements

    typename Core::Bases bases;
    Core::init_bases(bases, tw_fwd, tid);

    float2 v[1][V], w[V];
#pragma unroll
    for (int u = 0; u < Core::NU0; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHO0; ++q) {
            const gchar* ub = uniform_gptr(panel + (unsigned)(((q << Core::LOGR0) + u * Core::T) * 4));
            v[0][u * Core::RHO0 + q] = *reinterpret_cast<const float2*>((const char*)ub + lane_off);
        }
#pragma unroll
    for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
        for (int q = 0; q < Core::RHOL; ++q) {
            const gchar* ub = uniform_gptr(wpanel + (unsigned)(((q << Core::LOGOUT) + u * Core::T) * 4));
            w[u * Core::RHOL + q] = *reinterpret_cast<const float2*>((const char*)ub + lane_off);
        }

    Core::template run<0, false>(v, grp_lds, tw_fwd, bases, tid);

    constexpr int SEQ = Core::SLOTS;
    const bool packed_tile = packed0 && tl == 0;  // uniform per workgroup
    if (packed_tile) {  // column 0 of panel 0 carries DC + i Nyquist (see packed_column_filter): finished by its own thread group
        // (two buffers: the buffer of slot SEQ is free -- its last readers passed the barrier of the exchange after it;
        //  one buffer: it was read by the last exchange, hence the barrier)
        float2* bufc = grp_lds + (SEQ % NBUF) * St::BUF;
        if constexpr (NBUF == 1) __syncthreads();
        FDR_JITTER(4031);
        if (c == 0) {
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) bufc[Core::out_index(tid, u, q)] = v[0][u * Core::RHOL + q];
        }
        __syncthreads();
        FDR_JITTER(4032);
        if (c == 0) {
#pragma unroll
            for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
                for (int q = 0; q < Core::RHOL; ++q) {
                    const int s = u * Core::RHOL + q;
                    const int k = Core::out_index(tid, u, q);
                    const int km = (M - k) & (M - 1);
                    const float2 cc = v[0][s], cm = bufc[km], sl = w[s], sm = wcol[(size_t)km * 4];
                    const float2 f0 = make_float2(0.5f * (cc.x + cm.x), 0.5f * (cc.y - cm.y));
                    const float2 fn = make_float2(0.5f * (cc.y + cm.y), 0.5f * (cm.x - cc.x));
                    float2 a0, an;
                    if (k == 0 || k == M / 2) { a0 = make_float2(sl.x, 0.f); an = make_float2(sl.y, 0.f); }
                    else if (k < M / 2) { a0 = sl; an = sm; }
                    else { a0 = make_float2(sm.x, -sm.y); an = make_float2(sl.x, -sl.y); }
                    const float2 z0 = cmul_fma(f0, a0), zn = cmul_fma(fn, an);
                    v[0][s] = make_float2(z0.x - zn.y, z0.y + zn.x);
                    w[s] = make_float2(1.f, 0.f);
                }
        }
        __syncthreads();  // bufc was read: the inverse transform's first exchange writes it
    }
#pragma unroll
    for (int s = 0; s < V; ++s) v[0][s] = cmul_fma(v[0][s], w[s]);

    Core::permute_out_to_in(v);
    Core::template run<SEQ, true>(v, grp_lds, tw_fwd, bases, tid);

    {
        unsigned lo = (unsigned)threadIdx.x * 8u;  // opaque copy: recomputed here instead of living across both transforms
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int u = 0; u < Core::NUL; ++u)
#pragma unroll
            for (int q = 0; q < Core::RHOL; ++q) {
                gchar* ub = uniform_gptr(panel + (unsigned)(((q << Core::LOGOUT) + u * Core::T) * 4));
                *reinterpret_cast<float2*>((char*)ub + lo) = v[0][u * Core::RHOL + q];
            }
    }
}

#ifndef FDR_COLS_SPLIT
#define FDR_COLS_SPLIT 1  // single images with 256 .. 2048 rows: one thread group per COLUMN (A/B builds: 0)
#endif

template <int LOGM>
static hipError_t launch_cols_panel_t(ColKind kind, const ColArgs& a, const float2* tw, hipStream_t s) {
    using Geo = PanelGeom<LOGM>;
    const size_t ps = a.pstride;
    const int npanels = a.npanels > 0 ? a.npanels : a.N / 4;  // half spectrum: N/8
    const int ntiles = (npanels + Geo::G - 1) / Geo::G;
    if (kind == COL_FWD_FILTER) {
        hipLaunchKernelGGL((fft_cols_panel_fwd_filter_kernel<LOGM>), dim3(ntiles), dim3(Geo::THREADS), 0, s, a.data, tw, ps, npanels, a.nvalid, a.K,
                           a.packed0);
    } else if (kind == COL_FUSED) {
        PanelBatch pb = a.batch;
        if (pb.nimg <= 0) { pb.nimg = 1; pb.data[0] = a.data; }
        for (int k = pb.nimg; k < kMaxGroup; ++k) pb.data[k] = pb.data[0];
        if constexpr (FDR_COLS_SPLIT && LOGM >= 8 && LOGM <= 11) {
            if (pb.nimg == 1) {  // a single small image: latency, not bandwidth (see fft_cols_panel_split_kernel)
                hipLaunchKernelGGL((fft_cols_panel_split_kernel<LOGM, 2>), dim3(npanels), dim3(PanelSplitGeom<LOGM>::THREADS), 0, s, pb, a.filt, tw,
                                   (unsigned)ps, a.packed0, -1);
                return hipGetLastError();
            }
        }
        if constexpr (LOGM >= 10) {  // 16 values per thread: one workgroup per tile, grid (tiles, images)
            using G16 = Panel16Geom<LOGM>;
            const int nt16 = (npanels + G16::G - 1) / G16::G;
            const int ishift = FDR_SHARE_W && (nt16 % 8 == 0) ? (pb.nimg == 2 ? 1 : pb.nimg == 4 ? 2 : pb.nimg == 8 ? 3 : -1) : -1;
            const dim3 grid16 = ishift < 0 ? dim3(nt16, pb.nimg) : dim3(nt16 * pb.nimg);
            hipLaunchKernelGGL((fft_cols_panel_fused16_kernel<LOGM>), grid16, dim3(G16::THREADS), 0, s, pb, a.filt, tw,
                               (unsigned)ps, npanels, nt16, a.packed0, ishift);
        } else {                     // short columns: persistent radix-8 kernel, register double-buffered
            const int total = ntiles * pb.nimg;
            int grid = (a.num_cu > 0 ? a.num_cu : 256) * Geo::PIPE_WG_PER_CU;
            if (grid > total) grid = total;
            hipLaunchKernelGGL((fft_cols_panel_fused_kernel<LOGM>), dim3(grid), dim3(Geo::THREADS), 0, s, pb, a.filt, tw, (unsigned)ps,
                               npanels, ntiles, a.packed0);
        }
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_cols_panel(int logm, ColKind kind, const ColArgs& a, const float2* tw_fwd, hipStream_t s) {
    FDR_DISPATCH_LOG(logm, launch_cols_panel_t<LG>(kind, a, tw_fwd, s));
    return hipErrorInvalidValue;
}

}  // namespace fdr
