// fdr_fft_core.hpp -- register/LDS radix-2^k DIT core shared by the row and column kernels.
//
// Arithmetic contract (parity mode): every output equals, bit for bit, what
// fft_serial::fft_radix2_inplace (reference fft/fft_serial.cpp:40-68) computes: the same
// butterfly DAG  a' = u + v*w, b' = u - v*w  with v*w = (vr*wr - vi*wi, vr*wi + vi*wr), the same
// per-stage twiddle values (table replayed from the float recurrence `w *= wlen`), no FMA.
// Only the *storage* differs: instead of an in-place bit-reversed array we track
//     Y_s[r][k] = DFT_{2^s} of the decimated subsequence x[r + (L/2^s) j],  r < L/2^s, k < 2^s
// (block i of the reference's stage-s array is r = bitrev(i)), so the reference's recurrence is
//     Y_s[r][k]         = Y_{s-1}[r][k] + T_{2^s}[k] * Y_{s-1}[r + L/2^s][k]
//     Y_s[r][k + 2^s/2] = Y_{s-1}[r][k] - T_{2^s}[k] * Y_{s-1}[r + L/2^s][k]
// and no bit-reversal pass is needed: natural-order loads feed step 0, natural-order stores
// leave the last step (bit-exact bit-reversal *indexing* is implied by the r <-> i bijection).
//
// A thread holds 8 complex values per transform and runs 1..3 stages ("radix 2/4/8 step", each a
// literal composition of radix-2 butterflies) in registers; between steps the L values go
// through LDS in an "r fastest" layout, padded so that both the contiguous writes and the
// strided reads are bank-conflict free for ds_write_b64 / ds_read_b64 on gfx950.
#pragma once
#include <hip/hip_runtime.h>

namespace fdr {

// ---------------------------------------------------------------------------------------------
// compile-time step plan for L = 2^LOGL: small radix first (so only radix-8 steps ever read a
// padded layout), T = L/8 threads per transform.
// ---------------------------------------------------------------------------------------------
template <int LOGL>
struct Steps {
    static_assert(LOGL >= 1 && LOGL <= 13, "transform length 2..8192");
    static constexpr int L = 1 << LOGL;
    static constexpr int REM = LOGL % 3;
    static constexpr int S = LOGL / 3 + (REM ? 1 : 0);
    static constexpr int LOGT = LOGL >= 3 ? LOGL - 3 : 0;
    static constexpr int T = 1 << LOGT;  // threads per transform
    __host__ __device__ static constexpr int lr(int j) { return (REM != 0 && j == 0) ? REM : 3; }
    __host__ __device__ static constexpr int lprev(int j) {
        int s = 0;
        for (int i = 0; i < j; ++i) s += lr(i);
        return s;
    }
    __host__ __device__ static constexpr int logR(int j) { return LOGL - lprev(j) - lr(j); }
    // butterflies per thread at step j (L < 8: a single partial butterfly on one thread)
    __host__ __device__ static constexpr int nu(int j) { return LOGL >= 3 ? (8 >> lr(j)) : 1; }
    // LDS elements (float2) of one exchange buffer, including read-side padding
    static constexpr int BUF = L + (L >> 3) + 8;
};

// ---------------------------------------------------------------------------------------------
// arithmetic policies.  This translation unit is compiled with -ffp-contract=off, so a*b+c below
// is two roundings unless __builtin_fmaf is written explicitly.
// ---------------------------------------------------------------------------------------------
struct PolicyParity {
    static constexpr bool kFma = false;
    static __device__ __forceinline__ void bfly(float2& u, float2& v, const float2 w) {
        const float tr = v.x * w.x - v.y * w.y;
        const float ti = v.x * w.y + v.y * w.x;
        const float ur = u.x, ui = u.y;
        u.x = ur + tr; u.y = ui + ti;
        v.x = ur - tr; v.y = ui - ti;
    }
};
struct PolicyFast {
    static constexpr bool kFma = true;
    static __device__ __forceinline__ void bfly(float2& u, float2& v, const float2 w) {
        // u' = u + v*w in 4 FMAs, v' = 2u - u' in 2 FMAs
        const float ar = __builtin_fmaf(v.x, w.x, __builtin_fmaf(-v.y, w.y, u.x));
        const float ai = __builtin_fmaf(v.x, w.y, __builtin_fmaf(v.y, w.x, u.y));
        v.x = __builtin_fmaf(2.0f, u.x, -ar);
        v.y = __builtin_fmaf(2.0f, u.y, -ai);
        u.x = ar; u.y = ai;
    }
};

__device__ __forceinline__ void swap2(float2& a, float2& b) { const float2 t = a; a = b; b = t; }

// One radix-2^LR step on RHO = 2^LR values x[0..RHO) holding Y_s[r + R q][k], q = 0..RHO-1, leaving
// Y_{s+LR}[r][k + LP q'] in x[q'].  tw = per-stage table (stage len at offset len/2 - 1), LP = 2^s.
template <int LR, int LP, class Pol>
__device__ __forceinline__ void radix_step(float2* x, const float2* __restrict__ tw, const int k) {
    if constexpr (LR == 1) {
        Pol::bfly(x[0], x[1], tw[(LP - 1) + k]);
    } else if constexpr (LR == 2) {
        const float2 w1 = tw[(LP - 1) + k];
        const float2 w2a = tw[(2 * LP - 1) + k], w2b = tw[(2 * LP - 1) + k + LP];
        Pol::bfly(x[0], x[2], w1);
        Pol::bfly(x[1], x[3], w1);
        Pol::bfly(x[0], x[1], w2a);
        Pol::bfly(x[2], x[3], w2b);
        swap2(x[1], x[2]);
    } else {
        const float2 w1 = tw[(LP - 1) + k];
        const float2 w2a = tw[(2 * LP - 1) + k], w2b = tw[(2 * LP - 1) + k + LP];
        const float2 w3a = tw[(4 * LP - 1) + k], w3b = tw[(4 * LP - 1) + k + LP];
        const float2 w3c = tw[(4 * LP - 1) + k + 2 * LP], w3d = tw[(4 * LP - 1) + k + 3 * LP];
        Pol::bfly(x[0], x[4], w1);
        Pol::bfly(x[1], x[5], w1);
        Pol::bfly(x[2], x[6], w1);
        Pol::bfly(x[3], x[7], w1);
        Pol::bfly(x[0], x[2], w2a);
        Pol::bfly(x[1], x[3], w2a);
        Pol::bfly(x[4], x[6], w2b);
        Pol::bfly(x[5], x[7], w2b);
        Pol::bfly(x[0], x[1], w3a);
        Pol::bfly(x[4], x[5], w3b);
        Pol::bfly(x[2], x[3], w3c);
        Pol::bfly(x[6], x[7], w3d);
        swap2(x[1], x[4]);
        swap2(x[3], x[6]);
    }
}

// ---------------------------------------------------------------------------------------------
// FftCore<LOGL, B, NBUF, Pol>: B independent transforms per thread group of T threads.
//   v[b][u*RHO + q]  on entry : x_b[(tid + T u) + R_0 q]          (first-step operand order)
//   v[b][u*RHO + q'] on exit  : X_b[(tid + T u) + (L / RHO_last) q'] (last-step result order)
// lds: this group's NBUF * Steps::BUF float2 region.  SEQ0: how many exchange slots were already
// consumed on this region (keeps the double-buffer parity hazard-free across chained calls).
// ---------------------------------------------------------------------------------------------
template <int LOGL, int B, int NBUF, class Pol>
struct FftCore {
    using St = Steps<LOGL>;
    static constexpr int S = St::S;
    static constexpr int T = St::T;
    static constexpr int SLOTS = (S - 1) * B;  // exchange slots one run() consumes

    // index helpers for the first-step loads and last-step stores
    static constexpr int NU0 = St::nu(0), RHO0 = 1 << St::lr(0), LOGR0 = St::logR(0);
    static constexpr int NUL = St::nu(S - 1), RHOL = 1 << St::lr(S - 1), LOGOUT = LOGL - St::lr(S - 1);
    // element index of input slot (u, q) / output slot (u, q') for thread tid
    static __device__ __forceinline__ int in_index(int tid, int u, int q) { return (tid + u * T) + (q << LOGR0); }
    static __device__ __forceinline__ int out_index(int tid, int u, int q) { return (tid + u * T) + (q << LOGOUT); }

    template <int J>
    static __device__ __forceinline__ int pad(int a) {
        // layout read by step J+1: runs of R' elements every RHO'*R'; skew each run by R'
        constexpr int LRn = St::lr(J + 1), LOGRn = St::logR(J + 1);
        if constexpr (LOGRn < 5) return a + ((a >> (LRn + LOGRn)) << LOGRn);
        else return a;
    }

    template <int J>
    static __device__ __forceinline__ void butterflies(float2 (&v)[B][8], const float2* __restrict__ tw, int tid) {
        constexpr int LR = St::lr(J), NU = St::nu(J), RHO = 1 << LR, LOGR = St::logR(J), LP = 1 << St::lprev(J);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int k = (tid + u * T) >> LOGR;
#pragma unroll
            for (int b = 0; b < B; ++b) radix_step<LR, LP, Pol>(&v[b][u * RHO], tw, k);
        }
    }

    template <int J, int SEQ0>
    static __device__ __forceinline__ void exchange(float2 (&v)[B][8], float2* lds, int tid) {
        constexpr int LR = St::lr(J), NU = St::nu(J), RHO = 1 << LR;
        constexpr int LRn = St::lr(J + 1), NUn = St::nu(J + 1), RHOn = 1 << LRn, LOGRn = St::logR(J + 1);
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int slot = SEQ0 + J * B + b;
            float2* buf = lds + (slot % NBUF) * St::BUF;
            if (NBUF == 1 && slot != 0) __syncthreads();  // everyone finished reading the previous slot
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int q = 0; q < RHO; ++q) buf[pad<J>((tid + u * T) + (q << (LOGL - LR)))] = v[b][u * RHO + q];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NUn; ++u) {
                const int bid = tid + u * T;
                const int base = ((bid >> LOGRn) << (LRn + LOGRn)) + (bid & ((1 << LOGRn) - 1));
#pragma unroll
                for (int q = 0; q < RHOn; ++q) v[b][u * RHOn + q] = buf[pad<J>(base + (q << LOGRn))];
            }
        }
    }

    template <int J, int SEQ0>
    static __device__ __forceinline__ void steps_from(float2 (&v)[B][8], float2* lds, const float2* __restrict__ tw, int tid) {
        butterflies<J>(v, tw, tid);
        if constexpr (J + 1 < S) {
            exchange<J, SEQ0>(v, lds, tid);
            steps_from<J + 1, SEQ0>(v, lds, tw, tid);
        }
    }

    template <int SEQ0 = 0>
    static __device__ __forceinline__ void run(float2 (&v)[B][8], float2* lds, const float2* __restrict__ tw, int tid) {
        steps_from<0, SEQ0>(v, lds, tw, tid);
    }
};

// ---------------------------------------------------------------------------------------------
// min/max of the real plane: every producing workgroup writes ONE (min, max) partial; a single
// small kernel reduces the partials (deterministic, and no same-address atomics: 65k atomics on
// two words cost ~0.7 ms on MI355X, 15x the kernel that issued them).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_minmax_store(float mn, float mx, float2* __restrict__ part) {
    __shared__ float2 red[16];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = make_float2(mn, mx);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) {
            mn = fminf(mn, red[w].x);
            mx = fmaxf(mx, red[w].y);
        }
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        part[b] = make_float2(mn, mx);  // (+inf, -inf) when the block saw no counted element
    }
}

}  // namespace fdr
