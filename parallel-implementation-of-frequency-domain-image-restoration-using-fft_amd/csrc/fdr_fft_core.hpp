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

// Race fuzzer (timing-only debug builds, -DFDR_DEBUG_JITTER): every wave sleeps a pseudo-random 0..7 microseconds at the
// points where it is about to write or read shared LDS state, so that the waves of a workgroup drift apart by more than
// any phase lasts.  A missing barrier then corrupts data in (nearly) every workgroup instead of once in a thousand runs
// under a second stream's load (round 2: pass A's separation buffer).  The GPU test suite is run against such a build
// with tools/gpu_jitter.sh; product builds compile this to nothing.
#ifdef FDR_DEBUG_JITTER
__device__ __forceinline__ void fdr_jitter(unsigned salt) {
    unsigned h = (threadIdx.x >> 6) * 0x9E3779B1u + salt * 0x85EBCA77u + (blockIdx.x + 131u * blockIdx.y) * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    const unsigned n = __builtin_amdgcn_readfirstlane(h & 7u);
    for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);  // 32 x 64 clocks, about 1 us
}
#define FDR_JITTER(salt) fdr_jitter((unsigned)(salt))
#else
#define FDR_JITTER(salt) ((void)0)
#endif

// Stamps inside a transform (timing-only debug builds, -DFDR_DEBUG_STAMPS): shader-clock counter of thread 0 after every
// butterfly step and every exchange of the column pass's core (16 values x 4 columns per thread), slots 8.. (forward) and
// 16.. (inverse) of the workgroup's 32-entry record; see fdr_panel.hip.  The values are pinned at each stamp.
#ifdef FDR_DEBUG_STAMPS
static __device__ unsigned long long fdr_dbg_stamps[8192 * 32];
#define FDR_CORE_STAMP(cond, slot, v)                                                                                    \
    do {                                                                                                                 \
        if constexpr (cond) {                                                                                            \
            _Pragma("unroll") for (int b_ = 0; b_ < B; ++b_) _Pragma("unroll") for (int s_ = 0; s_ < V; ++s_)          \
                asm volatile("" : "+v"(v[b_][s_].x), "+v"(v[b_][s_].y));                                              \
            if (threadIdx.x == 0) fdr_dbg_stamps[((blockIdx.x + gridDim.x * blockIdx.y) & 8191) * 32 + (slot)] = __builtin_readcyclecounter(); \
        }                                                                                                                \
    } while (0)
#else
#define FDR_CORE_STAMP(cond, slot, v) ((void)0)
#endif

// ---------------------------------------------------------------------------------------------
// compile-time step plan for L = 2^LOGL: small radix first (so only radix-8 steps ever read a
// padded layout), T = L/8 threads per transform.
// ---------------------------------------------------------------------------------------------
// LOGV = log2 of the values a thread holds per transform: 3 (8 values, radix-8 steps; every kernel but one) or 4
// (16 values, radix-16 steps: half the threads and two LDS exchanges instead of three for 4096 points -- the
// column pass, where 4 columns x 16 values = 128 registers per lane buy two 256-thread workgroups per CU).
template <int LOGL, int LOGV = 3>
struct Steps {
    static_assert(LOGL >= 1 && LOGL <= 13, "transform length 2..8192");
    static_assert(LOGV == 3 || LOGV == 4, "8 or 16 values per thread");
    static constexpr int L = 1 << LOGL;
    static constexpr int V = 1 << LOGV;
    static constexpr int REM = LOGL % LOGV;
    static constexpr int S = LOGL / LOGV + (REM ? 1 : 0);
    static constexpr int LOGT = LOGL >= LOGV ? LOGL - LOGV : 0;
    static constexpr int T = 1 << LOGT;  // threads per transform
    __host__ __device__ static constexpr int lr(int j) { return (REM != 0 && j == 0) ? REM : LOGV; }
    __host__ __device__ static constexpr int lprev(int j) {
        int s = 0;
        for (int i = 0; i < j; ++i) s += lr(i);
        return s;
    }
    __host__ __device__ static constexpr int logR(int j) { return LOGL - lprev(j) - lr(j); }
    // butterflies per thread at step j (L < 8: a single partial butterfly on one thread)
    __host__ __device__ static constexpr int nu(int j) { return LOGL >= LOGV ? (V >> lr(j)) : 1; }
    // LDS elements (float2) of one exchange buffer, including read-side padding
    static constexpr int BUF = L + (L >> 3) + 8;  // (LOGV = 4 pads by at most L/16)
};

// ---------------------------------------------------------------------------------------------
// arithmetic policies.  This translation unit is compiled with -ffp-contract=off, so a*b+c below
// is two roundings unless __builtin_fmaf is written explicitly.
//
// A radix-2^LR step needs the twiddles of LR consecutive stages (LP = sub-transform size before
// the step, k = frequency index of the butterfly, k < LP):
//   w1 = T_{2LP}[k];  w2a = T_{4LP}[k], w2b = T_{4LP}[k+LP];  w3a..d = T_{8LP}[k + LP c], c = 0..3
// with T_len stored at table offset len/2 - 1.
// ---------------------------------------------------------------------------------------------
struct TwSet {
    float2 w1, w2a, w2b, w3a, w3b, w3c, w3d;
};

// Parity: every twiddle comes from the table (values replayed from the serial recurrence, so not
// even T_4[1] = (-4.37e-8, -1) is replaced by -i); every product and sum is rounded separately.
// The table passed in is already direction specific, INV is ignored.
struct PolicyParity {
    static constexpr bool kHoist = false;
    static __device__ __forceinline__ void bfly(float2& u, float2& v, const float2 w) {
        const float tr = v.x * w.x - v.y * w.y;
        const float ti = v.x * w.y + v.y * w.x;
        const float ur = u.x, ui = u.y;
        u.x = ur + tr; u.y = ui + ti;
        v.x = ur - tr; v.y = ui - ti;
    }
    template <int LR, int LP>
    static __device__ __forceinline__ float2 base(const float2* __restrict__, int) { return make_float2(1.f, 0.f); }
    template <int LR, int LP, bool INV>
    static __device__ __forceinline__ TwSet twiddles(const float2* __restrict__ tw, int k, float2) {
        TwSet t;
        t.w1 = tw[(LP - 1) + k];
        if constexpr (LR >= 2) { t.w2a = tw[(2 * LP - 1) + k]; t.w2b = tw[(2 * LP - 1) + k + LP]; }
        if constexpr (LR >= 3) {
            t.w3a = tw[(4 * LP - 1) + k]; t.w3b = tw[(4 * LP - 1) + k + LP];
            t.w3c = tw[(4 * LP - 1) + k + 2 * LP]; t.w3d = tw[(4 * LP - 1) + k + 3 * LP];
        }
        return t;
    }
};

// Fast: FMA butterflies (6 FMAs), and ONE table value per step -- the twiddle of the step's last
// stage, w = exp(-+2 pi i k / (2^LR LP)) from the double-generated forward table -- from which the
// others follow: w^2, w^4 and rotations by multiples of pi/4.  A thread's k never changes, so the
// base values are fetched once per kernel (kHoist) and no vector-memory load sits inside the
// transform: prefetches of the next tile stay in flight behind it (vmcnt is an in-order counter).
struct PolicyFast {
    static constexpr bool kHoist = true;
    typedef float v2f __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ void bfly(float2& u, float2& v, const float2 w) {
        // u' = u + v*w in 4 FMAs, v' = 2u - u' in 2 FMAs -- written on 2-vectors so that every butterfly becomes three
        // v_pk_fma_f32 (left to itself hipcc packs only about half of them; the rest are six scalar v_fma each):
        // 30 % fewer VALU instructions per transform, 1.37 -> 1.03 us per 4096-point transform per CU.  The rotated
        // twiddle (-w.y, w.x) is a register pair per twiddle; building it from operand modifiers instead (negated
        // broadcast of v.y) costs a v_mov per butterfly and was slower.  (Round 4: the three instructions written out as
        // inline asm with op_sel / neg_lo on the twiddle doing the rotation -- no second pair, 254 -> 238 registers in pass B' --
        // measured equal within noise, 33.8 / 35.3 vs 34.3 / 34.9 us per 4096^2 image: the passes are not bound by their VALU
        // work; hipcc also pads every inline-asm def-use pair closer than three instructions with an s_nop.  Not kept.)
        const v2f uu = {u.x, u.y}, vx = {v.x, v.x}, vy = {v.y, v.y}, ww = {w.x, w.y}, wr = {-w.y, w.x};
        const v2f t = __builtin_elementwise_fma(vy, wr, uu);   // (u.x - v.y w.y, u.y + v.y w.x)
        const v2f a = __builtin_elementwise_fma(vx, ww, t);    // (.. + v.x w.x, .. + v.x w.y)
        const v2f two = {2.0f, 2.0f};
        const v2f b = __builtin_elementwise_fma(two, uu, -a);
        u.x = a.x; u.y = a.y; v.x = b.x; v.y = b.y;
    }
    static __device__ __forceinline__ float2 csq(float2 a) {
        return make_float2(__builtin_fmaf(a.x, a.x, -(a.y * a.y)), 2.0f * a.x * a.y);
    }
    template <int LR, int LP>
    static __device__ __forceinline__ float2 base(const float2* __restrict__ tw_fwd, int k) {
        return tw_fwd[((LP << (LR - 1)) - 1) + k];  // T_{2^LR LP}[k], forward direction
    }
    template <int LR, int LP, bool INV>
    static __device__ __forceinline__ TwSet twiddles(const float2* __restrict__, int, float2 b) {
        // Opaque to the optimiser on purpose: in the persistent kernels the derived set is loop
        // invariant, and LICM would hoist all S x 2 sets (100+ VGPRs) out of the tile loop and
        // spill them -- scratch reloads are vector-memory loads and would drain the prefetch queue.
        asm volatile("" : "+v"(b.x), "+v"(b.y));
        const float2 w = INV ? make_float2(b.x, -b.y) : b;
        TwSet t;
        // multiplying by -i (forward) / +i (inverse): (x, y) -> (y, -x) / (-y, x)
        auto rot = [](float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); };
        if constexpr (LR == 1) {
            t.w1 = w;
        } else if constexpr (LR == 2) {
            t.w2a = w; t.w2b = rot(w); t.w1 = csq(w);
        } else {
            const float c = 0.70710678118654752440f;
            t.w3a = w; t.w3c = rot(w);
            // w * exp(-+i pi/4) and w * exp(-+3i pi/4)
            const float2 d = INV ? make_float2(c * (w.x - w.y), c * (w.x + w.y)) : make_float2(c * (w.x + w.y), c * (w.y - w.x));
            t.w3b = d; t.w3d = rot(d);
            t.w2a = csq(w); t.w2b = rot(t.w2a);
            t.w1 = csq(t.w2a);
        }
        return t;
    }
};

// The same with scalar FMAs (what hipcc packs by itself): the 4096-point column pass keeps 128 data registers per lane,
// where the extra register pair per rotated twiddle of the packed form spills (14 VGPRs, +15 % HBM traffic from scratch)
// for no gain in time -- that kernel alone uses this variant.
struct PolicyFastScalar : PolicyFast {
    static __device__ __forceinline__ void bfly(float2& u, float2& v, const float2 w) {
        const float ar = __builtin_fmaf(v.x, w.x, __builtin_fmaf(-v.y, w.y, u.x));
        const float ai = __builtin_fmaf(v.x, w.y, __builtin_fmaf(v.y, w.x, u.y));
        v.x = __builtin_fmaf(2.0f, u.x, -ar);
        v.y = __builtin_fmaf(2.0f, u.y, -ai);
        u.x = ar; u.y = ai;
    }
};

// One exchange value out of LDS as ONE ds_read_b64.  Left to itself hipcc pairs the strided reads of an exchange into
// ds_read2_b64, which the LDS serves as two 4 x 16-lane accesses: 8 cycles per wave-instruction for 1 KB, against 2 cycles
// per ds_read_b64 for 512 B (MI355X_MICROARCH.md, LDS table: 128 vs 256 B/clk per CU) -- half the read bandwidth in the
// phase of the transforms that is bound by the LDS.  A volatile 64-bit access is not merged.
#ifndef FDR_LDS_READ_B64
#define FDR_LDS_READ_B64 1
#endif
__device__ __forceinline__ float2 lds_read_b64(const float2* p) {
#if FDR_LDS_READ_B64
    typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64_ptr;  // (explicitly LDS: a volatile generic access would be a flat load)
    const unsigned long long u = *(lds_u64_ptr)(p);
    return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
#else
    return *p;
#endif
}

__device__ __forceinline__ void load4(const float2* p, float2& a, float2& b, float2& c, float2& d) {
    const float4 lo = *reinterpret_cast<const float4*>(p);
    const float4 hi = *reinterpret_cast<const float4*>(p + 2);
    a = make_float2(lo.x, lo.y); b = make_float2(lo.z, lo.w);
    c = make_float2(hi.x, hi.y); d = make_float2(hi.z, hi.w);
}
__device__ __forceinline__ void store4(float2* p, float2 a, float2 b, float2 c, float2 d) {
    *reinterpret_cast<float4*>(p) = make_float4(a.x, a.y, b.x, b.y);
    *reinterpret_cast<float4*>(p + 2) = make_float4(c.x, c.y, d.x, d.y);
}

__device__ __forceinline__ float2 cmul_fma(float2 a, float2 w) {
    return make_float2(__builtin_fmaf(a.x, w.x, -(a.y * w.y)), __builtin_fmaf(a.x, w.y, a.y * w.x));
}

__device__ __forceinline__ void swap2(float2& a, float2& b) { const float2 t = a; a = b; b = t; }

// fast mode: W = conj(H) / (|H|^2 + K), evaluated in double and rounded once (make_filter_fast_kernel and the fused
// PSF column pass share this, so both give the same bits)
// Filter slot S[k] of the packed DC / Nyquist column of the half spectrum (column 0 of panel 0 carries C = H0 + i HN with
// H0 = H[., 0], HN = H[., N/2], both Hermitian along the column): ck = C[k], cmk = C[M - k].
//   S[k] = W0[k] (0 < k < M/2),  S[k] = WN[M-k] (M/2 < k < M),  S[0] = (W0[0], WN[0]),  S[M/2] = (W0[M/2], WN[M/2])
// with W = conj(H) / (|H|^2 + K) in double, rounded once; evaluated at j = min(k, M - k) (pass B' reads it that way).
__device__ __forceinline__ float2 packed_column_filter_slot(float2 ck, float2 cmk, int k, int M, float K) {
    const bool upper = k > M / 2;
    const float2 c = upper ? cmk : ck, cm = upper ? ck : cmk;  // C[j], C[M - j]
    const double h0r = 0.5 * ((double)c.x + cm.x), h0i = 0.5 * ((double)c.y - cm.y);   // H0 = (C + conj Cm)/2
    const double hnr = 0.5 * ((double)c.y + cm.y), hni = 0.5 * ((double)cm.x - c.x);   // HN = (C - conj Cm)/(2i)
    const double d0 = h0r * h0r + h0i * h0i + (double)K, dn = hnr * hnr + hni * hni + (double)K;
    const double w0r = d0 != 0.0 ? h0r / d0 : 0.0, w0i = d0 != 0.0 ? -h0i / d0 : 0.0;
    const double wnr = dn != 0.0 ? hnr / dn : 0.0, wni = dn != 0.0 ? -hni / dn : 0.0;
    if (k == 0 || k == M / 2) return make_float2((float)w0r, (float)wnr);
    if (k < M / 2) return make_float2((float)w0r, (float)w0i);
    return make_float2((float)wnr, (float)wni);
}

__device__ __forceinline__ float2 wiener_filter_fast(float2 h, float K) {
    const double hr = h.x, hi = h.y;
    const double denom = hr * hr + hi * hi + (double)K;
    float2 w = make_float2(0.f, 0.f);
    if (denom != 0.0) { w.x = (float)(hr / denom); w.y = (float)(-hi / denom); }
    return w;
}

// One radix-2^LR step on RHO = 2^LR values x[0..RHO) holding Y_s[r + R q][k], q = 0..RHO-1, leaving
// Y_{s+LR}[r][k + LP q'] in x[q'].
template <int LR, class Pol>
__device__ __forceinline__ void radix_step(float2* x, const TwSet& t) {
    if constexpr (LR == 1) {
        Pol::bfly(x[0], x[1], t.w1);
    } else if constexpr (LR == 2) {
        Pol::bfly(x[0], x[2], t.w1);
        Pol::bfly(x[1], x[3], t.w1);
        Pol::bfly(x[0], x[1], t.w2a);
        Pol::bfly(x[2], x[3], t.w2b);
        swap2(x[1], x[2]);
    } else {
        Pol::bfly(x[0], x[4], t.w1);
        Pol::bfly(x[1], x[5], t.w1);
        Pol::bfly(x[2], x[6], t.w1);
        Pol::bfly(x[3], x[7], t.w1);
        Pol::bfly(x[0], x[2], t.w2a);
        Pol::bfly(x[1], x[3], t.w2a);
        Pol::bfly(x[4], x[6], t.w2b);
        Pol::bfly(x[5], x[7], t.w2b);
        Pol::bfly(x[0], x[1], t.w3a);
        Pol::bfly(x[4], x[5], t.w3b);
        Pol::bfly(x[2], x[3], t.w3c);
        Pol::bfly(x[6], x[7], t.w3d);
        swap2(x[1], x[4]);
        swap2(x[3], x[6]);
    }
}

// ---------------------------------------------------------------------------------------------
// First step of a transform in the fast mode: the sub-transforms have size LP = 1, so the step's base twiddle is
// T[0] = 1 and every derived twiddle is a constant -- 1, -+i, e^(-+i pi/4), e^(-+i pi/8) ...  Multiplications by 1 and
// by -+i are additions with swapped / negated operands; the others use literal constants.  No table value is loaded
// and no register holds a base for this step (a third of the butterflies of a three-step transform).
// (Parity mode never comes here: there every product with the table value is computed, as the reference does.)
// ---------------------------------------------------------------------------------------------
template <bool INV>
struct TrivialTw {
    // u' = u + v, v' = u - v
    static __device__ __forceinline__ void one(float2& u, float2& v) {
        const float2 a = make_float2(u.x + v.x, u.y + v.y), b = make_float2(u.x - v.x, u.y - v.y);
        u = a; v = b;
    }
    // v times -i (forward) / +i (inverse), then the butterfly
    static __device__ __forceinline__ void rot(float2& u, float2& v) {
        float2 a, b;
        if (INV) { a = make_float2(u.x - v.y, u.y + v.x); b = make_float2(u.x + v.y, u.y - v.x); }
        else     { a = make_float2(u.x + v.y, u.y - v.x); b = make_float2(u.x - v.y, u.y + v.x); }
        u = a; v = b;
    }
    static __device__ __forceinline__ float2 cis(float c, float s) { return make_float2(c, INV ? s : -s); }  // e^(-+i phi)
    static __device__ __forceinline__ float2 rotc(float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }
};

template <int LR, class Pol, bool INV>
__device__ __forceinline__ void radix_step_first(float2* x) {
    using Tw = TrivialTw<INV>;
    if constexpr (LR == 1) {
        Tw::one(x[0], x[1]);
    } else if constexpr (LR == 2) {
        Tw::one(x[0], x[2]);
        Tw::one(x[1], x[3]);
        Tw::one(x[0], x[1]);
        Tw::rot(x[2], x[3]);
        swap2(x[1], x[2]);
    } else {
        const float c = 0.70710678118654752440f;
        const float2 d = Tw::cis(c, c);  // e^(-+i pi/4)
        Tw::one(x[0], x[4]);
        Tw::one(x[1], x[5]);
        Tw::one(x[2], x[6]);
        Tw::one(x[3], x[7]);
        Tw::one(x[0], x[2]);
        Tw::one(x[1], x[3]);
        Tw::rot(x[4], x[6]);
        Tw::rot(x[5], x[7]);
        Tw::one(x[0], x[1]);
        Pol::bfly(x[4], x[5], d);
        Tw::rot(x[2], x[3]);
        Pol::bfly(x[6], x[7], Tw::rotc(d));
        swap2(x[1], x[4]);
        swap2(x[3], x[6]);
    }
}

template <class P, int B, int V, bool INV>
__device__ __forceinline__ void radix16_first(float2 (&v)[B][V]) {
    static_assert(V == 16, "16 values per thread");
    using Tw = TrivialTw<INV>;
    const float c4 = 0.70710678118654752440f, c8 = 0.92387953251128675613f, s8 = 0.38268343236508977173f;
    const float2 d = Tw::cis(c4, c4), e1 = Tw::cis(c8, s8), e3 = Tw::cis(s8, c8);
#pragma unroll
    for (int b = 0; b < B; ++b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) Tw::one(v[b][i], v[b][i + 8]);                     // stage 1: w^8 = 1
#pragma unroll
        for (int i = 0; i < 4; ++i) { Tw::one(v[b][i], v[b][i + 4]); Tw::rot(v[b][8 + i], v[b][12 + i]); }  // stage 2: {1, -+i}
        // stage 3: blocks j = 0..3 -> {1, -+i, e^(-+i pi/4), -+i e^(-+i pi/4)}
        Tw::one(v[b][0], v[b][2]); Tw::one(v[b][1], v[b][3]);
        Tw::rot(v[b][4], v[b][6]); Tw::rot(v[b][5], v[b][7]);
        P::bfly(v[b][8], v[b][10], d); P::bfly(v[b][9], v[b][11], d);
        P::bfly(v[b][12], v[b][14], Tw::rotc(d)); P::bfly(v[b][13], v[b][15], Tw::rotc(d));
        // stage 4: blocks j = 0..7 -> omega_16^bitrev3(j) = {0, 4, 2, 6, 1, 5, 3, 7}
        Tw::one(v[b][0], v[b][1]);
        Tw::rot(v[b][2], v[b][3]);
        P::bfly(v[b][4], v[b][5], d);
        P::bfly(v[b][6], v[b][7], Tw::rotc(d));
        P::bfly(v[b][8], v[b][9], e1);
        P::bfly(v[b][10], v[b][11], Tw::rotc(e1));
        P::bfly(v[b][12], v[b][13], e3);
        P::bfly(v[b][14], v[b][15], Tw::rotc(e3));
        swap2(v[b][1], v[b][8]); swap2(v[b][2], v[b][4]); swap2(v[b][3], v[b][12]);
        swap2(v[b][5], v[b][10]); swap2(v[b][7], v[b][14]); swap2(v[b][11], v[b][13]);
    }
}

// Radix-16 step of the fast policy on B transforms at once, STAGE-major (stage t of all B transforms before stage
// t+1), so that only one stage's twiddles are live: 1, 2, 4, then 8 complex values, all derived from the ONE hoisted
// table value w = exp(-+2 pi i k / (16 LP)):  stage 1: w^8;  stage 2: w^4 {1, -+i};  stage 3: w^2 {1, e^(-+i pi/4)} x
// {1, -+i};  stage 4: w {1, e^(-+i pi/8), e^(-+i pi/4), e^(-+3i pi/8)} x {1, -+i}.  Pair (i, i + 16/2^t) of block j
// uses T_{2^t LP}[k + LP bitrev(j)], exactly the radix-8 scheme one level deeper; outputs leave in bit-reversed
// order and are put back by six swaps.
template <class P, int B, int V, bool INV>
__device__ __forceinline__ void radix16_fast(float2 (&v)[B][V], float2 base) {
    static_assert(V == 16, "16 values per thread");
    asm volatile("" : "+v"(base.x), "+v"(base.y));  // as in PolicyFast::twiddles: keep the derived set out of LICM's reach
    const float2 w = INV ? make_float2(base.x, -base.y) : base;
    auto rot = [](float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); };  // times -+i
    const float2 w2 = P::csq(w), w4 = P::csq(w2);
    {   // stage 1
        const float2 w8 = P::csq(w4);
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int i = 0; i < 8; ++i) P::bfly(v[b][i], v[b][i + 8], w8);
    }
    {   // stage 2
        const float2 t0 = w4, t1 = rot(w4);
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                P::bfly(v[b][i], v[b][i + 4], t0);
                P::bfly(v[b][8 + i], v[b][12 + i], t1);
            }
    }
    const float c4 = 0.70710678118654752440f;
    {   // stage 3: blocks j = 0..3 -> w2 * omega_8^bitrev2(j) = {0, 2, 1, 3}
        const float2 d = INV ? make_float2(c4 * (w2.x - w2.y), c4 * (w2.x + w2.y)) : make_float2(c4 * (w2.x + w2.y), c4 * (w2.y - w2.x));
        const float2 t[4] = {w2, rot(w2), d, rot(d)};
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                P::bfly(v[b][4 * j], v[b][4 * j + 2], t[j]);
                P::bfly(v[b][4 * j + 1], v[b][4 * j + 3], t[j]);
            }
    }
    {   // stage 4: blocks j = 0..7 -> w * omega_16^bitrev3(j) = {0, 4, 2, 6, 1, 5, 3, 7}
        const float c8 = 0.92387953251128675613f, s8 = 0.38268343236508977173f;  // cos, sin of pi/8
        auto mulc = [](float2 a, float cr, float ci) {  // a * (cr + i ci)
            return make_float2(__builtin_fmaf(a.x, cr, -(a.y * ci)), __builtin_fmaf(a.x, ci, a.y * cr));
        };
        const float sg = INV ? 1.0f : -1.0f;                      // omega_16 = exp(-+i pi/8)
        const float2 e1 = mulc(w, c8, sg * s8);                   // w omega^1
        const float2 e2 = mulc(w, c4, sg * c4);                   // w omega^2
        const float2 e3 = mulc(w, s8, sg * c8);                   // w omega^3
        const float2 t[8] = {w, rot(w), e2, rot(e2), e1, rot(e1), e3, rot(e3)};
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) P::bfly(v[b][2 * j], v[b][2 * j + 1], t[j]);
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {  // 4-bit reversal
        swap2(v[b][1], v[b][8]); swap2(v[b][2], v[b][4]); swap2(v[b][3], v[b][12]);
        swap2(v[b][5], v[b][10]); swap2(v[b][7], v[b][14]); swap2(v[b][11], v[b][13]);
    }
}

// ---------------------------------------------------------------------------------------------
// FftCore<LOGL, B, NBUF, Pol>: B independent transforms per thread group of T threads.
//   v[b][u*RHO + q]  on entry : x_b[(tid + T u) + R_0 q]          (first-step operand order)
//   v[b][u*RHO + q'] on exit  : X_b[(tid + T u) + (L / RHO_last) q'] (last-step result order)
// lds: this group's NBUF * Steps::BUF float2 region.  SEQ0: how many exchange slots were already
// consumed on this region (keeps the double-buffer parity hazard-free across chained calls).
// ---------------------------------------------------------------------------------------------
// SWAP0: the exchange behind a radix-2 FIRST step (LOGL mod LOGV = 1, e.g. 8192 = 2 x 16 x 16 x 16) without LDS.  After
// that step thread t holds Y[t + T u][k], u < V/2, k < 2, and the next step wants Y[r0 + (T/2) q][k'], q < V, with
// r0 = t mod T/2 and k' = t div T/2: all sources of a thread are the thread itself and its partner t +- T/2.  With the
// logical thread index laid out so that the partners are lanes l and l + 32 of ONE wave (thread_index below), the
// exchange is `v_permlane32_swap_b32` on the register pair (Y[..][0], Y[..][1]) of every u: V/2 x 2 swaps per transform
// instead of a full LDS round trip with two barriers (the LDS store path, ~85 B/clk per CU, is what bounds the
// 8192-point passes: 7 round trips per column pair of transforms before, 4 now together with permute_out_to_in).
// Kernels that instantiate a SWAP0 core must take their thread index from thread_index(threadIdx.x).
template <int LOGL, int B, int NBUF, class Pol, int LOGV = 3, bool SWAP0 = false>
struct FftCore {
    using St = Steps<LOGL, LOGV>;
    static constexpr int V = St::V;
    static constexpr int S = St::S;
    static constexpr int T = St::T;
    static constexpr int SLOTS = (S - 1) * B;  // exchange slots one run() consumes
    static_assert(!SWAP0 || (St::lr(0) == 1 && S >= 2 && T >= 64), "SWAP0: radix-2 first step, at least one wave per transform");

    // logical thread index of physical thread p of a T-thread group (identity unless SWAP0)
    static __device__ __forceinline__ int thread_index(int p) {
        if constexpr (SWAP0) return (((p >> 6) << 5) | (p & 31)) + (T / 2) * ((p >> 5) & 1);
        else return p;
    }

    // index helpers for the first-step loads and last-step stores
    static constexpr int NU0 = St::nu(0), RHO0 = 1 << St::lr(0), LOGR0 = St::logR(0);
    static constexpr int NUL = St::nu(S - 1), RHOL = 1 << St::lr(S - 1), LOGOUT = LOGL - St::lr(S - 1);
    // element index of input slot (u, q) / output slot (u, q') for thread tid
    static __device__ __forceinline__ int in_index(int tid, int u, int q) { return (tid + u * T) + (q << LOGR0); }
    static __device__ __forceinline__ int out_index(int tid, int u, int q) { return (tid + u * T) + (q << LOGOUT); }

    // per-thread hoisted twiddle bases (fast policy): one value per step and butterfly slot
    struct Bases { float2 b[S][V / 2]; };

    template <int J>
    static __device__ __forceinline__ void init_bases_from(Bases& bs, const float2* __restrict__ tw, int tid) {
        constexpr int LR = St::lr(J), NU = St::nu(J), LOGR = St::logR(J), LP = 1 << St::lprev(J);
        if constexpr (LP != 1) {  // (the first step's twiddles are constants)
#pragma unroll
            for (int u = 0; u < NU; ++u) bs.b[J][u] = Pol::template base<LR, LP>(tw, (tid + u * T) >> LOGR);
        }
        if constexpr (J + 1 < S) init_bases_from<J + 1>(bs, tw, tid);
    }
    static __device__ __forceinline__ void init_bases(Bases& bs, const float2* __restrict__ tw, int tid) {
        if constexpr (Pol::kHoist) init_bases_from<0>(bs, tw, tid);
    }

    template <int J>
    static __device__ __forceinline__ int pad(int a) {
        // layout read by step J+1: runs of R' elements every RHO'*R'; skew each run by R'
        constexpr int LRn = St::lr(J + 1), LOGRn = St::logR(J + 1);
        if constexpr (LOGRn < 5) return a + ((a >> (LRn + LOGRn)) << LOGRn);
        else return a;
    }

    template <int J, bool INV>
    static __device__ __forceinline__ void butterflies(float2 (&v)[B][V], const float2* __restrict__ tw, const Bases& bs, int tid) {
        constexpr int LR = St::lr(J), NU = St::nu(J), RHO = 1 << LR, LOGR = St::logR(J), LP = 1 << St::lprev(J);
        if constexpr (Pol::kHoist && LP == 1) {  // first step of the fast mode: constant twiddles (see TrivialTw)
            if constexpr (LR == 4) {
                radix16_first<Pol, B, V, INV>(v);
            } else {
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int b = 0; b < B; ++b) radix_step_first<LR, Pol, INV>(&v[b][u * RHO]);
            }
        } else if constexpr (LR == 4) {
            static_assert(Pol::kHoist, "radix-16 steps exist for the fast policy only");
            radix16_fast<Pol, B, V, INV>(v, bs.b[J][0]);  // NU == 1: the 16 values of a thread are one butterfly
        } else {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int k = (tid + u * T) >> LOGR;
                const TwSet t = Pol::template twiddles<LR, LP, INV>(tw, k, bs.b[J][u]);
#pragma unroll
                for (int b = 0; b < B; ++b) radix_step<LR, Pol>(&v[b][u * RHO], t);
            }
        }
    }

    // Exchange between steps J and J+1 of the B transforms, one after the other.  Both index sets are AFFINE in the slot
    // number q once the thread part is fixed (pad() only looks at bits the thread part does not reach):
    //   write  pad(t + q S)        = pad(t) + q WS,   S = L / RHO,  WS = S + ((S >> K) << LOGR'),  K = LR' + LOGR'
    //   read   pad(base + q R')    = pad(base) + q R'
    // so each thread needs ONE write and ONE read address per u (plus the buffer offset) and every access is
    // base + immediate.  Written this way on purpose: left as pad(t + q S) hipcc materialises one address register per
    // slot, keeps the sets of all steps and columns alive and spills them (64 of them in the 8192-point column pass).
    template <int J, int SEQ0>
    static __device__ __forceinline__ void exchange(float2 (&v)[B][V], float2* lds, int tid) {
        constexpr int LR = St::lr(J), NU = St::nu(J), RHO = 1 << LR;
        constexpr int LRn = St::lr(J + 1), NUn = St::nu(J + 1), RHOn = 1 << LRn, LOGRn = St::logR(J + 1);
        constexpr int SW = 1 << (LOGL - LR), K = LRn + LOGRn;
        constexpr int WS = LOGRn < 5 ? SW + ((SW >> K) << LOGRn) : SW;
        int wbase[NU], rbase[NUn];
#pragma unroll
        for (int u = 0; u < NU; ++u) wbase[u] = pad<J>(tid + u * T);
#pragma unroll
        for (int u = 0; u < NUn; ++u) {
            const int bid = tid + u * T;
            rbase[u] = pad<J>(((bid >> LOGRn) << (LRn + LOGRn)) + (bid & ((1 << LOGRn) - 1)));
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int slot = SEQ0 + J * B + b;
            float2* buf = lds + (slot % NBUF) * St::BUF;
            if (NBUF == 1 && slot != 0) __syncthreads();  // everyone finished reading the previous slot
            FDR_JITTER(2 * slot);
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int q = 0; q < RHO; ++q) buf[wbase[u] + q * WS] = v[b][u * RHO + q];
            __syncthreads();
            FDR_JITTER(2 * slot + 1);
#pragma unroll
            for (int u = 0; u < NUn; ++u)
#pragma unroll
                for (int q = 0; q < RHOn; ++q) v[b][u * RHOn + q] = lds_read_b64(&buf[rbase[u] + (q << LOGRn)]);
        }
    }

    // exchange 0 of a SWAP0 core: lanes l < 32 keep k = 0 and collect the partner's k = 0 values, lanes l >= 32 keep k = 1;
    // v_permlane32_swap_b32 vdst, src swaps vdst's lanes 32..63 with src's lanes 0..31, so with vdst = Y[..][0] and
    // src = Y[..][1] both registers end up holding this lane's own k: vdst the even q (= 2u), src the odd q (= 2u + 1)
    static __device__ __forceinline__ void exchange_swap(float2 (&v)[B][V]) {
#if !defined(__HIP_DEVICE_COMPILE__)
        (void)v;  // host pass of the translation unit: device code only
#elif __has_builtin(__builtin_amdgcn_permlane32_swap)
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int u = 0; u < V / 2; ++u) {
                const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[b][2 * u].x), __float_as_uint(v[b][2 * u + 1].x), false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[b][2 * u].y), __float_as_uint(v[b][2 * u + 1].y), false, false);
                v[b][2 * u] = make_float2(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
                v[b][2 * u + 1] = make_float2(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
            }
#else
        static_assert(!SWAP0, "v_permlane32_swap_b32 (gfx950) is not available to this compiler");
#endif
    }

    template <int J, int SEQ0, bool INV>
    static __device__ __forceinline__ void steps_from(float2 (&v)[B][V], float2* lds, const float2* __restrict__ tw,
                                                      const Bases& bs, int tid) {
        butterflies<J, INV>(v, tw, bs, tid);
        FDR_CORE_STAMP((B == 4 && LOGV == 4), (INV ? 16 : 8) + 2 * J, v);
        if constexpr (J + 1 < S) {
            if constexpr (SWAP0 && J == 0) exchange_swap(v);
            else exchange<J, SEQ0>(v, lds, tid);
            FDR_CORE_STAMP((B == 4 && LOGV == 4), (INV ? 16 : 8) + 2 * J + 1, v);
            steps_from<J + 1, SEQ0, INV>(v, lds, tw, bs, tid);
        }
    }

    // Last-step result order -> first-step operand order, for chaining a second transform (forward . filter . inverse).
    // Both are the SAME thread's values: element indices are tid + T j, j < V, with j = u + (q << (LOGV - lr)) in either
    // order, so the change of order is a renaming of registers -- no exchange.
    static __device__ __forceinline__ void permute_out_to_in(float2 (&v)[B][V]) {
        constexpr int SH_OUT = LOGV - St::lr(S - 1), SH_IN = LOGV - St::lr(0);
        if constexpr (LOGL >= LOGV && SH_OUT != SH_IN) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                float2 t[V];
#pragma unroll
                for (int s = 0; s < V; ++s) t[s] = v[b][s];
#pragma unroll
                for (int u = 0; u < NU0; ++u)
#pragma unroll
                    for (int q = 0; q < RHO0; ++q) {
                        const int j = u + (q << SH_IN);                                   // element tid + T j
                        v[b][u * RHO0 + q] = t[(j & (NUL - 1)) * RHOL + (j >> SH_OUT)];   // out slot (u', q') with u' + (q' << SH_OUT) = j
                    }
            }
        }
    }

    // tw: parity -> the direction-specific recurrence table; fast -> unused once bases are hoisted
    template <int SEQ0, bool INV>
    static __device__ __forceinline__ void run(float2 (&v)[B][V], float2* lds, const float2* __restrict__ tw, const Bases& bs,
                                               int tid) {
#ifdef FDR_DEBUG_SKIP_FFT  // timing-only builds (tools/microbench): memory phases without the transform
        (void)lds; (void)tw; (void)bs; (void)tid; (void)v;
#else
        steps_from<0, SEQ0, INV>(v, lds, tw, bs, tid);
#endif
    }
};

// ---------------------------------------------------------------------------------------------
// min/max of the real plane: every producing workgroup writes ONE (min, max) partial; a single
// small kernel reduces the partials (deterministic, and no same-address atomics: 65k atomics on
// two words cost ~0.7 ms on MI355X, 15x the kernel that issued them).
// ---------------------------------------------------------------------------------------------
// min / max of three as ONE instruction.  Written as fminf / fmaxf chains hipcc quiets every operand it cannot prove
// canonical first (`v_max_f32 x, x, x`: transform results that went through LDS, DPP or an asm tie): pass C1's 64 values per
// lane cost 490 VALU instructions that way, 64 this way -- and that pass is bound by its VALU issue (4 waves per SIMD).
// NaN handling as v_min_f32 / v_max_f32 in IEEE mode (a quiet NaN operand is ignored, like fminf).
__device__ __forceinline__ float fdr_min3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return fminf(a, fminf(b, c));
#endif
}
__device__ __forceinline__ float fdr_max3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return fmaxf(a, fmaxf(b, c));
#endif
}

__device__ __forceinline__ void block_minmax_store(float mn, float mx, float2* __restrict__ part, int index = -1) {
    __shared__ float2 red[16];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    FDR_JITTER(1001);
    if ((threadIdx.x & 63) == 0) red[wave] = make_float2(mn, mx);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) {
            mn = fminf(mn, red[w].x);
            mx = fmaxf(mx, red[w].y);
        }
        const int b = index >= 0 ? index : (int)(blockIdx.x + gridDim.x * blockIdx.y);
        part[b] = make_float2(mn, mx);  // (+inf, -inf) when the block saw no counted element
    }
}

// cv::normalize(src, dst, 0, 1, NORM_MINMAX) scale / shift (fft/fft_serial.cpp:246): double min/max, scale rounded to
// float, shift = (float)0 - (float)(smin * scale); applied as a float multiply then a float add.
__device__ __forceinline__ void minmax_to_scale_shift(float mn, float mx, float& fscale, float& fshift) {
    const double smin = (double)mn, smax = (double)mx;
    double scale = ((smax - smin) > 2.2204460492503131e-16) ? 1.0 / (smax - smin) : 0.0;
    scale = (double)(float)scale;
    fscale = (float)scale;
    fshift = 0.0f - (float)(smin * scale);
}

// every workgroup folds the n_part per-workgroup (min, max) partials of an image itself (exact, order independent) and
// derives the normalisation constants: saves a reduce launch (normalize_kernel, pass C2)
__device__ __forceinline__ void block_fold_partials(const float2* __restrict__ part, int n_part, float& fscale, float& fshift) {
    __shared__ float2 fold_red[16];
    float mn = __builtin_inff(), mx = -__builtin_inff();
    // the first four partials per thread are requested unconditionally (clamped index: a repeated partial changes
    // nothing), so they travel together and behind whatever the caller has in flight
    float2 p4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = (int)threadIdx.x + k * (int)blockDim.x;
        p4[k] = part[i < n_part ? i : n_part - 1];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mn = fminf(mn, p4[k].x);
        mx = fmaxf(mx, p4[k].y);
    }
    for (int i = threadIdx.x + 4 * blockDim.x; i < n_part; i += blockDim.x) {
        const float2 p = part[i];
        mn = fminf(mn, p.x);
        mx = fmaxf(mx, p.y);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    const int nw = (blockDim.x + 63) >> 6;
    FDR_JITTER(1003);
    if ((threadIdx.x & 63) == 0) fold_red[threadIdx.x >> 6] = make_float2(mn, mx);
    __syncthreads();
    FDR_JITTER(1004);
    mn = fold_red[0].x; mx = fold_red[0].y;
    for (int w = 1; w < nw; ++w) {
        mn = fminf(mn, fold_red[w].x);
        mx = fmaxf(mx, fold_red[w].y);
    }
    minmax_to_scale_shift(mn, mx, fscale, fshift);
}

}  // namespace fdr
