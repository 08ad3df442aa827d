// fdr_api.hip -- host side of libfdr.so: plans, twiddle tables, pass sequencing, the C ABI of
// include/fdr.h.  No torch, no OpenCV; only the HIP runtime.  There is deliberately NO CPU
// fallback anywhere in this file: every entry point either launches HIP kernels or fails.
#include "../../include/fdr.h"
#include "fdr_kernels.hpp"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace fdr;

namespace {

thread_local std::string g_last_error;
std::atomic<bool> g_process_exiting{false};  // set by an atexit handler that runs before the HIP runtime's own (fdr_plan_destroy)

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define FDR_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            char buf_[512];                                                                        \
            snprintf(buf_, sizeof buf_, "%s:%d: %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return fail(FDR_ERR_HIP, buf_);                                                        \
        }                                                                                          \
    } while (0)

int ilog2(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}

// Per-stage twiddle table for transforms of length n: stage len = 2,4,..,n at offset len/2-1.
// parity: replay of fft/fft_serial.cpp:54-63 -- ang evaluated in double and rounded to float,
//         wlen = (cosf(ang), sinf(ang)), w advanced by the float recurrence w *= wlen.
// fast  : exp(-+2 pi i k / len) evaluated in double (as fft/fft_gpu.cu:206-212), rounded once.
void build_twiddles(int n, int mode, bool inverse, std::vector<float2>& out) {
    out.assign(n > 1 ? (size_t)n - 1 : 1, make_float2(1.f, 0.f));
    const double PI = 3.1415926535897932384626433832795;  // CV_PI
    for (int len = 2; len <= n; len <<= 1) {
        float2* t = out.data() + (len / 2 - 1);
        if (mode == FDR_MODE_PARITY) {
            const float ang = (float)((double)2.0f * PI / (double)len * (double)(inverse ? 1.0f : -1.0f));
            const float wlr = cosf(ang), wli = sinf(ang);
            float wr = 1.0f, wi = 0.0f;
            for (int k = 0; k < len / 2; ++k) {
                t[k] = make_float2(wr, wi);
                const float ac = wr * wlr, bd = wi * wli, ad = wr * wli, bc = wi * wlr;
                wr = ac - bd;
                wi = ad + bc;
            }
        } else {
            for (int k = 0; k < len / 2; ++k) {
                const double a = (inverse ? 2.0 : -2.0) * PI * (double)k / (double)len;
                t[k] = make_float2((float)cos(a), (float)sin(a));
            }
        }
    }
}

// n x n twiddle table of fft_serial::dft_naive_inplace (fft/fft_serial.cpp:71-87), forward direction, laid out
// [t][k] so that adjacent threads (adjacent k) read adjacent entries: ang = 2.0f * CV_PI * k * t / n * sign evaluated left
// to right in double, rounded to float, then the C library's cosf / sinf -- the calls the serial path makes.
void build_naive_table(int n, std::vector<float2>& out) {
    out.resize((size_t)n * n);
    const double PI = 3.1415926535897932384626433832795;
    for (int k = 0; k < n; ++k)
        for (int t = 0; t < n; ++t) {
            const float ang = (float)((double)2.0f * PI * (double)k * (double)t / (double)n * (double)-1.0f);
            out[(size_t)t * n + k] = make_float2(cosf(ang), sinf(ang));
        }
}
constexpr int kMaxNaiveLen = 4096;  // 128 MiB of table

struct PassTimer {
    static constexpr int kMaxRecords = 8192;
    struct Rec { hipEvent_t a, b; int pass; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    bool enabled = false;
    const char* names[FDR_MAX_PASSES] = {nullptr};
    int n_names = 0;

    int pass_id(const char* name) {
        for (int i = 0; i < n_names; ++i)
            if (names[i] == name) return i;
        if (n_names < FDR_MAX_PASSES) { names[n_names] = name; return n_names++; }
        return FDR_MAX_PASSES - 1;
    }
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void reset() {
        for (auto& r : recs) { pool.push_back(r.a); pool.push_back(r.b); }
        recs.clear();
    }
    void destroy() {
        reset();
        for (auto e : pool) (void)hipEventDestroy(e);
        pool.clear();
    }
};

}  // namespace

struct fdr_plan {
    int device = 0, M = 0, N = 0, logM = 0, logN = 0, mode = 0;
    unsigned flags = 0;
    bool simple = false;
    bool ppar = false;  // parity operator on a PANEL-major complex intermediate (round 4): contiguous column tiles
    bool big = false;  // a power-of-two dimension above 8192: simple sequence with the long row pass (fdr_aux.hip)
    int num_cu = 256;
    bool tables_only = false;  // FDR_FLAG_TABLES_ONLY: no workspaces, slab primitives only
    bool generic = false;  // FDR_FLAG_ANY_SIZE with a non-power-of-two dimension: naive DFT along that dimension
    float2 *naive_row = nullptr, *naive_col = nullptr;  // n x n tables of the non-power-of-two dimensions (length N / M)
    bool panel = false;
    size_t pstride = 0;  // panel stride (float2 elements)
    bool half = false;   // fast mode: only the non-redundant half of the Hermitian spectrum is kept (N/8 panels, Nyquist packed into column 0)
    int npanels = 0;  // fast mode: panel-major intermediate spectrum and filter
    float2 *tw_row_f = nullptr, *tw_row_i = nullptr, *tw_col_f = nullptr, *tw_col_i = nullptr;
    float2* work = nullptr;   // M x N complex working spectrum
    float2* work2 = nullptr;  // simple path: N x M transpose buffer
    float2* filt = nullptr;   // H (parity) or W (fast)
    float* raw = nullptr;     // M x N real plane before normalisation
    float* psf_dev = nullptr; // staging for host-pointer / generated PSFs
    float *stage_in = nullptr, *stage_out = nullptr;  // device staging of the host-pointer entry (kept between calls)
    size_t stage_cap = 0;
    size_t psf_cap = 0;
    float* mm = nullptr;       // final {min, max}
    float2* mm_part = nullptr; // per-workgroup partials
    int mm_part_cap = 0;
    float K = 0.f;
    bool have_psf = false;
    PassTimer timer;
    // the reference Profiler's buckets (fdr_plan_phase_times): resolved sums + event pairs not read back yet
    struct PhaseRec { hipEvent_t a, b; int phase; };
    double phase_ms[FDR_N_PHASES] = {0, 0, 0, 0, 0, 0};
    std::vector<PhaseRec> phase_pending;
    // batched mode: images alternate over `nslots` private workspaces, each on its own internal stream,
    // so the tail of one image's kernels overlaps the head of the next image's (slot 0 = the buffers above)
    struct Slot {
        float2* work = nullptr; float2* work2 = nullptr; float* raw = nullptr; float* mm = nullptr; float2* mm_part = nullptr;
        hipStream_t stream = nullptr; hipEvent_t done = nullptr;
    };
    static constexpr int kMaxSlots = 16;
    Slot slots[kMaxSlots];
    int nslots = 1;   // = nstreams * group
    int nstreams = 1;
    int group = 1;    // images per pass-B' launch (panel path)
    hipEvent_t fork = nullptr;
    size_t ws_elems = 0;  // elements of one work / raw buffer
    bool two_sweep = true;           // FDR_OPT_TWO_SWEEP_NORM: passes C1 + C2 instead of C' + E (fast half-spectrum path)
    size_t ce_chunk_bytes = (size_t)160 << 20;  // FDR_OPT_CE_CHUNK_MB: spectrum bytes per C1 + C2 launch pair of a multi-stream batch (0 = whole group)
    // FDR_OPT_BATCH_GRAPH: the launches of one fdr_wiener_batch_f32_dev call (fork, every pass of every group on the
    // internal streams, join) captured once as a hipGraph and replayed while the call's arguments stay the same
    struct GraphKey {
        const float* in; float* out; size_t in_pitch, out_pitch; int count, rows, cols, stride, out_stride, norm_area, nstreams, group;
        bool two_sweep; float K; size_t ce_cache;
        bool operator==(const GraphKey& o) const {
            return in == o.in && out == o.out && in_pitch == o.in_pitch && out_pitch == o.out_pitch && count == o.count && rows == o.rows &&
                   cols == o.cols && stride == o.stride && out_stride == o.out_stride && norm_area == o.norm_area && nstreams == o.nstreams &&
                   group == o.group && two_sweep == o.two_sweep && K == o.K && ce_cache == o.ce_cache;
        }
    };
    // host-pointer batch (fdr_wiener_batch_*_f32): three streams, three images in flight; created on first use and kept --
    // a driver that calls wienerDeblur_RGB_optimized once per picture (3 channels per call) would otherwise pay three
    // hipStreamCreate, six hipMalloc / hipFree and nine event creations per call: 16 of the 17.6 ms such a call took on a
    // 782 x 1920 picture whose device work is under 1 ms
    struct HostPipe {
        hipStream_t s_in = nullptr, s_cmp = nullptr, s_out = nullptr;
        float* d_in[3] = {nullptr, nullptr, nullptr};
        float* d_out[3] = {nullptr, nullptr, nullptr};
        hipEvent_t e_in[3] = {nullptr, nullptr, nullptr}, e_cmp[3] = {nullptr, nullptr, nullptr}, e_out[3] = {nullptr, nullptr, nullptr};
        size_t cap = 0;  // bytes of each d_in / d_out buffer
        bool ready = false;  // all three streams and nine events exist
    } pipe;
    bool batch_graph = false;
    hipGraphExec_t graph_exec = nullptr;
    GraphKey graph_key{};
    hipStream_t cap_stream = nullptr;
};

namespace {

struct ScopedPass {
    fdr_plan* p; hipStream_t s; PassTimer::Rec rec; bool on;
    ScopedPass(fdr_plan* plan, hipStream_t st, const char* name) : p(plan), s(st), on(false) {
        if (p->timer.enabled && (int)p->timer.recs.size() < PassTimer::kMaxRecords) {
            rec.a = p->timer.get(); rec.b = p->timer.get(); rec.pass = p->timer.pass_id(name);
            on = rec.a && rec.b;
            if (on) (void)hipEventRecord(rec.a, s);
        }
    }
    ~ScopedPass() {
        if (on) { (void)hipEventRecord(rec.b, s); p->timer.recs.push_back(rec); }
    }
};

// folds the pending pairs whose end event has already completed (no waiting) into the sums; keeps the others
void resolve_finished_phases(fdr_plan* p) {
    size_t keep = 0;
    for (auto& r : p->phase_pending) {
        float ms = 0.f;
        if (hipEventQuery(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            p->phase_ms[r.phase] += ms;
            p->timer.pool.push_back(r.a); p->timer.pool.push_back(r.b);
        } else {
            p->phase_pending[keep++] = r;
        }
    }
    p->phase_pending.resize(keep);
}

// One hipEvent pair on stream s around a phase of the reference's Profiler (fft/fft_gpu.cu:17-57); read back by
// resolve_phases.  Bounded at 1024 unread pairs; from 768 on, pairs that have completed are folded in first (no waiting), so only
// a caller with more than 1024 phases IN FLIGHT at once loses records.
struct ScopedPhase {
    fdr_plan* p; hipStream_t s; fdr_plan::PhaseRec rec; bool on;
    ScopedPhase(fdr_plan* plan, int phase, hipStream_t st) : p(plan), s(st), on(false) {
        if (p->phase_pending.size() >= 768) resolve_finished_phases(p);  // long host batches / many PSF rebuilds: fold what has completed
        if (p->phase_pending.size() < 1024) {
            rec.a = p->timer.get(); rec.b = p->timer.get(); rec.phase = phase;
            on = rec.a && rec.b;
            if (on) (void)hipEventRecord(rec.a, s);
        }
    }
    ~ScopedPhase() {
        if (on) { (void)hipEventRecord(rec.b, s); p->phase_pending.push_back(rec); }
    }
};
void resolve_phases(fdr_plan* p) {
    for (auto& r : p->phase_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) p->phase_ms[r.phase] += ms;
        p->timer.pool.push_back(r.a); p->timer.pool.push_back(r.b);
    }
    p->phase_pending.clear();
}

// names are static strings compared by pointer in PassTimer::pass_id
const char* const kPassRowsFwd = "A rows: pad+FFT (real->complex)";
const char* const kPassColsWiener = "B cols: FFT+Wiener";
const char* const kPassRowsInv = "C rows: IFFT (complex)";
const char* const kPassColsInvReal = "D cols: IFFT+real+minmax";
const char* const kPassColsFused = "B' cols: FFT*W*IFFT";
const char* const kPassColsFusedN[kMaxGroup + 1] = {
    nullptr,
    kPassColsFused,
    "B' cols: FFT*W*IFFT [2 images]",
    "B' cols: FFT*W*IFFT [3 images]",
    "B' cols: FFT*W*IFFT [4 images]",
    "B' cols: FFT*W*IFFT [5 images]",
    "B' cols: FFT*W*IFFT [6 images]",
    "B' cols: FFT*W*IFFT [7 images]",
    "B' cols: FFT*W*IFFT [8 images]"};
const char* const kPassRowsInvReal = "C' rows: IFFT+real+minmax";
const char* const kPassRowsFwdN[kMaxGroup + 1] = {
    nullptr,
    nullptr,
    "A rows: pad+FFT (real->complex) [2 images]",
    "A rows: pad+FFT (real->complex) [3 images]",
    "A rows: pad+FFT (real->complex) [4 images]",
    "A rows: pad+FFT (real->complex) [5 images]",
    "A rows: pad+FFT (real->complex) [6 images]",
    "A rows: pad+FFT (real->complex) [7 images]",
    "A rows: pad+FFT (real->complex) [8 images]"};
const char* const kPassRowsInvRealN[kMaxGroup + 1] = {
    nullptr,
    nullptr,
    "C' rows: IFFT+real+minmax [2 images]",
    "C' rows: IFFT+real+minmax [3 images]",
    "C' rows: IFFT+real+minmax [4 images]",
    "C' rows: IFFT+real+minmax [5 images]",
    "C' rows: IFFT+real+minmax [6 images]",
    "C' rows: IFFT+real+minmax [7 images]",
    "C' rows: IFFT+real+minmax [8 images]"};
const char* const kPassNormalizeN[kMaxGroup + 1] = {
    nullptr,
    nullptr,
    "E normalize+crop [2 images]",
    "E normalize+crop [3 images]",
    "E normalize+crop [4 images]",
    "E normalize+crop [5 images]",
    "E normalize+crop [6 images]",
    "E normalize+crop [7 images]",
    "E normalize+crop [8 images]"};
const char* const kPassNormalize = "E normalize+crop";
const char* const kPassRowsMinmax = "C1 rows: IFFT+minmax";
const char* const kPassRowsMinmaxN[kMaxGroup + 1] = {
    nullptr,
    nullptr,
    "C1 rows: IFFT+minmax [2 images]",
    "C1 rows: IFFT+minmax [3 images]",
    "C1 rows: IFFT+minmax [4 images]",
    "C1 rows: IFFT+minmax [5 images]",
    "C1 rows: IFFT+minmax [6 images]",
    "C1 rows: IFFT+minmax [7 images]",
    "C1 rows: IFFT+minmax [8 images]"};
const char* const kPassRowsNorm = "C2 rows: IFFT+normalize+crop";
const char* const kPassRowsNormN[kMaxGroup + 1] = {
    nullptr,
    nullptr,
    "C2 rows: IFFT+normalize+crop [2 images]",
    "C2 rows: IFFT+normalize+crop [3 images]",
    "C2 rows: IFFT+normalize+crop [4 images]",
    "C2 rows: IFFT+normalize+crop [5 images]",
    "C2 rows: IFFT+normalize+crop [6 images]",
    "C2 rows: IFFT+normalize+crop [7 images]",
    "C2 rows: IFFT+normalize+crop [8 images]"};
const char* const kPassSimple = "simple path (reference-shaped)";

int upload(float2** dst, const std::vector<float2>& v) {
    FDR_HIP(hipMalloc((void**)dst, v.size() * sizeof(float2)));
    FDR_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(float2), hipMemcpyHostToDevice));
    return FDR_OK;
}

// unscaled 2-D transform in place on d (M x N), rows then columns as fft/fft_serial.cpp:113-139
// `rows` transforms of L = 2^logl > 8192 points held contiguously in `buf`, `tmp` of the same size free: see fdr_aux.hip
// (long_gather_kernel).  twf / twi: the forward / inverse tables of the plan's mode for length L (their first 8191 entries
// are the tables of the 8192-point transform: build_twiddles stores stage `len` at offset len/2 - 1).  Result in `buf`.
hipError_t long_rows_dev(float2* buf, float2* tmp, size_t rows, int L, int logl, int mode, bool inverse, const float2* twf, const float2* twi,
                         hipStream_t s) {
    const int logs = logl - kMaxLdsLog, L0 = 1 << kMaxLdsLog;
    if (rows << logs > (size_t)0x7fffffff) return hipErrorInvalidValue;  // (the row kernels count rows in an int)
    hipError_t e = launch_long_gather(buf, tmp, rows, L, logs, s);
    if (e != hipSuccess) return e;
    RowArgs ra{};
    ra.src_c = tmp; ra.dst_c = tmp; ra.M = (int)(rows << logs);
    // the register kernels: parity -> the table of the direction; fast -> the forward table (they conjugate it)
    e = launch_rows(kMaxLdsLog, mode, ROW_IN_COMPLEX, ROW_OUT_COMPLEX, inverse, ra, mode == FDR_MODE_FAST ? twf : (inverse ? twi : twf), s);
    if (e != hipSuccess) return e;
    for (int half = L0; half < L; half <<= 1) {  // in place in `tmp` (a butterfly reads and writes its own pair), the last one into `buf`
        const bool last = (half << 1) == L;
        e = launch_long_stage(tmp, last ? buf : tmp, rows, L, half, inverse ? twi : twf, mode, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int dft2d_dev(fdr_plan* p, float2* d, float2* work2, bool inverse, hipStream_t s) {
    const float2* twr = inverse ? p->tw_row_i : p->tw_row_f;
    const float2* twc = inverse ? p->tw_col_i : p->tw_col_f;
    if (p->simple) {  // the reference's own sequence: rows, transpose, rows, transpose (fft/fft_serial.cpp:113-139)
        // one row pass over `rows` rows of length L held in `buf`, `tmp` free: radix-2 for powers of two, else the naive
        // DFT (transform_row_inplace, :100-101), which runs out of place and is copied back
        auto row_pass = [&](float2* buf, float2* tmp, int rows, int L, int logl, const float2* tw, const float2* twf, const float2* twi,
                            const float2* naive) -> int {
            if (naive) {
                FDR_HIP(launch_dft_naive_rows(buf, tmp, rows, L, naive, inverse ? 1 : 0, s));
                FDR_HIP(hipMemcpyAsync(buf, tmp, (size_t)rows * L * sizeof(float2), hipMemcpyDeviceToDevice, s));
            } else if (logl > kMaxLdsLog) {  // more than 8192 points: 8192-point blocks + global radix-2 stages
                FDR_HIP(long_rows_dev(buf, tmp, (size_t)rows, L, logl, p->mode, inverse, twf, twi, s));
            } else if ((p->generic || p->big) && L >= 8) {
                // register kernels, in place (parity: the table of the direction; fast: the forward table, they conjugate it)
                RowArgs ra{};
                ra.src_c = buf; ra.dst_c = buf; ra.M = rows;
                FDR_HIP(launch_rows(logl, p->mode, ROW_IN_COMPLEX, ROW_OUT_COMPLEX, inverse, ra, p->mode == FDR_MODE_FAST ? twf : tw, s));
            } else {
                FDR_HIP(launch_simple_rows(buf, rows, L, logl, tw, p->mode, s));
            }
            return FDR_OK;
        };
        int rc = row_pass(d, work2, p->M, p->N, p->logN, twr, p->tw_row_f, p->tw_row_i, p->naive_row);
        if (rc != FDR_OK) return rc;
        FDR_HIP(launch_transpose(d, work2, p->M, p->N, s));
        rc = row_pass(work2, d, p->N, p->M, p->logM, twc, p->tw_col_f, p->tw_col_i, p->naive_col);
        if (rc != FDR_OK) return rc;
        FDR_HIP(launch_transpose(work2, d, p->N, p->M, s));
        return FDR_OK;
    }
    RowArgs ra{};
    ra.src_c = d; ra.dst_c = d; ra.M = p->M;
    FDR_HIP(launch_rows(p->logN, p->mode, ROW_IN_COMPLEX, ROW_OUT_COMPLEX, inverse, ra,
                        p->mode == FDR_MODE_FAST ? p->tw_row_f : twr, s));
    ColArgs ca{};
    ca.data = d; ca.N = p->N;
    FDR_HIP(launch_cols(p->logM, p->mode, inverse ? COL_INV : COL_FWD, ca, p->tw_col_f, p->tw_col_i, s));
    return FDR_OK;
}

int set_psf_dev_impl(fdr_plan* p, const float* d_psf, int prows, int pcols, int pstride, float K, hipStream_t s) {
    if (p->tables_only) return fail(FDR_ERR_STATE, "fdr_set_psf: plan was created with FDR_FLAG_TABLES_ONLY (slab primitives only)");
    if (prows <= 0 || pcols <= 0 || pstride < pcols) return fail(FDR_ERR_ARG, "fdr_set_psf: bad PSF shape");
    if (prows > p->M || pcols > p->N)
        return fail(FDR_ERR_ARG, "fdr_set_psf: PSF larger than the padded image (copyMakeBorder would throw, fft_serial.cpp:168)");
    ScopedPhase phase(p, FDR_PHASE_PRE, s);
    // pad top-left + forward 2-D FFT (fft/fft_serial.cpp:166-171,182)
    if (p->simple) {
        FDR_HIP(launch_pad_real_to_complex(d_psf, prows, pcols, pstride, p->filt, p->M, p->N, s));
        int rc = dft2d_dev(p, p->filt, p->work2, false, s);
        if (rc != FDR_OK) return rc;
    } else if (p->panel) {
        // the PSF reaches only the first `prows` rows of the padded field: the row pass transforms just those row groups,
        // the column pass takes every row below as zero (unread) and turns the spectrum into W on its way out
        const int nvalid = (prows + 3) & ~3;  // <= M (M is a multiple of 8 on this path)
        RowArgs ra{};
        ra.src_real = d_psf; ra.src_rows = prows; ra.src_cols = pcols; ra.src_stride = pstride;
        ra.dst_c = p->filt; ra.M = nvalid; ra.pstride = p->pstride; ra.half = p->half; ra.num_cu = p->num_cu;
        FDR_HIP(launch_rows4(p->logN, ROW_IN_REAL, ROW_OUT_COMPLEX, ra, p->tw_row_f, s));
        ColArgs ca{};
        ca.data = p->filt; ca.N = p->N; ca.num_cu = p->num_cu; ca.pstride = p->pstride; ca.npanels = p->npanels;
        ca.nvalid = nvalid; ca.K = K; ca.packed0 = p->half ? 1 : 0;
        FDR_HIP(launch_cols_panel(p->logM, COL_FWD_FILTER, ca, p->tw_col_f, s));
    } else {
        RowArgs ra{};
        ra.src_real = d_psf; ra.src_rows = prows; ra.src_cols = pcols; ra.src_stride = pstride;
        ra.dst_c = p->filt; ra.M = p->M; ra.panel_c = p->ppar ? 1 : 0; ra.pstride = p->pstride;
        FDR_HIP(launch_rows(p->logN, p->mode, ROW_IN_REAL, ROW_OUT_COMPLEX, false, ra, p->tw_row_f, s));
        ColArgs ca{};
        ca.data = p->filt; ca.N = p->N; ca.panel_c = p->ppar ? 1 : 0; ca.pstride = p->pstride;
        FDR_HIP(launch_cols(p->logM, p->mode, COL_FWD, ca, p->tw_col_f, p->tw_col_i, s));
    }
    if (p->mode == FDR_MODE_FAST && !p->panel)  // (the panel path's column pass has written W already)
        FDR_HIP(launch_make_filter_fast(p->filt, p->filt, (size_t)p->M * p->N, K, s));
    p->K = K;
    p->have_psf = true;
    return FDR_OK;
}



// ---- fast panel path in three stages, so that pass B' can be launched once for a group of images ----
int panel_stage_A(fdr_plan* p, fdr_plan::Slot& w, const float* d_img, int rows, int cols, int stride, hipStream_t s) {
    ScopedPass t(p, s, kPassRowsFwd);   // A: 4 rows per thread group, real -> panel-major (half) spectrum
    RowArgs a{};
    a.src_real = d_img; a.src_rows = rows; a.src_cols = cols; a.src_stride = stride;
    a.dst_c = w.work; a.M = p->M; a.pstride = p->pstride; a.half = p->half; a.num_cu = p->num_cu;
    FDR_HIP(launch_rows4(p->logN, ROW_IN_REAL, ROW_OUT_COMPLEX, a, p->tw_row_f, s));
    return FDR_OK;
}
int panel_stage_B(fdr_plan* p, fdr_plan::Slot* const* ws, int n, hipStream_t s) {
    ScopedPass t(p, s, kPassColsFusedN[n]);  // B': per panel, columns forward * W * inverse
    ColArgs c{};
    c.data = ws[0]->work; c.filt = p->filt; c.K = p->K; c.N = p->N; c.num_cu = p->num_cu;
    c.pstride = p->pstride; c.npanels = p->npanels; c.packed0 = p->half ? 1 : 0;
    c.batch.nimg = n;
    for (int k = 0; k < n; ++k) c.batch.data[k] = ws[k]->work;
    FDR_HIP(launch_cols_panel(p->logM, COL_FUSED, c, p->tw_col_f, s));
    return FDR_OK;
}
int panel_stage_CE(fdr_plan* p, fdr_plan::Slot& w, int rows, int cols, float* d_out, int out_stride, int mm_rows,
                   int mm_cols, hipStream_t s) {
    if (p->two_sweep && p->half) {
        // C1 + C2: the inverse row transform runs twice -- once for the min/max alone, once more with the normalisation
        // applied on store -- so the raw real plane never exists: 4 + 8 bytes per pixel instead of 8 + 8
        RowArgs a{};
        a.src_c = w.work; a.mm_part = w.mm_part; a.mm_rows = mm_rows; a.mm_cols = mm_cols; a.M = p->M;
        a.pstride = p->pstride; a.half = 1; a.num_cu = p->num_cu;
        a.out = d_out; a.out_rows = rows; a.out_cols = cols; a.out_stride = out_stride;
        a.n_part = rows4_minmax_partials(p->logN, p->M, p->num_cu, 1, p->half ? 1 : 0);
        if (a.n_part <= 0 || a.n_part > p->mm_part_cap) return fail(FDR_ERR_STATE, "fdr_wiener: min/max partial count out of range");
        {
            ScopedPass t(p, s, kPassRowsMinmax);
            FDR_HIP(launch_rows4(p->logN, ROW_IN_COMPLEX, ROW_OUT_MINMAX_ONLY, a, p->tw_row_f, s));
        }
        {
            ScopedPass t(p, s, kPassRowsNorm);
            FDR_HIP(launch_rows4(p->logN, ROW_IN_COMPLEX, ROW_OUT_NORMALIZED, a, p->tw_row_f, s));
        }
        return FDR_OK;
    }
    {   // C': 4 rows rebuilt from the panels, inverse, real plane, min/max partials
        ScopedPass t(p, s, kPassRowsInvReal);
        RowArgs a{};
        a.src_c = w.work; a.dst_real = w.raw; a.mm_part = w.mm_part; a.mm_rows = mm_rows; a.mm_cols = mm_cols; a.M = p->M;
        a.pstride = p->pstride; a.half = p->half; a.num_cu = p->num_cu;
        FDR_HIP(launch_rows4(p->logN, ROW_IN_COMPLEX, ROW_OUT_REAL_MINMAX, a, p->tw_row_f, s));
    }
    {   // E: normalise to [0,1] and crop
        ScopedPass t(p, s, kPassNormalize);
        const int n_part = rows4_minmax_partials(p->logN, p->M, p->num_cu, 1, p->half ? 1 : 0);
        if (n_part <= 0 || n_part > p->mm_part_cap || n_part > 4096) return fail(FDR_ERR_STATE, "fdr_wiener: min/max partial count out of range");
        FDR_HIP(launch_normalize(w.raw, p->N, w.mm_part, n_part, nullptr, d_out, rows, cols, out_stride, s));
    }
    return FDR_OK;
}

// the same passes for a GROUP of 2..4 images in one launch each (blockIdx.y = image); packed half-spectrum path only
bool can_batch_rows(const fdr_plan* p) { return p->half; }
int panel_stage_A_batch(fdr_plan* p, fdr_plan::Slot* const* ws, int n, const float* const* d_imgs, int rows, int cols, int stride,
                        hipStream_t s) {
    ScopedPass t(p, s, kPassRowsFwdN[n]);
    RowArgs a{};
    a.src_real = d_imgs[0]; a.src_rows = rows; a.src_cols = cols; a.src_stride = stride;
    a.dst_c = ws[0]->work; a.M = p->M; a.pstride = p->pstride; a.half = 1; a.num_cu = p->num_cu;
    a.batch.nimg = n;
    for (int k = 0; k < kMaxGroup; ++k) { a.batch.src_real[k] = d_imgs[k < n ? k : 0]; a.batch.spec[k] = ws[k < n ? k : 0]->work; }
    FDR_HIP(launch_rows4(p->logN, ROW_IN_REAL, ROW_OUT_COMPLEX, a, p->tw_row_f, s));
    return FDR_OK;
}
int panel_stage_CE_batch(fdr_plan* p, fdr_plan::Slot* const* ws, int n, int rows, int cols, float* const* d_outs, int out_stride,
                         int mm_rows, int mm_cols, hipStream_t s) {
    if (p->two_sweep) {  // C1 + C2 (see panel_stage_CE)
        RowArgs a{};
        a.src_c = ws[0]->work; a.mm_part = ws[0]->mm_part; a.mm_rows = mm_rows; a.mm_cols = mm_cols; a.M = p->M;
        a.pstride = p->pstride; a.half = 1; a.num_cu = p->num_cu;
        a.out = d_outs[0]; a.out_rows = rows; a.out_cols = cols; a.out_stride = out_stride;
        a.n_part = rows4_minmax_partials(p->logN, p->M, p->num_cu, n, p->half ? 1 : 0);
        if (a.n_part <= 0 || a.n_part > p->mm_part_cap) return fail(FDR_ERR_STATE, "fdr_wiener: min/max partial count out of range");
        a.batch.nimg = n;
        for (int k = 0; k < kMaxGroup; ++k) {
            const fdr_plan::Slot* w = ws[k < n ? k : 0];
            a.batch.spec[k] = w->work; a.batch.mm_part[k] = w->mm_part; a.batch.out[k] = d_outs[k < n ? k : 0];
        }
        {
            ScopedPass t(p, s, kPassRowsMinmaxN[n]);
            FDR_HIP(launch_rows4(p->logN, ROW_IN_COMPLEX, ROW_OUT_MINMAX_ONLY, a, p->tw_row_f, s));
        }
        {
            ScopedPass t(p, s, kPassRowsNormN[n]);
            FDR_HIP(launch_rows4(p->logN, ROW_IN_COMPLEX, ROW_OUT_NORMALIZED, a, p->tw_row_f, s));
        }
        return FDR_OK;
    }
    {
        ScopedPass t(p, s, kPassRowsInvRealN[n]);
        RowArgs a{};
        a.src_c = ws[0]->work; a.dst_real = ws[0]->raw; a.mm_part = ws[0]->mm_part; a.mm_rows = mm_rows; a.mm_cols = mm_cols; a.M = p->M;
        a.pstride = p->pstride; a.half = 1; a.num_cu = p->num_cu;
        a.batch.nimg = n;
        for (int k = 0; k < kMaxGroup; ++k) {
            const fdr_plan::Slot* w = ws[k < n ? k : 0];
            a.batch.spec[k] = w->work; a.batch.raw[k] = w->raw; a.batch.mm_part[k] = w->mm_part;
        }
        FDR_HIP(launch_rows4(p->logN, ROW_IN_COMPLEX, ROW_OUT_REAL_MINMAX, a, p->tw_row_f, s));
    }
    {
        ScopedPass t(p, s, kPassNormalizeN[n]);
        const int n_part = rows4_minmax_partials(p->logN, p->M, p->num_cu, n, p->half ? 1 : 0);
        if (n_part <= 0 || n_part > p->mm_part_cap || n_part > 4096) return fail(FDR_ERR_STATE, "fdr_wiener: min/max partial count out of range");
        NormBatch nb{};
        nb.nimg = n;
        for (int k = 0; k < kMaxGroup; ++k) {
            const fdr_plan::Slot* w = ws[k < n ? k : 0];
            nb.raw[k] = w->raw; nb.part[k] = w->mm_part; nb.out[k] = d_outs[k < n ? k : 0];
        }
        FDR_HIP(launch_normalize(ws[0]->raw, p->N, ws[0]->mm_part, n_part, nullptr, d_outs[0], rows, cols, out_stride, s, &nb));
    }
    return FDR_OK;
}

int check_image_args(fdr_plan* p, const float* d_img, int rows, int cols, int stride, float* d_out, int out_stride) {
    if (p->tables_only) return fail(FDR_ERR_STATE, "fdr_wiener: plan was created with FDR_FLAG_TABLES_ONLY (slab primitives only)");
    if (!p->have_psf) return fail(FDR_ERR_STATE, "fdr_wiener: no PSF set on this plan (call fdr_set_psf* first)");
    if (!d_img || !d_out) return fail(FDR_ERR_ARG, "fdr_wiener: null image pointer");
    if (rows <= 0 || cols <= 0 || rows > p->M || cols > p->N || stride < cols || out_stride < cols)
        return fail(FDR_ERR_ARG, "fdr_wiener: image shape does not fit the plan");
    return FDR_OK;
}

int wiener_dev_impl(fdr_plan* p, fdr_plan::Slot& w, const float* d_img, int rows, int cols, int stride, float* d_out,
                    int out_stride, int norm_area, hipStream_t s) {
    int vrc = check_image_args(p, d_img, rows, cols, stride, d_out, out_stride);
    if (vrc != FDR_OK) return vrc;
    const int mm_rows = norm_area == FDR_NORM_PADDED ? p->M : rows;
    const int mm_cols = norm_area == FDR_NORM_PADDED ? p->N : cols;
    const size_t P = (size_t)p->M * p->N;
    int n_part = 0;

    if (p->simple) {
        ScopedPass t(p, s, kPassSimple);
        FDR_HIP(launch_pad_real_to_complex(d_img, rows, cols, stride, w.work, p->M, p->N, s));
        int rc = dft2d_dev(p, w.work, w.work2, false, s);
        if (rc != FDR_OK) return rc;
        FDR_HIP(launch_wiener_pointwise(w.work, p->filt, P, p->K, p->mode, s));
        rc = dft2d_dev(p, w.work, w.work2, true, s);
        if (rc != FDR_OK) return rc;
        FDR_HIP(launch_real_minmax(w.work, w.raw, p->M, p->N, mm_rows, mm_cols, w.mm_part, &n_part, s));
    } else if (p->mode == FDR_MODE_PARITY) {
        {   // A: rows, real -> complex (fft/fft_serial.cpp:157-165,176 first half)
            ScopedPass t(p, s, kPassRowsFwd);
            RowArgs a{};
            a.src_real = d_img; a.src_rows = rows; a.src_cols = cols; a.src_stride = stride;
            a.dst_c = w.work; a.M = p->M; a.panel_c = p->ppar ? 1 : 0; a.pstride = p->pstride;
            FDR_HIP(launch_rows(p->logN, p->mode, ROW_IN_REAL, ROW_OUT_COMPLEX, false, a, p->tw_row_f, s));
        }
        {   // B: columns forward + Wiener quotient (:176 second half, :186-224)
            ScopedPass t(p, s, kPassColsWiener);
            ColArgs c{};
            c.data = w.work; c.filt = p->filt; c.K = p->K; c.N = p->N; c.panel_c = p->ppar ? 1 : 0; c.pstride = p->pstride;
            FDR_HIP(launch_cols(p->logM, p->mode, COL_FWD_WIENER, c, p->tw_col_f, p->tw_col_i, s));
        }
        {   // C: rows inverse (:229 first half)
            ScopedPass t(p, s, kPassRowsInv);
            RowArgs a{};
            a.src_c = w.work; a.dst_c = w.work; a.M = p->M; a.panel_c = p->ppar ? 1 : 0; a.pstride = p->pstride;
            FDR_HIP(launch_rows(p->logN, p->mode, ROW_IN_COMPLEX, ROW_OUT_COMPLEX, true, a, p->tw_row_i, s));
        }
        {   // D: columns inverse, real plane, min/max (:229 second half, :236-240, minMaxIdx of :246)
            ScopedPass t(p, s, kPassColsInvReal);
            ColArgs c{};
            c.data = w.work; c.dst_real = w.raw; c.mm_part = w.mm_part; c.mm_rows = mm_rows; c.mm_cols = mm_cols; c.N = p->N;
            c.panel_c = p->ppar ? 1 : 0; c.pstride = p->pstride;
            FDR_HIP(launch_cols(p->logM, p->mode, COL_INV_REAL, c, p->tw_col_f, p->tw_col_i, s));
            n_part = cols_minmax_partials(p->logM, p->N);
        }
    } else {  // fast mode runs on the panel path (or the simple path for dimensions below 8)
        fdr_plan::Slot* one[1] = {&w};
        int rc = panel_stage_A(p, w, d_img, rows, cols, stride, s);
        if (rc == FDR_OK) rc = panel_stage_B(p, one, 1, s);
        if (rc == FDR_OK) rc = panel_stage_CE(p, w, rows, cols, d_out, out_stride, mm_rows, mm_cols, s);
        return rc;
    }
    {   // E: normalise to [0,1] and crop (fft/fft_serial.cpp:246, serial.cpp:38)
        ScopedPass t(p, s, kPassNormalize);
        if (n_part <= 0 || n_part > p->mm_part_cap) return fail(FDR_ERR_STATE, "fdr_wiener: min/max partial count out of range");
        const bool pp = p->ppar && p->mode == FDR_MODE_PARITY && !p->simple;  // the raw plane is panel-major then
        if (n_part <= 4096) {
            if (pp) FDR_HIP(launch_normalize_panels(w.raw, p->M, p->N, w.mm_part, n_part, nullptr, d_out, rows, cols, out_stride, s));
            else FDR_HIP(launch_normalize(w.raw, p->N, w.mm_part, n_part, nullptr, d_out, rows, cols, out_stride, s));
        } else {  // many partials (reference-shaped path): fold them once in a separate launch
            FDR_HIP(launch_reduce_minmax(w.mm_part, n_part, w.mm, s));
            if (pp) FDR_HIP(launch_normalize_panels(w.raw, p->M, p->N, nullptr, 0, w.mm, d_out, rows, cols, out_stride, s));
            else FDR_HIP(launch_normalize(w.raw, p->N, nullptr, 0, w.mm, d_out, rows, cols, out_stride, s));
        }
    }
    return FDR_OK;
}

int ensure_psf_staging(fdr_plan* p, size_t elems) {
    if (p->psf_cap >= elems) return FDR_OK;
    if (p->psf_dev) { (void)hipFree(p->psf_dev); p->psf_dev = nullptr; p->psf_cap = 0; }
    FDR_HIP(hipMalloc((void**)&p->psf_dev, elems * sizeof(float)));
    p->psf_cap = elems;
    return FDR_OK;
}

}  // namespace

extern "C" {

int fdr_version(void) { return FDR_VERSION; }
const char* fdr_last_error(void) { return g_last_error.c_str(); }

int fdr_device_count(int* count) {
    if (!count) return fail(FDR_ERR_ARG, "fdr_device_count: null");
    FDR_HIP(hipGetDeviceCount(count));
    return FDR_OK;
}

int fdr_next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }  // utils.hpp:27-37
int fdr_is_pow2(int n) { return n > 0 && ((n & (n - 1)) == 0); }           // utils.hpp:50-52

// cv::getOptimalDFTSize as the serial path uses it (fft/fft_serial.cpp:153-154): smallest 2^a 3^b 5^c >= n
int fdr_optimal_dft_size(int n) {
    if (n <= 1) return n < 0 ? -1 : 1;
    for (long long best = -1, p2 = 1; p2 < 2LL * n; p2 *= 2) {
        for (long long p3 = p2; p3 < 2LL * n; p3 *= 3)
            for (long long p5 = p3; p5 < 2LL * n; p5 *= 5)
                if (p5 >= n && (best < 0 || p5 < best)) best = p5;
        if (p2 * 2 >= 2LL * n) return (int)best;
    }
    return -1;
}

static int plan_create_impl(fdr_plan* p, int device, int M, int N, int mode, unsigned flags) {
    p->device = device; p->M = M; p->N = N; p->mode = mode; p->flags = flags;
    const bool pow2 = fdr_is_pow2(M) && fdr_is_pow2(N);
    p->generic = !pow2;  // only reachable with FDR_FLAG_ANY_SIZE: reference-shaped passes, parity arithmetic
    if (p->generic) p->mode = mode = FDR_MODE_PARITY;
    p->logM = fdr_is_pow2(M) ? ilog2(M) : -1;
    p->logN = fdr_is_pow2(N) ? ilog2(N) : -1;
    // dimensions above 8192 (one row no longer fits the LDS): the reference-shaped sequence rows / transpose / rows / transpose
    // with the long row pass (fdr_aux.hip, long_gather_kernel) -- as the serial path, correct at any power of two and slower
    p->simple = p->generic || (flags & FDR_FLAG_SIMPLE_PATH) != 0 || M < 8 || N < 8 || M > (1 << kMaxLdsLog) || N > (1 << kMaxLdsLog);
    p->big = (fdr_is_pow2(M) && M > (1 << kMaxLdsLog)) || (fdr_is_pow2(N) && N > (1 << kMaxLdsLog));
    p->panel = mode == FDR_MODE_FAST && !p->simple;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) p->num_cu = cus;
    }
    size_t P = (size_t)M * N;
    if (p->panel) {  // panel-major buffers: panels of 4 columns, PS elements apart (not a power of two: channel skew)
        p->pstride = (size_t)M * 4 + 16;
        p->half = N >= 32 && (flags & FDR_FLAG_FULL_SPECTRUM) == 0;
        p->npanels = p->half ? N / 8 : N / 4;
        P = (size_t)p->npanels * p->pstride;
    } else if (mode == FDR_MODE_PARITY && !p->simple) {
        // the bit-identical mode keeps the reference's pass order and full complex spectrum, but its intermediate is panel-major
        // too since round 4 (all N/4 panels): the column passes B and D read and write contiguous M x 32-byte tiles instead of
        // 32 bytes of every row (B 169 -> 88, D 67 -> 55 us per 4096^2 image; A and C pay 9 us each for their 32-byte pieces,
        // which they reach through an XCD-aware workgroup order, fdr_rows.hip; the raw real plane is panel-major as well and
        // normalize_panels_kernel turns it back).  381 -> 276 us per 4096^2 image, 1767 -> 1266 at 8192^2.  Same butterflies,
        // same tables, same bits (every parity test compares with ==).
        p->ppar = true;
        p->pstride = (size_t)M * 4 + 16;
        p->npanels = N / 4;
        P = (size_t)p->npanels * p->pstride;
    }
    std::vector<float2> t;
    int rc = FDR_OK;
    if (fdr_is_pow2(N)) {
        build_twiddles(N, mode, false, t); if ((rc = upload(&p->tw_row_f, t)) != FDR_OK) return rc;
        build_twiddles(N, mode, true, t);  if ((rc = upload(&p->tw_row_i, t)) != FDR_OK) return rc;
    } else {
        build_naive_table(N, t); if ((rc = upload(&p->naive_row, t)) != FDR_OK) return rc;
    }
    if (fdr_is_pow2(M)) {
        build_twiddles(M, mode, false, t); if ((rc = upload(&p->tw_col_f, t)) != FDR_OK) return rc;
        build_twiddles(M, mode, true, t);  if ((rc = upload(&p->tw_col_i, t)) != FDR_OK) return rc;
    } else if (M == N) {
        p->naive_col = p->naive_row;
    } else {
        build_naive_table(M, t); if ((rc = upload(&p->naive_col, t)) != FDR_OK) return rc;
    }
    p->tables_only = (flags & FDR_FLAG_TABLES_ONLY) != 0;
    if ((!p->tables_only && (hipMalloc((void**)&p->work, P * sizeof(float2)) != hipSuccess ||
                             hipMalloc((void**)&p->filt, P * sizeof(float2)) != hipSuccess ||
                             hipMalloc((void**)&p->raw, (size_t)M * N * sizeof(float)) != hipSuccess)) ||
        hipMalloc((void**)&p->mm, 2 * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&p->mm_part, (size_t)(p->mm_part_cap = (int)(((size_t)N + 255) / 256 * M + 8192)) * sizeof(float2)) != hipSuccess ||
        (p->simple && !p->tables_only && hipMalloc((void**)&p->work2, P * sizeof(float2)) != hipSuccess))
        return fail(FDR_ERR_ALLOC, "fdr_plan_create: hipMalloc of the plan workspace failed");
    p->ws_elems = P;
    p->slots[0].work = p->work; p->slots[0].work2 = p->work2; p->slots[0].raw = p->raw; p->slots[0].mm = p->mm;
    p->slots[0].mm_part = p->mm_part;
    return FDR_OK;
}

int fdr_plan_create(int device, int M, int N, int mode, unsigned flags, fdr_plan** out) {
    if (!out) return fail(FDR_ERR_ARG, "fdr_plan_create: null out");
    *out = nullptr;
    if (M <= 0 || N <= 0) return fail(FDR_ERR_ARG, "fdr_plan_create: non-positive dimension");
    if (mode != FDR_MODE_PARITY && mode != FDR_MODE_FAST) return fail(FDR_ERR_ARG, "fdr_plan_create: unknown mode");
    if (!fdr_is_pow2(M) || !fdr_is_pow2(N)) {
        if ((flags & FDR_FLAG_ANY_SIZE) == 0)
            return fail(FDR_ERR_NOT_POW2, "fdr_plan_create: M and N must be powers of two (pad first, utils.hpp:40-47) unless FDR_FLAG_ANY_SIZE is set");
        if ((!fdr_is_pow2(M) && M > kMaxNaiveLen) || (!fdr_is_pow2(N) && N > kMaxNaiveLen))
            return fail(FDR_ERR_ARG, "fdr_plan_create: non-power-of-two dimension above 4096 (naive-DFT twiddle table)");
    }
    if (M > (1 << kMaxLongLog) || N > (1 << kMaxLongLog))
        return fail(FDR_ERR_ARG, "fdr_plan_create: dimension above 32768");
    FDR_HIP(hipSetDevice(device));
    // registered behind the first HIP call, i.e. after the HIP runtime's own exit handlers: it runs BEFORE them
    static std::once_flag exit_hook;
    std::call_once(exit_hook, [] { std::atexit([] { g_process_exiting.store(true); }); });
    const auto t0 = std::chrono::steady_clock::now();
    fdr_plan* p = new (std::nothrow) fdr_plan();
    if (!p) return fail(FDR_ERR_ALLOC, "fdr_plan_create: out of host memory");
    const int rc = plan_create_impl(p, device, M, N, mode, flags);
    if (rc != FDR_OK) {
        const std::string msg = g_last_error;  // fdr_plan_destroy does not touch it, but keep the first failure's text
        fdr_plan_destroy(p);
        g_last_error = msg;
        return rc;
    }
    p->phase_ms[FDR_PHASE_ALLOC] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = p;
    return FDR_OK;
}

int fdr_plan_destroy(fdr_plan* p) {
    if (!p) return FDR_OK;
    if (g_process_exiting.load()) {  // static destructors / late atexit handlers: the HIP runtime may be gone; the
        delete p;                    // process's device memory goes with it, only the host side is ours to free
        return FDR_OK;
    }
    (void)hipSetDevice(p->device);
    for (int k = 1; k < fdr_plan::kMaxSlots; ++k) {
        fdr_plan::Slot& w = p->slots[k];
        (void)hipFree(w.work); (void)hipFree(w.work2); (void)hipFree(w.raw); (void)hipFree(w.mm); (void)hipFree(w.mm_part);
    }
    for (int k = 0; k < fdr_plan::kMaxSlots; ++k) {
        if (p->slots[k].stream) (void)hipStreamDestroy(p->slots[k].stream);
        if (p->slots[k].done) (void)hipEventDestroy(p->slots[k].done);
    }
    for (int k = 0; k < 3; ++k) {
        (void)hipFree(p->pipe.d_in[k]); (void)hipFree(p->pipe.d_out[k]);
        if (p->pipe.e_in[k]) (void)hipEventDestroy(p->pipe.e_in[k]);
        if (p->pipe.e_cmp[k]) (void)hipEventDestroy(p->pipe.e_cmp[k]);
        if (p->pipe.e_out[k]) (void)hipEventDestroy(p->pipe.e_out[k]);
    }
    if (p->pipe.s_in) (void)hipStreamDestroy(p->pipe.s_in);
    if (p->pipe.s_cmp) (void)hipStreamDestroy(p->pipe.s_cmp);
    if (p->pipe.s_out) (void)hipStreamDestroy(p->pipe.s_out);
    if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
    if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
    if (p->fork) (void)hipEventDestroy(p->fork);
    (void)hipFree(p->tw_row_f); (void)hipFree(p->tw_row_i); (void)hipFree(p->tw_col_f); (void)hipFree(p->tw_col_i);
    if (p->naive_col != p->naive_row) (void)hipFree(p->naive_col);
    (void)hipFree(p->naive_row);
    for (auto& r : p->phase_pending) { p->timer.pool.push_back(r.a); p->timer.pool.push_back(r.b); }
    p->phase_pending.clear();
    p->timer.destroy();
    (void)hipFree(p->work); (void)hipFree(p->work2); (void)hipFree(p->filt); (void)hipFree(p->raw);
    (void)hipFree(p->psf_dev); (void)hipFree(p->mm); (void)hipFree(p->mm_part);
    (void)hipFree(p->stage_in); (void)hipFree(p->stage_out);
    delete p;
    return FDR_OK;
}

int fdr_plan_dims(const fdr_plan* p, int* M, int* N, int* mode) {
    if (!p) return fail(FDR_ERR_ARG, "fdr_plan_dims: null plan");
    if (M) *M = p->M;
    if (N) *N = p->N;
    if (mode) *mode = p->mode;
    return FDR_OK;
}

int fdr_plan_set_option(fdr_plan* p, int option, long long value) {
    if (!p) return fail(FDR_ERR_ARG, "fdr_plan_set_option: null plan");
    switch (option) {
        case FDR_OPT_BATCH_GRAPH:
            if (value != 0 && value != 1) return fail(FDR_ERR_ARG, "fdr_plan_set_option: FDR_OPT_BATCH_GRAPH takes 0 or 1");
            p->batch_graph = value != 0;
            return FDR_OK;
        case FDR_OPT_CE_CHUNK_MB:
            if (value < 0 || value > (1 << 20)) return fail(FDR_ERR_ARG, "fdr_plan_set_option: FDR_OPT_CE_CHUNK_MB takes 0 .. 1048576");
            p->ce_chunk_bytes = (size_t)value << 20;
            return FDR_OK;
        case FDR_OPT_TWO_SWEEP_NORM:
            if (value != 0 && value != 1) return fail(FDR_ERR_ARG, "fdr_plan_set_option: FDR_OPT_TWO_SWEEP_NORM takes 0 or 1");
            p->two_sweep = value != 0;
            return FDR_OK;
        default:
            return fail(FDR_ERR_ARG, "fdr_plan_set_option: unknown option");
    }
}

int fdr_plan_phase_times(fdr_plan* p, float ms[FDR_N_PHASES], int reset) {
    if (!p || !ms) return fail(FDR_ERR_ARG, "fdr_plan_phase_times: null argument");
    FDR_HIP(hipSetDevice(p->device));
    resolve_phases(p);
    for (int i = 0; i < FDR_N_PHASES; ++i) ms[i] = (float)p->phase_ms[i];
    if (reset)
        for (int i = 0; i < FDR_N_PHASES; ++i) p->phase_ms[i] = 0.0;
    return FDR_OK;
}

int fdr_psf_motion_dev(int device, int size, double angle_deg, float* d_out, void* stream) {
    if (size <= 0 || !d_out) return fail(FDR_ERR_ARG, "fdr_psf_motion_dev: bad argument");
    FDR_HIP(hipSetDevice(device));
    FDR_HIP(launch_psf_motion(size, angle_deg, d_out, (hipStream_t)stream));
    return FDR_OK;
}

int fdr_psf_motion(int size, double angle_deg, float* out_host) {
    if (size <= 0 || !out_host) return fail(FDR_ERR_ARG, "fdr_psf_motion: bad argument");
    float* d = nullptr;
    const size_t bytes = (size_t)size * size * sizeof(float);
    FDR_HIP(hipMalloc((void**)&d, bytes));
    hipError_t e = launch_psf_motion(size, angle_deg, d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_host, d, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    FDR_HIP(e);
    return FDR_OK;
}

int fdr_warp_affine_f32(const float* src_host, int srows, int scols, int sstride, const double M[6], float* dst_host, int drows, int dcols,
                        int dstride) {
    if (!src_host || !dst_host || !M || srows <= 0 || scols <= 0 || sstride < scols || drows <= 0 || dcols <= 0 || dstride < dcols)
        return fail(FDR_ERR_ARG, "fdr_warp_affine_f32: bad argument");
    if (srows > 32767 || scols > 32767 || drows > 32767 || dcols > 32767)
        return fail(FDR_ERR_ARG, "fdr_warp_affine_f32: image dimension above 32767 (cv::warpAffine's short coordinates)");
    float *d_src = nullptr, *d_dst = nullptr;
    const size_t sb = (size_t)scols * sizeof(float), db = (size_t)dcols * sizeof(float);
    FDR_HIP(hipMalloc((void**)&d_src, sb * srows));
    if (hipMalloc((void**)&d_dst, db * drows) != hipSuccess) { (void)hipFree(d_src); return fail(FDR_ERR_ALLOC, "fdr_warp_affine_f32: hipMalloc"); }
    hipError_t e = hipMemcpy2D(d_src, sb, src_host, (size_t)sstride * sizeof(float), sb, srows, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_warp_affine(d_src, srows, scols, scols, M, d_dst, drows, dcols, dcols, nullptr);
    if (e == hipSuccess) e = hipMemcpy2D(dst_host, (size_t)dstride * sizeof(float), d_dst, db, db, drows, hipMemcpyDeviceToHost);
    (void)hipFree(d_src); (void)hipFree(d_dst);
    FDR_HIP(e);
    return FDR_OK;
}

int fdr_set_psf_dev(fdr_plan* p, const float* d_psf, int prows, int pcols, int pstride, float K, void* stream) {
    if (!p || !d_psf) return fail(FDR_ERR_ARG, "fdr_set_psf_dev: null argument");
    FDR_HIP(hipSetDevice(p->device));
    return set_psf_dev_impl(p, d_psf, prows, pcols, pstride, K, (hipStream_t)stream);
}

int fdr_set_psf(fdr_plan* p, const float* psf_host, int prows, int pcols, int pstride, float K) {
    if (!p || !psf_host) return fail(FDR_ERR_ARG, "fdr_set_psf: null argument");
    if (prows <= 0 || pcols <= 0 || pstride < pcols) return fail(FDR_ERR_ARG, "fdr_set_psf: bad PSF shape");
    FDR_HIP(hipSetDevice(p->device));
    int rc = ensure_psf_staging(p, (size_t)prows * pcols);
    if (rc != FDR_OK) return rc;
    {
        ScopedPhase ph(p, FDR_PHASE_H2D, nullptr);
        FDR_HIP(hipMemcpy2D(p->psf_dev, (size_t)pcols * sizeof(float), psf_host, (size_t)pstride * sizeof(float),
                            (size_t)pcols * sizeof(float), prows, hipMemcpyHostToDevice));
    }
    rc = set_psf_dev_impl(p, p->psf_dev, prows, pcols, pcols, K, nullptr);
    if (rc != FDR_OK) return rc;
    FDR_HIP(hipStreamSynchronize(nullptr));
    resolve_phases(p);
    return FDR_OK;
}

int fdr_set_psf_motion(fdr_plan* p, int size, double angle_deg, float K, void* stream) {
    if (!p || size <= 0) return fail(FDR_ERR_ARG, "fdr_set_psf_motion: bad argument");
    FDR_HIP(hipSetDevice(p->device));
    int rc = ensure_psf_staging(p, (size_t)size * size);
    if (rc != FDR_OK) return rc;
    FDR_HIP(launch_psf_motion(size, angle_deg, p->psf_dev, (hipStream_t)stream));
    return set_psf_dev_impl(p, p->psf_dev, size, size, size, K, (hipStream_t)stream);
}

int fdr_plan_filter_bytes(const fdr_plan* p, size_t* bytes) {
    if (!p || !bytes) return fail(FDR_ERR_ARG, "fdr_plan_filter_bytes: null argument");
    if (p->tables_only) return fail(FDR_ERR_STATE, "fdr_plan_filter_bytes: plan was created with FDR_FLAG_TABLES_ONLY");
    *bytes = p->ws_elems * sizeof(float2);
    return FDR_OK;
}

int fdr_plan_export_filter_dev(fdr_plan* p, void* d_dst, size_t bytes, void* stream) {
    if (!p || !d_dst) return fail(FDR_ERR_ARG, "fdr_plan_export_filter_dev: null argument");
    if (p->tables_only) return fail(FDR_ERR_STATE, "fdr_plan_export_filter_dev: plan was created with FDR_FLAG_TABLES_ONLY");
    if (!p->have_psf) return fail(FDR_ERR_STATE, "fdr_plan_export_filter_dev: no PSF set on this plan");
    if (bytes != p->ws_elems * sizeof(float2)) return fail(FDR_ERR_ARG, "fdr_plan_export_filter_dev: size differs from fdr_plan_filter_bytes");
    FDR_HIP(hipSetDevice(p->device));
    FDR_HIP(hipMemcpyAsync(d_dst, p->filt, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FDR_OK;
}

int fdr_plan_import_filter_dev(fdr_plan* p, const void* d_src, size_t bytes, float K, void* stream) {
    if (!p || !d_src) return fail(FDR_ERR_ARG, "fdr_plan_import_filter_dev: null argument");
    if (p->tables_only) return fail(FDR_ERR_STATE, "fdr_plan_import_filter_dev: plan was created with FDR_FLAG_TABLES_ONLY");
    if (bytes != p->ws_elems * sizeof(float2)) return fail(FDR_ERR_ARG, "fdr_plan_import_filter_dev: size differs from fdr_plan_filter_bytes");
    FDR_HIP(hipSetDevice(p->device));
    ScopedPhase phase(p, FDR_PHASE_PRE, (hipStream_t)stream);
    FDR_HIP(hipMemcpyAsync(p->filt, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    p->K = K;
    p->have_psf = true;
    return FDR_OK;
}

int fdr_wiener_f32_dev(fdr_plan* p, const float* d_img, int rows, int cols, int stride, float* d_out, int out_stride,
                       int norm_area, void* stream) {
    if (!p) return fail(FDR_ERR_ARG, "fdr_wiener_f32_dev: null plan");
    FDR_HIP(hipSetDevice(p->device));
    return wiener_dev_impl(p, p->slots[0], d_img, rows, cols, stride, d_out, out_stride, norm_area, (hipStream_t)stream);
}

}  // extern "C"
namespace {
// fork, every pass of every group, join -- all relative to `us` (the caller's stream, or the capturing stream)
int batch_enqueue(fdr_plan* p, const float* d_imgs, size_t img_pitch, int count, int rows, int cols, int stride, float* d_out,
                  size_t out_pitch, int out_stride, int norm_area, hipStream_t us);
}
extern "C" {

int fdr_wiener_batch_f32_dev(fdr_plan* p, const float* d_imgs, size_t img_pitch, int count, int rows, int cols, int stride,
                             float* d_out, size_t out_pitch, int out_stride, int norm_area, void* stream) {
    if (!p) return fail(FDR_ERR_ARG, "fdr_wiener_batch_f32_dev: null plan");
    if (count < 0) return fail(FDR_ERR_ARG, "fdr_wiener_batch_f32_dev: negative count");
    if (count == 0) return FDR_OK;
    FDR_HIP(hipSetDevice(p->device));
    hipStream_t us = (hipStream_t)stream;
    int rc = check_image_args(p, d_imgs, rows, cols, stride, d_out, out_stride);
    if (rc != FDR_OK) return rc;
    // graph replay: launch-bound batches (small images) pay one graph launch instead of 4 kernel launches per group.
    // Not with per-kernel profiling (host-side event pairs).
    if (p->batch_graph && p->panel && !p->timer.enabled) {
        const fdr_plan::GraphKey key{d_imgs, d_out, img_pitch, out_pitch, count, rows, cols, stride, out_stride, norm_area, p->nstreams, p->group,
                                     p->two_sweep, p->K, p->ce_chunk_bytes};
        if (!(p->graph_exec && key == p->graph_key)) {
            if (p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; }
            if (!p->cap_stream) FDR_HIP(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
            FDR_HIP(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal));
            rc = batch_enqueue(p, d_imgs, img_pitch, count, rows, cols, stride, d_out, out_pitch, out_stride, norm_area, p->cap_stream);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(p->cap_stream, &g);  // (always: the stream has to leave capture mode)
            if (rc != FDR_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
            FDR_HIP(e);
            const hipError_t ei = hipGraphInstantiate(&p->graph_exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ei != hipSuccess) { p->graph_exec = nullptr; FDR_HIP(ei); }
            p->graph_key = key;
        }
        FDR_HIP(hipGraphLaunch(p->graph_exec, us));
        return FDR_OK;
    }
    return batch_enqueue(p, d_imgs, img_pitch, count, rows, cols, stride, d_out, out_pitch, out_stride, norm_area, us);
}

}  // extern "C"
namespace {
int batch_enqueue(fdr_plan* p, const float* d_imgs, size_t img_pitch, int count, int rows, int cols, int stride, float* d_out,
                  size_t out_pitch, int out_stride, int norm_area, hipStream_t us) {
    int rc = FDR_OK;
    const int mm_rows = norm_area == FDR_NORM_PADDED ? p->M : rows;
    const int mm_cols = norm_area == FDR_NORM_PADDED ? p->N : cols;
    const int group = p->panel ? p->group : 1;
    // per-kernel profiling wants un-overlapped durations: keep everything on the caller's stream then
    const int ns = (p->timer.enabled || count <= group) ? 1 : p->nstreams;
    if (ns > 1) {  // fork: internal streams wait for everything queued so far on the caller's stream
        FDR_HIP(hipEventRecord(p->fork, us));
        for (int k = 0; k < ns; ++k) FDR_HIP(hipStreamWaitEvent(p->slots[k * group].stream, p->fork, 0));
    }
    int chunk = 0;
    for (int i0 = 0; i0 < count && rc == FDR_OK; i0 += group, ++chunk) {
        const int n = count - i0 < group ? count - i0 : group;
        const int sidx = chunk % ns;
        fdr_plan::Slot* ws[fdr_plan::kMaxSlots];
        for (int k = 0; k < n; ++k) ws[k] = &p->slots[sidx * group + k];
        hipStream_t s = ns > 1 ? p->slots[sidx * group].stream : us;
        if (!p->panel) {
            rc = wiener_dev_impl(p, *ws[0], d_imgs + (size_t)i0 * img_pitch, rows, cols, stride, d_out + (size_t)i0 * out_pitch,
                                 out_stride, norm_area, s);
            continue;
        }
        if (n > 1 && can_batch_rows(p)) {  // every pass once for the whole group
            const float* ins[kMaxGroup]; float* outs[kMaxGroup];
            for (int k = 0; k < n; ++k) { ins[k] = d_imgs + (size_t)(i0 + k) * img_pitch; outs[k] = d_out + (size_t)(i0 + k) * out_pitch; }
            rc = panel_stage_A_batch(p, ws, n, ins, rows, cols, stride, s);
            if (rc == FDR_OK) rc = panel_stage_B(p, ws, n, s);
            // The two inverse row passes (C1: extremes; C2: the same transform again, normalised) go in CHUNKS of the group
            // when the batch alternates over two or more streams: C1 is the pass with exposed compute, and in launches of half
            // the size it interleaves better with the memory-bound passes of the other stream's group.  Measured at 4096^2,
            // 2 streams x 4 images, alternating runs on one box: chunks of 4 / 2 / 1 images 89.1 / 87.9 / 87.5 us per image;
            // with ONE stream the chunks only make the launches smaller (91.5 -> 93.1 us), at 2048^2 too (21.3 -> 25.3 us), and
            // passes A / B' lose in chunks at any size (89.3 -> 91.1 / 93.4 us).  Hence: >= 2 streams, and a chunk holds at least
            // ce_chunk_bytes of spectrum (FDR_OPT_CE_CHUNK_MB, default 160 MiB: pairs at 4096^2, the whole group below, no
            // split where one image alone is larger).  (Not an Infinity-Cache effect, although the 256 MiB suggest it: the
            // passes' own durations get LONGER in chunks, C1 14.0 -> 16.4 us per image; the gain is in the overlap.)
            int chunk = n;
            if (ns > 1) {
                const size_t spec_bytes = p->ws_elems * sizeof(float2);
                if (p->ce_chunk_bytes > 0 && spec_bytes <= p->ce_chunk_bytes) {
                    const size_t c = p->ce_chunk_bytes / spec_bytes;
                    if (c < (size_t)n) chunk = (int)c;
                }
            }
            for (int k0 = 0; k0 < n && rc == FDR_OK; k0 += chunk) {
                const int m = n - k0 < chunk ? n - k0 : chunk;
                if (m == 1) rc = panel_stage_CE(p, *ws[k0], rows, cols, outs[k0], out_stride, mm_rows, mm_cols, s);
                else rc = panel_stage_CE_batch(p, ws + k0, m, rows, cols, outs + k0, out_stride, mm_rows, mm_cols, s);
            }
            continue;
        }
        for (int k = 0; k < n && rc == FDR_OK; ++k)
            rc = panel_stage_A(p, *ws[k], d_imgs + (size_t)(i0 + k) * img_pitch, rows, cols, stride, s);
        if (rc == FDR_OK) rc = panel_stage_B(p, ws, n, s);
        for (int k = 0; k < n && rc == FDR_OK; ++k)
            rc = panel_stage_CE(p, *ws[k], rows, cols, d_out + (size_t)(i0 + k) * out_pitch, out_stride, mm_rows, mm_cols, s);
    }
    if (ns > 1) {  // join -- also after an error: the caller's stream continues only after every internal stream has
                   // drained, so work already queued there cannot still be writing d_out when the caller goes on
        const std::string first_error = g_last_error;
        for (int k = 0; k < ns; ++k) {
            hipError_t e = hipEventRecord(p->slots[k * group].done, p->slots[k * group].stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(us, p->slots[k * group].done, 0);
            if (e != hipSuccess) {  // cannot order the streams: drain them on the host instead
                (void)hipStreamSynchronize(p->slots[k * group].stream);
                if (rc == FDR_OK) rc = fail(FDR_ERR_HIP, std::string("fdr_wiener_batch_f32_dev: join failed: ") + hipGetErrorString(e));
            }
        }
        if (rc != FDR_OK && !first_error.empty() && first_error != g_last_error && rc != FDR_ERR_HIP) g_last_error = first_error;
    }
    return rc;
}
}  // namespace
extern "C" {

int fdr_plan_set_batching(fdr_plan* p, int nstreams, int group) {
    if (!p) return fail(FDR_ERR_ARG, "fdr_plan_set_batching: null plan");
    if (nstreams < 1 || group < 1 || group > kMaxGroup || nstreams * group > fdr_plan::kMaxSlots)
        return fail(FDR_ERR_ARG, "fdr_plan_set_batching: need 1 <= group <= 8 and nstreams * group <= 16");
    FDR_HIP(hipSetDevice(p->device));
    if (!p->fork) FDR_HIP(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming));
    const int nslots = nstreams * group;
    for (int k = 0; k < nslots; ++k) {
        fdr_plan::Slot& w = p->slots[k];
        if (!w.stream) FDR_HIP(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        if (!w.done) FDR_HIP(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
        if (k > 0 && !w.work) {
            if (hipMalloc((void**)&w.work, p->ws_elems * sizeof(float2)) != hipSuccess ||
                hipMalloc((void**)&w.raw, (size_t)p->M * p->N * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&w.mm, 2 * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&w.mm_part, (size_t)p->mm_part_cap * sizeof(float2)) != hipSuccess ||
                (p->simple && hipMalloc((void**)&w.work2, p->ws_elems * sizeof(float2)) != hipSuccess))
                return fail(FDR_ERR_ALLOC, "fdr_plan_set_batching: hipMalloc of an extra workspace failed");
        }
    }
    p->nslots = nslots; p->nstreams = nstreams; p->group = group;
    return FDR_OK;
}

int fdr_plan_set_concurrency(fdr_plan* p, int nstreams) { return fdr_plan_set_batching(p, nstreams, 1); }

int fdr_wiener_f32(fdr_plan* p, const float* img_host, int rows, int cols, int stride, float* out_host, int out_stride,
                   int norm_area) {
    if (!p || !img_host || !out_host) return fail(FDR_ERR_ARG, "fdr_wiener_f32: null argument");
    if (rows <= 0 || cols <= 0 || rows > p->M || cols > p->N || stride < cols || out_stride < cols)
        return fail(FDR_ERR_ARG, "fdr_wiener_f32: image shape does not fit the plan");
    FDR_HIP(hipSetDevice(p->device));
    // device staging of the image and of the result: owned by the plan and kept between calls (the per-channel loop
    // of the drivers calls this three times; the reference's _optimized version hoists its buffers the same way,
    // fft/fft_gpu.cu:304-322)
    const size_t bytes = (size_t)rows * cols * sizeof(float);
    if (p->stage_cap < bytes) {
        (void)hipFree(p->stage_in); (void)hipFree(p->stage_out);
        p->stage_in = p->stage_out = nullptr; p->stage_cap = 0;
        const size_t cap = (size_t)p->M * p->N * sizeof(float);
        if (hipMalloc((void**)&p->stage_in, cap) != hipSuccess || hipMalloc((void**)&p->stage_out, cap) != hipSuccess) {
            (void)hipFree(p->stage_in); p->stage_in = nullptr;
            return fail(FDR_ERR_ALLOC, "fdr_wiener_f32: hipMalloc of the staging buffers failed");
        }
        p->stage_cap = cap;
    }
    float *d_in = p->stage_in, *d_out = p->stage_out;
    hipError_t e;
    {
        ScopedPhase ph(p, FDR_PHASE_H2D, nullptr);
        // (a dense image is ONE linear copy: the 2-D form of a pageable buffer goes row by row -- measured 2.9 ms against
        // 0.8 ms for the 8 MB channel of a padded 1024 x 2048 picture)
        if (stride == cols) e = hipMemcpy(d_in, img_host, bytes, hipMemcpyHostToDevice);
        else e = hipMemcpy2D(d_in, (size_t)cols * sizeof(float), img_host, (size_t)stride * sizeof(float),
                             (size_t)cols * sizeof(float), rows, hipMemcpyHostToDevice);
    }
    int rc = FDR_OK;
    if (e == hipSuccess) {
        ScopedPhase ph(p, FDR_PHASE_COMPUTE, nullptr);
        rc = wiener_dev_impl(p, p->slots[0], d_in, rows, cols, cols, d_out, cols, norm_area, nullptr);
    }
    if (e == hipSuccess && rc == FDR_OK) {
        ScopedPhase ph(p, FDR_PHASE_D2H, nullptr);
        if (out_stride == cols) e = hipMemcpy(out_host, d_out, bytes, hipMemcpyDeviceToHost);
        else e = hipMemcpy2D(out_host, (size_t)out_stride * sizeof(float), d_out, (size_t)cols * sizeof(float),
                             (size_t)cols * sizeof(float), rows, hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess) resolve_phases(p);
    if (rc != FDR_OK) return rc;
    FDR_HIP(e);
    return FDR_OK;
}

// ---- host-pointer batch: H2D, restore, D2H of consecutive images overlap on three streams ----------------------
// (the pipeline fft/fft_gpu.cu:306-350,372-385 sets out to build with pinned staging buffers and cudaMemcpyAsync)
int fdr_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return fail(FDR_ERR_ARG, "fdr_host_alloc: bad argument");
    FDR_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return FDR_OK;
}
int fdr_host_free(void* p) {
    if (p) FDR_HIP(hipHostFree(p));
    return FDR_OK;
}

int fdr_wiener_batch_f32(fdr_plan* p, const float* imgs_host, size_t img_pitch, int count, int rows, int cols, int stride,
                         float* out_host, size_t out_pitch, int out_stride, int norm_area) {
    if (!p || !imgs_host || !out_host) return fail(FDR_ERR_ARG, "fdr_wiener_batch_f32: null argument");
    if (count < 0) return fail(FDR_ERR_ARG, "fdr_wiener_batch_f32: negative count");
    if (count == 0) return FDR_OK;
    std::vector<const float*> ins((size_t)count);
    std::vector<float*> outs((size_t)count);
    for (int i = 0; i < count; ++i) { ins[i] = imgs_host + (size_t)i * img_pitch; outs[i] = out_host + (size_t)i * out_pitch; }
    return fdr_wiener_batch_ptrs_f32(p, ins.data(), outs.data(), count, rows, cols, stride, out_stride, norm_area);
}

int fdr_wiener_batch_ptrs_f32(fdr_plan* p, const float* const* imgs_host, float* const* outs_host, int count, int rows, int cols,
                              int stride, int out_stride, int norm_area) {
    if (!p || !imgs_host || !outs_host) return fail(FDR_ERR_ARG, "fdr_wiener_batch_ptrs_f32: null argument");
    if (count < 0) return fail(FDR_ERR_ARG, "fdr_wiener_batch_ptrs_f32: negative count");
    if (count == 0) return FDR_OK;
    for (int i = 0; i < count; ++i)
        if (!imgs_host[i] || !outs_host[i]) return fail(FDR_ERR_ARG, "fdr_wiener_batch_ptrs_f32: null image pointer");
    if (!p->have_psf) return fail(FDR_ERR_STATE, "fdr_wiener: no PSF set on this plan (call fdr_set_psf* first)");
    if (rows <= 0 || cols <= 0 || rows > p->M || cols > p->N || stride < cols || out_stride < cols)
        return fail(FDR_ERR_ARG, "fdr_wiener_batch_f32: image shape does not fit the plan");
    FDR_HIP(hipSetDevice(p->device));
    // Three images in flight: one arriving, one being restored, one leaving, each on its own stream.  Pinned buffers
    // (fdr_host_alloc) are read / written by DMA and all three stages overlap (measured 1.84 ms per 4096^2 image,
    // 36 GB/s each way at once); with pageable buffers the runtime stages every copy itself and the copy calls
    // block, which leaves the synchronous rate (8 ms) -- an own staging ring with one memcpy thread was slower.
    constexpr int D = 3;
    const size_t bytes = (size_t)rows * cols * sizeof(float), rowb = (size_t)cols * sizeof(float);
    // streams, events and device staging live in the plan (created on first use, sized for the plan's M x N)
    fdr_plan::HostPipe& hp = p->pipe;
    if (!hp.ready) {  // built into locals and committed only when every stream and event exists (failure-atomic)
        hipStream_t st[3] = {nullptr, nullptr, nullptr};
        hipEvent_t ev[3 * D] = {};
        hipError_t ce = hipSuccess;
        for (int k = 0; k < 3 && ce == hipSuccess; ++k) ce = hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking);
        for (int k = 0; k < 3 * D && ce == hipSuccess; ++k) ce = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
        if (ce != hipSuccess) {
            for (int k = 0; k < 3; ++k) if (st[k]) (void)hipStreamDestroy(st[k]);
            for (int k = 0; k < 3 * D; ++k) if (ev[k]) (void)hipEventDestroy(ev[k]);
            FDR_HIP(ce);
        }
        hp.s_in = st[0]; hp.s_cmp = st[1]; hp.s_out = st[2];
        for (int k = 0; k < D; ++k) { hp.e_in[k] = ev[3 * k]; hp.e_cmp[k] = ev[3 * k + 1]; hp.e_out[k] = ev[3 * k + 2]; }
        hp.ready = true;
    }
    if (hp.cap < bytes) {
        const size_t cap = (size_t)p->M * p->N * sizeof(float);
        for (int k = 0; k < D; ++k) { (void)hipFree(hp.d_in[k]); (void)hipFree(hp.d_out[k]); hp.d_in[k] = hp.d_out[k] = nullptr; }
        hp.cap = 0;
        for (int k = 0; k < D; ++k)
            if (hipMalloc((void**)&hp.d_in[k], cap) != hipSuccess || hipMalloc((void**)&hp.d_out[k], cap) != hipSuccess)
                return fail(FDR_ERR_ALLOC, "fdr_wiener_batch_f32: hipMalloc of the staging buffers failed");
        hp.cap = cap;
    }
    float* const* d_in = hp.d_in;
    float* const* d_out = hp.d_out;
    hipStream_t s_in = hp.s_in, s_cmp = hp.s_cmp, s_out = hp.s_out;
    hipEvent_t *e_in = hp.e_in, *e_cmp = hp.e_cmp, *e_out = hp.e_out;
    int rc = FDR_OK;
    hipError_t e = hipSuccess;
    auto bad = [&](hipError_t err) { e = err; return err != hipSuccess; };
    for (int i = 0; i < count; ++i) {
        const int k = i % D;
        const float* src = imgs_host[i];
        float* dst = outs_host[i];
        // slot k is free again once image i-D has left the device (its D2H read d_out[k], its kernels read d_in[k])
        if (i >= D && bad(hipStreamWaitEvent(s_in, e_out[k], 0))) break;
        {
            ScopedPhase ph(p, FDR_PHASE_H2D, s_in);
            if (bad(stride == cols ? hipMemcpyAsync(d_in[k], src, bytes, hipMemcpyHostToDevice, s_in)
                                   : hipMemcpy2DAsync(d_in[k], rowb, src, (size_t)stride * sizeof(float), rowb, rows, hipMemcpyHostToDevice, s_in))) break;
        }
        if (bad(hipEventRecord(e_in[k], s_in)) || bad(hipStreamWaitEvent(s_cmp, e_in[k], 0))) break;
        {
            ScopedPhase ph(p, FDR_PHASE_COMPUTE, s_cmp);
            rc = wiener_dev_impl(p, p->slots[0], d_in[k], rows, cols, cols, d_out[k], cols, norm_area, s_cmp);
        }
        if (rc != FDR_OK) break;
        if (bad(hipEventRecord(e_cmp[k], s_cmp)) || bad(hipStreamWaitEvent(s_out, e_cmp[k], 0))) break;
        {
            ScopedPhase ph(p, FDR_PHASE_D2H, s_out);
            if (bad(out_stride == cols ? hipMemcpyAsync(dst, d_out[k], bytes, hipMemcpyDeviceToHost, s_out)
                                       : hipMemcpy2DAsync(dst, (size_t)out_stride * sizeof(float), d_out[k], rowb, rowb, rows, hipMemcpyDeviceToHost, s_out))) break;
        }
        if (bad(hipEventRecord(e_out[k], s_out))) break;
    }
    // everything queued must have left the device before the call returns (also on the error paths: the buffers are reused)
    (void)hipStreamSynchronize(s_in);
    (void)hipStreamSynchronize(s_cmp);
    { hipError_t es = hipStreamSynchronize(s_out); if (e == hipSuccess) e = es; }
    resolve_phases(p);
    if (rc != FDR_OK) return rc;
    FDR_HIP(e);
    return FDR_OK;
}

int fdr_fft2d_c2c_dev(fdr_plan* p, float* d_data, int inverse, void* stream) {
    if (!p || !d_data) return fail(FDR_ERR_ARG, "fdr_fft2d_c2c_dev: null argument");
    if (p->tables_only) return fail(FDR_ERR_STATE, "fdr_fft2d_c2c_dev: plan was created with FDR_FLAG_TABLES_ONLY");
    FDR_HIP(hipSetDevice(p->device));
    return dft2d_dev(p, reinterpret_cast<float2*>(d_data), p->work2, inverse != 0, (hipStream_t)stream);
}

int fdr_fft2d_c2c(fdr_plan* p, float* data_host, int inverse) {
    if (!p || !data_host) return fail(FDR_ERR_ARG, "fdr_fft2d_c2c: null argument");
    if (p->tables_only) return fail(FDR_ERR_STATE, "fdr_fft2d_c2c: plan was created with FDR_FLAG_TABLES_ONLY");
    FDR_HIP(hipSetDevice(p->device));
    const size_t elems = (size_t)p->M * p->N, bytes = elems * sizeof(float2);
    // p->work is free between operator calls and serves as the staging buffer -- unless the plan keeps only the
    // half spectrum there (fast panel mode: about M*N/2 elements), where a full-size buffer is allocated for the call
    float2* buf = p->work;
    const bool own = p->ws_elems < elems;
    if (own) FDR_HIP(hipMalloc((void**)&buf, bytes));
    hipError_t e = hipMemcpy(buf, data_host, bytes, hipMemcpyHostToDevice);
    int rc = FDR_OK;
    if (e == hipSuccess) rc = dft2d_dev(p, buf, p->work2, inverse != 0, nullptr);
    if (e == hipSuccess && rc == FDR_OK) e = hipMemcpy(data_host, buf, bytes, hipMemcpyDeviceToHost);
    if (own) (void)hipFree(buf);
    if (rc != FDR_OK) return rc;
    FDR_HIP(e);
    return FDR_OK;
}

int fdr_dft_naive_c2c(float* data_host, int n, int inverse) {
    if (!data_host || n < 0) return fail(FDR_ERR_ARG, "fdr_dft_naive_c2c: bad argument");
    if (n <= 1) return FDR_OK;  // fft/fft_serial.cpp:74
    float2 *a = nullptr, *b = nullptr, *tab = nullptr;
    const size_t bytes = (size_t)n * sizeof(float2);
    FDR_HIP(hipMalloc((void**)&a, bytes));
    if (hipMalloc((void**)&b, bytes) != hipSuccess) { (void)hipFree(a); return fail(FDR_ERR_ALLOC, "fdr_dft_naive_c2c: hipMalloc"); }
    hipError_t e = hipMemcpy(a, data_host, bytes, hipMemcpyHostToDevice);
    if (n <= kMaxNaiveLen) {  // host-generated twiddles: the bits of the serial path's cosf / sinf
        std::vector<float2> t;
        build_naive_table(n, t);
        if (e == hipSuccess) e = hipMalloc((void**)&tab, t.size() * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(tab, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = launch_dft_naive_rows(a, b, 1, n, tab, inverse, nullptr);
    } else if (e == hipSuccess) {
        e = launch_dft_naive(a, b, n, inverse, nullptr);
    }
    if (e == hipSuccess) e = hipMemcpy(data_host, b, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(tab);
    FDR_HIP(e);
    return FDR_OK;
}

int fdr_fft1d_c2c(float* data_host, int n, int inverse, int mode) {
    if (!data_host || n < 0) return fail(FDR_ERR_ARG, "fdr_fft1d_c2c: bad argument");
    if (mode != FDR_MODE_PARITY && mode != FDR_MODE_FAST) return fail(FDR_ERR_ARG, "fdr_fft1d_c2c: unknown mode");
    if (n <= 1) return FDR_OK;                                        // fft/fft_serial.cpp:43
    if (!fdr_is_pow2(n)) return fdr_dft_naive_c2c(data_host, n, inverse);  // fft/fft_serial.cpp:100-101
    if (n > (1 << kMaxLongLog)) return fail(FDR_ERR_ARG, "fdr_fft1d_c2c: power-of-two length above 32768");
    std::vector<float2> t;
    if (n > (1 << kMaxLdsLog)) {  // 8192-point blocks + global stages (fdr_aux.hip): both tables of the mode
        std::vector<float2> ti;
        build_twiddles(n, mode, false, t);
        build_twiddles(n, mode, true, ti);
        float2 *twf = nullptr, *twi = nullptr, *d = nullptr, *tmp = nullptr;
        const size_t bytes = (size_t)n * sizeof(float2), tb = t.size() * sizeof(float2);
        hipError_t e = hipMalloc((void**)&twf, tb);
        if (e == hipSuccess) e = hipMalloc((void**)&twi, tb);
        if (e == hipSuccess) e = hipMalloc((void**)&d, bytes);
        if (e == hipSuccess) e = hipMalloc((void**)&tmp, bytes);
        if (e == hipSuccess) e = hipMemcpy(twf, t.data(), tb, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(twi, ti.data(), tb, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d, data_host, bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = long_rows_dev(d, tmp, 1, n, ilog2(n), mode, inverse != 0, twf, twi, nullptr);
        if (e == hipSuccess) e = hipMemcpy(data_host, d, bytes, hipMemcpyDeviceToHost);
        (void)hipFree(twf); (void)hipFree(twi); (void)hipFree(d); (void)hipFree(tmp);
        FDR_HIP(e);
        return FDR_OK;
    }
    // the register kernels take the forward table in fast mode (they conjugate it); the simple kernel
    // (n < 8) and parity mode take the table of the requested direction
    build_twiddles(n, mode, (mode == FDR_MODE_FAST && n >= 8) ? false : (inverse != 0), t);
    float2 *tw = nullptr, *d = nullptr;
    const size_t bytes = (size_t)n * sizeof(float2);
    FDR_HIP(hipMalloc((void**)&tw, t.size() * sizeof(float2)));
    if (hipMalloc((void**)&d, bytes) != hipSuccess) { (void)hipFree(tw); return fail(FDR_ERR_ALLOC, "fdr_fft1d_c2c: hipMalloc"); }
    hipError_t e = hipMemcpy(tw, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d, data_host, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int logn = ilog2(n);
        if (n >= 8) {
            RowArgs ra{};
            ra.src_c = d; ra.dst_c = d; ra.M = 1;
            e = launch_rows(logn, mode, ROW_IN_COMPLEX, ROW_OUT_COMPLEX, inverse != 0, ra, tw, nullptr);
        } else {
            e = launch_simple_rows(d, 1, n, logn, tw, mode, nullptr);
        }
    }
    if (e == hipSuccess) e = hipMemcpy(data_host, d, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(tw); (void)hipFree(d);
    FDR_HIP(e);
    return FDR_OK;
}

int fdr_white_balance_u8_dev(int device, const float* const d_orig_bgr[3], const float* const d_restored_bgr[3], int rows,
                             int cols, int stride, unsigned char* d_out_bgr8, int out_stride_bytes, void* stream) {
    if (!d_orig_bgr || !d_restored_bgr || !d_out_bgr8) return fail(FDR_ERR_ARG, "fdr_white_balance_u8_dev: null argument");
    if (rows <= 0 || cols <= 0 || stride < cols || out_stride_bytes < 3 * cols) return fail(FDR_ERR_ARG, "fdr_white_balance_u8_dev: bad shape");
    ColorArgs a{};
    for (int c = 0; c < 3; ++c) {
        if (!d_orig_bgr[c] || !d_restored_bgr[c]) return fail(FDR_ERR_ARG, "fdr_white_balance_u8_dev: null plane");
        a.orig[c] = d_orig_bgr[c]; a.rest[c] = d_restored_bgr[c];
    }
    a.rows = rows; a.cols = cols; a.stride = stride; a.out = d_out_bgr8; a.out_stride = out_stride_bytes;
    FDR_HIP(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    double2* part = nullptr;
    FDR_HIP(hipMallocAsync((void**)&part, (size_t)color_partials(rows, cols) * sizeof(double2), s));
    hipError_t e = launch_color_epilogue(a, part, s);
    (void)hipFreeAsync(part, s);
    FDR_HIP(e);
    return FDR_OK;
}

int fdr_white_balance_u8(int device, const float* const orig_bgr[3], const float* const restored_bgr[3], int rows, int cols,
                         int stride, unsigned char* out_bgr8, int out_stride_bytes) {
    if (!orig_bgr || !restored_bgr || !out_bgr8) return fail(FDR_ERR_ARG, "fdr_white_balance_u8: null argument");
    if (rows <= 0 || cols <= 0 || stride < cols || out_stride_bytes < 3 * cols) return fail(FDR_ERR_ARG, "fdr_white_balance_u8: bad shape");
    FDR_HIP(hipSetDevice(device));
    const size_t plane = (size_t)rows * cols * sizeof(float), rowb = (size_t)cols * sizeof(float);
    float* d = nullptr; unsigned char* d_out = nullptr;
    FDR_HIP(hipMalloc((void**)&d, 6 * plane));
    if (hipMalloc((void**)&d_out, (size_t)rows * cols * 3) != hipSuccess) { (void)hipFree(d); return fail(FDR_ERR_ALLOC, "fdr_white_balance_u8: hipMalloc"); }
    const float* dp[6];
    hipError_t e = hipSuccess;
    for (int c = 0; c < 6 && e == hipSuccess; ++c) {
        const float* src = c < 3 ? orig_bgr[c] : restored_bgr[c - 3];
        if (!src) { (void)hipFree(d); (void)hipFree(d_out); return fail(FDR_ERR_ARG, "fdr_white_balance_u8: null plane"); }
        float* dst = d + (size_t)c * rows * cols;
        dp[c] = dst;
        e = hipMemcpy2D(dst, rowb, src, (size_t)stride * sizeof(float), rowb, rows, hipMemcpyHostToDevice);
    }
    int rc = FDR_OK;
    if (e == hipSuccess) rc = fdr_white_balance_u8_dev(device, dp, dp + 3, rows, cols, cols, d_out, 3 * cols, nullptr);
    if (e == hipSuccess && rc == FDR_OK)
        e = hipMemcpy2D(out_bgr8, (size_t)out_stride_bytes, d_out, (size_t)cols * 3, (size_t)cols * 3, rows, hipMemcpyDeviceToHost);
    (void)hipFree(d); (void)hipFree(d_out);
    if (rc != FDR_OK) return rc;
    FDR_HIP(e);
    return FDR_OK;
}

int fdr_synth_image_dev(int device, uint64_t seed, uint64_t first_index, size_t count, float* d_out, void* stream) {
    if (!d_out && count) return fail(FDR_ERR_ARG, "fdr_synth_image_dev: null output");
    FDR_HIP(hipSetDevice(device));
    FDR_HIP(launch_synth(seed, first_index, count, d_out, (hipStream_t)stream));
    return FDR_OK;
}

// ---- slab primitives of the single-image multi-GPU mode (see fdr.h) ----
int fdr_slab_pad_dev(const float* d_src, int valid_rows, int valid_cols, int src_stride, float* d_dst, int rows, int N, void* stream) {
    if (!d_dst || rows < 0 || N <= 0 || valid_rows < 0 || valid_cols < 0 || valid_rows > rows || valid_cols > N || (valid_rows && valid_cols && (!d_src || src_stride < valid_cols)))
        return fail(FDR_ERR_ARG, "fdr_slab_pad_dev: bad argument");
    if (rows == 0) return FDR_OK;
    FDR_HIP(launch_pad_real_to_complex(d_src ? d_src : reinterpret_cast<const float*>(d_dst), valid_rows, valid_cols, src_stride > 0 ? src_stride : 1,
                                       reinterpret_cast<float2*>(d_dst), rows, N, (hipStream_t)stream));
    return FDR_OK;
}

int fdr_slab_rows_fft_dev(fdr_plan* p, float* d_complex, int rows, int dim, int inverse, void* stream) {
    if (!p || !d_complex || rows < 0 || (dim != 0 && dim != 1)) return fail(FDR_ERR_ARG, "fdr_slab_rows_fft_dev: bad argument");
    if (rows == 0) return FDR_OK;
    FDR_HIP(hipSetDevice(p->device));
    const int L = dim == 0 ? p->N : p->M, logl = dim == 0 ? p->logN : p->logM;
    const float2* twf = dim == 0 ? p->tw_row_f : p->tw_col_f;
    const float2* twi = dim == 0 ? p->tw_row_i : p->tw_col_i;
    const float2* naive = dim == 0 ? p->naive_row : p->naive_col;
    hipStream_t s = (hipStream_t)stream;
    if (naive) return fail(FDR_ERR_ARG, "fdr_slab_rows_fft_dev: power-of-two dimensions only");
    float2* d = reinterpret_cast<float2*>(d_complex);
    if (logl > kMaxLdsLog) {  // more than 8192 points: 8192-point blocks + global radix-2 stages (fdr_aux.hip); stream-ordered scratch
        float2* tmp = nullptr;
        FDR_HIP(hipMallocAsync((void**)&tmp, (size_t)rows * L * sizeof(float2), s));
        const hipError_t e = long_rows_dev(d, tmp, (size_t)rows, L, logl, p->mode, inverse != 0, twf, twi, s);
        (void)hipFreeAsync(tmp, s);
        FDR_HIP(e);
        return FDR_OK;
    }
    if (L >= 8) {
        RowArgs ra{};
        ra.src_c = d; ra.dst_c = d; ra.M = rows;
        // fast mode: the register kernels take the forward table and conjugate it; parity: the table of the direction
        FDR_HIP(launch_rows(logl, p->mode, ROW_IN_COMPLEX, ROW_OUT_COMPLEX, inverse != 0, ra, p->mode == FDR_MODE_FAST ? twf : (inverse ? twi : twf), s));
    } else {
        FDR_HIP(launch_simple_rows(d, rows, L, logl, inverse ? twi : twf, p->mode, s));
    }
    return FDR_OK;
}

int fdr_slab_pack_dev(const void* d_src, int rows, int ld, int parts, const int* counts, int elem_size, void* d_dst, void* stream) {
    if (!d_src || !d_dst || !counts || rows < 0 || ld <= 0) return fail(FDR_ERR_ARG, "fdr_slab_pack_dev: bad argument");
    hipError_t e = launch_slab_pack(d_src, rows, ld, parts, counts, elem_size, d_dst, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return fail(FDR_ERR_ARG, "fdr_slab_pack_dev: 1..16 parts with non-negative counts that sum to ld, element size 4 or 8");
    FDR_HIP(e);
    return FDR_OK;
}

int fdr_slab_transpose_dev(const void* d_src, void* d_dst, int rows, int cols, int elem_size, void* stream) {
    if (!d_src || !d_dst || rows < 0 || cols < 0 || d_src == d_dst) return fail(FDR_ERR_ARG, "fdr_slab_transpose_dev: bad argument");
    hipError_t e = launch_transpose_any(d_src, d_dst, rows, cols, elem_size, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return fail(FDR_ERR_ARG, "fdr_slab_transpose_dev: element size 4 or 8");
    FDR_HIP(e);
    return FDR_OK;
}

int fdr_slab_wiener_dev(fdr_plan* p, float* d_g, const float* d_h, size_t count, float K, void* stream) {
    if (!p || !d_g || !d_h) return fail(FDR_ERR_ARG, "fdr_slab_wiener_dev: null argument");
    if (count == 0) return FDR_OK;
    FDR_HIP(hipSetDevice(p->device));
    // parity: the quotient against H in the reference's operation order; fast: H is turned into W in a scratch-free
    // second launch first?  No: the slab mode keeps H and uses the parity quotient in both modes (one pointwise pass).
    FDR_HIP(launch_wiener_pointwise(reinterpret_cast<float2*>(d_g), reinterpret_cast<const float2*>(d_h), count, K, FDR_MODE_PARITY, (hipStream_t)stream));
    return FDR_OK;
}

int fdr_slab_real_dev(const float* d_complex, float* d_real, size_t count, void* stream) {
    if (!d_complex || !d_real) return fail(FDR_ERR_ARG, "fdr_slab_real_dev: null argument");
    FDR_HIP(launch_real_part(reinterpret_cast<const float2*>(d_complex), d_real, count, (hipStream_t)stream));
    return FDR_OK;
}

int fdr_slab_minmax_dev(fdr_plan* p, const float* d_real, int rows, int ld, int mm_rows, int mm_cols, float* d_mm, void* stream) {
    if (!p || !d_real || !d_mm || rows <= 0 || ld <= 0) return fail(FDR_ERR_ARG, "fdr_slab_minmax_dev: bad argument");
    FDR_HIP(hipSetDevice(p->device));
    const long long need = (long long)((ld + 255) / 256) * rows;
    if (need > p->mm_part_cap) return fail(FDR_ERR_ARG, "fdr_slab_minmax_dev: slab larger than the plan's M x N");
    int n_part = 0;
    FDR_HIP(launch_minmax_real(d_real, rows, ld, mm_rows, mm_cols, p->mm_part, &n_part, (hipStream_t)stream));
    FDR_HIP(launch_reduce_minmax(p->mm_part, n_part, d_mm, (hipStream_t)stream));
    return FDR_OK;
}

int fdr_slab_normalize_dev(const float* d_real, int ld, const float* d_mm, float* d_out, int rows, int cols, int out_stride, void* stream) {
    if (!d_real || !d_mm || !d_out || rows < 0 || cols < 0 || cols > ld || out_stride < cols) return fail(FDR_ERR_ARG, "fdr_slab_normalize_dev: bad argument");
    FDR_HIP(launch_normalize(d_real, ld, nullptr, 0, d_mm, d_out, rows, cols, out_stride, (hipStream_t)stream));
    return FDR_OK;
}

// ---- multi-GPU batched mode for C / C++ callers: one host thread, one plan, one PSF spectrum per device entry ----
namespace {

struct BatchWorker {
    int index = 0, device = 0, first = 0, count = 0;
    int status = FDR_OK;
    std::string error;
    double elapsed_ms = 0.0, checksum = 0.0;
    std::chrono::steady_clock::time_point t_end;
};

// deterministic checksum of `count` floats on the device: per-block partial sums in double, folded on the host
__global__ void checksum_kernel(const float* __restrict__ x, size_t count, double* __restrict__ part) {
    __shared__ double red[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) acc += (double)x[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// `prepared`: a plan that already holds its filter (fdr_batch_desc::bcast_filter: created and filled by the calling thread,
// which has synchronised the device); the worker owns it from here on.  nullptr: the worker builds plan and filter itself.
// Start line of fdr_batch_run's workers: set-up (plan, PSF spectrum, synthesis, warm-up) differs from device to device, so
// every worker waits here until all of them are ready and the timed regions start together; `wall_ms` then spans the work
// itself, not the set-up skew.  A worker that fails before the line still arrives (without waiting), so nobody waits for it.
struct StartGate {
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, total = 0;
    explicit StartGate(int n) : total(n) {}
    void arrive(bool wait) {
        std::unique_lock<std::mutex> lk(m);
        if (++arrived >= total) { cv.notify_all(); return; }
        if (wait) cv.wait(lk, [&] { return arrived >= total; });
    }
};

int batch_worker_run(const fdr_batch_desc* d, BatchWorker* w, std::chrono::steady_clock::time_point* t_start_out, fdr_plan* prepared, StartGate* gate) {
    fdr_plan* plan = prepared;
    bool at_gate = false;
    auto start_line = [&] { at_gate = true; gate->arrive(true); };
    float *d_in = nullptr, *d_out = nullptr;
    double* d_part = nullptr;
    hipStream_t stream = nullptr;
    int rc = FDR_OK;
    auto body = [&]() -> int {
        if (w->count == 0) return FDR_OK;
        int r = FDR_OK;
        if (!plan) {
            r = fdr_plan_create(w->device, d->M, d->N, d->mode, d->flags, &plan);
            if (r != FDR_OK) return r;
        }
        // the worker's stream exists BEFORE the PSF spectrum is queued, and the generated PSF is prepared ON it: the batches
        // below run on this (non-blocking) stream and its forks, which never synchronise with the null stream by themselves
        // (fdr_set_psf with a host PSF synchronises before it returns)
        FDR_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        if (!prepared) {
            if (d->psf_host) r = fdr_set_psf(plan, d->psf_host, d->psf_rows, d->psf_cols, d->psf_stride, d->K);
            else r = fdr_set_psf_motion(plan, d->psf_size, d->psf_angle_deg, d->K, stream);
        }
        if (r != FDR_OK) return r;
        if (d->imgs_host) {  // host images: the pipelined host batch over this worker's shard
            FDR_HIP(hipStreamSynchronize(stream));  // (the PSF spectrum is part of the set-up)
            start_line();
            const auto t0 = std::chrono::steady_clock::now();
            *t_start_out = t0;
            r = fdr_wiener_batch_ptrs_f32(plan, d->imgs_host + w->first, d->outs_host + w->first, w->count, d->rows, d->cols, d->stride,
                                          d->out_stride, d->norm_area);
            w->t_end = std::chrono::steady_clock::now();
            w->elapsed_ms = std::chrono::duration<double, std::milli>(w->t_end - t0).count();
            if (r != FDR_OK) return r;
            double acc = 0.0;
            for (int i = 0; i < w->count; ++i)
                for (int y = 0; y < d->rows; ++y) {
                    const float* row = d->outs_host[w->first + i] + (size_t)y * d->out_stride;
                    for (int x = 0; x < d->cols; ++x) acc += (double)row[x];
                }
            w->checksum = acc;
            return FDR_OK;
        }
        // synthetic, device resident
        // defaults as bench.py's (measured): 2 streams; 8 images per launch up to 1024^2, 4 up to 4096^2, larger 2
        const size_t px = (size_t)d->M * (size_t)d->N;
        const int ns = d->nstreams > 0 ? d->nstreams : (d->mode == FDR_MODE_FAST ? 2 : 3);
        const int gr = d->group > 0 ? d->group : (px <= (size_t)1024 * 1024 ? 8 : (px <= (size_t)4096 * 4096 ? 4 : 2));
        r = fdr_plan_set_batching(plan, ns, d->mode == FDR_MODE_FAST ? gr : 1);
        if (r != FDR_OK) return r;
        const size_t P = (size_t)d->rows * d->cols, total = P * (size_t)w->count;
        FDR_HIP(hipMalloc((void**)&d_in, total * sizeof(float)));
        FDR_HIP(hipMalloc((void**)&d_out, total * sizeof(float)));
        FDR_HIP(hipMalloc((void**)&d_part, 1024 * sizeof(double)));
        FDR_HIP(launch_synth(d->synth_seed, (uint64_t)w->first * P, total, d_in, stream));
        for (int k = 0; k < d->warmup && r == FDR_OK; ++k)
            r = fdr_wiener_batch_f32_dev(plan, d_in, P, w->count, d->rows, d->cols, d->cols, d_out, P, d->cols, d->norm_area, stream);
        FDR_HIP(hipStreamSynchronize(stream));
        if (r != FDR_OK) return r;
        start_line();
        const auto t0 = std::chrono::steady_clock::now();
        *t_start_out = t0;
        for (int k = 0; k < d->steps && r == FDR_OK; ++k)
            r = fdr_wiener_batch_f32_dev(plan, d_in, P, w->count, d->rows, d->cols, d->cols, d_out, P, d->cols, d->norm_area, stream);
        FDR_HIP(hipStreamSynchronize(stream));
        w->t_end = std::chrono::steady_clock::now();
        w->elapsed_ms = std::chrono::duration<double, std::milli>(w->t_end - t0).count();
        if (r != FDR_OK) return r;
        hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, stream, d_out, total, d_part);
        FDR_HIP(hipGetLastError());
        std::vector<double> part(1024);
        FDR_HIP(hipMemcpyAsync(part.data(), d_part, 1024 * sizeof(double), hipMemcpyDeviceToHost, stream));
        FDR_HIP(hipStreamSynchronize(stream));
        double acc = 0.0;
        for (double v : part) acc += v;
        w->checksum = acc;
        return FDR_OK;
    };
    if (hipSetDevice(w->device) != hipSuccess) rc = fail(FDR_ERR_HIP, "fdr_batch_run: hipSetDevice failed");
    else rc = body();
    if (!at_gate) gate->arrive(false);  // no images, or failed during set-up: count as arrived, do not hold the others up
    if (rc != FDR_OK) w->error = g_last_error;  // thread-local: hand it to the calling thread
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_part);
    if (stream) (void)hipStreamDestroy(stream);
    fdr_plan_destroy(plan);
    w->status = rc;
    return rc;
}

}  // namespace

namespace {

// RCCL, resolved at run time (no link-time dependency: a process that never broadcasts a filter never loads it, and inside
// a PyTorch process the copy of the library that torch has already mapped is the one that answers)
struct Rccl {
    typedef void* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)(void) = nullptr;
    int (*GroupEnd)(void) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    Rccl() {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!h) return;
        *(void**)&CommInitAll = dlsym(h, "ncclCommInitAll");
        *(void**)&CommDestroy = dlsym(h, "ncclCommDestroy");
        *(void**)&GroupStart = dlsym(h, "ncclGroupStart");
        *(void**)&GroupEnd = dlsym(h, "ncclGroupEnd");
        *(void**)&Broadcast = dlsym(h, "ncclBroadcast");
        *(void**)&GetErrorString = dlsym(h, "ncclGetErrorString");
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast;
    }
};

// fdr_batch_desc::bcast_filter: plans[0] holds the filter; every other plan gets its bytes.  Distinct devices: ONE
// ncclBroadcast over a communicator of all of them (RCCL over xGMI: the MPI_Bcast of fft/fft_mpi.cpp:334-378); an ordinal
// that repeats (two workers on one device -- RCCL refuses that) or a missing RCCL: device-to-device / peer copies.
// Returns the path taken (FDR_FILTER_*) or a negative status.
int distribute_filter(const std::vector<fdr_plan*>& plans, float K, bool force_rccl) {
    const int G = (int)plans.size();
    const size_t bytes = plans[0]->ws_elems * sizeof(float2);
    bool distinct = true;
    for (int a = 0; a < G; ++a)
        for (int b = a + 1; b < G; ++b) distinct = distinct && plans[a]->device != plans[b]->device;
    int path = FDR_FILTER_PEER_COPY;
    static Rccl rccl;  // (thread-safe initialisation; loaded on first use)
    if (distinct && (G > 1 || force_rccl) && rccl.ok) {
        std::vector<int> devs(G);
        for (int g = 0; g < G; ++g) devs[g] = plans[g]->device;
        std::vector<Rccl::comm_t> comms(G, nullptr);
        int nr = rccl.CommInitAll(comms.data(), G, devs.data());
        if (nr == 0) {
            nr = rccl.GroupStart();
            for (int g = 0; g < G && nr == 0; ++g) {
                if (hipSetDevice(devs[g]) != hipSuccess) { nr = -1; break; }
                nr = rccl.Broadcast(plans[g]->filt, plans[g]->filt, bytes, 0 /* ncclChar */, 0, comms[g], nullptr);
            }
            const int ne = rccl.GroupEnd();
            if (nr == 0) nr = ne;
            for (int g = 0; g < G; ++g)
                if (hipSetDevice(devs[g]) == hipSuccess && hipDeviceSynchronize() != hipSuccess && nr == 0) nr = -1;
            for (int g = 0; g < G; ++g)
                if (comms[g]) (void)rccl.CommDestroy(comms[g]);
        }
        if (nr == 0) path = FDR_FILTER_RCCL_BROADCAST;
        else  // the G > 1 RCCL path has not met multi-GPU hardware yet: an error there must not cost the batch
            std::fprintf(stderr, "fdr_batch_run: RCCL broadcast of the filter failed (%s); falling back to peer copies\n",
                         rccl.GetErrorString && nr > 0 ? rccl.GetErrorString(nr) : "error");
    }
    if (path != FDR_FILTER_RCCL_BROADCAST) {
        for (int g = 1; g < G; ++g) {
            FDR_HIP(hipSetDevice(plans[g]->device));
            if (plans[g]->device == plans[0]->device) FDR_HIP(hipMemcpy(plans[g]->filt, plans[0]->filt, bytes, hipMemcpyDeviceToDevice));
            else FDR_HIP(hipMemcpyPeer(plans[g]->filt, plans[g]->device, plans[0]->filt, plans[0]->device, bytes));
        }
    }
    for (int g = 1; g < G; ++g) { plans[g]->K = K; plans[g]->have_psf = true; }
    return path;
}

}  // namespace

extern "C" int fdr_batch_run(const fdr_batch_desc* d, fdr_batch_stats* st) {
    if (!d) return fail(FDR_ERR_ARG, "fdr_batch_run: null descriptor");
    if (d->n_devices < 1 || d->n_devices > FDR_BATCH_MAX_DEVICES || !d->devices)
        return fail(FDR_ERR_ARG, "fdr_batch_run: need 1..16 device entries");
    if (d->count < 0 || d->rows <= 0 || d->cols <= 0 || d->rows > d->M || d->cols > d->N)
        return fail(FDR_ERR_ARG, "fdr_batch_run: bad batch shape");
    if (d->imgs_host && (!d->outs_host || d->stride < d->cols || d->out_stride < d->cols))
        return fail(FDR_ERR_ARG, "fdr_batch_run: host images need outs_host and strides >= cols");
    if (!d->imgs_host && d->steps < 1) return fail(FDR_ERR_ARG, "fdr_batch_run: synthetic run needs steps >= 1");
    if (!d->psf_host && d->psf_size <= 0) return fail(FDR_ERR_ARG, "fdr_batch_run: no PSF given");
    int ndev = 0;
    FDR_HIP(hipGetDeviceCount(&ndev));
    for (int g = 0; g < d->n_devices; ++g)
        if (d->devices[g] < 0 || d->devices[g] >= ndev) return fail(FDR_ERR_ARG, "fdr_batch_run: device ordinal out of range");
    const int G = d->n_devices;
    std::vector<BatchWorker> ws((size_t)G);
    std::vector<std::chrono::steady_clock::time_point> starts((size_t)G);
    // fft/fft_mpi.cpp:89-100 applied to images: floor(count / G) each, the first count % G workers one more
    for (int g = 0, first = 0; g < G; ++g) {
        ws[g].index = g; ws[g].device = d->devices[g];
        ws[g].count = d->count / G + (g < d->count % G ? 1 : 0);
        ws[g].first = first;
        first += ws[g].count;
    }
    // bcast_filter: worker 0's filter for everyone -- plans created and the filter distributed here, before the workers start
    std::vector<fdr_plan*> prepared((size_t)G, nullptr);
    int filter_path = FDR_FILTER_LOCAL;
    if (d->bcast_filter && (G > 1 || d->bcast_filter == 2) && d->count >= G) {  // (every worker has at least one image, so every plan is used)
        int prc = FDR_OK;
        for (int g = 0; g < G && prc == FDR_OK; ++g) prc = fdr_plan_create(ws[g].device, d->M, d->N, d->mode, d->flags, &prepared[g]);
        if (prc == FDR_OK) {
            if (d->psf_host) prc = fdr_set_psf(prepared[0], d->psf_host, d->psf_rows, d->psf_cols, d->psf_stride, d->K);
            else prc = fdr_set_psf_motion(prepared[0], d->psf_size, d->psf_angle_deg, d->K, nullptr);
        }
        if (prc == FDR_OK && (hipSetDevice(prepared[0]->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess))
            prc = fail(FDR_ERR_HIP, "fdr_batch_run: preparing the filter on worker 0's device failed");
        if (prc == FDR_OK) { filter_path = distribute_filter(prepared, d->K, d->bcast_filter == 2); if (filter_path < 0) prc = filter_path; }
        if (prc != FDR_OK) {
            const std::string msg0 = g_last_error;
            for (auto* pl : prepared) fdr_plan_destroy(pl);
            return fail(prc, "fdr_batch_run: " + msg0);
        }
    }
    const auto t_launch = std::chrono::steady_clock::now();
    for (int g = 0; g < G; ++g) { starts[g] = t_launch; ws[g].t_end = t_launch; }
    std::vector<std::thread> threads;
    StartGate gate(G);
    for (int g = 1; g < G; ++g) threads.emplace_back(batch_worker_run, d, &ws[g], &starts[g], prepared[g], &gate);
    batch_worker_run(d, &ws[0], &starts[0], prepared[0], &gate);  // worker 0 on the calling thread
    for (auto& t : threads) t.join();
    int rc = FDR_OK;
    std::string msg;
    auto t_first = starts[0], t_last = ws[0].t_end;
    bool any = false;
    long long done = 0;
    for (int g = 0; g < G; ++g) {
        if (ws[g].status != FDR_OK && rc == FDR_OK) { rc = ws[g].status; msg = "worker " + std::to_string(g) + " (device " + std::to_string(ws[g].device) + "): " + ws[g].error; }
        if (ws[g].count > 0 && ws[g].status == FDR_OK) {
            if (!any || starts[g] < t_first) t_first = starts[g];
            if (!any || ws[g].t_end > t_last) t_last = ws[g].t_end;
            any = true;
            done += (long long)ws[g].count * (d->imgs_host ? 1 : d->steps);
        }
    }
    if (st) {
        memset(st, 0, sizeof *st);
        st->n_devices = G;
        for (int g = 0; g < G; ++g) {
            st->first[g] = ws[g].first; st->images[g] = ws[g].count; st->elapsed_ms[g] = ws[g].elapsed_ms;
            st->checksum[g] = ws[g].checksum; st->status[g] = ws[g].status;
        }
        st->wall_ms = any ? std::chrono::duration<double, std::milli>(t_last - t_first).count() : 0.0;
        st->images_done = done;
        st->mpixels_per_s = st->wall_ms > 0.0 ? (double)done * d->rows * d->cols / 1e6 / (st->wall_ms * 1e-3) : 0.0;
        st->filter_path = filter_path;
    }
    if (rc != FDR_OK) return fail(rc, "fdr_batch_run: " + msg);
    return FDR_OK;
}


int fdr_plan_profile(fdr_plan* p, int enable) {
    if (!p) return fail(FDR_ERR_ARG, "fdr_plan_profile: null plan");
    FDR_HIP(hipSetDevice(p->device));
    p->timer.reset();
    p->timer.enabled = enable != 0;
    return FDR_OK;
}

int fdr_plan_pass_times(fdr_plan* p, int* n_passes, float* mean_ms, const char** names, int* launches) {
    if (!p || !n_passes) return fail(FDR_ERR_ARG, "fdr_plan_pass_times: null argument");
    FDR_HIP(hipSetDevice(p->device));
    double sum[FDR_MAX_PASSES] = {0};
    int cnt[FDR_MAX_PASSES] = {0};
    for (auto& r : p->timer.recs) {
        FDR_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        FDR_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        sum[r.pass] += ms;
        cnt[r.pass] += 1;
    }
    *n_passes = p->timer.n_names;
    for (int i = 0; i < p->timer.n_names; ++i) {
        if (mean_ms) mean_ms[i] = cnt[i] ? (float)(sum[i] / cnt[i]) : 0.f;
        if (names) names[i] = p->timer.names[i];
        if (launches) launches[i] = cnt[i];
    }
    p->timer.reset();
    return FDR_OK;
}

}  // extern "C"

#ifdef FDR_DIAG  // diagnostic builds only (tools/diag): the addresses of a slot's intermediates
extern "C" int fdr_debug_slot_ptrs(fdr_plan* p, int slot, void** work, void** raw, void** mm_part, size_t* ws_elems) {
    if (!p || slot < 0 || slot >= fdr_plan::kMaxSlots) return FDR_ERR_ARG;
    *work = p->slots[slot].work; *raw = p->slots[slot].raw; *mm_part = p->slots[slot].mm_part; *ws_elems = p->ws_elems;
    return FDR_OK;
}
#endif
