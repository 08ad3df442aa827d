// fdr_kernels.hpp -- argument blocks and launcher declarations shared between the kernel
// translation units (compiled for gfx950 with -ffp-contract=off) and the host-side plan code.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace fdr {

enum RowIn { ROW_IN_REAL = 0, ROW_IN_COMPLEX = 1 };
enum RowOut {
    ROW_OUT_COMPLEX = 0,
    ROW_OUT_REAL_MINMAX = 1,  // pass C': real plane + min/max partials
    ROW_OUT_MINMAX_ONLY = 2,  // pass C1 (two-sweep normalisation): min/max partials, nothing stored
    ROW_OUT_NORMALIZED = 3    // pass C2: the transform again, normalised with the folded partials and cropped on store
};
enum ColKind {
    COL_FWD = 0,         // forward column FFT, complex in place (PSF spectrum, fft2d)
    COL_INV = 1,         // inverse column FFT, complex in place (fft2d)
    COL_FWD_WIENER = 2,  // parity pass B: forward column FFT then the Wiener quotient against H
    COL_INV_REAL = 3,    // parity pass D: inverse column FFT, real part to the raw plane, min/max
    COL_FUSED = 4,       // fast pass B' (panel kernels only): forward column FFT, multiply by W, inverse column FFT
    COL_FWD_FILTER = 5   // PSF spectrum (panel kernels only): rows >= nvalid taken as zero, forward column FFT, W = conj(H)/(|H|^2+K)
};

// several images per launch of the fast row passes / the normalisation (blockIdx.y = image): small images are launch
// bound, one launch per pass and GROUP of images fills the chip.  nimg <= 1: the single-image fields are used.
constexpr int kMaxGroup = 8;  // images per launch (fdr_plan_set_batching)
// entry `i` (uniform over the workgroup) of a kernel-argument array as a select chain on scalars: a dynamic index would
// send the array to scratch memory
template <class P>
__host__ __device__ __forceinline__ P pick_image(P const (&p)[kMaxGroup], int i) {
    static_assert(kMaxGroup == 8, "select chain written for 8 entries");
    const P lo = i == 0 ? p[0] : i == 1 ? p[1] : i == 2 ? p[2] : p[3];
    const P hi = i == 4 ? p[4] : i == 5 ? p[5] : i == 6 ? p[6] : p[7];
    return i < 4 ? lo : hi;
}
struct RowBatch {
    const float* src_real[kMaxGroup];  // pass A input
    float2* spec[kMaxGroup];           // pass A output / pass C' input (panel-major spectrum)
    float* raw[kMaxGroup];             // pass C' output
    float2* mm_part[kMaxGroup];        // pass C' min/max partials
    float* out[kMaxGroup];             // pass C2 output (normalised, cropped)
    int nimg;
};
struct NormBatch {
    const float* raw[kMaxGroup];
    const float2* part[kMaxGroup];
    float* out[kMaxGroup];
    int nimg;
};

// Workgroups that touch the same 128-byte lines -- `share` consecutive tiles -- are placed on ONE XCD (workgroup b lands
// on XCD b % 8) and dispatched back to back, so that their partial lines meet in that XCD's L2: tile of workgroup b.
__host__ __device__ __forceinline__ int col_tile_of_block(int b, int ntiles, int share) {
    if (share <= 1 || (ntiles % (8 * share)) != 0) return b;
    const int grp = b / (8 * share), r = b % (8 * share);
    return grp * 8 * share + (r % 8) * share + (r / 8);
}

struct RowArgs {
    // input
    const float* src_real;  // ROW_IN_REAL: rows x cols image, zero-padded on the fly to M x L
    int src_rows, src_cols, src_stride;
    const float2* src_c;  // ROW_IN_COMPLEX: M x L
    // output
    float2* dst_c;    // ROW_OUT_COMPLEX: M x L
    float* dst_real;  // ROW_OUT_REAL_MINMAX: M x L real plane
    float2* mm_part;  // one (min, max) partial per workgroup
    int mm_rows, mm_cols;
    float* out;       // ROW_OUT_NORMALIZED: out_rows x out_cols result, row stride out_stride; mm_part holds n_part partials
    int out_rows, out_cols, out_stride, n_part;
    int M;              // number of rows to transform
    size_t pstride;     // rows4 kernels: panel stride of the panel-major spectrum, in float2 elements
    int half;           // rows4 packed kernels: half (Hermitian) spectrum, N/8 panels
    int num_cu;         // rows4 persistent kernels: CUs of the device
    RowBatch batch;     // rows4 packed kernels: several images per launch
    int panel_c;        // launch_rows (parity operator): the complex side(s) are PANEL-major, full spectrum: element (m, n) at
                        // (n >> 2) * pstride + m * 4 + (n & 3) -- the column passes then work on contiguous tiles
};

// up to kMaxGroup images' spectra handled by ONE pass-B' launch (their panels form one tile sequence)
struct PanelBatch {
    float2* data[kMaxGroup];
    int nimg;  // 0 or 1: use ColArgs::data only
};

struct ColArgs {
    float2* data;        // M x N complex, transformed in place
    const float2* filt;  // H (parity) or W (fast), M x N
    float K;
    float* dst_real;  // COL_INV_REAL: M x N real plane
    float2* mm_part;  // one (min, max) partial per workgroup
    int mm_rows, mm_cols;
    int N;  // row length (number of columns)
    int npanels;      // panel kernels: number of panels (0 = N/4)
    PanelBatch batch; // panel kernels, COL_FUSED: several images per launch
    int packed0;      // panel kernels: column 0 of panel 0 is the packed DC + i Nyquist column (half spectrum)
    int nvalid;       // COL_FWD_FILTER: rows of the panels that hold data (a multiple of 4); the others are read as zero
    size_t pstride;   // panel kernels: panel stride in float2 elements
    int num_cu;       // CUs of the device (persistent pass B' launches one workgroup per CU)
    int panel_c;      // launch_cols (parity operator): data / filt panel-major (pstride), dst_real panel-major with stride 4 M floats
};

// launchers (fdr_rows.hip / fdr_cols.hip); logl = log2 of the transform length, 3..13
// tw: parity mode -> table of the requested direction; fast mode -> the forward table (inverse = conjugate)
hipError_t launch_rows(int logl, int mode, RowIn in, RowOut out, bool inverse, const RowArgs& a, const float2* tw,
                       hipStream_t s);
hipError_t launch_cols(int logm, int mode, ColKind kind, const ColArgs& a, const float2* tw_fwd, const float2* tw_inv,
                       hipStream_t s);

// fast-mode passes on the panel-major intermediate (fdr_panel.hip); tw_fwd = forward table
// rows4: (ROW_IN_REAL -> ROW_OUT_COMPLEX[panel]) forward, (ROW_IN_COMPLEX[panel] -> ROW_OUT_REAL_MINMAX) inverse
hipError_t launch_rows4(int logl, RowIn in, RowOut out, const RowArgs& a, const float2* tw_fwd, hipStream_t s);
// min/max partials pass C' writes per image when `nimg` images share a launch on a device with num_cu CUs
int rows4_minmax_partials(int logl, int M, int num_cu, int nimg, int half);
// cols_panel: COL_FWD_FILTER (PSF spectrum -> W, in place) or COL_FUSED (FFT . W . IFFT)
hipError_t launch_cols_panel(int logm, ColKind kind, const ColArgs& a, const float2* tw_fwd, hipStream_t s);

// reference-shaped and auxiliary kernels (fdr_aux.hip)
hipError_t launch_pad_real_to_complex(const float* src, int rows, int cols, int stride, float2* dst, int M, int N,
                                      hipStream_t s);
hipError_t launch_simple_rows(float2* data, int rows, int L, int logl, const float2* tw, int mode, hipStream_t s);
hipError_t launch_transpose(const float2* src, float2* dst, int rows, int cols, hipStream_t s);
// transforms of more than 8192 points (fdr_aux.hip): subsequences gathered into 8192-point blocks, and one radix-2 stage
// (butterfly distance `half`) over rows of length L in global memory; tw = table of the requested direction (both modes)
constexpr int kMaxLdsLog = 13;    // longest transform the row / column kernels hold on chip
constexpr int kMaxLongLog = 15;   // longest transform at all (32768 points)
hipError_t launch_long_gather(const float2* src, float2* dst, size_t rows, int L, int logs, hipStream_t s);
hipError_t launch_long_stage(const float2* src, float2* dst, size_t rows, int L, int half, const float2* tw, int mode, hipStream_t s);
hipError_t launch_wiener_pointwise(float2* g, const float2* filt, size_t count, float K, int mode, hipStream_t s);
hipError_t launch_make_filter_fast(const float2* H, float2* W, size_t count, float K, hipStream_t s);
hipError_t launch_real_minmax(const float2* src, float* dst, int M, int N, int mm_rows, int mm_cols, float2* mm_part,
                              int* n_part, hipStream_t s);
hipError_t launch_reduce_minmax(const float2* mm_part, int n_part, float* mm, hipStream_t s);
// number of (min,max) partials the row / column real-output passes write for an M x N plan
int rows_minmax_partials(int logl, int M);
int cols_minmax_partials(int logm, int N);
// mm_part != nullptr: every workgroup folds the n_part partials itself; else mm = {min, max} from launch_reduce_minmax
hipError_t launch_normalize(const float* raw, int N, const float2* mm_part, int n_part, const float* mm, float* out,
                            int rows, int cols, int out_stride, hipStream_t s, const NormBatch* batch = nullptr);
// the same from a PANEL-major real plane (panel p = columns 4 p .. 4 p + 3, M rows of 4 floats, panels 4 M floats apart)
hipError_t launch_normalize_panels(const float* raw, int M, int N, const float2* mm_part, int n_part, const float* mm, float* out,
                                   int rows, int cols, int out_stride, hipStream_t s);
hipError_t launch_psf_motion(int size, double angle_deg, float* d_out, hipStream_t s);
// cv::warpAffine defaults (bilinear, constant 0 border) on a single-channel float image; fwd = the 2 x 3 matrix as cv::warpAffine takes it
hipError_t launch_warp_affine(const float* src, int srows, int scols, int sstride, const double fwd[6], float* dst, int drows, int dcols,
                              int dstride, hipStream_t s);
// slab mode (single image over several GPUs): column blocks of a row slab packed for the all-to-all, dense transposes
// of 4- or 8-byte elements, real part, min/max partials of a real plane
hipError_t launch_slab_pack(const void* src, int rows, int ld, int parts, const int* counts, int elem_size, void* dst, hipStream_t s);
hipError_t launch_transpose_any(const void* src, void* dst, int rows, int cols, int elem_size, hipStream_t s);
hipError_t launch_real_part(const float2* src, float* dst, size_t count, hipStream_t s);
hipError_t launch_minmax_real(const float* src, int rows, int ld, int mm_rows, int mm_cols, float2* part, int* n_part, hipStream_t s);

// colour epilogue of the drivers (fdr_color.hip): planar float BGR in [0,1] -> Lab white balance -> interleaved 8-bit BGR
struct ColorArgs {
    const float* orig[3];  // blurred input planes B, G, R (the white-balance reference)
    const float* rest[3];  // restored planes B, G, R
    int rows, cols, stride;
    unsigned char* out;    // rows x cols x 3, row stride out_stride bytes
    int out_stride;
};
int color_partials(int rows, int cols);
hipError_t launch_color_epilogue(const ColorArgs& a, double2* part, hipStream_t s);
hipError_t launch_synth(uint64_t seed, uint64_t first, size_t count, float* d_out, hipStream_t s);
hipError_t launch_dft_naive(const float2* src, float2* dst, int n, int inverse, hipStream_t s);
// table[t * n + k], forward direction, host generated (see fdr_aux.hip); rows transforms of length n, src != dst
hipError_t launch_dft_naive_rows(const float2* src, float2* dst, int rows, int n, const float2* table, int inverse, hipStream_t s);

}  // namespace fdr
