/*
 * fdr.h -- C ABI of libfdr.so: the MI355X (gfx950) frequency-domain restoration path.
 *
 * This is the drop-in boundary for the reference's GPU operator surface.  Every entry point
 * names the reference interface it stands in for (paths relative to the reference repo).
 * Plain pointers and sizes only; no C++ or torch types; every function returns an int status
 * (FDR_OK or a negative FDR_ERR_*), never throws, and leaves a message for fdr_last_error().
 * `stream` arguments are a hipStream_t passed as void* (NULL = the null stream).
 *
 * The C++ shim with the reference's own names (fft_gpu::wienerDeblur_RGB_optimized etc.,
 * fft/fft.hpp:31-45) lives in include/fft/fft.hpp and calls only these functions.
 */
#ifndef FDR_H
#define FDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDR_VERSION 300 /* 0.3.0 */

/* status codes (reference: CHECK_CUDA prints and exit(1)s, fft/fft_gpu.cu:59-66; the C++ shim
 * reproduces that on any non-zero status) */
#define FDR_OK 0
#define FDR_ERR_ARG (-1)      /* null pointer, non-positive size, PSF larger than the plan ... */
#define FDR_ERR_NOT_POW2 (-2) /* plan dimensions must be powers of two (callers pad first,
                                 as fft/fft_gpu.cu:287-288 and serial.cpp:36 do) unless
                                 FDR_FLAG_ANY_SIZE asks for the reference's naive-DFT path  */
#define FDR_ERR_HIP (-3)      /* a HIP runtime call failed; message holds hipGetErrorString  */
#define FDR_ERR_STATE (-4)    /* e.g. fdr_wiener_* before fdr_set_psf*                        */
#define FDR_ERR_ALLOC (-5)

/* arithmetic modes of a plan */
#define FDR_MODE_PARITY 0 /* reference operation order (rows, cols, Wiener, rows, cols), per-stage
                             twiddles replayed from the float recurrence of fft/fft_serial.cpp:54-63,
                             no FMA contraction: bit-identical to the serial path's FFT arithmetic */
#define FDR_MODE_FAST 1   /* fused column pass (FFT . Wiener . IFFT), double-generated twiddles (as
                             fft/fft_gpu.cu:206-212), FMA butterflies; within 1e-4 of the serial path */

/* plan flags */
#define FDR_FLAG_SIMPLE_PATH 1u /* reference-shaped kernels: row FFT, transpose, row FFT, transpose
                                   (fft/fft_gpu.cu:214-240); slow, used as an on-device cross-check */

#define FDR_FLAG_FULL_SPECTRUM 32u /* fast mode: keep all N columns of the (Hermitian) spectrum instead of the
                                      non-redundant half (the complex-to-complex byte count of SURVEY.md 8d)     */

#define FDR_FLAG_TABLES_ONLY 1024u /* twiddle tables and min/max scratch only, no M x N workspaces: a plan for the slab
                                      primitives (fdr_slab_*) of the single-image multi-GPU mode, where every rank
                                      holds only its rows of the image; fdr_wiener_* / fdr_fft2d_* / fdr_set_psf*
                                      return FDR_ERR_STATE on such a plan                                          */

#define FDR_FLAG_ANY_SIZE 512u /* accept plan dimensions that are not powers of two: such a dimension is transformed by
                                  the O(n^2) DFT of fft_serial::dft_naive_inplace (fft/fft_serial.cpp:71-87), as
                                  transform_row_inplace dispatches (:100-101) when wienerDeblur_myfft pads to
                                  getOptimalDFTSize (2^a 3^b 5^c, :153-154) instead of to a power of two.  Reference-shaped
                                  passes (rows, transpose, rows, transpose); both modes use the parity arithmetic.        */

/* normalisation area selector for fdr_wiener_* */
#define FDR_NORM_PADDED 1  /* serial semantics: min/max over the padded M x N area, then crop
                              (serial.cpp:36-38 + fft/fft_serial.cpp:243-246)                 */
#define FDR_NORM_CROPPED 0 /* reference GPU semantics: min/max over the cropped rows x cols
                              area only (fft/fft_gpu.cu:379-381)                              */

typedef struct fdr_plan fdr_plan;

int fdr_version(void);
/* thread-local message of the last failing call on this thread */
const char* fdr_last_error(void);
int fdr_device_count(int* count);

/* -- utils.hpp:27-37,50-52 ------------------------------------------------------------- */
int fdr_next_pow2(int n);
int fdr_is_pow2(int n);

/* -- cv::getOptimalDFTSize as fft/fft_serial.cpp:153-154 uses it: smallest 2^a 3^b 5^c >= n -- */
int fdr_optimal_dft_size(int n);

/* -- plan: owns twiddle tables, the M x N complex workspace, the filter spectrum and the
 *    min/max scratch for one device.  Replaces the per-call cudaMalloc/cudaFree block of
 *    fft/fft_gpu.cu:304-322,389-393.  One host thread at a time per plan.  fdr_plan_destroy
 *    called while the process is already running its exit handlers (static destructors of a
 *    caller) frees the host side only: the HIP runtime may be gone by then.
 *    Dimensions: powers of two up to 32768 (a dimension above 8192 is transformed in 8192-point
 *    blocks plus radix-2 stages in global memory: the reference's serial path takes any power
 *    of two, fft/fft_serial.cpp:90-108); with FDR_FLAG_ANY_SIZE also non-powers of two up to
 *    4096 (naive DFT).                                                                    */
int fdr_plan_create(int device, int M, int N, int mode, unsigned flags, fdr_plan** out);
int fdr_plan_destroy(fdr_plan* plan);
int fdr_plan_dims(const fdr_plan* plan, int* M, int* N, int* mode);
/* tunables of a plan */
#define FDR_OPT_TWO_SWEEP_NORM 2   /* fast mode, half spectrum: 1 = the inverse row pass runs twice (min/max only, then
                                      again with the normalisation applied on store) instead of writing a raw real plane
                                      that a normalise pass reads back: 12 instead of 16 bytes per pixel for the last two
                                      passes, same bits (default).  0 = passes C' + E. */
#define FDR_OPT_BATCH_GRAPH 3      /* 1 = fdr_wiener_batch_f32_dev captures its launches (fork, every pass of every group on
                                      the internal streams, join) as a hipGraph on first use and replays it while the call's
                                      arguments stay the same: for small images, whose batches are bound by the host's
                                      launch rate.  0 (default) = plain launches. */
#define FDR_OPT_CE_CHUNK_MB 4      /* batched fast mode on two or more streams: the inverse row passes C1 + C2 of a group of images are
                                      launched in chunks of the group -- as many images as make up at least this many MiB of spectrum
                                      (default 160: pairs at 4096^2, the whole group below) -- because launches of half the size
                                      interleave better with the other stream's memory-bound passes; 0 = always the whole group.
                                      Same bits either way. */
int fdr_plan_set_option(fdr_plan* plan, int option, long long value);

/* -- PSF generation: utils.hpp:15-24 motionBlurKernel(size, angle) ------------------- */
/* host result, size*size floats (computed on the device by the psf kernel, copied back) */
int fdr_psf_motion(int size, double angle_deg, float* out_host);
/* device result into d_out (size*size floats), asynchronous on stream */
int fdr_psf_motion_dev(int device, int size, double angle_deg, float* d_out, void* stream);

/* -- the OpenCV call inside motionBlurKernel (utils.hpp:22): cv::warpAffine(src, dst, M, dsize) with its defaults
 *    (INTER_LINEAR, BORDER_CONSTANT 0) on a single-channel float image, evaluated on the device in OpenCV's classic
 *    fixed-point form (10-bit coordinates rounded to 1/32 pixel, 32 x 32 float weight table).
 *    M: the 2 x 3 forward matrix (row-major) as cv::getRotationMatrix2D returns it; inverted in double as
 *    cv::warpAffine does.  Host pointers; strides in elements.  fdr_psf_motion(size, angle) is this call applied to
 *    the line kernel of utils.hpp:17-19 with the matrix of :20 (same bits).                                          */
int fdr_warp_affine_f32(const float* src_host, int srows, int scols, int sstride, const double M[6],
                        float* dst_host, int drows, int dcols, int dstride);

/* -- PSF spectrum: pad top-left + forward 2-D FFT (fft/fft_serial.cpp:166-171,182;
 *    fft/fft_gpu.cu:340,356), kept in the plan together with K.                        */
int fdr_set_psf(fdr_plan* plan, const float* psf_host, int prows, int pcols, int pstride, float K);
int fdr_set_psf_dev(fdr_plan* plan, const float* d_psf, int prows, int pcols, int pstride, float K, void* stream);
/* motionBlurKernel on the device straight into the plan (no host round trip) */
int fdr_set_psf_motion(fdr_plan* plan, int size, double angle_deg, float K, void* stream);

/* -- the prepared filter of a plan as an opaque block of bytes, for a caller that distributes ONE rank's PSF spectrum to the
 *    others instead of recomputing it everywhere -- the role of the MPI_Bcast / MPI_Scatterv of the padded PSF in the
 *    reference's MPI variant (fft/fft_mpi.cpp:334-378); in the batched mode a broadcast over RCCL (bench.py --bcast-filter).
 *    The layout is private to the library (mode, flags and dimensions select it): a block exported from one plan may only be
 *    imported into a plan created with the same (M, N, mode, flags) -- on any device.  `bytes` is the exact size.
 *      fdr_plan_filter_bytes   size of the block (FDR_ERR_STATE on a tables-only plan)
 *      fdr_plan_export_filter_dev  copy the plan's filter into d_dst (device memory of the plan's device), asynchronous on stream;
 *                                  needs a PSF set on the plan
 *      fdr_plan_import_filter_dev  copy d_src in as the plan's filter and take K with it: the plan then behaves as after
 *                                  fdr_set_psf* with the exporting plan's PSF                                              */
int fdr_plan_filter_bytes(const fdr_plan* plan, size_t* bytes);
int fdr_plan_export_filter_dev(fdr_plan* plan, void* d_dst, size_t bytes, void* stream);
int fdr_plan_import_filter_dev(fdr_plan* plan, const void* d_src, size_t bytes, float K, void* stream);

/* -- the operator: fft_serial::wienerDeblur_myfft (fft/fft_serial.cpp:141-261) wrapped as
 *    serial.cpp:34-39 does (pad -> restore -> crop), one channel.  img rows x cols with row
 *    stride `stride` (elements); out rows x cols with row stride `out_stride`, values in [0,1].
 *    rows <= M, cols <= N.  Host-pointer form copies in and out synchronously; the _dev form
 *    is asynchronous on `stream` and touches only device memory.                         */
int fdr_wiener_f32(fdr_plan* plan, const float* img_host, int rows, int cols, int stride,
                   float* out_host, int out_stride, int norm_area);
int fdr_wiener_f32_dev(fdr_plan* plan, const float* d_img, int rows, int cols, int stride,
                       float* d_out, int out_stride, int norm_area, void* stream);
/* `count` independent images, image i at d_imgs + i*img_pitch / d_out + i*out_pitch (elements);
 * the batched mode of BASELINE config 5 (one plan, one PSF spectrum, many images).      */
int fdr_wiener_batch_f32_dev(fdr_plan* plan, const float* d_imgs, size_t img_pitch, int count,
                             int rows, int cols, int stride,
                             float* d_out, size_t out_pitch, int out_stride, int norm_area, void* stream);

/* Host-pointer batch: `count` images at imgs_host + i*img_pitch, results to out_host + i*out_pitch (elements).
 * H2D copy, restoration and D2H copy of consecutive images overlap on three internal streams with three images in
 * flight -- the pinned-buffer / cudaMemcpyAsync pipeline fft/fft_gpu.cu:306-350,372-385 sets out to build.  Buffers
 * from fdr_host_alloc (pinned; replaces cudaMallocHost, fft/fft_gpu.cu:306-308) are copied by DMA in place and the
 * three stages overlap; pageable buffers work too, at the rate of the synchronous copies the runtime then makes.
 * Synchronous: returns when every result is in out_host.                                                        */
int fdr_host_alloc(size_t bytes, void** out);
int fdr_host_free(void* p);
int fdr_wiener_batch_f32(fdr_plan* plan, const float* imgs_host, size_t img_pitch, int count,
                         int rows, int cols, int stride,
                         float* out_host, size_t out_pitch, int out_stride, int norm_area);
/* the same with one pointer per image (the channel Mats of fft_gpu::wienerDeblur_RGB_*, fft/fft_gpu.cu:325-385) */
int fdr_wiener_batch_ptrs_f32(fdr_plan* plan, const float* const* imgs_host, float* const* outs_host, int count,
                              int rows, int cols, int stride, int out_stride, int norm_area);

/* Batched mode only: let consecutive images of fdr_wiener_batch_f32_dev alternate over `nstreams`
 * (1..4) private workspaces on internal HIP streams, forked from / joined to the caller's stream,
 * so one image's kernel tails overlap the next image's kernel heads.  Costs (nstreams-1) extra
 * workspaces of 12 bytes per padded pixel.  Default 1.                                         */
int fdr_plan_set_concurrency(fdr_plan* plan, int nstreams);
/* The same with `group` (1..8) images per launch in the fast mode: every pass handles `group` images in one launch
 * (small images are launch bound; one launch per pass and group fills the chip; pass B' shares the filter between
 * the images of a launch).  nstreams * group <= 16 workspaces.  fdr_plan_set_concurrency(n) == fdr_plan_set_batching(n, 1).              */
int fdr_plan_set_batching(fdr_plan* plan, int nstreams, int group);

/* -- fft_gpu::my_dft2D(Mat&, bool) (fft/fft.hpp:40; empty body at fft/fft_gpu.cu:515):
 *    in-place unscaled 2-D transform of M x N interleaved complex.                       */
int fdr_fft2d_c2c(fdr_plan* plan, float* data_host, int inverse);
int fdr_fft2d_c2c_dev(fdr_plan* plan, float* d_data, int inverse, void* stream);

/* -- fft_gpu::fft_radix2_kernel / transform_row_kernel / dft_naive_kernel (fft/fft.hpp:35-39;
 *    declared, never defined in the reference): 1-D unscaled transform of n interleaved
 *    complex values given by host pointer.  fft1d: power-of-two n up to 32768 (radix-2) else
 *    naive DFT (n <= 4096), as fft_serial::transform_row_inplace dispatches
 *    (fft/fft_serial.cpp:100-101).                                                        */
int fdr_fft1d_c2c(float* data_host, int n, int inverse, int mode);
int fdr_dft_naive_c2c(float* data_host, int n, int inverse);

/* -- colour epilogue of the drivers (serial.cpp:43-54, gpu.cpp:123-137; utils.hpp:55-71 applyWhiteBalance) on the
 *    device: restored planes B, G, R in [0,1] -> Lab -> L scaled so that its mean matches the blurred input's, clamped
 *    to [0,100] -> BGR -> 8 bit interleaved (convertTo(CV_8U, 255)).  Planes: rows x cols, row stride `stride`
 *    (elements); out: rows x cols x 3 bytes, row stride `out_stride_bytes`.  The Lab formulae are OpenCV's (third
 *    party, version unpinned by the reference): expect +-1 at 8 bit against a given OpenCV build.               */
int fdr_white_balance_u8_dev(int device, const float* const d_orig_bgr[3], const float* const d_restored_bgr[3],
                             int rows, int cols, int stride, unsigned char* d_out_bgr8, int out_stride_bytes, void* stream);
int fdr_white_balance_u8(int device, const float* const orig_bgr[3], const float* const restored_bgr[3],
                         int rows, int cols, int stride, unsigned char* out_bgr8, int out_stride_bytes);

/* -- synthetic input: pixel i = top 24 bits of splitmix64(seed+first+i) / 2^24 */
int fdr_synth_image_dev(int device, uint64_t seed, uint64_t first_index, size_t count, float* d_out, void* stream);

/* -- per-pass device timing (the reference's Profiler buckets, fft/fft_gpu.cu:17-57).
 *    With profiling on, every fdr_wiener_*_dev call records a hipEvent pair around each
 *    kernel on the call's stream; fdr_plan_pass_times synchronises and returns the mean
 *    duration in ms of each pass since the last reset, names[i] a static string.         */
#define FDR_MAX_PASSES 16
int fdr_plan_profile(fdr_plan* plan, int enable);
int fdr_plan_pass_times(fdr_plan* plan, int* n_passes, float* mean_ms, const char** names, int* launches);

/* -- the six buckets of the reference's Profiler (fft/fft_gpu.cu:17-57: alloc / h2d / pre / compute / d2h / post),
 *    accumulated per plan since its creation (or the last reset) from hipEvent pairs on the streams the work ran on:
 *      ALLOC   host wall time of fdr_plan_create (hipMalloc is synchronous)      [fft_gpu.cu:304-322]
 *      H2D     image / PSF uploads of the host-pointer entry points              [:330-335,346-350]
 *      PRE     PSF generation, padding, PSF spectrum and filter (fdr_set_psf*)   [:337-343,356]
 *      COMPUTE the restoration passes of fdr_wiener_*                            [:354-369]
 *      D2H     result downloads of the host-pointer entry points                 [:372-375]
 *      POST    0 here: normalisation runs on the device inside COMPUTE           [:378-384 is a CPU cv::normalize]
 *    In the pipelined host batch (fdr_wiener_batch_*_f32) the three streams overlap, so H2D + COMPUTE + D2H exceeds the
 *    wall time, exactly as per-phase sums do.  The call synchronises the device.                                       */
#define FDR_PHASE_ALLOC 0
#define FDR_PHASE_H2D 1
#define FDR_PHASE_PRE 2
#define FDR_PHASE_COMPUTE 3
#define FDR_PHASE_D2H 4
#define FDR_PHASE_POST 5
#define FDR_N_PHASES 6
int fdr_plan_phase_times(fdr_plan* plan, float ms[FDR_N_PHASES], int reset);

/* -- the multi-GPU batched mode for C and C++ callers: `count` independent images sharded over
 *    `n_devices` devices of this process, contiguous blocks by the reference's calculate_distribution rule
 *    (fft/fft_mpi.cpp:89-100 applied to images: count[g] = count / G + (g < count % G)), one host thread, one plan and
 *    one PSF spectrum per entry of `devices` (an ordinal may repeat: {0, 0} runs two workers on device 0).  No data-path
 *    collective: images are independent.
 *      imgs_host != NULL: image i is read from imgs_host[i] and its result written to outs_host[i] (rows x cols,
 *                         strides in elements) through the pipelined host batch of each worker;
 *      imgs_host == NULL: device-resident synthetic run (BASELINE config 5 shape): every worker generates its shard with
 *                         fdr_synth_image_dev (global image index = position in the batch), restores it `steps` times
 *                         (after `warmup` untimed passes) and keeps the results on its device; only statistics return.
 *    psf_host == NULL: motionBlurKernel(psf_size, psf_angle_deg) generated on each device.                            */
typedef struct fdr_batch_desc {
    int n_devices;
    const int* devices;
    int M, N;            /* plan dimensions (powers of two) */
    int mode;            /* FDR_MODE_* */
    unsigned flags;      /* FDR_FLAG_* */
    const float* psf_host;
    int psf_rows, psf_cols, psf_stride;
    int psf_size;        /* used when psf_host == NULL */
    double psf_angle_deg;
    float K;
    int count;           /* images in the batch */
    int rows, cols;      /* image size, rows <= M, cols <= N */
    int stride, out_stride;
    const float* const* imgs_host;
    float* const* outs_host;
    uint64_t synth_seed; /* synthetic run */
    int steps, warmup;   /* synthetic run: timed / untimed passes over the shard (steps >= 1) */
    int nstreams, group; /* fdr_plan_set_batching of every worker (0, 0 = defaults) */
    int norm_area;       /* FDR_NORM_* */
    int bcast_filter;    /* 0: every worker prepares the PSF spectrum / filter itself (default; cheaper than moving it);
                            1: worker 0 prepares it and the others RECEIVE its bytes -- ncclBroadcast over RCCL (xGMI) when
                            the device ordinals are distinct, device / peer copies when an ordinal repeats or RCCL cannot be
                            loaded: the MPI_Bcast / MPI_Scatterv of the padded PSF in the reference's MPI variant
                            (fft/fft_mpi.cpp:334-378).  Needs count >= n_devices.  fdr_batch_stats::filter_path says which.
                            2: the same, and the RCCL path is taken even for a single device entry (a broadcast to itself:
                            exercises the library and the call on a one-GPU machine).
                            UNEXERCISED ON HARDWARE for more than one distinct device (no multi-GPU machine has been
                            available to the builds so far): any RCCL error there falls back to the peer copies (a line
                            on stderr says so, filter_path reports FDR_FILTER_PEER_COPY) instead of failing the batch. */
} fdr_batch_desc;

#define FDR_BATCH_MAX_DEVICES 16
typedef struct fdr_batch_stats {
    int n_devices;
    int first[FDR_BATCH_MAX_DEVICES];      /* first image of worker g */
    int images[FDR_BATCH_MAX_DEVICES];     /* images of worker g (per pass) */
    double elapsed_ms[FDR_BATCH_MAX_DEVICES]; /* worker g: wall time of its timed region */
    double checksum[FDR_BATCH_MAX_DEVICES];   /* sum of worker g's restored pixels (last pass) */
    int status[FDR_BATCH_MAX_DEVICES];     /* FDR_OK or the failing status of worker g */
    double wall_ms;                        /* all workers: from the common start line (every worker has finished its set-up
                                              and warm-up and waits for the others there) to the last one's finish */
    long long images_done;                 /* sum over workers of images x passes */
    double mpixels_per_s;                  /* images_done * rows * cols / wall_ms */
    int filter_path;                       /* FDR_FILTER_*: how the workers came by their filter */
} fdr_batch_stats;
#define FDR_FILTER_LOCAL 0          /* prepared by every worker */
#define FDR_FILTER_RCCL_BROADCAST 1 /* worker 0's, by ncclBroadcast */
#define FDR_FILTER_PEER_COPY 2      /* worker 0's, by device-to-device / peer copies */
int fdr_batch_run(const fdr_batch_desc* desc, fdr_batch_stats* stats);

/* -- single-image multi-GPU mode: the reference's MPI variant splits ONE image into row slabs and
 *    transposes through MPI_Alltoallv (fft/fft_mpi.cpp:89-100 distribution, :170-279 distributed transpose, :284-307 the
 *    2-D driver rows -> transpose -> rows -> transpose).  These are the per-rank device steps of that scheme; the exchange
 *    itself belongs to the caller's communicator (RCCL all-to-all in ..._amd/slab.py).  All asynchronous on `stream`,
 *    device pointers only; `plan` supplies the twiddle tables (and the arithmetic mode) for dimensions M and N.
 *      pad       : real rows (valid_rows x valid_cols, row stride src_stride) -> rows x N complex, zero padded
 *                  (copyMakeBorder + merge of fft/fft_mpi.cpp:357-366, for the rows this rank owns)
 *      rows_fft  : `rows` contiguous transforms of length N (dim 0) or M (dim 1), in place, unscaled   (:291-294, :301-304)
 *      pack      : column blocks of a rows x ld array, block p = columns [displs[p], displs[p] + counts[p]) stored
 *                  rows x counts[p], blocks in rank order: the send buffer of :118-135; elem_size 4 (real) or 8 (complex)
 *      transpose : dense rows x cols -> cols x rows, elem_size 4 or 8                                   (:154-166)
 *      wiener    : G <- Wiener quotient of G against H, pointwise, with the plan's mode and K given     (fft_serial.cpp:186-224)
 *      real      : real part of `count` complex values
 *      minmax    : {min, max} of the window [0, mm_rows) x [0, mm_cols) of a real rows x ld plane into d_mm[2]
 *      normalize : cv::normalize(0, 1, MINMAX) with the given {min, max}, cropped to rows x cols         (fft_serial.cpp:246) */
int fdr_slab_pad_dev(const float* d_src, int valid_rows, int valid_cols, int src_stride, float* d_dst_complex, int rows, int N, void* stream);
int fdr_slab_rows_fft_dev(fdr_plan* plan, float* d_complex, int rows, int dim, int inverse, void* stream);
int fdr_slab_pack_dev(const void* d_src, int rows, int ld, int parts, const int* counts, int elem_size, void* d_dst, void* stream);
int fdr_slab_transpose_dev(const void* d_src, void* d_dst, int rows, int cols, int elem_size, void* stream);
int fdr_slab_wiener_dev(fdr_plan* plan, float* d_g, const float* d_h, size_t count, float K, void* stream);
int fdr_slab_real_dev(const float* d_complex, float* d_real, size_t count, void* stream);
int fdr_slab_minmax_dev(fdr_plan* plan, const float* d_real, int rows, int ld, int mm_rows, int mm_cols, float* d_mm, void* stream);
int fdr_slab_normalize_dev(const float* d_real, int ld, const float* d_mm, float* d_out, int rows, int cols, int out_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FDR_H */
