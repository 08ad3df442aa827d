/*
 * fdr_oracle.c -- CPU restatement of the reference's serial restoration path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under the product package may include, link,
 * dlopen or execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED.  The reference (a) ships no golden vectors, known-answer tests or
 * expected outputs, and (b) cannot be built in this image: every
 * translation unit includes <opencv2/opencv.hpp> (fft/fft.hpp:2, utils.hpp:2) and OpenCV
 * is absent; writing a stand-in header is not a reference build.  This file is therefore
 * a line-by-line restatement of the reference *source text*, cross-checked only against an
 * independent float64 model (tests/test_oracle.py), not against reference outputs.
 * Third-party arithmetic restated here (OpenCV 4.x, version not pinned by the reference
 * Makefile:3-4): cv::magnitude, Mat::mul, operator/, cv::normalize, cv::getOptimalDFTSize,
 * cv::getRotationMatrix2D, cv::warpAffine (classic fixed-point bilinear path).
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off (no -march / -mfma / -ffast-math): the
 * reference's MODE=serial flags are `-std=c++17 -O2` on baseline x86-64 (Makefile:3,57-59),
 * i.e. SSE2 scalar float arithmetic with every product and sum rounded separately.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define FDR_PI 3.1415926535897932384626433832795 /* CV_PI */

/* ---- utils.hpp:27-37 nextPowerOfTwo / getNextPowerOf2 ; :50-52 isPowerOfTwo ---- */
int fdr_oracle_next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
int fdr_oracle_is_pow2(int n) { return n > 0 && ((n & (n - 1)) == 0); }

/* ---- cv::getOptimalDFTSize (used at fft/fft_serial.cpp:153-154): smallest 2^a*3^b*5^c >= n ---- */
int fdr_oracle_optimal_dft_size(int n) {
    if (n <= 1) return n < 0 ? -1 : 1;
    long best = -1;
    for (long p5 = 1; p5 < 2L * n; p5 *= 5)
        for (long p3 = p5; p3 < 2L * n; p3 *= 3) {
            long v = p3;
            while (v < n) v *= 2;
            if (best < 0 || v < best) best = v;
        }
    return (int)best;
}

/* complex<float> product as libstdc++/GCC evaluate it without -ffast-math and without FMA:
 * (ar*br - ai*bi, ar*bi + ai*br), each product and each sum rounded to float.            */
static inline void cmulf(float ar, float ai, float br, float bi, float* cr, float* ci) {
    float ac = ar * br, bd = ai * bi, ad = ar * bi, bc = ai * br;
    *cr = ac - bd;
    *ci = ad + bc;
}

/* ---- fft/fft_serial.cpp:40-68  fft_radix2_inplace ----
 * a: n interleaved (re,im) floats, n a power of two; unscaled in both directions.       */
void fdr_oracle_fft_radix2(float* a, int n, int inverse) {
    if (n <= 1) return;
    /* :45-51 bit reversal permutation by incremental counter */
    int j = 0;
    for (int i = 1; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            float tr = a[2 * i], ti = a[2 * i + 1];
            a[2 * i] = a[2 * j]; a[2 * i + 1] = a[2 * j + 1];
            a[2 * j] = tr; a[2 * j + 1] = ti;
        }
    }
    /* :53-66 butterflies; twiddle by float recurrence w *= wlen */
    for (int len = 2; len <= n; len <<= 1) {
        /* :54 `float ang = 2.0f * CV_PI / len * (inverse ? 1.0f : -1.0f);` -- double expression, rounded to float */
        float ang = (float)((double)2.0f * FDR_PI / (double)len * (double)(inverse ? 1.0f : -1.0f));
        float wlr = cosf(ang), wli = sinf(ang); /* :55 std::cos(float)/std::sin(float) */
        int half = len / 2;
        for (int i = 0; i < n; i += len) {
            float wr = 1.0f, wi = 0.0f;
            for (int k = 0; k < half; ++k) {
                float ur = a[2 * (i + k)], ui = a[2 * (i + k) + 1];
                float vr, vi;
                cmulf(a[2 * (i + k + half)], a[2 * (i + k + half) + 1], wr, wi, &vr, &vi);
                a[2 * (i + k)] = ur + vr;
                a[2 * (i + k) + 1] = ui + vi;
                a[2 * (i + k + half)] = ur - vr;
                a[2 * (i + k + half) + 1] = ui - vi;
                float nr, ni;
                cmulf(wr, wi, wlr, wli, &nr, &ni); /* :63 w *= wlen */
                wr = nr; wi = ni;
            }
        }
    }
}

/* The per-stage twiddles the recurrence above produces: out holds n-1 complex values,
 * stage len (2,4,..,n) at offset len/2-1, len/2 entries each.  Used by tests to check the
 * table the HIP library builds for its parity mode.                                       */
void fdr_oracle_twiddle_recurrence(int n, int inverse, float* out) {
    for (int len = 2; len <= n; len <<= 1) {
        float ang = (float)((double)2.0f * FDR_PI / (double)len * (double)(inverse ? 1.0f : -1.0f));
        float wlr = cosf(ang), wli = sinf(ang);
        float wr = 1.0f, wi = 0.0f;
        float* t = out + 2 * (len / 2 - 1);
        for (int k = 0; k < len / 2; ++k) {
            t[2 * k] = wr; t[2 * k + 1] = wi;
            float nr, ni;
            cmulf(wr, wi, wlr, wli, &nr, &ni);
            wr = nr; wi = ni;
        }
    }
}

/* ---- fft/fft_serial.cpp:71-87  dft_naive_inplace (O(n^2), arbitrary n) ---- */
void fdr_oracle_dft_naive(float* a, int n, int inverse) {
    if (n <= 1) return;
    float* out = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    const float sign = inverse ? 1.0f : -1.0f;
    for (int k = 0; k < n; ++k) {
        float sr = 0.f, si = 0.f;
        for (int t = 0; t < n; ++t) {
            /* :79 `2.0f * CV_PI * k * t / n * sign`: left-to-right in double, k,t,n ints promoted one at a time */
            float ang = (float)((double)2.0f * FDR_PI * (double)k * (double)t / (double)n * (double)sign);
            float wr = cosf(ang), wi = sinf(ang);
            float pr, pi;
            cmulf(a[2 * t], a[2 * t + 1], wr, wi, &pr, &pi);
            sr += pr; si += pi;
        }
        out[2 * k] = sr; out[2 * k + 1] = si;
    }
    memcpy(a, out, sizeof(float) * 2 * (size_t)n);
    free(out);
}

/* ---- fft/fft_serial.cpp:90-108  transform_row_inplace ---- */
void fdr_oracle_transform_row(float* row, int n, int inverse) {
    if (fdr_oracle_is_pow2(n)) fdr_oracle_fft_radix2(row, n, inverse);
    else fdr_oracle_dft_naive(row, n, inverse);
}

static void transpose_c(const float* src, float* dst, int rows, int cols) {
    const int B = 32;
    for (int r0 = 0; r0 < rows; r0 += B)
        for (int c0 = 0; c0 < cols; c0 += B) {
            int r1 = r0 + B < rows ? r0 + B : rows, c1 = c0 + B < cols ? c0 + B : cols;
            for (int r = r0; r < r1; ++r)
                for (int c = c0; c < c1; ++c) {
                    dst[2 * ((size_t)c * rows + r)] = src[2 * ((size_t)r * cols + c)];
                    dst[2 * ((size_t)c * rows + r) + 1] = src[2 * ((size_t)r * cols + c) + 1];
                }
        }
}

/* ---- fft/fft_serial.cpp:113-139  my_dft2D: rows, transpose, rows, transpose back ---- */
void fdr_oracle_dft2d(float* data, int M, int N, int inverse) {
    for (int r = 0; r < M; ++r) fdr_oracle_transform_row(data + 2 * (size_t)r * N, N, inverse);
    float* t = (float*)malloc(sizeof(float) * 2 * (size_t)M * N);
    transpose_c(data, t, M, N);
    for (int r = 0; r < N; ++r) fdr_oracle_transform_row(t + 2 * (size_t)r * M, M, inverse);
    transpose_c(t, data, N, M);
    free(t);
}

/* ---- cv::normalize(src, dst, 0, 1, NORM_MINMAX) as used at fft/fft_serial.cpp:246 ----
 * OpenCV 4.x: minMaxIdx in double; scale = (dmax-dmin)*(smax-smin > DBL_EPSILON ? 1/(smax-smin) : 0);
 * for CV_32F output scale is rounded to float and shift = (float)dmin - (float)(smin*scale);
 * convertTo applies dst = src*scale + shift in float.  (OpenCV version unpinned: parity unpinned.) */
void fdr_oracle_minmax_scale_shift(double smin, double smax, float* scale_out, float* shift_out) {
    const double dmin = 0.0, dmax = 1.0;
    double scale = (dmax - dmin) * ((smax - smin) > DBL_EPSILON ? 1.0 / (smax - smin) : 0.0);
    scale = (double)(float)scale;
    double shift = (double)((float)dmin - (float)(smin * scale));
    *scale_out = (float)scale;
    *shift_out = (float)shift;
}

void fdr_oracle_normalize_minmax(float* a, size_t n, float* min_out, float* max_out) {
    if (n == 0) return;
    float mn = a[0], mx = a[0];
    for (size_t i = 1; i < n; ++i) { if (a[i] < mn) mn = a[i]; if (a[i] > mx) mx = a[i]; }
    float scale, shift;
    fdr_oracle_minmax_scale_shift((double)mn, (double)mx, &scale, &shift);
    for (size_t i = 0; i < n; ++i) { float p = a[i] * scale; a[i] = p + shift; }
    if (min_out) *min_out = mn;
    if (max_out) *max_out = mx;
}

/* ---- fft/fft_serial.cpp:141-261  wienerDeblur_myfft ----
 * img rows x cols (stride = cols), psf prows x pcols, out rows x cols in [0,1].
 * spectrum_out (optional, 2*optRows*optCols floats): the Wiener quotient before the IFFT.
 * raw_out (optional, optRows*optCols floats): real plane after the unscaled IFFT.          */
int fdr_oracle_wiener(const float* img, int rows, int cols, const float* psf, int prows, int pcols,
                      float K, float* out, float* spectrum_out, float* raw_out) {
    int M = fdr_oracle_optimal_dft_size(rows), N = fdr_oracle_optimal_dft_size(cols); /* :153-154 */
    if (prows > M || pcols > N) return -1; /* copyMakeBorder would throw on a negative border */
    size_t P = (size_t)M * N;
    float* G = (float*)calloc(2 * P, sizeof(float));
    float* H = (float*)calloc(2 * P, sizeof(float));
    if (!G || !H) { free(G); free(H); return -2; }
    /* :157-171 zero-pad bottom/right, zero imaginary plane; PSF anchored top-left */
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) G[2 * ((size_t)r * N + c)] = img[(size_t)r * cols + c];
    for (int r = 0; r < prows; ++r)
        for (int c = 0; c < pcols; ++c) H[2 * ((size_t)r * N + c)] = psf[(size_t)r * pcols + c];
    fdr_oracle_dft2d(G, M, N, 0); /* :176 */
    fdr_oracle_dft2d(H, M, N, 0); /* :182 */
    /* :186-224 Wiener quotient, whole-Mat OpenCV expression order */
    for (size_t i = 0; i < P; ++i) {
        float hr = H[2 * i], hi = H[2 * i + 1];
        float gr = G[2 * i], gi = G[2 * i + 1];
        float hr2 = hr * hr, hi2 = hi * hi;
        float mag = sqrtf(hr2 + hi2);     /* :195 cv::magnitude */
        float mag2 = mag * mag;           /* :196 mag2.mul(mag2) */
        float denom = mag2 + K;           /* :197 */
        float chi = -hi;                  /* :200 conj */
        float p0 = gr * hr, p1 = gi * chi, p2 = gr * chi, p3 = gi * hr;
        float nr = p0 - p1;               /* :210 */
        float ni = p2 + p3;               /* :211 */
        /* :220-221 Mat / Mat: cv::divide yields 0 where the divisor is 0 */
        G[2 * i] = denom != 0.0f ? nr / denom : 0.0f;
        G[2 * i + 1] = denom != 0.0f ? ni / denom : 0.0f;
    }
    if (spectrum_out) memcpy(spectrum_out, G, sizeof(float) * 2 * P);
    fdr_oracle_dft2d(G, M, N, 1); /* :229 unscaled inverse */
    if (raw_out) for (size_t i = 0; i < P; ++i) raw_out[i] = G[2 * i];
    /* :236-247 real plane, crop to the *input* size, normalise */
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) out[(size_t)r * cols + c] = G[2 * ((size_t)r * N + c)];
    fdr_oracle_normalize_minmax(out, (size_t)rows * cols, NULL, NULL);
    free(G); free(H);
    return 0;
}

/* ---- serial.cpp:34-39: autoPadToPowerOfTwo -> wienerDeblur_myfft -> crop ----
 * (normalisation therefore spans the padded power-of-two area)            */
int fdr_oracle_serial_channel(const float* img, int rows, int cols, const float* psf, int prows, int pcols,
                              float K, float* out) {
    int M = fdr_oracle_next_pow2(rows), N = fdr_oracle_next_pow2(cols); /* utils.hpp:40-47 */
    float* padded = (float*)calloc((size_t)M * N, sizeof(float));
    float* res = (float*)malloc(sizeof(float) * (size_t)M * N);
    if (!padded || !res) { free(padded); free(res); return -2; }
    for (int r = 0; r < rows; ++r) memcpy(padded + (size_t)r * N, img + (size_t)r * cols, sizeof(float) * cols);
    int rc = fdr_oracle_wiener(padded, M, N, psf, prows, pcols, K, res, NULL, NULL);
    if (rc == 0)
        for (int r = 0; r < rows; ++r) memcpy(out + (size_t)r * cols, res + (size_t)r * N, sizeof(float) * cols);
    free(padded); free(res);
    return rc;
}

/* cvRound: round half to even (lrint under the default rounding mode), saturating to int */
static int cv_round(double v) {
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)lrint(v);
}

/* ---- utils.hpp:15-24 motionBlurKernel(size, angle) ----
 * OpenCV pieces restated: getRotationMatrix2D, invertAffineTransform inside warpAffine,
 * WarpAffineInvoker's 10-bit fixed-point coordinates (INTER_BITS=5, INTER_TAB_SIZE=32),
 * remapBilinear<float> with the 32x32 float weight table and BORDER_CONSTANT 0.
 * out: size x size floats.  Not renormalised (the reference has no `/= sum`).            */
void fdr_oracle_motion_blur_kernel(int size, double angle, float* out) {
    size_t n = (size_t)size * size;
    float* kernel = (float*)calloc(n, sizeof(float));
    int cx = size / 2, cy = size / 2;
    for (int i = 0; i < size; ++i) kernel[(size_t)cy * size + i] = (float)(1.0 / size); /* :18-19 */
    /* getRotationMatrix2D(center, angle, 1) */
    double a = angle * FDR_PI / 180.0;
    double alpha = cos(a), beta = sin(a);
    double cxf = (double)(float)cx, cyf = (double)(float)cy;
    double Mx[6] = { alpha, beta, (1 - alpha) * cxf - beta * cyf, -beta, alpha, beta * cxf + (1 - alpha) * cyf };
    /* warpAffine: invert (dst -> src map) */
    double D = Mx[0] * Mx[4] - Mx[1] * Mx[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = Mx[4] * D, A22 = Mx[0] * D;
    Mx[0] = A11; Mx[1] *= -D; Mx[3] *= -D; Mx[4] = A22;
    double b1 = -Mx[0] * Mx[2] - Mx[1] * Mx[5];
    double b2 = -Mx[3] * Mx[2] - Mx[4] * Mx[5];
    Mx[2] = b1; Mx[5] = b2;
    const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, TAB = 32;
    const int round_delta = AB_SCALE / TAB / 2; /* 16 */
    const float tscale = 1.f / TAB;
    for (int y = 0; y < size; ++y) {
        int X0 = cv_round((Mx[1] * y + Mx[2]) * AB_SCALE) + round_delta;
        int Y0 = cv_round((Mx[4] * y + Mx[5]) * AB_SCALE) + round_delta;
        for (int x = 0; x < size; ++x) {
            int adelta = cv_round(Mx[0] * x * AB_SCALE), bdelta = cv_round(Mx[3] * x * AB_SCALE);
            int X = (X0 + adelta) >> (AB_BITS - INTER_BITS);
            int Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
            int sx = X >> INTER_BITS, sy = Y >> INTER_BITS; /* saturate_cast<short>: no-op at these sizes */
            int ax = X & (TAB - 1), ay = Y & (TAB - 1);
            float fx = ax * tscale, fy = ay * tscale;
            float vx0 = 1.f - fx, vx1 = fx, vy0 = 1.f - fy, vy1 = fy;
            float w0 = vy0 * vx0, w1 = vy0 * vx1, w2 = vy1 * vx0, w3 = vy1 * vx1;
            float s00 = 0, s01 = 0, s10 = 0, s11 = 0;
            if (sy >= 0 && sy < size) {
                if (sx >= 0 && sx < size) s00 = kernel[(size_t)sy * size + sx];
                if (sx + 1 >= 0 && sx + 1 < size) s01 = kernel[(size_t)sy * size + sx + 1];
            }
            if (sy + 1 >= 0 && sy + 1 < size) {
                if (sx >= 0 && sx < size) s10 = kernel[(size_t)(sy + 1) * size + sx];
                if (sx + 1 >= 0 && sx + 1 < size) s11 = kernel[(size_t)(sy + 1) * size + sx + 1];
            }
            float t0 = s00 * w0, t1 = s01 * w1, t2 = s10 * w2, t3 = s11 * w3;
            float acc = t0 + t1; acc = acc + t2; acc = acc + t3;
            out[(size_t)y * size + x] = acc;
        }
    }
    free(kernel);
}

/* Counter-based synthetic image: pixel i of image b = top 24 bits of
 * splitmix64(seed + b*P + i) scaled to [0,1).  Same bits on CPU and GPU, no transfer.   */
static unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
void fdr_oracle_synth_image(unsigned long long seed, unsigned long long first_index, size_t count, float* out) {
    for (size_t i = 0; i < count; ++i)
        out[i] = (float)(splitmix64(seed + first_index + i) >> 40) * (1.0f / 16777216.0f);
}
