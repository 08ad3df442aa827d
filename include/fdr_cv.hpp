// fdr_cv.hpp -- the part of OpenCV's free-function surface that the reference's drivers and utils.hpp call
// (serial.cpp:22-60, gpu.cpp:23-33,68,127-137, utils.hpp:15-47), on the bundled Mat (fdr_mat.hpp), so that the
// reference's own serial.cpp / gpu.cpp compile UNCHANGED against include/ when OpenCV is not installed
// (include/opencv2/opencv.hpp forwards here).  Link with -lfdr -lz.
//   namespace fdr_io : 8-bit PNG (gray, RGB, RGBA, non-interlaced; zlib for inflate/deflate/crc32) and binary PPM/PGM
//                      codecs, BGR <-> Lab for float images
//   namespace cv     : imread, imwrite, imshow / waitKey (no GUI: see imshow), cvtColor(COLOR_BGR2Lab | COLOR_Lab2BGR),
//                      norm(a, b, NORM_INF | NORM_L1 | NORM_L2 | NORM_L2SQR), copyMakeBorder(BORDER_CONSTANT),
//                      getRotationMatrix2D, warpAffine (on the DEVICE: fdr_warp_affine_f32, the kernel that also serves
//                      motionBlurKernel -- PSF generation is on the hot path, SURVEY.md 8a row 7, and has no CPU version here)
// With FDR_WITH_OPENCV the real library provides namespace cv and only fdr_io is defined.
#pragma once
#include "fdr.h"
#include "fdr_mat.hpp"
#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace fdr_io {

inline bool read_file(const std::string& path, std::vector<unsigned char>& buf) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    return got == buf.size() && !buf.empty();
}
inline unsigned be32(const unsigned char* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }

inline cv::Mat decode_png(const std::vector<unsigned char>& buf) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (buf.size() < 33 || std::memcmp(buf.data(), sig, 8) != 0) return cv::Mat();
    size_t pos = 8; unsigned w = 0, h = 0; int bitdepth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat;
    while (pos + 12 <= buf.size()) {
        const unsigned len = be32(&buf[pos]);
        const char* tag = reinterpret_cast<const char*>(&buf[pos + 4]);
        if (pos + 12 + len > buf.size()) return cv::Mat();
        const unsigned char* d = &buf[pos + 8];
        if (!std::memcmp(tag, "IHDR", 4)) { w = be32(d); h = be32(d + 4); bitdepth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!std::memcmp(tag, "IEND", 4)) break;
        pos += 12 + len;
    }
    if (bitdepth != 8 || interlace != 0 || w == 0 || h == 0) return cv::Mat();
    const int cn = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!cn) return cv::Mat();
    const size_t stride = (size_t)w * cn;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return cv::Mat();
    std::vector<unsigned char> img(stride * h);
    for (unsigned y = 0; y < h; ++y) {  // undo the per-row filters
        const unsigned char* in = &raw[y * (stride + 1)];
        unsigned char* cur = &img[y * stride];
        const unsigned char* up = y ? &img[(y - 1) * stride] : nullptr;
        const int ft = in[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)cn ? cur[i - cn] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)cn) ? up[i - cn] : 0;
            int pred = 0;
            if (ft == 1) pred = a; else if (ft == 2) pred = b; else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            cur[i] = (unsigned char)(in[1 + i] + pred);
        }
    }
    cv::Mat m((int)h, (int)w, cv::CV_8UC3);  // BGR, as cv::imread(IMREAD_COLOR)
    for (unsigned y = 0; y < h; ++y)
        for (unsigned x = 0; x < w; ++x) {
            const unsigned char* p = &img[y * stride + (size_t)x * cn];
            unsigned char r = p[0], g = p[0], b = p[0];
            if (cn >= 3) { g = p[1]; b = p[2]; }
            unsigned char* q = m.ptr<unsigned char>((int)y) + 3 * x;
            q[0] = b; q[1] = g; q[2] = r;
        }
    return m;
}

inline cv::Mat decode_pnm(const std::vector<unsigned char>& buf) {
    if (buf.size() < 8 || buf[0] != 'P' || (buf[1] != '6' && buf[1] != '5')) return cv::Mat();
    const int cn = buf[1] == '6' ? 3 : 1;
    size_t pos = 2; int vals[3], k = 0;
    while (k < 3 && pos < buf.size()) {
        while (pos < buf.size() && (buf[pos] == ' ' || buf[pos] == '\n' || buf[pos] == '\r' || buf[pos] == '\t')) ++pos;
        if (pos < buf.size() && buf[pos] == '#') { while (pos < buf.size() && buf[pos] != '\n') ++pos; continue; }
        int v = 0; while (pos < buf.size() && buf[pos] >= '0' && buf[pos] <= '9') v = v * 10 + (buf[pos++] - '0');
        vals[k++] = v;
    }
    ++pos;
    const int w = vals[0], h = vals[1];
    if (k < 3 || vals[2] != 255 || pos + (size_t)w * h * cn > buf.size()) return cv::Mat();
    cv::Mat m(h, w, cv::CV_8UC3);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const unsigned char* p = &buf[pos + ((size_t)y * w + x) * cn];
            unsigned char* q = m.ptr<unsigned char>(y) + 3 * x;
            if (cn == 3) { q[0] = p[2]; q[1] = p[1]; q[2] = p[0]; } else { q[0] = q[1] = q[2] = p[0]; }
        }
    return m;
}

// cv::imread(path, IMREAD_COLOR): 8-bit BGR, empty Mat on failure
inline cv::Mat imread(const std::string& path) {
    std::vector<unsigned char> buf;
    if (!read_file(path, buf)) return cv::Mat();
    cv::Mat m = decode_png(buf);
    return m.empty() ? decode_pnm(buf) : m;
}

inline void put_chunk(std::vector<unsigned char>& out, const char* tag, const unsigned char* d, unsigned len) {
    unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8), (unsigned char)len,
                            (unsigned char)tag[0], (unsigned char)tag[1], (unsigned char)tag[2], (unsigned char)tag[3]};
    out.insert(out.end(), hdr, hdr + 8);
    if (len) out.insert(out.end(), d, d + len);
    uLong c = crc32(0L, hdr + 4, 4);
    if (len) c = crc32(c, d, len);
    unsigned char tail[4] = {(unsigned char)(c >> 24), (unsigned char)(c >> 16), (unsigned char)(c >> 8), (unsigned char)c};
    out.insert(out.end(), tail, tail + 4);
}

// 8-bit BGR Mat -> .png (RGB) or .ppm by extension
inline bool imwrite(const std::string& path, const cv::Mat& bgr) {
    if (bgr.empty() || bgr.type() != cv::CV_8UC3) return false;
    const int w = bgr.cols, h = bgr.rows;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = true;
    if (path.size() > 4 && path.substr(path.size() - 4) == ".ppm") {
        std::fprintf(f, "P6\n%d %d\n255\n", w, h);
        for (int y = 0; y < h && ok; ++y)
            for (int x = 0; x < w; ++x) { const unsigned char* q = bgr.ptr<unsigned char>(y) + 3 * x; const unsigned char rgb[3] = {q[2], q[1], q[0]}; ok = std::fwrite(rgb, 1, 3, f) == 3; }
    } else {
        std::vector<unsigned char> raw(((size_t)w * 3 + 1) * h);
        for (int y = 0; y < h; ++y) {
            unsigned char* r = &raw[(size_t)y * (w * 3 + 1)];
            r[0] = 0;
            for (int x = 0; x < w; ++x) { const unsigned char* q = bgr.ptr<unsigned char>(y) + 3 * x; r[1 + 3 * x] = q[2]; r[2 + 3 * x] = q[1]; r[3 + 3 * x] = q[0]; }
        }
        uLongf clen = compressBound((uLong)raw.size());
        std::vector<unsigned char> comp(clen);
        ok = compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) == Z_OK;
        std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
        unsigned char ihdr[13] = {(unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8), (unsigned char)w,
                                  (unsigned char)(h >> 24), (unsigned char)(h >> 16), (unsigned char)(h >> 8), (unsigned char)h, 8, 2, 0, 0, 0};
        put_chunk(out, "IHDR", ihdr, 13);
        put_chunk(out, "IDAT", comp.data(), (unsigned)clen);
        put_chunk(out, "IEND", nullptr, 0);
        ok = ok && std::fwrite(out.data(), 1, out.size(), f) == out.size();
    }
    std::fclose(f);
    return ok;
}

// cv::cvtColor(COLOR_BGR2Lab / COLOR_Lab2BGR) for float images in [0,1]: sRGB companding, D65,
// L in [0,100], a/b around 0 (OpenCV's float path; version unpinned, +-1 LSB at 8 bit expected).
inline float srgb_to_lin(float c) { return c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f); }
inline float lin_to_srgb(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f; }
inline cv::Mat bgr2lab(const cv::Mat& bgr) {
    cv::Mat lab(bgr.rows, bgr.cols, cv::CV_32FC3);
    const float Xn = 0.950456f, Zn = 1.088754f;
    for (int y = 0; y < bgr.rows; ++y)
        for (int x = 0; x < bgr.cols; ++x) {
            const float* p = bgr.ptr<float>(y) + 3 * x;
            const float b = srgb_to_lin(std::fmin(std::fmax(p[0], 0.f), 1.f)), g = srgb_to_lin(std::fmin(std::fmax(p[1], 0.f), 1.f)),
                        r = srgb_to_lin(std::fmin(std::fmax(p[2], 0.f), 1.f));
            const float X = (0.412453f * r + 0.357580f * g + 0.180423f * b) / Xn, Y = 0.212671f * r + 0.715160f * g + 0.072169f * b,
                        Z = (0.019334f * r + 0.119193f * g + 0.950227f * b) / Zn;
            auto f = [](float t) { return t > 0.008856f ? std::cbrt(t) : 7.787f * t + 16.f / 116.f; };
            const float fx = f(X), fy = f(Y), fz = f(Z);
            float* q = lab.ptr<float>(y) + 3 * x;
            q[0] = Y > 0.008856f ? 116.f * fy - 16.f : 903.3f * Y; q[1] = 500.f * (fx - fy); q[2] = 200.f * (fy - fz);
        }
    return lab;
}
inline cv::Mat lab2bgr(const cv::Mat& lab) {
    cv::Mat bgr(lab.rows, lab.cols, cv::CV_32FC3);
    const float Xn = 0.950456f, Zn = 1.088754f;
    for (int y = 0; y < lab.rows; ++y)
        for (int x = 0; x < lab.cols; ++x) {
            const float* p = lab.ptr<float>(y) + 3 * x;
            const float fy = (p[0] + 16.f) / 116.f, fx = fy + p[1] / 500.f, fz = fy - p[2] / 200.f;
            auto finv = [](float t) { return t > 0.206893f ? t * t * t : (t - 16.f / 116.f) / 7.787f; };
            const float Y = p[0] > 7.9996f ? fy * fy * fy : p[0] / 903.3f;
            const float X = finv(fx) * Xn, Z = finv(fz) * Zn;
            const float r = 3.240479f * X - 1.537150f * Y - 0.498535f * Z, g = -0.969256f * X + 1.875991f * Y + 0.041556f * Z,
                        b = 0.055648f * X - 0.204043f * Y + 1.057311f * Z;
            float* q = bgr.ptr<float>(y) + 3 * x;
            q[0] = lin_to_srgb(std::fmin(std::fmax(b, 0.f), 1.f)); q[1] = lin_to_srgb(std::fmin(std::fmax(g, 0.f), 1.f));
            q[2] = lin_to_srgb(std::fmin(std::fmax(r, 0.f), 1.f));
        }
    return bgr;
}

}  // namespace fdr_io

#ifndef FDR_WITH_OPENCV
#include <cstdlib>
namespace cv {

#ifndef CV_Assert
#define CV_Assert(expr) do { if (!(expr)) { std::fprintf(stderr, "Error: %s:%d, assertion failed: %s\n", __FILE__, __LINE__, #expr); std::exit(1); } } while (0)
#endif

// serial.cpp:22 / gpu.cpp:68: 8-bit BGR (IMREAD_COLOR) or gray; empty Mat when the file cannot be read or decoded
inline Mat imread(const std::string& path, int flags = IMREAD_COLOR) {
    Mat bgr = fdr_io::imread(path);
    if (bgr.empty() || flags != IMREAD_GRAYSCALE) return bgr;
    Mat g(bgr.rows, bgr.cols, CV_8UC1);  // OpenCV's BGR -> gray weights, 8 bit
    for (int y = 0; y < bgr.rows; ++y)
        for (int x = 0; x < bgr.cols; ++x) {
            const unsigned char* q = bgr.ptr<unsigned char>(y) + 3 * x;
            g.ptr<unsigned char>(y)[x] = (unsigned char)std::lrint(0.114 * q[0] + 0.587 * q[1] + 0.299 * q[2]);
        }
    return g;
}
inline bool imwrite(const std::string& path, const Mat& bgr8) { return fdr_io::imwrite(path, bgr8); }

// serial.cpp:59-60, gpu.cpp:136-137.  There is no GUI here: imshow writes "<FDR_IMSHOW_DIR>/<window name>.png" when that
// environment variable is set (spaces in the name become '_') and does nothing otherwise; waitKey returns at once.
inline void imshow(const std::string& winname, const Mat& img) {
    const char* dir = std::getenv("FDR_IMSHOW_DIR");
    if (!dir || !*dir || img.empty()) return;
    std::string name = winname;
    for (char& c : name) if (c == ' ' || c == '/') c = '_';
    Mat u8 = img;
    if (img.depth() != CV_8U) img.convertTo(u8, CV_8U, 255.0);
    if (u8.channels() == 1) { std::vector<Mat> planes; Mat f; u8.convertTo(f, CV_32F); planes.assign(3, f); Mat m; merge(planes, m); m.convertTo(u8, CV_8U); }
    (void)fdr_io::imwrite(std::string(dir) + "/" + name + ".png", u8);
}
inline int waitKey(int = 0) { return -1; }

// serial.cpp:47-53, gpu.cpp:127-133: float BGR in [0,1] <-> Lab (L in [0,100])
inline void cvtColor(const Mat& src, Mat& dst, int code) {
    CV_Assert(src.type() == CV_32FC3 && (code == COLOR_BGR2Lab || code == COLOR_Lab2BGR));
    Mat out = code == COLOR_BGR2Lab ? fdr_io::bgr2lab(src) : fdr_io::lab2bgr(src);
    dst = out;
}

// gpu.cpp:29,33: cv::norm(a, b, NORM_INF) and NORM_L2SQR (also L1 / L2) of the difference, accumulated in double
inline double norm(const Mat& a, const Mat& b, int normType = NORM_L2) {
    CV_Assert(a.rows == b.rows && a.cols == b.cols && a.type() == b.type() && a.depth() == CV_32F);
    double mx = 0.0, s1 = 0.0, s2 = 0.0;
    const int n = a.cols * a.channels();
    for (int r = 0; r < a.rows; ++r) {
        const float *pa = a.ptr<float>(r), *pb = b.ptr<float>(r);
        for (int i = 0; i < n; ++i) {
            const double d = std::fabs((double)pa[i] - (double)pb[i]);
            if (d > mx) mx = d;
            s1 += d; s2 += d * d;
        }
    }
    switch (normType) {
        case NORM_INF: return mx;
        case NORM_L1: return s1;
        case NORM_L2SQR: return s2;
        default: return std::sqrt(s2);
    }
}

// utils.hpp:44-45: constant border (the drivers pad bottom / right with zeros)
inline void copyMakeBorder(const Mat& src, Mat& dst, int top, int bottom, int left, int right, int borderType = BORDER_CONSTANT,
                           const Scalar& value = Scalar()) {
    CV_Assert(borderType == BORDER_CONSTANT && top >= 0 && bottom >= 0 && left >= 0 && right >= 0);
    Mat out(src.rows + top + bottom, src.cols + left + right, src.type());
    const int cn = src.channels();
    const size_t es = src.elemSize();
    for (int r = 0; r < out.rows; ++r)
        for (int c = 0; c < out.cols; ++c)
            for (int k = 0; k < cn; ++k) {
                if (src.depth() == CV_32F) out.ptr<float>(r)[c * cn + k] = (float)value[k < 4 ? k : 3];
                else out.ptr<unsigned char>(r)[c * cn + k] = (unsigned char)value[k < 4 ? k : 3];
            }
    for (int r = 0; r < src.rows; ++r) std::memcpy(out.data + (size_t)(r + top) * out.step + (size_t)left * es, src.data + (size_t)r * src.step, (size_t)src.cols * es);
    dst = out;
}

// utils.hpp:20: 2 x 3 CV_64F matrix of a rotation by `angle` degrees (counter-clockwise, y down) about `center`
inline Mat getRotationMatrix2D(Point2f center, double angle, double scale) {
    const double a = angle * CV_PI / 180.0;
    const double alpha = std::cos(a) * scale, beta = std::sin(a) * scale;
    Mat M(2, 3, CV_64F);
    double* m = M.ptr<double>(0);
    m[0] = alpha; m[1] = beta; m[2] = (1 - alpha) * center.x - beta * center.y;
    m[3] = -beta; m[4] = alpha; m[5] = beta * center.x + (1 - alpha) * center.y;
    return M;
}

// utils.hpp:22: cv::warpAffine with its defaults (INTER_LINEAR, BORDER_CONSTANT 0), single-channel float.  Runs on the
// device (fdr_warp_affine_f32: OpenCV's classic fixed-point bilinear path); print-and-exit on failure like CHECK_CUDA.
inline void warpAffine(const Mat& src, Mat& dst, const Mat& M, Size dsize, int flags = INTER_LINEAR, int borderMode = BORDER_CONSTANT,
                       const Scalar& = Scalar()) {
    CV_Assert(src.type() == CV_32F && M.rows == 2 && M.cols == 3 && flags == INTER_LINEAR && borderMode == BORDER_CONSTANT);
    double m[6];
    for (int i = 0; i < 6; ++i) m[i] = M.depth() == CV_64F ? M.ptr<double>(i / 3)[i % 3] : (double)M.ptr<float>(i / 3)[i % 3];
    Mat s = src.isContinuous() ? src : src.clone();
    Mat out(dsize.height, dsize.width, CV_32F);
    if (fdr_warp_affine_f32(s.ptr<float>(0), s.rows, s.cols, s.cols, m, out.ptr<float>(0), out.rows, out.cols, out.cols) != FDR_OK) {
        std::fprintf(stderr, "Error: %s:%d, %s\n", __FILE__, __LINE__, fdr_last_error());
        std::exit(1);
    }
    dst = out;
}

}  // namespace cv
#endif  // FDR_WITH_OPENCV
