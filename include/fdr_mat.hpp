// fdr_mat.hpp -- the small dense-matrix container the drop-in C++ surface uses when OpenCV is not
// available at build time (define FDR_WITH_OPENCV to use cv::Mat instead).  It carries exactly the
// subset of cv::Mat that the reference's drivers touch (serial.cpp / gpu.cpp / utils.hpp): rows, cols,
// type(), ptr<T>(r), at<T>(r,c), isContinuous(), clone(), operator()(Rect), zeros, convertTo, /=, split,
// merge, mean, min/max with a scalar, scaling.  Element storage is float (CV_32F, 1..3 channels) or
// unsigned char (CV_8U, 1..3 channels), row-major, reference counted.
#pragma once
#ifdef FDR_WITH_OPENCV
#include <opencv2/opencv.hpp>
#else
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace cv {

enum { CV_8U = 0, CV_32F = 5, CV_64F = 6 };
constexpr int CV_MAKETYPE(int depth, int cn) { return depth + ((cn - 1) << 3); }
enum { CV_8UC1 = 0, CV_8UC3 = 16, CV_32FC1 = 5, CV_32FC2 = 13, CV_32FC3 = 21, CV_64FC1 = 6 };
// the enumerators of cv:: the reference's drivers and utils.hpp name (values as in OpenCV 4)
enum { IMREAD_GRAYSCALE = 0, IMREAD_COLOR = 1 };
enum { COLOR_BGR2Lab = 44, COLOR_Lab2BGR = 56 };
enum { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4, NORM_L2SQR = 5, NORM_MINMAX = 32 };
enum { BORDER_CONSTANT = 0 };
enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };
constexpr double CV_PI = 3.1415926535897932384626433832795;

struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size& o) const { return !(*this == o); }
};
struct Rect { int x = 0, y = 0, width = 0, height = 0; Rect() {} Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {} };
struct Point { int x = 0, y = 0; Point() {} Point(int x_, int y_) : x(x_), y(y_) {} };
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} Point2f(const Point& p) : x((float)p.x), y((float)p.y) {} };
struct Vec2f { float v[2]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
struct Vec3f { float v[3]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
struct Vec3b { unsigned char v[3]; unsigned char& operator[](int i) { return v[i]; } const unsigned char& operator[](int i) const { return v[i]; } };
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {} double operator[](int i) const { return v[i]; } static Scalar all(double a) { return Scalar(a, a, a, a); } };

class Mat {
public:
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;
    size_t step = 0;  // bytes per row

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    static Mat zeros(int r, int c, int type) { Mat m(r, c, type); if (m.data) std::memset(m.data, 0, m.step * r); return m; }
    static Mat zeros(Size s, int type) { return zeros(s.height, s.width, type); }

    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type;
        step = (size_t)c * elemSize();
        store_ = std::shared_ptr<unsigned char>(new unsigned char[step * (size_t)(r > 0 ? r : 0) + 16], std::default_delete<unsigned char[]>());
        data = store_.get();
    }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { return (size_t)channels() * (depth() == CV_64F ? 8 : depth() == CV_32F ? 4 : 1); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Size size() const { return Size(cols, rows); }
    size_t total() const { return (size_t)rows * cols; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }

    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
    template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }

    Mat operator()(const Rect& roi) const {  // view sharing storage
        Mat m; m.rows = roi.height; m.cols = roi.width; m.type_ = type_; m.step = step; m.store_ = store_;
        m.data = data + (size_t)roi.y * step + (size_t)roi.x * elemSize();
        return m;
    }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * elemSize());
        return m;
    }
    // dst = saturate(src * alpha + beta) with a depth change (float <-> uchar), channels kept
    void convertTo(Mat& dst, int rtype, double alpha = 1.0, double beta = 0.0) const {
        const int cn = channels();
        Mat out(rows, cols, CV_MAKETYPE(rtype & 7, cn));
        for (int r = 0; r < rows; ++r)
            for (int i = 0; i < cols * cn; ++i) {
                const double v = (depth() == CV_32F ? (double)ptr<float>(r)[i] : (double)ptr<unsigned char>(r)[i]) * alpha + beta;
                if ((rtype & 7) == CV_32F) out.ptr<float>(r)[i] = (float)v;
                else { const long q = std::lrint(v); out.ptr<unsigned char>(r)[i] = (unsigned char)(q < 0 ? 0 : (q > 255 ? 255 : q)); }
            }
        dst = out;
    }
    Mat& operator/=(double d) { scale((float)(1.0 / d), true, d); return *this; }
    Mat operator*(double g) const { Mat m = clone(); m.scale((float)g, false, 1.0); return m; }

private:
    void scale(float f, bool divide, double d) {
        assert(depth() == CV_32F);
        for (int r = 0; r < rows; ++r)
            for (int i = 0; i < cols * channels(); ++i) ptr<float>(r)[i] = divide ? (float)(ptr<float>(r)[i] / d) : ptr<float>(r)[i] * f;
    }
    int type_ = 0;
    std::shared_ptr<unsigned char> store_;
};

inline void split(const Mat& src, std::vector<Mat>& planes) {
    const int cn = src.channels();
    planes.assign(cn, Mat());
    for (int k = 0; k < cn; ++k) planes[k].create(src.rows, src.cols, CV_32F);
    for (int r = 0; r < src.rows; ++r)
        for (int c = 0; c < src.cols; ++c)
            for (int k = 0; k < cn; ++k) planes[k].at<float>(r, c) = src.ptr<float>(r)[c * cn + k];
}
inline void merge(const std::vector<Mat>& planes, Mat& dst) {
    const int cn = (int)planes.size();
    Mat out(planes[0].rows, planes[0].cols, CV_MAKETYPE(CV_32F, cn));
    for (int r = 0; r < out.rows; ++r)
        for (int c = 0; c < out.cols; ++c)
            for (int k = 0; k < cn; ++k) out.ptr<float>(r)[c * cn + k] = planes[k].at<float>(r, c);
    dst = out;
}
inline Scalar mean(const Mat& m) {
    double s = 0;
    for (int r = 0; r < m.rows; ++r)
        for (int c = 0; c < m.cols; ++c) s += m.at<float>(r, c);
    return Scalar(m.total() ? s / (double)m.total() : 0.0);
}
inline void min(const Mat& a, double v, Mat& dst) { Mat o = a.clone(); for (int r = 0; r < o.rows; ++r) for (int c = 0; c < o.cols; ++c) if (o.at<float>(r, c) > (float)v) o.at<float>(r, c) = (float)v; dst = o; }
inline void max(const Mat& a, double v, Mat& dst) { Mat o = a.clone(); for (int r = 0; r < o.rows; ++r) for (int c = 0; c < o.cols; ++c) if (o.at<float>(r, c) < (float)v) o.at<float>(r, c) = (float)v; dst = o; }

}  // namespace cv
#endif  // FDR_WITH_OPENCV
