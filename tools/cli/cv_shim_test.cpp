// cv_shim_test -- host-only checks of the OpenCV stand-in (include/opencv2/opencv.hpp -> fdr_cv.hpp): the cv:: free functions
// the reference's drivers call that do NOT touch the device (imread / imwrite / imshow, cvtColor, norm, copyMakeBorder,
// getRotationMatrix2D, Size comparison, convertTo, split / merge).  cv::warpAffine runs on the device and is covered by the
// GPU tests.  usage: cv_shim_test <tmp dir>; prints "cv shim ok" and returns 0.
#include <opencv2/opencv.hpp>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
using namespace cv;

#define REQUIRE(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    // a small BGR picture with a gradient, through PNG and PPM and back (lossless, BGR order kept)
    Mat img(37, 53, CV_8UC3);
    for (int y = 0; y < img.rows; ++y)
        for (int x = 0; x < img.cols; ++x) {
            unsigned char* q = img.ptr<unsigned char>(y) + 3 * x;
            q[0] = (unsigned char)(x * 4); q[1] = (unsigned char)(y * 6); q[2] = (unsigned char)((x + y) * 2);
        }
    for (const char* ext : {".png", ".ppm"}) {
        const std::string path = dir + "/cv_shim_roundtrip" + ext;
        REQUIRE(imwrite(path, img));
        Mat back = imread(path, IMREAD_COLOR);
        REQUIRE(!back.empty() && back.size() == img.size() && back.type() == CV_8UC3);
        REQUIRE(!(back.size() != img.size()));
        for (int y = 0; y < img.rows; ++y) REQUIRE(std::memcmp(back.ptr<unsigned char>(y), img.ptr<unsigned char>(y), 3 * (size_t)img.cols) == 0);
        Mat gray = imread(path, IMREAD_GRAYSCALE);
        REQUIRE(gray.type() == CV_8UC1 && gray.rows == img.rows);
    }
    REQUIRE(imread(dir + "/does_not_exist.png", IMREAD_COLOR).empty());  // serial.cpp:23 relies on this
    // imshow: nothing without FDR_IMSHOW_DIR, a PNG with it
    imshow("no gui here", img);
    setenv("FDR_IMSHOW_DIR", dir.c_str(), 1);
    imshow("Deblurred Color Image", img);
    REQUIRE(!imread(dir + "/Deblurred_Color_Image.png").empty());
    REQUIRE(waitKey(0) == -1);
    // float conversions as the drivers do them (serial.cpp:24-25, :54)
    Mat f;
    img.convertTo(f, CV_32F);
    f /= 255.0;
    REQUIRE(f.type() == CV_32FC3 && std::fabs(f.ptr<float>(3)[3 * 5 + 1] - (3 * 6) / 255.0f) < 1e-6f);
    Mat u8;
    f.convertTo(u8, CV_8U, 255.0);
    for (int y = 0; y < img.rows; ++y) REQUIRE(std::memcmp(u8.ptr<unsigned char>(y), img.ptr<unsigned char>(y), 3 * (size_t)img.cols) == 0);
    // BGR -> Lab -> BGR round trip; L in [0, 100]
    Mat lab, bgr;
    cvtColor(f, lab, COLOR_BGR2Lab);
    cvtColor(lab, bgr, COLOR_Lab2BGR);
    REQUIRE(norm(f, bgr, NORM_INF) < 2e-3);
    std::vector<Mat> ch;
    split(lab, ch);
    REQUIRE(ch.size() == 3);
    for (int y = 0; y < lab.rows; ++y)
        for (int x = 0; x < lab.cols; ++x) REQUIRE(ch[0].at<float>(y, x) >= 0.f && ch[0].at<float>(y, x) <= 100.f);
    Mat white = Mat::zeros(2, 2, CV_32FC3);
    for (int i = 0; i < 12; ++i) white.ptr<float>(i / 6)[i % 6] = 1.0f;
    cvtColor(white, lab, COLOR_BGR2Lab);
    REQUIRE(std::fabs(lab.ptr<float>(0)[0] - 100.f) < 1e-2f && std::fabs(lab.ptr<float>(0)[1]) < 1e-2f && std::fabs(lab.ptr<float>(0)[2]) < 1e-2f);
    Mat merged;
    merge(ch, merged);
    cvtColor(f, lab, COLOR_BGR2Lab);
    REQUIRE(norm(merged, lab, NORM_INF) == 0.0);
    // norms (gpu.cpp:29,33)
    Mat a = Mat::zeros(3, 4, CV_32F), b = Mat::zeros(3, 4, CV_32F);
    b.at<float>(1, 2) = 3.f; b.at<float>(2, 3) = -4.f;
    REQUIRE(norm(a, b, NORM_INF) == 4.0 && norm(a, b, NORM_L1) == 7.0 && norm(a, b, NORM_L2SQR) == 25.0 && norm(a, b, NORM_L2) == 5.0);
    // copyMakeBorder as utils.hpp:44-45 calls it
    Mat padded;
    copyMakeBorder(b, padded, 0, 5, 0, 4, BORDER_CONSTANT, Scalar::all(0));
    REQUIRE(padded.rows == 8 && padded.cols == 8 && padded.at<float>(1, 2) == 3.f && padded.at<float>(7, 7) == 0.f && padded.at<float>(2, 3) == -4.f);
    // getRotationMatrix2D (utils.hpp:20): rotation by 90 degrees about (2, 2) maps (3, 2) to (2, 1) (y down)
    Mat r = getRotationMatrix2D(Point(2, 2), 90.0, 1.0);
    REQUIRE(r.rows == 2 && r.cols == 3 && r.type() == CV_64F);
    const double* m = r.ptr<double>(0);
    REQUIRE(std::fabs(m[0] * 3 + m[1] * 2 + m[2] - 2.0) < 1e-12 && std::fabs(m[3] * 3 + m[4] * 2 + m[5] - 1.0) < 1e-12);
    std::printf("cv shim ok\n");
    return 0;
}
