// opencv2/opencv.hpp -- stands where OpenCV's umbrella header would be when OpenCV is not installed, so that the
// reference's sources (#include <opencv2/opencv.hpp> at utils.hpp:2, fft/fft.hpp:2, serial.cpp:3, gpu.cpp:3) compile
// unchanged with -I<this repo>/include: the bundled Mat (fdr_mat.hpp) plus the free functions the drivers call
// (fdr_cv.hpp).  Define FDR_WITH_OPENCV and put the real OpenCV ahead on the include path to use cv::Mat instead.
#pragma once
#include "../fdr_cv.hpp"
