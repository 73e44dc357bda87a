// TEST-ONLY stand-in, declarations without definitions (see core.hpp in this directory): the drawing and
// filtering calls tests/cpp/opencv_diff.cpp makes.
#ifndef SLAM_AMD_TEST_CV_STANDIN_IMGPROC_HPP
#define SLAM_AMD_TEST_CV_STANDIN_IMGPROC_HPP

#include "core.hpp"

namespace cv {
void circle(InputOutputArray img, Point center, int radius, const Scalar& color, int thickness = 1);
void rectangle(InputOutputArray img, Rect rec, const Scalar& color, int thickness = 1);
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0);
}  // namespace cv

#endif
