// TEST-ONLY stand-in, declarations without definitions (see core.hpp in this directory): cv::imread.
#ifndef SLAM_AMD_TEST_CV_STANDIN_IMGCODECS_HPP
#define SLAM_AMD_TEST_CV_STANDIN_IMGCODECS_HPP

#include <string>

#include "core.hpp"

namespace cv {
enum ImreadModes { IMREAD_GRAYSCALE = 0, IMREAD_COLOR = 1 };
Mat imread(const std::string& filename, int flags = IMREAD_COLOR);
}  // namespace cv

#endif
