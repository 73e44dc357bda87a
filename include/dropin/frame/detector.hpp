// Drop-in replacement of pxl-th/slam's include/frame/detector.hpp: put this directory in front
// of the reference's include path (-I <slam_amd>/include/dropin -I <slam_amd>/include) and remove
// src/frame/detector.cpp from the build.  slam::Detector keeps the reference's interface
// (include/frame/detector.hpp:15-37) and runs on the GPU.
#ifndef DETECTOR_H
#define DETECTOR_H
#define SLAM_AMD_OPENCV_DETECTOR_ONLY
#include "slam_amd_opencv.hpp"
#undef SLAM_AMD_OPENCV_DETECTOR_ONLY
#endif
