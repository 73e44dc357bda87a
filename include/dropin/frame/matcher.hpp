// Drop-in replacement of pxl-th/slam's include/frame/matcher.hpp (see detector.hpp beside this
// file); remove src/frame/matcher.cpp from the build.  The reference's headers include each other
// by file name relative to their own directory, which finds the reference's copies whatever the
// include path says; force this file in front of every translation unit instead
// (-include frame/matcher.hpp): it claims both include guards, so the reference's detector.hpp and
// matcher.hpp become empty.  The frame / keyframe includes are the reference's own
// (include/frame/matcher.hpp:10-11): slam::KeyFrame must be complete before slam::Matcher's
// inline methods.
#ifndef MATCHER_H
#define MATCHER_H
#include "detector.hpp"
#include "frame/frame.hpp"
#include "map/keyframe.hpp"
#include "slam_amd_opencv.hpp"
#endif
