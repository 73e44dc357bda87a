// Syntax check of the drop-in against the reference's OWN headers (only where the reference tree
// is present; nothing is linked or run): include/map/keyframe.hpp, include/map/mappoint.hpp and
// include/frame/frame.hpp of pxl-th/slam are compiled with include/dropin/frame/matcher.hpp forced
// in front (-include frame/matcher.hpp), so slam::Detector / slam::Matcher are this library's and
// everything else is the reference's.  The functions below are the reference's call shapes.
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>

#include "frame/frame.hpp"
#include "frame/matcher.hpp"
#include "map/keyframe.hpp"
#include "map/mappoint.hpp"

namespace {

using KF = std::shared_ptr<slam::KeyFrame>;

struct TrackerLike { slam::Matcher matcher; };                                   // tracker.hpp:29
struct MapperLike { slam::Matcher matcher; MapperLike(slam::Matcher m) : matcher(m) {} };   // mapper.cpp:17

[[maybe_unused]] std::shared_ptr<slam::Frame> pack_image(std::shared_ptr<cv::Mat> image, std::shared_ptr<slam::Detector> detector,
                                                         std::shared_ptr<cv::Mat> K, std::shared_ptr<cv::Mat> D) {
    return std::make_shared<slam::Frame>(image, detector, K, D);                 // tracker.cpp:147
}

[[maybe_unused]] size_t call_shapes(TrackerLike& tracker, MapperLike& mapper, const KF& last, const KF& cur) {
    size_t n = 0;
    n += tracker.matcher.mappointsFrameMatch(last, cur, 300, 50).size();          // tracker.cpp:74
    n += tracker.matcher.mappointsFrameMatch(last, cur, 300, -1, -1).size();      // tracker.cpp:77
    n += tracker.matcher.projectionMatch(last, cur, 300, 50).size();              // tracker.cpp:84, :96
    n += tracker.matcher.projectionMatch(last, cur, 300, -1, -1).size();          // tracker.cpp:99
    n += mapper.matcher.frameMatch(last, cur, {}, 300, -1, 4).size();             // mapper.cpp:34
    n += mapper.matcher.inverseMappointsFrameMatch(cur, last, 280, -1, 3).size(); // mapper.cpp:115
    n += mapper.matcher.mappointsFrameMatch(last, cur).size();                    // mapper.cpp:178
    n += tracker.matcher.frameMatch(last, cur, {}, 300, 50).size();               // main.cpp:82
    std::vector<cv::DMatch> match;
    cv::Mat d1, d2;
    mapper.matcher.matcher->match(d1, d2, match);                                 // mapper.cpp:336
    slam::Detector detector(cv::ORB::create(1000, 1.2f, 8, 31, 0, 2, cv::ORB::FAST_SCORE));   // main.cpp:100
    n += static_cast<size_t>(detector.getLevels()) + static_cast<size_t>(detector.getScaleFactor());
    return n;
}

}  // namespace
