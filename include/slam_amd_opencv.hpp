// slam_amd_opencv.hpp -- drop-in slam::Detector and slam::Matcher for pxl-th/slam.
//
// The classes keep the reference's signatures, default arguments and public members
// (include/frame/detector.hpp:15-37, include/frame/matcher.hpp:15-117), so src/frame/frame.cpp,
// src/tracking/*.cpp and main.cpp compile unchanged; src/frame/detector.cpp and
// src/frame/matcher.cpp drop out of the build.  Needs OpenCV headers (core, features2d) and links
// libslam_amd.so.  The two shims under include/dropin/frame/ carry the reference's file names and
// include guards (put include/dropin before the reference's include directory and drop detector.cpp / matcher.cpp from the sources).
//
// slam::Matcher needs the complete slam::KeyFrame (include "map/keyframe.hpp" first, as the
// reference's matcher.hpp does); define SLAM_AMD_OPENCV_DETECTOR_ONLY to get slam::Detector alone.
//
// Status: compiled and run in tests/cpp/test_dropin.cpp against a test-only stand-in of the
// OpenCV types (OpenCV itself is not installed where this library is built).
#ifndef SLAM_AMD_OPENCV_DETECTOR_HPP
#define SLAM_AMD_OPENCV_DETECTOR_HPP

#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>

#include "slam_amd.hpp"

namespace slam {

static_assert(sizeof(cv::KeyPoint) == sizeof(slam_amd::KeyPoint), "cv::KeyPoint layout changed");
static_assert(sizeof(cv::DMatch) == sizeof(slam_amd::DMatch), "cv::DMatch layout changed");

// reference: include/frame/detector.hpp:15-37
class Detector {
private:
    cv::Ptr<cv::ORB> detector;  // kept as the carrier of the parameters
    std::shared_ptr<slam_amd::Detector> gpu;
public:
    Detector() {}
    Detector(cv::Ptr<cv::ORB> detector) : detector(detector) {
        slam_amd_orb_params p;
        p.nfeatures = detector->getMaxFeatures();
        p.scale_factor = static_cast<float>(detector->getScaleFactor());
        p.nlevels = detector->getNLevels();
        p.edge_threshold = detector->getEdgeThreshold();
        p.first_level = detector->getFirstLevel();
        p.wta_k = detector->getWTA_K();
        p.score_type = static_cast<int>(detector->getScoreType());  // HARRIS_SCORE 0, FAST_SCORE 1
        p.patch_size = detector->getPatchSize();
        p.fast_threshold = detector->getFastThreshold();
        gpu = std::make_shared<slam_amd::Detector>(p);
    }

    // reference: src/frame/detector.cpp:6-12
    void detect(std::shared_ptr<cv::Mat> image, std::vector<cv::KeyPoint>& keypoints,
                std::shared_ptr<cv::Mat> descriptors) {
        CV_Assert(image->empty() || image->type() == CV_8UC1);
        slam_amd::Image8 view;
        view.data = image->data; view.cols = image->cols; view.rows = image->rows; view.step = image->step;
        slam_amd::Descriptors d;
        gpu->detect(view, keypoints, d);
        descriptors->create(d.rows, 32, CV_8U);
        if (d.rows) std::memcpy(descriptors->data, d.bytes.data(), d.bytes.size());
    }

    inline int getLevels() { return detector->getNLevels(); }
    inline double getScaleFactor() { return detector->getScaleFactor(); }
};

}  // namespace slam

#endif  // SLAM_AMD_OPENCV_DETECTOR_HPP

#if !defined(SLAM_AMD_OPENCV_DETECTOR_ONLY) && !defined(SLAM_AMD_OPENCV_MATCHER_HPP)
#define SLAM_AMD_OPENCV_MATCHER_HPP

namespace slam {

class KeyFrame;

// reference: include/frame/matcher.hpp:15-117, src/frame/matcher.cpp
class Matcher {
public:
    // The reference reaches through this member for its 1 x 32 against 1 x 32 duplicate test
    // (src/tracking/mapper.cpp:336); that call stays OpenCV's.  Frame matching does not use it.
    cv::Ptr<cv::BFMatcher> matcher;
private:
    std::shared_ptr<slam_amd::Matcher> gpu;   // shared: Mapper and Tracker hold copies of one Matcher

    struct Packed {
        slam_amd::KeyFrameView view;
        std::vector<int> ids;
        std::vector<slam_amd::Point3f> positions;
        cv::Mat pose32;
        std::vector<double> dist;
    };
    template <class KF>
    static void pack(const std::shared_ptr<KF>& kf, Packed& p, bool with_points) {
        const auto frame = kf->getFrame();
        p.view.descriptors = frame->descriptors->data;
        p.view.undistortedKeypoints = reinterpret_cast<const slam_amd::KeyPoint*>(frame->undistortedKeypoints.data());
        p.view.n = static_cast<int>(frame->undistortedKeypoints.size());
        for (const auto& [id, mappoint] : kf->mappoints) {
            p.ids.push_back(id);
            if (with_points) {
                const cv::Point3f w = mappoint->getWorldPos();
                p.positions.push_back({w.x, w.y, w.z});
            }
        }
        p.view.mappointIds = p.ids.data();
        p.view.mappointPositions = p.positions.empty() ? nullptr : p.positions.data();
        p.view.nMappoints = static_cast<int>(p.ids.size());
        kf->getPose().convertTo(p.pose32, CV_32F);
        p.view.pose = p.pose32.template ptr<float>();
        cv::Mat K;
        frame->cameraMatrix->convertTo(K, CV_64F);
        p.view.fx = K.template at<double>(0, 0); p.view.fy = K.template at<double>(1, 1);
        p.view.cx = K.template at<double>(0, 2); p.view.cy = K.template at<double>(1, 2);
        cv::Mat D;
        frame->distortions->convertTo(D, CV_64F);
        p.dist.assign(5, 0.0);
        for (int i = 0; i < 5 && i < static_cast<int>(D.total()); i++) p.dist[i] = D.template at<double>(i);
        p.view.distortions = p.dist.data();
    }
    static std::vector<cv::DMatch> convert(const std::vector<slam_amd::DMatch>& in) {
        std::vector<cv::DMatch> out(in.size());
        if (!in.empty()) std::memcpy(static_cast<void*>(out.data()), in.data(), in.size() * sizeof(cv::DMatch));
        return out;
    }
public:
    // reference: include/frame/matcher.hpp:19 -- brute force, Hamming, crossCheck (which radiusMatch ignores)
    Matcher() : matcher(cv::BFMatcher::create(cv::BFMatcher::BRUTEFORCE_HAMMING, true)),
                gpu(std::make_shared<slam_amd::Matcher>()) {}
    // reference: include/frame/matcher.hpp:20, src/frame/matcher.cpp:13
    Matcher(cv::Ptr<cv::BFMatcher> matcher) : matcher(matcher), gpu(std::make_shared<slam_amd::Matcher>()) {}

    std::vector<cv::DMatch> frameMatch(const std::shared_ptr<KeyFrame>& keyframe1, const std::shared_ptr<KeyFrame>& keyframe2,
                                       const std::vector<int>& ids, float maximumDistance = 300,
                                       float areaSize = -1, int maxLevel = 4) const {
        Packed a, b; pack(keyframe1, a, false); pack(keyframe2, b, false);
        return convert(gpu->frameMatch(a.view, b.view, ids, maximumDistance, areaSize, maxLevel));
    }
    std::vector<cv::DMatch> mappointsFrameMatch(const std::shared_ptr<KeyFrame>& keyframe1,
                                                const std::shared_ptr<KeyFrame>& keyframe2, float maximumDistance = 300,
                                                float areaSize = -1, int maxLevel = 4) const {
        Packed a, b; pack(keyframe1, a, false); pack(keyframe2, b, false);
        return convert(gpu->mappointsFrameMatch(a.view, b.view, maximumDistance, areaSize, maxLevel));
    }
    std::vector<cv::DMatch> inverseMappointsFrameMatch(const std::shared_ptr<KeyFrame>& keyframe1,
                                                       const std::shared_ptr<KeyFrame>& keyframe2,
                                                       float maximumDistance = 300, float areaSize = -1,
                                                       int maxLevel = 4) const {
        Packed a, b; pack(keyframe1, a, false); pack(keyframe2, b, false);
        return convert(gpu->inverseMappointsFrameMatch(a.view, b.view, maximumDistance, areaSize, maxLevel));
    }
    std::vector<cv::DMatch> projectionMatch(const std::shared_ptr<KeyFrame>& fromKeyFrame,
                                            const std::shared_ptr<KeyFrame>& toKeyFrame, float maximumDistance = 300,
                                            float areaSize = -1, int maxLevel = 4) const {
        Packed a, b; pack(fromKeyFrame, a, true); pack(toKeyFrame, b, false);
        return convert(gpu->projectionMatch(a.view, b.view, maximumDistance, areaSize, maxLevel));
    }
};

}  // namespace slam

#endif  // SLAM_AMD_OPENCV_MATCHER_HPP
