// Compiles include/slam_amd_opencv.hpp (the drop-in slam::Detector / slam::Matcher) against a
// test-only stand-in of the OpenCV types it touches, and drives it with the exact call shapes of
// the reference:
//   main.cpp:100           cv::ORB::create(1000, 1.2f, 8, 31, 0, 2, cv::ORB::FAST_SCORE)
//   frame.cpp:23, 28-37    detector->detect(image, keypoints, descriptors); getLevels / getScaleFactor
//   tracker.cpp:74 / :77   matcher.mappointsFrameMatch(last, cur, 300, 50) / (last, cur, 300, -1, -1)
//   tracker.cpp:84 / :99   matcher.projectionMatch(last, cur, 300, 50) / (last, cur, 300, -1, -1)
//   mapper.cpp:17          Mapper(Matcher matcher) : matcher(matcher)        (Matcher is copied)
//   mapper.cpp:34          matcher.frameMatch(initial, current, {}, 300, -1, 4)
//   mapper.cpp:115         matcher.inverseMappointsFrameMatch(current, keyframe, 280, -1, 3)
//   mapper.cpp:178         matcher.mappointsFrameMatch(connection, current)
//   mapper.cpp:336         matcher.matcher->match(descriptor1, descriptor2, match)
//   main.cpp:82            tracker.matcher.frameMatch(keyframe1, keyframe2, {}, 300, 50)
// The Frame / KeyFrame / MapPoint classes below restate only the members of the reference's
// classes that slam::Matcher reads (include/frame/frame.hpp:20-43, include/map/keyframe.hpp:20-50,
// include/map/mappoint.hpp:19-33).  Results are compared with the CPU oracle.  The stand-in pins
// nothing about OpenCV; it proves that the header parses, that the signatures bind and that the
// adapter moves the data correctly.  Needs a GPU.  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <vector>

#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>

#define SLAM_AMD_OPENCV_DETECTOR_ONLY
#include "slam_amd_opencv.hpp"      // slam::Detector, as include/frame/frame.hpp needs it
#undef SLAM_AMD_OPENCV_DETECTOR_ONLY

namespace slam {

class Frame {   // include/frame/frame.hpp:20-43
public:
    std::shared_ptr<Detector> detector;
    std::shared_ptr<cv::Mat> image, cameraMatrix, distortions, descriptors;
    std::vector<cv::KeyPoint> keypoints, undistortedKeypoints;
    std::vector<float> scales, sigma, invSigma;
    Frame(std::shared_ptr<cv::Mat> image, std::shared_ptr<Detector> detector, std::shared_ptr<cv::Mat> cameraMatrix,
          std::shared_ptr<cv::Mat> distortions)
        : detector(detector), image(image), cameraMatrix(cameraMatrix), distortions(distortions) {
        descriptors = std::make_shared<cv::Mat>();
        detector->detect(image, keypoints, descriptors);              // frame.cpp:23
        if (keypoints.empty()) return;
        undistortedKeypoints = keypoints;                              // frame.cpp:42-45 (zero distortion)
        const float scale = static_cast<float>(detector->getScaleFactor());
        sigma.push_back(1.0f); invSigma.push_back(1.0f); scales.push_back(1.0f);
        for (int i = 1; i < detector->getLevels(); i++) {              // frame.cpp:33-37
            scales.push_back(scales[i - 1] * scale);
            sigma.push_back(scales[i] * scales[i]);
            invSigma.push_back(1.0f / sigma[i]);
        }
    }
};

class KeyFrame;
class MapPoint {   // include/map/mappoint.hpp:19-33
    cv::Point3f position;
public:
    MapPoint(const cv::Point3f& position, std::shared_ptr<KeyFrame>) : position(position) {}
    cv::Point3f getWorldPos() const { return position; }
};

class KeyFrame {   // include/map/keyframe.hpp:20-50
    std::shared_ptr<Frame> frame;
    cv::Mat pose;
public:
    std::map<int, std::shared_ptr<MapPoint>> mappoints;
    KeyFrame(std::shared_ptr<Frame> frame, const cv::Mat& pose) : frame(frame), pose(pose) {}
    cv::Mat getPose() const { return pose; }
    std::shared_ptr<Frame> getFrame() const { return frame; }
};

}  // namespace slam

#include "slam_amd_opencv.hpp"      // slam::Matcher, now that slam::KeyFrame is complete

extern "C" {
#include "../../oracle/orb_oracle.h"
}

static int g_fail = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { g_fail++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static unsigned rng_state = 12345;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static std::shared_ptr<cv::Mat> make_scene(int w, int h, int shift) {
    auto img = std::make_shared<cv::Mat>(h, w, CV_8UC1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            img->at<uint8_t>(y, x) = static_cast<uint8_t>(60 + 40 * std::sin((x + shift) * 0.05) + 30 * std::cos(y * 0.07));
    rng_state = 777;
    for (int i = 0; i < 700; i++) {
        int cx = static_cast<int>(rnd() % w) + shift, cy = static_cast<int>(rnd() % h), r = 3 + static_cast<int>(rnd() % 14);
        uint8_t v = static_cast<uint8_t>(rnd() % 256);
        for (int y = std::max(0, cy - r); y < std::min(h, cy + r); y++)
            for (int x = std::max(0, cx - r); x < std::min(w, cx + r); x++) img->at<uint8_t>(y, x) = v;
    }
    return img;
}

// A class that takes a Matcher by value and keeps a copy, as Mapper does (mapper.cpp:17).
struct MapperLike {
    slam::Matcher matcher;
    MapperLike(slam::Matcher matcher) : matcher(matcher) {}
};

static std::vector<orc_dmatch> reference_match(const std::shared_ptr<slam::KeyFrame>& a, const std::shared_ptr<slam::KeyFrame>& b,
                                               const std::vector<int>& ids, float md, float area, int lvl, const float* qxy = nullptr) {
    const auto fa = a->getFrame(), fb = b->getFrame();
    const int n1 = static_cast<int>(fa->undistortedKeypoints.size()), n2 = static_cast<int>(fb->undistortedKeypoints.size());
    std::vector<orc_keypoint> k1(n1), k2(n2);
    std::memcpy(k1.data(), fa->undistortedKeypoints.data(), sizeof(orc_keypoint) * n1);
    std::memcpy(k2.data(), fb->undistortedKeypoints.data(), sizeof(orc_keypoint) * n2);
    if (qxy) for (size_t i = 0; i < ids.size(); i++) { k1[ids[i]].x = qxy[2 * i]; k1[ids[i]].y = qxy[2 * i + 1]; }
    std::vector<orc_dmatch> out(static_cast<size_t>(std::max(n1, 1)));
    int n = 0;
    orc_frame_match(fa->descriptors->data, k1.data(), n1, fb->descriptors->data, k2.data(), n2,
                    ids.empty() ? nullptr : ids.data(), static_cast<int>(ids.size()), md, area, lvl, out.data(), &n);
    out.resize(static_cast<size_t>(n));
    return out;
}

static bool same(const std::vector<cv::DMatch>& a, const std::vector<orc_dmatch>& b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(orc_dmatch)) == 0);
}

int main() {
    const int W = 480, H = 360;
    try {
        auto detector = std::make_shared<slam::Detector>(cv::ORB::create(1000, 1.2f, 8, 31, 0, 2, cv::ORB::FAST_SCORE));  // main.cpp:100
        auto K = std::make_shared<cv::Mat>(3, 3, CV_64F);
        for (int i = 0; i < 9; i++) K->at<double>(i) = 0.0;
        K->at<double>(0, 0) = 963.0; K->at<double>(1, 1) = 950.4; K->at<double>(0, 2) = 539.5; K->at<double>(1, 2) = 404.5; K->at<double>(2, 2) = 1.0;
        auto dist = std::make_shared<cv::Mat>(1, 5, CV_64F);
        for (int i = 0; i < 5; i++) dist->at<double>(i) = 0.0;
        std::shared_ptr<slam::KeyFrame> kf[2];
        for (int f = 0; f < 2; f++) {
            auto img = make_scene(W, H, 9 * f);
            auto frame = std::make_shared<slam::Frame>(img, detector, K, dist);
            // extraction against the oracle
            orc_params op; orc_default_params(&op);
            std::vector<orc_keypoint> okp(8192); std::vector<uint8_t> od(8192 * 32); int on = 0;
            orc_detect_and_compute(&op, img->data, W, H, static_cast<int>(img->step), okp.data(), od.data(), 8192, &on);
            EXPECT(on == static_cast<int>(frame->keypoints.size()) && on > 300, "keypoint count %d vs %zu", on, frame->keypoints.size());
            EXPECT(std::memcmp(okp.data(), frame->keypoints.data(), static_cast<size_t>(on) * 28) == 0, "keypoints differ");
            EXPECT(frame->descriptors->rows == on && frame->descriptors->cols == 32 &&
                   std::memcmp(od.data(), frame->descriptors->data, static_cast<size_t>(on) * 32) == 0, "descriptors differ");
            EXPECT(frame->scales.size() == 8 && frame->scales[1] == 1.2f, "scale table");
            cv::Mat pose(4, 4, CV_64F);
            const double c = std::cos(0.05 * (f + 1)), s = std::sin(0.05 * (f + 1));
            const double p[16] = {c, 0, s, 0.1, 0, 1, 0, -0.05, -s, 0, c, 0.2, 0, 0, 0, 1};
            for (int i = 0; i < 16; i++) pose.at<double>(i) = static_cast<double>(static_cast<float>(p[i]));
            kf[f] = std::make_shared<slam::KeyFrame>(frame, pose);
            for (int i = f; i < on; i += 3)
                kf[f]->mappoints[i] = std::make_shared<slam::MapPoint>(
                    cv::Point3f((rnd() % 4000) / 1000.f - 2.f, (rnd() % 3000) / 1000.f - 1.5f, 2.f + (rnd() % 5000) / 1000.f), kf[f]);
        }
        std::vector<int> ids0;
        for (const auto& kv : kf[0]->mappoints) ids0.push_back(kv.first);
        slam::Matcher matcher;                                     // tracker.hpp:29
        MapperLike mapper(matcher);                                // mapper.cpp:17
        slam::Matcher from_ptr(cv::BFMatcher::create(cv::BFMatcher::BRUTEFORCE_HAMMING, true));   // matcher.hpp:20
        const auto& last = kf[0]; const auto& cur = kf[1];
        EXPECT(same(matcher.mappointsFrameMatch(last, cur, 300, 50), reference_match(last, cur, ids0, 300, 50, 4)), "tracker.cpp:74");
        EXPECT(same(matcher.mappointsFrameMatch(last, cur, 300, -1, -1), reference_match(last, cur, ids0, 300, -1, -1)), "tracker.cpp:77");
        {
            // projected positions: map points of `last` with the pose of `cur`
            std::vector<float> xyz, uv(ids0.size() * 2);
            for (const auto& kv : last->mappoints) { auto w = kv.second->getWorldPos(); xyz.push_back(w.x); xyz.push_back(w.y); xyz.push_back(w.z); }
            double Rt[12]; cv::Mat p32; cur->getPose().convertTo(p32, CV_32F);
            for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) Rt[r * 4 + c] = static_cast<double>(p32.at<float>(r, c));
            const double K4[4] = {963.0, 950.4, 539.5, 404.5};
            orc_project_points(Rt, K4, nullptr, xyz.data(), static_cast<int>(ids0.size()), uv.data());
            EXPECT(same(matcher.projectionMatch(last, cur, 300, 50), reference_match(last, cur, ids0, 300, 50, 4, uv.data())), "tracker.cpp:84");
        }
        EXPECT(same(matcher.projectionMatch(last, cur, 300, -1, -1), reference_match(last, cur, ids0, 300, -1, -1)), "tracker.cpp:99");
        EXPECT(same(mapper.matcher.frameMatch(last, cur, {}, 300, -1, 4), reference_match(last, cur, {}, 300, -1, 4)), "mapper.cpp:34");
        {
            std::vector<int> inv;
            const int n = static_cast<int>(cur->getFrame()->undistortedKeypoints.size());
            for (int i = 0; i < n; i++) if (!cur->mappoints.count(i)) inv.push_back(i);
            EXPECT(same(mapper.matcher.inverseMappointsFrameMatch(cur, last, 280, -1, 3), reference_match(cur, last, inv, 280, -1, 3)), "mapper.cpp:115");
        }
        EXPECT(same(mapper.matcher.mappointsFrameMatch(last, cur), reference_match(last, cur, ids0, 300, -1, 4)), "mapper.cpp:178");
        EXPECT(same(from_ptr.frameMatch(last, cur, {}, 300, 50), reference_match(last, cur, {}, 300, 50, 4)), "main.cpp:82");
        {   // mapper.cpp:336: one descriptor row against one descriptor row through the public member
            std::vector<cv::DMatch> match;
            const cv::Mat d1 = last->getFrame()->descriptors->row(3), d2 = cur->getFrame()->descriptors->row(5);
            mapper.matcher.matcher->match(d1, d2, match);
            EXPECT(match.size() == 1 && match[0].distance == static_cast<float>(orc_hamming256(d1.data, d2.data)), "mapper.cpp:336");
        }
        {   // empty image: empty outputs (frame.cpp:24)
            auto empty = std::make_shared<cv::Mat>();
            std::vector<cv::KeyPoint> k; auto d = std::make_shared<cv::Mat>();
            detector->detect(empty, k, d);
            EXPECT(k.empty() && d->rows == 0, "empty image");
        }
    } catch (const std::exception& e) {
        printf("exception: %s\n", e.what());
        return 2;
    }
    printf(g_fail ? "%d check(s) failed\n" : "drop-in header ok (%d failures)\n", g_fail);
    return g_fail ? 1 : 0;
}
