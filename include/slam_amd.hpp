// slam_amd.hpp -- C++ host layer over the C ABI (slam_amd.h), header-only, C++17.
//
// It mirrors the two classes of pxl-th/slam that carry the frontend hot path,
// with the same method names, argument meaning, defaults and result order:
//
//   slam::Detector  (reference: include/frame/detector.hpp:15-37, src/frame/detector.cpp)
//   slam::Matcher   (reference: include/frame/matcher.hpp:15-117, src/frame/matcher.cpp)
//   the per-frame tables of slam::Frame (reference: src/frame/frame.cpp:14-38)
//
// The classes here are written against plain views (KeyFrameView, Image8) and
// POD records whose layouts equal cv::KeyPoint / cv::DMatch / cv::Point2f /
// cv::Point3f, so this header builds without OpenCV.  slam_amd_opencv.hpp maps
// the reference's exact OpenCV-typed signatures onto these classes.
//
// Errors: the reference has no error convention on this path (OpenCV throws
// cv::Exception); this layer throws slam_amd::Error carrying the C status.
// An empty image yields empty outputs, as the reference relies on at
// src/frame/frame.cpp:24.  Calls block until results are on the host, one
// caller thread per object.
#ifndef SLAM_AMD_HPP
#define SLAM_AMD_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <iterator>
#include <stdexcept>
#include <string>
#include <vector>

#include "slam_amd.h"

namespace slam_amd {

struct Point2f { float x, y; };
struct Point3f { float x, y, z; };

// Layout of cv::KeyPoint.
struct KeyPoint {
    Point2f pt;
    float size;
    float angle;
    float response;
    int octave;
    int class_id;
};

// Layout of cv::DMatch.
struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
};

static_assert(sizeof(KeyPoint) == sizeof(slam_amd_keypoint) && sizeof(KeyPoint) == 28, "KeyPoint layout");
static_assert(sizeof(DMatch) == sizeof(slam_amd_dmatch) && sizeof(DMatch) == 16, "DMatch layout");

class Error : public std::runtime_error {
public:
    Error(int status, const std::string& what) : std::runtime_error(what), status(status) {}
    int status;
};

inline void check(int status) {
    if (status != SLAM_AMD_OK)
        throw Error(status, std::string(slam_amd_status_string(status)) + ": " + slam_amd_last_error());
}

// View of an 8-bit single-channel image (cv::Mat of type CV_8UC1).
struct Image8 {
    const uint8_t* data = nullptr;
    int cols = 0, rows = 0;
    size_t step = 0;  // bytes per row
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
};

// N x 32 descriptor bytes, row-major (cv::Mat N x 32 CV_8U).
struct Descriptors {
    std::vector<uint8_t> bytes;
    int rows = 0;
    const uint8_t* row(int i) const { return bytes.data() + static_cast<size_t>(i) * 32; }
};

// ---------------------------------------------------------------------------
// Detector: replaces cv::ORB::detectAndCompute behind slam::Detector::detect.
// ---------------------------------------------------------------------------
class Detector {
public:
    Detector() { slam_amd_default_orb_params(&params_); }
    // reference: Detector(cv::Ptr<cv::ORB>); the ORB parameters arrive as a POD here.
    explicit Detector(const slam_amd_orb_params& params, int device = 0) : params_(params), device_(device) {}
    Detector(const Detector&) = delete;
    Detector& operator=(const Detector&) = delete;
    ~Detector() { slam_amd_extractor_destroy(ex_); }

    // reference: Detector::detect(image, keypoints, descriptors), src/frame/detector.cpp:6-12.
    // Clears and refills `keypoints`, (re)allocates `descriptors`.  KP is any
    // 28-byte record with the layout of cv::KeyPoint.
    template <class KP>
    void detect(const Image8& image, std::vector<KP>& keypoints, Descriptors& descriptors) {
        static_assert(sizeof(KP) == sizeof(slam_amd_keypoint), "keypoint record must match cv::KeyPoint");
        keypoints.clear();
        descriptors.bytes.clear();
        descriptors.rows = 0;
        if (image.empty()) return;
        ensure(image.cols, image.rows);
        int capacity = std::max(2 * params_.nfeatures, 1024);
        for (;;) {
            keypoints.resize(static_cast<size_t>(capacity));
            descriptors.bytes.resize(static_cast<size_t>(capacity) * 32);
            int n = 0;
            const int rc = slam_amd_extract(ex_, image.data, image.cols, image.rows, static_cast<int>(image.step),
                                            reinterpret_cast<slam_amd_keypoint*>(keypoints.data()),
                                            descriptors.bytes.data(), capacity, &n);
            if (rc == SLAM_AMD_ERR_CAPACITY && n > capacity && n <= max_keypoints_) {
                // score ties pushed the count above the guess; the results are still on the
                // device, so they are fetched into buffers of the exact size -- no second extraction
                capacity = n;
                keypoints.resize(static_cast<size_t>(capacity));
                descriptors.bytes.resize(static_cast<size_t>(capacity) * 32);
                check(slam_amd_extract_fetch(ex_, reinterpret_cast<slam_amd_keypoint*>(keypoints.data()),
                                             descriptors.bytes.data(), capacity, &n));
                descriptors.rows = n;
                return;
            }
            if (rc == SLAM_AMD_ERR_CAPACITY && n > max_keypoints_) {
                // more ties than the context was sized for: rebuild it larger, once
                slam_amd_extractor_destroy(ex_);
                ex_ = nullptr;
                max_keypoints_ = n;
                ensure(image.cols, image.rows);
                capacity = n;
                continue;
            }
            check(rc);
            keypoints.resize(static_cast<size_t>(n));
            descriptors.bytes.resize(static_cast<size_t>(n) * 32);
            descriptors.rows = n;
            return;
        }
    }

    // Not in the reference: keep only the best corner of every cell x cell block of level
    // pixels (cell = 8, 16 or 32; 0 = off, the default) before the per-level best-N cut.
    void setGridCell(int cell) {
        if (cell != 0 && cell != 8 && cell != 16 && cell != 32) throw Error(SLAM_AMD_ERR_ARG, "grid cell must be 0, 8, 16 or 32");
        grid_cell_ = cell;
        if (ex_) check(slam_amd_extractor_set_grid(ex_, cell));
    }

    // Not in the reference either: quadtree distribution with an exact per-level count
    // (slam_amd_extractor_set_quadtree); excludes the grid option.
    void setQuadtree(bool on) {
        if (on && grid_cell_) throw Error(SLAM_AMD_ERR_ARG, "the grid and the quadtree distribution exclude each other");
        quadtree_ = on;
        if (ex_) check(slam_amd_extractor_set_quadtree(ex_, on ? 1 : 0));
    }

    // reference: include/frame/detector.hpp:35-36
    int getLevels() const { return params_.nlevels; }
    double getScaleFactor() const { return static_cast<double>(params_.scale_factor); }
    const slam_amd_orb_params& params() const { return params_; }

private:
    void ensure(int w, int h) {
        if (ex_ && w == width_ && h == height_) return;
        slam_amd_extractor_destroy(ex_);
        ex_ = nullptr;
        if (max_keypoints_ == 0) max_keypoints_ = std::max(4 * params_.nfeatures, 4096);
        check(slam_amd_extractor_create(&params_, w, h, 1, max_keypoints_, device_, &ex_));
        check(slam_amd_extractor_set_grid(ex_, grid_cell_));
        check(slam_amd_extractor_set_quadtree(ex_, quadtree_ ? 1 : 0));
        width_ = w;
        height_ = h;
    }
    slam_amd_orb_params params_;
    int device_ = 0;
    slam_amd_extractor* ex_ = nullptr;
    int width_ = 0, height_ = 0, max_keypoints_ = 0;
    int grid_cell_ = 0;
    bool quadtree_ = false;
};

// ---------------------------------------------------------------------------
// The per-frame products of slam::Frame::Frame (src/frame/frame.cpp:14-38).
// ---------------------------------------------------------------------------
struct FrameData {
    std::vector<KeyPoint> keypoints, undistortedKeypoints;
    Descriptors descriptors;
    std::vector<float> scales, sigma, invSigma;
};

// distortion0: first distortion coefficient; the reference copies the keypoints
// when it is zero (frame.cpp:42-45), which holds for the shipped calibration.
// A non-zero value needs cv::undistortPoints, which this layer does not replace.
inline FrameData makeFrame(Detector& detector, const Image8& image, float distortion0 = 0.0f) {
    FrameData f;
    detector.detect(image, f.keypoints, f.descriptors);
    if (f.keypoints.empty()) return f;
    if (distortion0 != 0.0f)
        throw Error(SLAM_AMD_ERR_UNSUPPORTED, "keypoint undistortion is outside this layer (frame.cpp:46-65)");
    f.undistortedKeypoints = f.keypoints;
    const float scale = static_cast<float>(detector.getScaleFactor());
    f.sigma.push_back(1.0f);
    f.invSigma.push_back(1.0f);
    f.scales.push_back(1.0f);
    for (int i = 1; i < detector.getLevels(); i++) {
        f.scales.push_back(f.scales[i - 1] * scale);
        f.sigma.push_back(f.scales[i] * f.scales[i]);
        f.invSigma.push_back(1.0f / f.sigma[i]);
    }
    return f;
}

// ---------------------------------------------------------------------------
// FrontEnd: the per-frame preparation of main.cpp:124-129 (cvtColor BGR2GRAY, then
// resize to the working width), one kernel.  Created lazily per source size.
// ---------------------------------------------------------------------------
class FrontEnd {
public:
    explicit FrontEnd(int targetWidth, int device = 0) : target_(targetWidth), device_(device) {}
    FrontEnd(const FrontEnd&) = delete;
    FrontEnd& operator=(const FrontEnd&) = delete;
    ~FrontEnd() { slam_amd_frontend_destroy(fe_); }

    // bgr: rows x cols pixels of `channels` interleaved bytes (3 = BGR, 1 = grey).
    // Returns the grey image, cols = round(cols * scale), rows = round(rows * scale).
    std::vector<uint8_t> run(const uint8_t* data, int cols, int rows, size_t step, int channels,
                             int* outCols, int* outRows) {
        if (!data || cols <= 0 || rows <= 0) throw Error(SLAM_AMD_ERR_ARG, "empty frame");
        if (!fe_ || cols != cols_ || rows != rows_ || channels != channels_) {
            slam_amd_frontend_destroy(fe_);
            fe_ = nullptr;
            check(slam_amd_frontend_create(cols, rows, channels, target_, device_, &fe_));
            cols_ = cols; rows_ = rows; channels_ = channels;
        }
        int w = 0, h = 0;
        check(slam_amd_frontend_output_size(fe_, &w, &h));
        std::vector<uint8_t> grey(static_cast<size_t>(w) * static_cast<size_t>(h));
        check(slam_amd_frontend_run(fe_, data, static_cast<int>(step), grey.data(), w));
        if (outCols) *outCols = w;
        if (outRows) *outRows = h;
        return grey;
    }

private:
    int target_, device_;
    slam_amd_frontend* fe_ = nullptr;
    int cols_ = 0, rows_ = 0, channels_ = 0;
};

// ---------------------------------------------------------------------------
// What Matcher reads from a slam::KeyFrame.
// ---------------------------------------------------------------------------
struct KeyFrameView {
    const uint8_t* descriptors = nullptr;       // frame->descriptors, N x 32
    const KeyPoint* undistortedKeypoints = nullptr;  // frame->undistortedKeypoints, N records
    int n = 0;
    // keys of keyframe->mappoints in std::map order (ascending keypoint id) and the
    // world position of each (MapPoint::getWorldPos)
    const int* mappointIds = nullptr;
    const Point3f* mappointPositions = nullptr;
    int nMappoints = 0;
    const float* pose = nullptr;                // KeyFrame::getPose(): 4 x 4, row-major, float
    double fx = 0, fy = 0, cx = 0, cy = 0;      // frame->cameraMatrix
    const double* distortions = nullptr;        // k1 k2 p1 p2 k3, or null for none
};

// ---------------------------------------------------------------------------
// Matcher: replaces cv::BFMatcher::radiusMatch + Matcher::_filterMatches.
// ---------------------------------------------------------------------------
class Matcher {
public:
    explicit Matcher(int device = 0) { check(slam_amd_matcher_create(device, &m_)); }
    Matcher(const Matcher&) = delete;
    Matcher& operator=(const Matcher&) = delete;
    ~Matcher() { slam_amd_matcher_destroy(m_); }

    // reference: Matcher::frameMatch, src/frame/matcher.cpp:15-55
    std::vector<DMatch> frameMatch(const KeyFrameView& keyframe1, const KeyFrameView& keyframe2,
                                   const std::vector<int>& ids, float maximumDistance = 300,
                                   float areaSize = -1, int maxLevel = 4) const {
        return run(keyframe1, keyframe2, ids, nullptr, maximumDistance, areaSize, maxLevel);
    }

    // reference: Matcher::mappointsFrameMatch, src/frame/matcher.cpp:57-66
    std::vector<DMatch> mappointsFrameMatch(const KeyFrameView& keyframe1, const KeyFrameView& keyframe2,
                                            float maximumDistance = 300, float areaSize = -1,
                                            int maxLevel = 4) const {
        std::vector<int> ids(keyframe1.mappointIds, keyframe1.mappointIds + keyframe1.nMappoints);
        // an empty id list means "all keypoints" in frameMatch (matcher.cpp:26,37-40)
        return frameMatch(keyframe1, keyframe2, ids, maximumDistance, areaSize, maxLevel);
    }

    // reference: Matcher::inverseMappointsFrameMatch, src/frame/matcher.cpp:68-88
    std::vector<DMatch> inverseMappointsFrameMatch(const KeyFrameView& keyframe1, const KeyFrameView& keyframe2,
                                                   float maximumDistance = 300, float areaSize = -1,
                                                   int maxLevel = 4) const {
        if (keyframe1.nMappoints == 0)
            return frameMatch(keyframe1, keyframe2, {}, maximumDistance, areaSize, maxLevel);
        std::vector<int> all(static_cast<size_t>(keyframe1.n)), matchIds;
        for (int i = 0; i < keyframe1.n; i++) all[static_cast<size_t>(i)] = i;
        std::set_difference(all.begin(), all.end(), keyframe1.mappointIds,
                            keyframe1.mappointIds + keyframe1.nMappoints, std::back_inserter(matchIds));
        // like the reference, an empty difference falls through to "all keypoints"
        return frameMatch(keyframe1, keyframe2, matchIds, maximumDistance, areaSize, maxLevel);
    }

    // reference: Matcher::projectionMatch, src/frame/matcher.cpp:90-129.  Queries are
    // the map points of fromKeyFrame (ascending keypoint id); with areaSize != -1
    // their positions are the map points projected with toKeyFrame's pose.
    std::vector<DMatch> projectionMatch(const KeyFrameView& fromKeyFrame, const KeyFrameView& toKeyFrame,
                                        float maximumDistance = 300, float areaSize = -1,
                                        int maxLevel = 4) const {
        const int nq = fromKeyFrame.nMappoints;
        if (nq == 0) return {};
        std::vector<int> ids(fromKeyFrame.mappointIds, fromKeyFrame.mappointIds + nq);
        std::vector<Point2f> projected;
        if (areaSize != -1) projected = projectMapPoints(fromKeyFrame, toKeyFrame);
        return run(fromKeyFrame, toKeyFrame, ids, projected.empty() ? nullptr : &projected[0].x,
                   maximumDistance, areaSize, maxLevel);
    }

    // reference: Matcher::_projectMapPoints, src/frame/matcher.cpp:163-184
    std::vector<Point2f> projectMapPoints(const KeyFrameView& fromKeyFrame, const KeyFrameView& toKeyFrame) const {
        std::vector<Point2f> points(static_cast<size_t>(fromKeyFrame.nMappoints));
        if (points.empty()) return points;
        if (!toKeyFrame.pose || !fromKeyFrame.mappointPositions)
            throw Error(SLAM_AMD_ERR_ARG, "projection needs the target pose and the map point positions");
        double Rt[12];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) Rt[r * 4 + c] = static_cast<double>(toKeyFrame.pose[r * 4 + c]);
        const double K4[4] = {toKeyFrame.fx, toKeyFrame.fy, toKeyFrame.cx, toKeyFrame.cy};
        check(slam_amd_project_points(m_, Rt, K4, toKeyFrame.distortions, &fromKeyFrame.mappointPositions[0].x,
                                      fromKeyFrame.nMappoints, &points[0].x));
        return points;
    }

    // reference: Mapper::_isDuplicate over every pair of Mapper::_keyframeDuplicates,
    // src/tracking/mapper.cpp:292-346.  flag[i * n2 + j] != 0 where the descriptors are
    // within descriptorDistance and the world points closer than pointDistance.
    std::vector<uint8_t> duplicatePairs(const uint8_t* desc1, const Point3f* pos1, int n1,
                                        const uint8_t* desc2, const Point3f* pos2, int n2,
                                        int descriptorDistance = 100, double pointDistance = 1e-2) const {
        std::vector<uint8_t> flag(static_cast<size_t>(n1) * static_cast<size_t>(n2));
        if (flag.empty()) return flag;
        check(slam_amd_duplicate_pairs(m_, desc1, &pos1[0].x, n1, desc2, &pos2[0].x, n2, descriptorDistance,
                                       pointDistance, flag.data()));
        return flag;
    }

    // reference: isOutlier(point, queryKeyframe, trainKeyframe, match) for every point,
    // src/map/mappoint.cpp:54-101 (called per point at src/tracking/mapper.cpp:48,121).
    // flag != 0 = outlier.  Camera centres are -R^T t of each pose (KeyFrame::setPose,
    // src/map/keyframe.cpp:19-22), products summed in double and rounded to float.
    std::vector<uint8_t> outlierFlags(const std::vector<Point3f>& points, const std::vector<DMatch>& matches,
                                      const KeyFrameView& queryKeyFrame, const KeyFrameView& trainKeyFrame) const {
        if (points.size() != matches.size()) throw Error(SLAM_AMD_ERR_ARG, "one match per point");
        std::vector<uint8_t> flags(points.size());
        if (flags.empty()) return flags;
        if (!queryKeyFrame.pose || !trainKeyFrame.pose) throw Error(SLAM_AMD_ERR_ARG, "both poses are needed");
        double Rt[2][12], K4[2][4];
        float centre[2][3];
        const KeyFrameView* kf[2] = {&queryKeyFrame, &trainKeyFrame};
        for (int v = 0; v < 2; v++) {
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 4; c++) Rt[v][r * 4 + c] = static_cast<double>(kf[v]->pose[r * 4 + c]);
            for (int c = 0; c < 3; c++) {
                double acc = 0;
                for (int r = 0; r < 3; r++) acc += -Rt[v][r * 4 + c] * Rt[v][r * 4 + 3];
                centre[v][c] = static_cast<float>(acc);
            }
            K4[v][0] = kf[v]->fx; K4[v][1] = kf[v]->fy; K4[v][2] = kf[v]->cx; K4[v][3] = kf[v]->cy;
        }
        check(slam_amd_outlier_flags(m_, &points[0].x, reinterpret_cast<const slam_amd_dmatch*>(matches.data()),
                                     static_cast<int>(points.size()),
                                     reinterpret_cast<const slam_amd_keypoint*>(queryKeyFrame.undistortedKeypoints),
                                     queryKeyFrame.n,
                                     reinterpret_cast<const slam_amd_keypoint*>(trainKeyFrame.undistortedKeypoints),
                                     trainKeyFrame.n, centre[0], centre[1], Rt[0], K4[0], queryKeyFrame.distortions,
                                     Rt[1], K4[1], trainKeyFrame.distortions, flags.data()));
        return flags;
    }

    // Not in the reference: Lowe's ratio test (nnRatio > 0) and the rotation-consistency
    // histogram (histoLength > 0, e.g. 30) applied to every later match call; see
    // slam_amd_matcher_set_options.  Both are off by default.
    void setOptions(float nnRatio, int histoLength) { check(slam_amd_matcher_set_options(m_, nnRatio, histoLength)); }

    slam_amd_matcher* handle() const { return m_; }

private:
    std::vector<DMatch> run(const KeyFrameView& kf1, const KeyFrameView& kf2, const std::vector<int>& ids,
                            const float* q_xy, float maximumDistance, float areaSize, int maxLevel) const {
        const int nq = ids.empty() ? kf1.n : static_cast<int>(ids.size());
        std::vector<DMatch> matches(static_cast<size_t>(std::max(nq, 0)));
        int n = 0;
        check(slam_amd_frame_match(m_, kf1.descriptors, reinterpret_cast<const slam_amd_keypoint*>(kf1.undistortedKeypoints),
                                   kf1.n, kf2.descriptors,
                                   reinterpret_cast<const slam_amd_keypoint*>(kf2.undistortedKeypoints), kf2.n,
                                   ids.empty() ? nullptr : ids.data(), static_cast<int>(ids.size()), q_xy,
                                   maximumDistance, areaSize, maxLevel,
                                   reinterpret_cast<slam_amd_dmatch*>(matches.data()), &n));
        matches.resize(static_cast<size_t>(n));
        return matches;
    }
    slam_amd_matcher* m_ = nullptr;
};

// ---------------------------------------------------------------------------
// FramePipeline: extraction + matching against the previous frame over a stream of frames in
// host memory, with upload, compute and download overlapped (three HIP streams, a ring of
// slots).  The reference's per-frame loop (main.cpp:110-140) without its serial copies.
// ---------------------------------------------------------------------------
class FramePipeline {
public:
    struct Result {
        const KeyPoint* keypoints = nullptr;   // n records, pinned host memory of the slot
        const uint8_t* descriptors = nullptr;  // n x 32
        int n = 0;
        const DMatch* matches = nullptr;       // frameMatch(previous frame, this frame): queryIdx = previous frame's keypoint
        int nMatches = 0;
    };
    // batch > 1: frames are queued `batch` at a time (see slam_amd_pipeline_create_batched); flush() queues an
    // incomplete batch
    FramePipeline(const slam_amd_orb_params& params, int width, int height, int capacity, int depth = 3,
                  float maximumDistance = 300, float areaSize = -1, int maxLevel = -1, int device = 0, int batch = 1)
        : width_(width), height_(height) {
        check(slam_amd_pipeline_create_batched(&params, width, height, capacity, depth, batch, device, maximumDistance,
                                               areaSize, maxLevel, &p_));
    }
    FramePipeline(const FramePipeline&) = delete;
    FramePipeline& operator=(const FramePipeline&) = delete;
    ~FramePipeline() { slam_amd_pipeline_destroy(p_); }

    // Copies `image` (width x height, 8-bit grey) into the next slot and enqueues its processing.
    // Returns the frame's index.  For zero-copy producers: input() + submit().
    long long push(const Image8& image) {
        if (image.cols != width_ || image.rows != height_) throw Error(SLAM_AMD_ERR_ARG, "frame size differs from the pipeline's");
        int stride = 0;
        uint8_t* dst = input(&stride);
        for (int y = 0; y < height_; y++)
            std::memcpy(dst + static_cast<size_t>(y) * static_cast<size_t>(stride), image.data + static_cast<size_t>(y) * image.step,
                        static_cast<size_t>(width_));
        return submit();
    }
    uint8_t* input(int* rowStride) {
        uint8_t* buf = nullptr;
        check(slam_amd_pipeline_input(p_, next_, &buf, rowStride));
        return buf;
    }
    long long submit() {
        check(slam_amd_pipeline_submit(p_, next_));
        return next_++;
    }
    void flush() { check(slam_amd_pipeline_flush(p_)); }
    // Blocks until frame `index` is done.  The pointers stay valid until `depth` more frames were pushed.
    Result result(long long index) const {
        Result r;
        const slam_amd_keypoint* kp = nullptr;
        const slam_amd_dmatch* mt = nullptr;
        check(slam_amd_pipeline_result(p_, index, &kp, &r.descriptors, &r.n, &mt, &r.nMatches));
        r.keypoints = reinterpret_cast<const KeyPoint*>(kp);
        r.matches = reinterpret_cast<const DMatch*>(mt);
        return r;
    }

private:
    slam_amd_pipeline* p_ = nullptr;
    int width_, height_;
    long long next_ = 0;
};

}  // namespace slam_amd

#endif
