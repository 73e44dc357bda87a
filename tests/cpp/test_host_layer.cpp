// Drives the C++ host layer (include/slam_amd.hpp) the way the reference's callers
// drive slam::Detector / slam::Matcher, and checks every result against the CPU
// oracle.  The parameter sets are the reference's call sites:
//   main.cpp:82            frameMatch(kf1, kf2, {}, 300, 50)
//   tracker.cpp:74 / :77   mappointsFrameMatch(last, cur, 300, 50) / (..., 300, -1, -1)
//   tracker.cpp:84 / :99   projectionMatch(last, cur, 300, 50) / (..., 300, -1, -1)
//   mapper.cpp:34          frameMatch(initial, current, {}, 300, -1, 4)
//   mapper.cpp:115         inverseMappointsFrameMatch(current, keyframe, 280, -1, 3)
//   mapper.cpp:178         mappointsFrameMatch(connection, current)
//   mapper.cpp:292-346     duplicate map points
// Needs a GPU.  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/slam_amd.hpp"
extern "C" {
#include "../../oracle/orb_oracle.h"
}

static int g_fail = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { g_fail++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static unsigned rng_state = 12345;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static std::vector<uint8_t> make_scene(int w, int h, int shift) {
    std::vector<uint8_t> img(static_cast<size_t>(w) * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            img[static_cast<size_t>(y) * w + x] = static_cast<uint8_t>(60 + 40 * std::sin((x + shift) * 0.05) + 30 * std::cos(y * 0.07));
    rng_state = 777;
    for (int i = 0; i < 700; i++) {
        int cx = static_cast<int>(rnd() % w) + shift, cy = static_cast<int>(rnd() % h), r = 3 + static_cast<int>(rnd() % 14);
        uint8_t v = static_cast<uint8_t>(rnd() % 256);
        for (int y = std::max(0, cy - r); y < std::min(h, cy + r); y++)
            for (int x = std::max(0, cx - r); x < std::min(w, cx + r); x++) img[static_cast<size_t>(y) * w + x] = v;
    }
    return img;
}

struct Kf {
    slam_amd::FrameData frame;
    std::vector<int> ids;
    std::vector<slam_amd::Point3f> pos;
    float pose[16];
    slam_amd::KeyFrameView view() const {
        slam_amd::KeyFrameView v;
        v.descriptors = frame.descriptors.bytes.data();
        v.undistortedKeypoints = frame.undistortedKeypoints.data();
        v.n = static_cast<int>(frame.undistortedKeypoints.size());
        v.mappointIds = ids.data();
        v.mappointPositions = pos.data();
        v.nMappoints = static_cast<int>(ids.size());
        v.pose = pose;
        v.fx = 963.0; v.fy = 950.4; v.cx = 539.5; v.cy = 404.5;  // data/calibration.yaml:9-15
        return v;
    }
};

static bool same(const std::vector<slam_amd::DMatch>& a, const std::vector<orc_dmatch>& b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(orc_dmatch)) == 0);
}

static std::vector<orc_dmatch> ref_frame_match(const Kf& a, const Kf& b, const std::vector<int>& ids,
                                               float md, float area, int lvl, const float* qxy = nullptr) {
    const auto* k1 = reinterpret_cast<const orc_keypoint*>(a.frame.undistortedKeypoints.data());
    const auto* k2 = reinterpret_cast<const orc_keypoint*>(b.frame.undistortedKeypoints.data());
    const int n1 = static_cast<int>(a.frame.undistortedKeypoints.size()), n2 = static_cast<int>(b.frame.undistortedKeypoints.size());
    std::vector<orc_keypoint> kq(k1, k1 + n1);
    if (qxy) for (size_t i = 0; i < ids.size(); i++) { kq[ids[i]].x = qxy[2 * i]; kq[ids[i]].y = qxy[2 * i + 1]; }
    std::vector<orc_dmatch> out(static_cast<size_t>(std::max<int>(n1, 1)));
    int n = 0;
    orc_frame_match(a.frame.descriptors.bytes.data(), kq.data(), n1, b.frame.descriptors.bytes.data(), k2, n2,
                    ids.empty() ? nullptr : ids.data(), static_cast<int>(ids.size()), md, area, lvl, out.data(), &n);
    out.resize(static_cast<size_t>(n));
    return out;
}

int main() {
    const int W = 480, H = 360;
    slam_amd_orb_params p;
    slam_amd_default_orb_params(&p);  // main.cpp:100
    try {
        slam_amd::Detector detector(p);
        slam_amd::Matcher matcher;
        Kf kf[2];
        for (int f = 0; f < 2; f++) {
            std::vector<uint8_t> img = make_scene(W, H, f * 9);
            slam_amd::Image8 view; view.data = img.data(); view.cols = W; view.rows = H; view.step = W;
            kf[f].frame = slam_amd::makeFrame(detector, view);
            // oracle extraction
            orc_params op; orc_default_params(&op);
            std::vector<orc_keypoint> okp(8192); std::vector<uint8_t> od(8192 * 32); int on = 0;
            int rc = orc_detect_and_compute(&op, img.data(), W, H, W, okp.data(), od.data(), 8192, &on);
            EXPECT(rc == 0, "oracle rc %d", rc);
            EXPECT(on == static_cast<int>(kf[f].frame.keypoints.size()), "keypoint count %d vs %zu", on, kf[f].frame.keypoints.size());
            EXPECT(on > 300, "too few keypoints: %d", on);
            EXPECT(std::memcmp(okp.data(), kf[f].frame.keypoints.data(), static_cast<size_t>(on) * 28) == 0, "keypoints differ");
            EXPECT(std::memcmp(od.data(), kf[f].frame.descriptors.bytes.data(), static_cast<size_t>(on) * 32) == 0, "descriptors differ");
            EXPECT(kf[f].frame.undistortedKeypoints.size() == kf[f].frame.keypoints.size(), "undistorted copy");
            float s[8], g[8], inv[8];
            orc_frame_scale_tables(1.2f, 8, s, g, inv);
            EXPECT(kf[f].frame.scales.size() == 8 && std::memcmp(s, kf[f].frame.scales.data(), 32) == 0, "scales");
            EXPECT(std::memcmp(g, kf[f].frame.sigma.data(), 32) == 0 && std::memcmp(inv, kf[f].frame.invSigma.data(), 32) == 0, "sigma tables");
            // every third keypoint carries a map point
            const int n = static_cast<int>(kf[f].frame.keypoints.size());
            for (int i = f; i < n; i += 3) {
                kf[f].ids.push_back(i);
                kf[f].pos.push_back({(rnd() % 4000) / 1000.f - 2.f, (rnd() % 3000) / 1000.f - 1.5f, 2.f + (rnd() % 5000) / 1000.f});
            }
            const float c = std::cos(0.05f * (f + 1)), sn = std::sin(0.05f * (f + 1));
            const float pose[16] = {c, 0, sn, 0.1f, 0, 1, 0, -0.05f, -sn, 0, c, 0.2f, 0, 0, 0, 1};
            std::memcpy(kf[f].pose, pose, sizeof pose);
        }
        EXPECT(detector.getLevels() == 8 && detector.getScaleFactor() == static_cast<double>(1.2f), "detector getters");
        {   // empty image -> empty outputs (frame.cpp:24)
            std::vector<slam_amd::KeyPoint> k; slam_amd::Descriptors d;
            detector.detect(slam_amd::Image8(), k, d);
            EXPECT(k.empty() && d.rows == 0, "empty image");
        }
        const auto a = kf[0].view(), b = kf[1].view();
        // frameMatch
        EXPECT(same(matcher.frameMatch(a, b, {}, 300, 50), ref_frame_match(kf[0], kf[1], {}, 300, 50, 4)), "frameMatch 300/50/4");
        EXPECT(same(matcher.frameMatch(a, b, {}, 300, -1, 4), ref_frame_match(kf[0], kf[1], {}, 300, -1, 4)), "frameMatch 300/-1/4");
        EXPECT(same(matcher.frameMatch(a, b, {}), ref_frame_match(kf[0], kf[1], {}, 300, -1, 4)), "frameMatch defaults");
        // mappointsFrameMatch
        EXPECT(same(matcher.mappointsFrameMatch(a, b, 300, 50), ref_frame_match(kf[0], kf[1], kf[0].ids, 300, 50, 4)), "mappoints 300/50");
        EXPECT(same(matcher.mappointsFrameMatch(a, b, 300, -1, -1), ref_frame_match(kf[0], kf[1], kf[0].ids, 300, -1, -1)), "mappoints loose");
        EXPECT(same(matcher.mappointsFrameMatch(a, b), ref_frame_match(kf[0], kf[1], kf[0].ids, 300, -1, 4)), "mappoints defaults");
        // inverseMappointsFrameMatch: ids = all \ mappoint ids
        std::vector<int> inv;
        for (int i = 0, k = 0; i < a.n; i++) {
            while (k < a.nMappoints && a.mappointIds[k] < i) k++;
            if (!(k < a.nMappoints && a.mappointIds[k] == i)) inv.push_back(i);
        }
        EXPECT(same(matcher.inverseMappointsFrameMatch(a, b, 280, -1, 3), ref_frame_match(kf[0], kf[1], inv, 280, -1, 3)), "inverse 280/-1/3");
        {
            Kf none = kf[0]; none.ids.clear(); none.pos.clear();
            EXPECT(same(matcher.inverseMappointsFrameMatch(none.view(), b, 280, -1, 3), ref_frame_match(kf[0], kf[1], {}, 280, -1, 3)), "inverse without map points");
        }
        // projectionMatch: map points of `from` projected with the pose of `to`
        {
            double Rt[12]; for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) Rt[r * 4 + c] = kf[1].pose[r * 4 + c];
            const double K4[4] = {963.0, 950.4, 539.5, 404.5};
            std::vector<float> uv(kf[0].pos.size() * 2);
            orc_project_points(Rt, K4, nullptr, &kf[0].pos[0].x, static_cast<int>(kf[0].pos.size()), uv.data());
            auto got = matcher.projectMapPoints(a, b);
            EXPECT(got.size() == kf[0].pos.size() && std::memcmp(got.data(), uv.data(), uv.size() * 4) == 0, "projected points");
            EXPECT(same(matcher.projectionMatch(a, b, 300, 50), ref_frame_match(kf[0], kf[1], kf[0].ids, 300, 50, 4, uv.data())), "projection 300/50");
            // area -1: no projection at all, plain map-point matching (matcher.cpp:114-120)
            EXPECT(same(matcher.projectionMatch(a, b, 300, -1, -1), ref_frame_match(kf[0], kf[1], kf[0].ids, 300, -1, -1)), "projection loose");
            Kf none = kf[0]; none.ids.clear(); none.pos.clear();
            EXPECT(matcher.projectionMatch(none.view(), b, 300, 50).empty(), "projection without map points");
        }
        // duplicates
        {
            const int n1 = static_cast<int>(kf[0].ids.size()), n2 = static_cast<int>(kf[1].ids.size());
            std::vector<uint8_t> d1(static_cast<size_t>(n1) * 32), d2(static_cast<size_t>(n2) * 32);
            for (int i = 0; i < n1; i++) std::memcpy(&d1[static_cast<size_t>(i) * 32], kf[0].frame.descriptors.row(kf[0].ids[i]), 32);
            for (int i = 0; i < n2; i++) std::memcpy(&d2[static_cast<size_t>(i) * 32], kf[1].frame.descriptors.row(kf[1].ids[i]), 32);
            std::vector<slam_amd::Point3f> p2 = kf[1].pos;
            for (int i = 0; i < std::min(n1, n2); i += 2) { p2[i] = kf[0].pos[i]; p2[i].x += 1e-3f; std::memcpy(&d2[static_cast<size_t>(i) * 32], &d1[static_cast<size_t>(i) * 32], 32); }
            auto got = matcher.duplicatePairs(d1.data(), kf[0].pos.data(), n1, d2.data(), p2.data(), n2);
            std::vector<uint8_t> ref(static_cast<size_t>(n1) * n2);
            orc_duplicate_pairs(d1.data(), &kf[0].pos[0].x, n1, d2.data(), &p2[0].x, n2, 100, 1e-2, ref.data());
            EXPECT(got == ref, "duplicate flags");
            int dups = 0; for (uint8_t v : ref) dups += v;
            EXPECT(dups >= n1 / 4, "expected duplicates, got %d", dups);
        }
        // unsupported ORB variant: loud error, no fallback
        {
            slam_amd_orb_params q = p; q.score_type = 0;  // HARRIS_SCORE
            slam_amd::Detector harris(q);
            std::vector<uint8_t> img = make_scene(W, H, 0);
            slam_amd::Image8 view; view.data = img.data(); view.cols = W; view.rows = H; view.step = W;
            std::vector<slam_amd::KeyPoint> k; slam_amd::Descriptors d;
            bool threw = false;
            try { harris.detect(view, k, d); } catch (const slam_amd::Error& e) { threw = e.status == SLAM_AMD_ERR_UNSUPPORTED; }
            EXPECT(threw, "Harris score must be rejected");
        }
    } catch (const slam_amd::Error& e) {
        printf("slam_amd::Error %d: %s\n", e.status, e.what());
        return 2;
    }
    printf(g_fail ? "%d check(s) failed\n" : "host layer ok (%d failures)\n", g_fail);
    return g_fail ? 1 : 0;
}
