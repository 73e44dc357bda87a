// opencv_diff -- the library against OpenCV itself, for whoever has OpenCV (it is not installed where
// this library is built, so this file has never been compiled there: `make opencv_diff` builds it
// only when pkg-config finds opencv4).  It is the one way to pin this repository's CPU oracle, and
// with it the GPU path, to what the reference really computes:
//   slam::Detector::detect   = cv::ORB::detectAndCompute        (src/frame/detector.cpp:6-12)
//   slam::Matcher::frameMatch = cv::BFMatcher(NORM_HAMMING)::match with a mask, then the
//                               distance cut                     (src/frame/matcher.cpp:38-55, 131-161)
//
//   opencv_diff [image.png ...]
// Without arguments it runs on procedurally generated frames (blobs, bars and noise, 640x480 and
// 1241x376).  For every image: cv::ORB::create(2000, 1.2f, 8, 31, 0, 2, cv::ORB::FAST_SCORE, 31, 20)
// against slam_amd_extract, field by field, bit for bit; then every frame against the one before,
// gates off, cv::BFMatcher against slam_amd_frame_match.  Prints the first difference of every kind
// and exits with the number of differing images / pairs.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgcodecs.hpp>
#include <opencv2/imgproc.hpp>

#include "slam_amd.h"

static_assert(sizeof(cv::KeyPoint) == sizeof(slam_amd_keypoint), "cv::KeyPoint is not the 28-byte record");
static_assert(sizeof(cv::DMatch) == sizeof(slam_amd_dmatch), "cv::DMatch is not the 16-byte record");

static cv::Mat synthetic_frame(int w, int h, unsigned seed, int shift) {
    cv::Mat img(h, w, CV_8UC1, cv::Scalar(96));
    std::mt19937 rng(seed);
    auto rnd = [&](int n) { return (int)(rng() % (unsigned)n); };
    for (int i = 0; i < 400; i++) {
        const int x = rnd(w) + shift, y = rnd(h), r = 2 + rnd(18), c = rnd(256);
        if (i & 1) cv::circle(img, cv::Point(x % w, y), r, cv::Scalar(c), -1);
        else cv::rectangle(img, cv::Rect(x % w, y, 3 + rnd(40), 3 + rnd(40)) & cv::Rect(0, 0, w, h), cv::Scalar(c), -1);
    }
    cv::Mat noise(h, w, CV_8UC1);
    cv::randn(noise, 0, 6);
    cv::add(img, noise, img, cv::noArray(), CV_8UC1);
    cv::GaussianBlur(img, img, cv::Size(3, 3), 0.8);
    return img;
}

struct Extracted {
    std::vector<cv::KeyPoint> kp;
    cv::Mat desc;
};

static bool same_bits(float a, float b) { return std::memcmp(&a, &b, 4) == 0; }

// 0 if identical
static int diff_extract(const Extracted& cvr, const std::vector<slam_amd_keypoint>& kp, const std::vector<uint8_t>& desc,
                        const char* name) {
    int bad = 0;
    if (cvr.kp.size() != kp.size()) {
        printf("%s: %zu keypoints from OpenCV, %zu from the library\n", name, cvr.kp.size(), kp.size());
        bad = 1;
    }
    const size_t n = std::min(cvr.kp.size(), kp.size());
    const char* fields[] = {"pt.x", "pt.y", "size", "angle", "response", "octave", "class_id", "descriptor"};
    bool seen[8] = {};
    for (size_t i = 0; i < n; i++) {
        const cv::KeyPoint& a = cvr.kp[i];
        const slam_amd_keypoint& b = kp[i];
        const bool eq[8] = {same_bits(a.pt.x, b.x), same_bits(a.pt.y, b.y), same_bits(a.size, b.size),
                            same_bits(a.angle, b.angle), same_bits(a.response, b.response), a.octave == b.octave,
                            a.class_id == b.class_id, std::memcmp(cvr.desc.ptr(i), &desc[32 * i], 32) == 0};
        for (int f = 0; f < 8; f++)
            if (!eq[f] && !seen[f]) {
                seen[f] = true;
                bad = 1;
                printf("%s: first difference in %s at keypoint %zu: OpenCV (%.9g, %.9g) octave %d angle %.9g response %.9g, "
                       "library (%.9g, %.9g) octave %d angle %.9g response %.9g\n",
                       name, fields[f], i, a.pt.x, a.pt.y, a.octave, a.angle, a.response, b.x, b.y, b.octave, b.angle, b.response);
            }
    }
    return bad;
}

int main(int argc, char** argv) {
    std::vector<cv::Mat> frames;
    std::vector<std::string> names;
    for (int i = 1; i < argc; i++) {
        cv::Mat m = cv::imread(argv[i], cv::IMREAD_GRAYSCALE);
        if (m.empty()) { printf("cannot read %s\n", argv[i]); return 100; }
        frames.push_back(m);
        names.push_back(argv[i]);
    }
    if (frames.empty()) {
        for (int i = 0; i < 4; i++) { frames.push_back(synthetic_frame(640, 480, 7u, 3 * i)); names.push_back("synthetic 640x480 #" + std::to_string(i)); }
        for (int i = 0; i < 2; i++) { frames.push_back(synthetic_frame(1241, 376, 11u, 5 * i)); names.push_back("synthetic 1241x376 #" + std::to_string(i)); }
    }
    cv::Ptr<cv::ORB> orb = cv::ORB::create(2000, 1.2f, 8, 31, 0, 2, cv::ORB::FAST_SCORE, 31, 20);
    slam_amd_orb_params p;
    slam_amd_default_orb_params(&p);
    p.nfeatures = 2000;
    p.fast_threshold = 20;
    slam_amd_matcher* matcher = nullptr;
    if (slam_amd_matcher_create(0, &matcher) != SLAM_AMD_OK) { printf("matcher: %s\n", slam_amd_last_error()); return 101; }

    int differing = 0;
    Extracted prev_cv;
    std::vector<slam_amd_keypoint> prev_kp;
    std::vector<uint8_t> prev_desc;
    cv::Size prev_size;
    for (size_t f = 0; f < frames.size(); f++) {
        const cv::Mat& img = frames[f];
        Extracted e;
        orb->detectAndCompute(img, cv::noArray(), e.kp, e.desc);
        slam_amd_extractor* ex = nullptr;
        const int cap = 8192;
        if (slam_amd_extractor_create(&p, img.cols, img.rows, 1, cap, 0, &ex) != SLAM_AMD_OK) { printf("extractor: %s\n", slam_amd_last_error()); return 102; }
        std::vector<slam_amd_keypoint> kp(cap);
        std::vector<uint8_t> desc((size_t)cap * 32);
        int n = 0;
        const int rc = slam_amd_extract(ex, img.data, img.cols, img.rows, (int)img.step, kp.data(), desc.data(), cap, &n);
        slam_amd_extractor_destroy(ex);
        if (rc != SLAM_AMD_OK) { printf("%s: slam_amd_extract: %s\n", names[f].c_str(), slam_amd_last_error()); return 103; }
        kp.resize(n);
        desc.resize((size_t)n * 32);
        const int bad = diff_extract(e, kp, desc, names[f].c_str());
        printf("%s: %zu keypoints, extraction %s\n", names[f].c_str(), e.kp.size(), bad ? "DIFFERS" : "identical");
        differing += bad;

        if (f > 0 && img.size() == prev_size && !e.kp.empty() && !prev_cv.kp.empty()) {
            // the reference's frameMatch with all gates off: query = previous frame, train = this one
            cv::BFMatcher bf(cv::NORM_HAMMING);
            std::vector<cv::DMatch> m_cv;
            bf.match(prev_cv.desc, e.desc, m_cv);
            std::vector<slam_amd_dmatch> m_lib(prev_kp.size());
            int nm = 0;
            if (slam_amd_frame_match(matcher, prev_desc.data(), prev_kp.data(), (int)prev_kp.size(), desc.data(), kp.data(), n,
                                     nullptr, 0, nullptr, 300.0f, -1.0f, -1, m_lib.data(), &nm) != SLAM_AMD_OK) {
                printf("frame_match: %s\n", slam_amd_last_error());
                return 104;
            }
            int mbad = 0;
            if (bad || prev_cv.kp.size() != prev_kp.size()) {
                printf("  (matching compared on differing descriptor sets: differences below follow from those above)\n");
            }
            if ((size_t)nm != m_cv.size()) { printf("  %zu matches from OpenCV, %d from the library\n", m_cv.size(), nm); mbad = 1; }
            for (size_t i = 0; i < std::min(m_cv.size(), (size_t)nm); i++) {
                if (m_cv[i].queryIdx != m_lib[i].queryIdx || m_cv[i].trainIdx != m_lib[i].trainIdx ||
                    !same_bits(m_cv[i].distance, m_lib[i].distance)) {
                    printf("  first differing match %zu: OpenCV %d -> %d (%g), library %d -> %d (%g)\n", i, m_cv[i].queryIdx,
                           m_cv[i].trainIdx, m_cv[i].distance, m_lib[i].queryIdx, m_lib[i].trainIdx, m_lib[i].distance);
                    mbad = 1;
                    break;
                }
            }
            printf("  match against the frame before: %zu matches, %s\n", m_cv.size(), mbad ? "DIFFERS" : "identical");
            differing += mbad;
        }
        prev_cv = e; prev_kp = kp; prev_desc = desc; prev_size = img.size();
    }
    slam_amd_matcher_destroy(matcher);
    printf("%d differing image(s) / pair(s)\n", differing);
    return differing;
}
