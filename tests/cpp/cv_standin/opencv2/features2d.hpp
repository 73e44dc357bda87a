// TEST-ONLY stand-in for cv::ORB (a carrier of its creation parameters) and cv::BFMatcher
// (Hamming distance, k = 1), see core.hpp in this directory.
#ifndef SLAM_AMD_TEST_CV_STANDIN_FEATURES2D_HPP
#define SLAM_AMD_TEST_CV_STANDIN_FEATURES2D_HPP

#include "core.hpp"

namespace cv {

class ORB {
public:
    enum ScoreType { HARRIS_SCORE = 0, FAST_SCORE = 1 };
    static Ptr<ORB> create(int nfeatures = 500, float scaleFactor = 1.2f, int nlevels = 8, int edgeThreshold = 31,
                           int firstLevel = 0, int WTA_K = 2, ScoreType scoreType = HARRIS_SCORE, int patchSize = 31,
                           int fastThreshold = 20) {
        Ptr<ORB> o(new ORB);
        o->nfeatures = nfeatures; o->scaleFactor = scaleFactor; o->nlevels = nlevels; o->edgeThreshold = edgeThreshold;
        o->firstLevel = firstLevel; o->wta_k = WTA_K; o->scoreType = scoreType; o->patchSize = patchSize;
        o->fastThreshold = fastThreshold;
        return o;
    }
    // declared only (syntax checks of code written against OpenCV)
    void detectAndCompute(InputArray image, InputArray mask, std::vector<KeyPoint>& keypoints, OutputArray descriptors,
                          bool useProvidedKeypoints = false);
    int getMaxFeatures() const { return nfeatures; }
    double getScaleFactor() const { return scaleFactor; }
    int getNLevels() const { return nlevels; }
    int getEdgeThreshold() const { return edgeThreshold; }
    int getFirstLevel() const { return firstLevel; }
    int getWTA_K() const { return wta_k; }
    ScoreType getScoreType() const { return scoreType; }
    int getPatchSize() const { return patchSize; }
    int getFastThreshold() const { return fastThreshold; }
private:
    int nfeatures, nlevels, edgeThreshold, firstLevel, wta_k, patchSize, fastThreshold;
    float scaleFactor;
    ScoreType scoreType;
};

class BFMatcher {
public:
    enum { BRUTEFORCE_HAMMING = 4 };
    BFMatcher() {}
    explicit BFMatcher(int normType, bool crossCheck = false) : norm(normType), cross(crossCheck) {}
    static Ptr<BFMatcher> create(int normType = BRUTEFORCE_HAMMING, bool crossCheck = false) {
        Ptr<BFMatcher> m(new BFMatcher);
        m->norm = normType; m->cross = crossCheck;
        return m;
    }
    // nearest train row of every query row (the reference only calls this with 1 x 32 rows)
    void match(const Mat& q, const Mat& t, std::vector<DMatch>& out) const {
        out.clear();
        for (int i = 0; i < q.rows; i++) {
            int best = -1, bd = 1 << 30;
            for (int j = 0; j < t.rows; j++) {
                int d = 0;
                for (int k = 0; k < q.cols; k++) d += __builtin_popcount(static_cast<unsigned>(q.ptr<uint8_t>(i)[k] ^ t.ptr<uint8_t>(j)[k]));
                if (d < bd) { bd = d; best = j; }
            }
            if (best >= 0) out.push_back(DMatch(i, best, 0, static_cast<float>(bd)));
        }
    }
    int norm = BRUTEFORCE_HAMMING;
    bool cross = false;
};

}  // namespace cv

#endif
