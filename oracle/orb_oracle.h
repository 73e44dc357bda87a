/*
 * orb_oracle.h -- CPU restatement of the ORB frontend hot path of pxl-th/slam.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing shipped (the HIP library, its host layer)
 * may include, link or call this; only tests/, the smoke check and the CPU
 * baseline leg of the benchmark do, and there only as the checker / baseline.
 *
 * PARITY UNPINNED.  The reference delegates this path to OpenCV
 * (cv::ORB::detectAndCompute at src/frame/detector.cpp:11 and
 * cv::BFMatcher::radiusMatch at src/frame/matcher.cpp:43-45,111-113).  OpenCV is
 * not part of the reference tree, its version is not pinned (CMakeLists.txt:10)
 * and it is not installed where this oracle was written; the reference holds no
 * tests, golden vectors or fixtures for the path.  The OpenCV side is therefore
 * restated from the published algorithm of OpenCV 4.x (ORB, FAST-9/16,
 * KeyPointsFilter, INTER_LINEAR_EXACT resize, separable Gaussian blur,
 * BFMatcher) and checked only by analytic known-answer tests.  The wrapper and
 * gate semantics (src/frame/matcher.cpp:15-161, src/frame/frame.cpp:14-38) are
 * restated from the reference's own sources and are exact.
 *
 * Plain C99, single thread, integer-exact; float steps are written without
 * fused multiply-add (build with -ffp-contract=off).
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16

/* Layout of cv::KeyPoint as ORB fills it (28 bytes). */
typedef struct {
    float x, y;      /* pt, in level-0 pixel units (level coords * level scale) */
    float size;      /* patchSize * level scale */
    float angle;     /* degrees, [0, 360] */
    float response;  /* FAST score */
    int32_t octave;  /* pyramid level */
    int32_t class_id;/* -1 */
} orc_keypoint;

/* Layout of cv::DMatch (16 bytes). */
typedef struct {
    int32_t queryIdx, trainIdx, imgIdx;
    float distance;
} orc_dmatch;

/* Arguments of cv::ORB::create (reference: main.cpp:100) plus fastThreshold. */
typedef struct {
    int32_t nfeatures;      /* 1000 in main.cpp:100 */
    float   scale_factor;   /* 1.2f */
    int32_t nlevels;        /* 8 */
    int32_t edge_threshold; /* 31 */
    int32_t first_level;    /* 0 (only 0 is restated) */
    int32_t wta_k;          /* 2 (only 2 is restated) */
    int32_t score_type;     /* 1 = FAST_SCORE (only this is restated) */
    int32_t patch_size;     /* 31 (only 31 is restated) */
    int32_t fast_threshold; /* 20 */
} orc_params;

enum { ORC_OK = 0, ORC_ERR_ARG = -1, ORC_ERR_UNSUPPORTED = -2, ORC_ERR_CAPACITY = -3, ORC_ERR_NOMEM = -4 };

void orc_default_params(orc_params* p);

/* Pyramid geometry and per-level quotas. */
float orc_level_scale(const orc_params* p, int level);
int orc_pyramid_sizes(const orc_params* p, int w, int h, int* widths, int* heights);
int orc_level_quotas(const orc_params* p, int* quotas);

/* INTER_LINEAR_EXACT coefficient table of one axis: for every destination index
 * the source offset and the 8.8 fixed-point weight of the second tap. */
int orc_linear_exact_coeffs(int src_size, int dst_size, int32_t* offsets, uint16_t* c1);

/* 8-bit single-channel resize, INTER_LINEAR_EXACT. */
int orc_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride,
                            uint8_t* dst, int dw, int dh, int dstride);

/* All levels, tightly packed one after another (level l has widths[l] columns,
 * stride widths[l]).  `out` must hold sum(widths[l]*heights[l]) bytes. */
int orc_build_pyramid(const orc_params* p, const uint8_t* img, int w, int h, int stride, uint8_t* out);

/* FAST-9/16 score of every pixel (0 where not a corner or within 3 px of the edge). */
int orc_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score);

/* FAST-9/16 detection with 3x3 non-maximum suppression, raster order.
 * Writes up to `capacity` (x, y, score) triples; *n is the number found. */
int orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold,
                    int32_t* xys, int capacity, int* n);

/* 7x7 Gaussian blur, sigma 2, 8.8 fixed-point kernel, reflect-101 border. */
int orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);

/* Degree-7 polynomial atan2 in degrees. */
float orc_fast_atan2(float y, float x);

/* Intensity-centroid moments and angle at integer (x, y) of an unblurred level. */
int orc_ic_moments(const uint8_t* img, int stride, int x, int y, int* m01, int* m10);

/* 32-byte rBRIEF descriptor at integer (cx, cy) of a blurred level, angle in degrees. */
int orc_brief_descriptor(const uint8_t* blurred, int stride, int cx, int cy, float angle_deg, uint8_t* desc32);

/* cv::ORB::detectAndCompute without mask.  Keypoints in canonical order
 * (octave, y, x ascending).  Returns ORC_ERR_CAPACITY (and the needed count in
 * *n) when more than `capacity` keypoints result. */
int orc_detect_and_compute(const orc_params* p, const uint8_t* img, int w, int h, int stride,
                           orc_keypoint* kps, uint8_t* desc, int capacity, int* n);

/* The same with an option the reference does not have: grid_cell > 0 keeps, per level
 * and after the border filter, only the best NMS survivor of every grid_cell x
 * grid_cell cell (cells anchored at the level origin; highest score, first in raster
 * order among equals) before the per-level quota is applied.  grid_cell 0 = off. */
int orc_detect_and_compute_ex(const orc_params* p, int grid_cell, int quadtree, const uint8_t* img, int w, int h,
                              int stride, orc_keypoint* kps, uint8_t* desc, int capacity, int* n);
/* Quadtree distribution of one level's survivors with an exact target count (option beyond the
 * reference; see orb_oracle.c).  cand[3 i] = x, y, score inside [rx0, rx1) x [ry0, ry1). */
int orc_quadtree_select(const int32_t* cand, int n, int rx0, int ry0, int rx1, int ry1, int target, uint8_t* keep);
int orc_detect_and_compute_grid(const orc_params* p, int grid_cell, const uint8_t* img, int w, int h, int stride,
                                orc_keypoint* kps, uint8_t* desc, int capacity, int* n);

/* n_frames independent frames of one size (frame f at imgs + f*frame_pitch), each
 * processed by orc_detect_and_compute; frames run in parallel when the oracle is
 * built with OpenMP.  Outputs are [n_frames][capacity] records; counts[f] = n. */
int orc_detect_and_compute_batch(const orc_params* p, const uint8_t* imgs, int n_frames, size_t frame_pitch,
                                 int w, int h, int stride, orc_keypoint* kps, uint8_t* desc,
                                 int capacity, int* counts);

/* orc_match_gated with the queries split over OpenMP threads (same result). */
int orc_match_gated_mt(const uint8_t* qd, const orc_keypoint* qkp, int nq,
                       const uint8_t* td, const orc_keypoint* tkp, int nt,
                       float max_dist, float area, int max_level,
                       orc_dmatch* out, int* n_out);

/* Threads the two calls above use (1 when built without OpenMP). */
int orc_max_threads(void);

/* Hamming distance of two 32-byte rows. */
int orc_hamming256(const uint8_t* a, const uint8_t* b);

/* Literal restatement of radiusMatch + _filterMatches (src/frame/matcher.cpp:43-50,
 * 131-161): materialises every query's candidate list sorted by distance, ties
 * by ascending train index, then walks it.  O(nq*nt) memory: small inputs only. */
int orc_radius_match_filter(const uint8_t* qd, const orc_keypoint* qkp, int nq,
                            const uint8_t* td, const orc_keypoint* tkp, int nt,
                            float max_dist, float area, int max_level,
                            orc_dmatch* out, int* n_out);

/* The same result computed as a gated arg-min, no intermediate lists. */
int orc_match_gated(const uint8_t* qd, const orc_keypoint* qkp, int nq,
                    const uint8_t* td, const orc_keypoint* tkp, int nt,
                    float max_dist, float area, int max_level,
                    orc_dmatch* out, int* n_out);

/* Options beyond the reference (it has neither; both default off in the library):
 *  ratio > 0: Lowe's ratio test -- keep a match only if best < ratio * second best,
 *    second best = next smallest distance among the train rows that pass the gates
 *    (a query with a single candidate is kept);
 *  histo_length > 0: rotation consistency as in ORB-SLAM -- bin = round((angle_q -
 *    angle_t mod 360) * histo_length / 360) mod histo_length over all matches, keep
 *    the matches of the three fullest bins (second / third only if >= 0.1 * first).
 * Order of application: gates and max_dist, ratio, then rotation histogram. */
int orc_match_gated_ex(const uint8_t* qd, const orc_keypoint* qkp, int nq,
                       const uint8_t* td, const orc_keypoint* tkp, int nt,
                       float max_dist, float area, int max_level,
                       float ratio, int histo_length,
                       orc_dmatch* out, int* n_out);

/* Matcher::frameMatch (src/frame/matcher.cpp:15-55) on plain arrays: optional id
 * gather of the query side, match, remap of queryIdx. */
int orc_frame_match(const uint8_t* d1, const orc_keypoint* kp1, int n1,
                    const uint8_t* d2, const orc_keypoint* kp2, int n2,
                    const int32_t* ids, int n_ids,
                    float max_dist, float area, int max_level,
                    orc_dmatch* out, int* n_out);

/* Pinhole projection of world points with pose [R|t] (row-major 3x4, double),
 * intrinsics fx, fy, cx, cy and distortion k1 k2 p1 p2 k3
 * (src/frame/matcher.cpp:163-184). */
int orc_project_points(const double* Rt, const double* K4, const double* dist5,
                       const float* xyz, int n, float* uv);

/* isOutlier over n triangulated points (src/map/mappoint.cpp:54-101): flags[i] = 1 when
 * the point is NaN, its parallax cosine is outside [0, 0.999f], it lies within 1e-6 of
 * a camera centre, or it reprojects more than 1 pixel from matches[i]'s keypoint in
 * either keyframe.  Poses as row-major 3x4 [R|t] (the reference passes R through
 * cv::Rodrigues and back, which this restatement does not repeat). */
int orc_outlier_flags(const float* xyz, const orc_dmatch* matches, int n,
                      const orc_keypoint* kp_q, int nq, const orc_keypoint* kp_t, int nt,
                      const float* center_q, const float* center_t,
                      const double* Rt_q, const double* K4_q, const double* dist5_q,
                      const double* Rt_t, const double* K4_t, const double* dist5_t,
                      uint8_t* flags);

/* Mapper::_isDuplicate over all pairs (src/tracking/mapper.cpp:328-346):
 * flag[i*n2+j] = hamming(d1[i], d2[j]) <= desc_dist && |p1[i]-p2[j]| < point_dist. */
int orc_duplicate_pairs(const uint8_t* d1, const float* p1, int n1,
                        const uint8_t* d2, const float* p2, int n2,
                        int desc_dist, double point_dist, uint8_t* flag);

/* Input front end of the reference's main loop (main.cpp:124-129): BGR -> grey, then
 * cv::resize by scale = (float)target_width / (float)width with INTER_LINEAR.  PARITY
 * UNPINNED like the rest of the OpenCV side: restated from OpenCV 4.x (15-bit grey
 * coefficients, 11-bit resize taps). */
int orc_bgr2gray(const uint8_t* bgr, int w, int h, int sstride, uint8_t* gray, int dstride);
int orc_resize_scale_size(int src_w, int src_h, float scale, int* dst_w, int* dst_h);
int orc_linear_taps(int src_size, int dst_size, double inv_scale, int clamp_frac, int32_t* ofs, int16_t* taps);
int orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride, float scale,
                      uint8_t* dst, int dw, int dh, int dstride);
int orc_frontend(const uint8_t* bgr, int w, int h, int sstride, int target_width,
                 uint8_t* gray_out, int dw, int dh, int dstride);

/* Frame::Frame scale tables (src/frame/frame.cpp:28-37). */
int orc_frame_scale_tables(float scale_factor, int nlevels, float* scales, float* sigma, float* inv_sigma);

#ifdef __cplusplus
}
#endif
#endif
