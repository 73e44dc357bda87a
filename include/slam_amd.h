/*
 * slam_amd.h -- C ABI of the MI355X (gfx950) ORB frontend: pyramid, FAST-9/16,
 * per-level best-N selection, intensity-centroid orientation, rBRIEF descriptors
 * and gated brute-force Hamming matching, as hand-written HIP kernels.
 *
 * This is the boundary a maintainer of pxl-th/slam binds to in place of the two
 * OpenCV calls that carry the reference's frontend arithmetic:
 *   cv::ORB::detectAndCompute      (reference: src/frame/detector.cpp:11)
 *   cv::BFMatcher::radiusMatch     (reference: src/frame/matcher.cpp:43-45, 111-113)
 * plus the post-filter and projection around them (src/frame/matcher.cpp:131-184).
 * The C++ classes that keep the reference's slam::Detector / slam::Matcher
 * signatures on top of this ABI are in slam_amd.hpp.
 *
 * Conventions: every function returns an int status (0 = ok, negative = error),
 * never throws, never takes ownership of caller memory.  Record layouts equal
 * cv::KeyPoint (28 bytes) and cv::DMatch (16 bytes).  "Host" entry points take
 * host pointers and block until results are on the host, as the reference's
 * callers expect (src/tracking/tracker.cpp:74-75).  "_async" entry points take
 * device pointers and a hipStream_t (passed as void*), enqueue work and return.
 * One context serves one host thread at a time.  There is no CPU fallback: with
 * no usable GPU the create calls fail with SLAM_AMD_ERR_NO_DEVICE.
 *
 * Keypoint order is canonical: (octave, y, x) ascending.  Match ties (equal
 * Hamming distance) resolve to the lowest train index.
 */
#ifndef SLAM_AMD_H
#define SLAM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_AMD_MAX_LEVELS 16
#define SLAM_AMD_VERSION 100

enum {
    SLAM_AMD_OK = 0,
    SLAM_AMD_ERR_ARG = -1,         /* null pointer, bad size, bad parameter value */
    SLAM_AMD_ERR_UNSUPPORTED = -2, /* ORB variant not implemented (Harris score, WTA_K != 2, ...) */
    SLAM_AMD_ERR_CAPACITY = -3,    /* more results than the caller's buffer holds; *n_out = needed */
    SLAM_AMD_ERR_NOMEM = -4,
    SLAM_AMD_ERR_HIP = -5,         /* a HIP runtime call failed; see slam_amd_last_error() */
    SLAM_AMD_ERR_NO_DEVICE = -6    /* no gfx950-compatible GPU visible */
};

/* Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} slam_amd_keypoint;

/* Same 16-byte layout as cv::DMatch (queryIdx, trainIdx, imgIdx, distance). */
typedef struct {
    int32_t queryIdx, trainIdx, imgIdx;
    float distance;
} slam_amd_dmatch;

/* Arguments of cv::ORB::create as the reference passes them (main.cpp:100:
 * 1000, 1.2f, 8, 31, 0, 2, FAST_SCORE) plus patchSize and fastThreshold. */
typedef struct {
    int32_t nfeatures;
    float   scale_factor;
    int32_t nlevels;        /* 1..SLAM_AMD_MAX_LEVELS */
    int32_t edge_threshold; /* >= 31 */
    int32_t first_level;    /* must be 0 */
    int32_t wta_k;          /* must be 2 */
    int32_t score_type;     /* must be 1 (cv::ORB::FAST_SCORE) */
    int32_t patch_size;     /* must be 31 */
    int32_t fast_threshold; /* 0..255, OpenCV default 20 */
} slam_amd_orb_params;

typedef struct slam_amd_extractor slam_amd_extractor;
typedef struct slam_amd_matcher slam_amd_matcher;

/* ---- general ---------------------------------------------------------- */

int slam_amd_version(void);
const char* slam_amd_status_string(int status);
/* Message of the last failing call on this thread ("" if none). */
const char* slam_amd_last_error(void);
int slam_amd_device_count(int* count);
/* Fills the parameters of main.cpp:100 (nfeatures 1000, 1.2, 8 levels, ...). */
void slam_amd_default_orb_params(slam_amd_orb_params* p);

/* ---- extraction: replaces cv::ORB::detectAndCompute ---------------------
 * reference: slam::Detector::detect, src/frame/detector.cpp:6-12 */

/* A context is bound to one image size, one device, a maximum batch of frames
 * per call and a maximum number of keypoints per frame. */
int slam_amd_extractor_create(const slam_amd_orb_params* params, int width, int height,
                              int max_batch, int max_keypoints, int device,
                              slam_amd_extractor** out);
int slam_amd_extractor_destroy(slam_amd_extractor* ex);

/* Pyramid geometry: any output pointer may be null.  Arrays hold nlevels entries.
 * reference: Detector::getLevels / getScaleFactor, include/frame/detector.hpp:35-36 */
int slam_amd_extractor_levels(const slam_amd_extractor* ex, int* nlevels, int* widths,
                              int* heights, int* quotas, float* scales);

/* Host, blocking.  gray: 8-bit single channel, `stride` bytes per row.  Writes at
 * most `capacity` keypoints (28-byte records) and descriptors (row-major N x 32).
 * An image of the wrong size is SLAM_AMD_ERR_ARG.  More keypoints than capacity
 * (score ties can exceed nfeatures, include/frame/detector.hpp:25-27) is
 * SLAM_AMD_ERR_CAPACITY with *n_out = the number needed and nothing written. */
int slam_amd_extract(slam_amd_extractor* ex, const uint8_t* gray, int width, int height,
                     int stride, slam_amd_keypoint* kp_out, uint8_t* desc_out,
                     int capacity, int* n_out);
/* Host, blocking.  Copies out the keypoints and descriptors of the last slam_amd_extract
 * call of this context again (they stay on the device until the next call): the way to
 * complete a call that ended in SLAM_AMD_ERR_CAPACITY without extracting a second time. */
int slam_amd_extract_fetch(slam_amd_extractor* ex, slam_amd_keypoint* kp_out, uint8_t* desc_out,
                           int capacity, int* n_out);

/* Device, asynchronous, batched.  d_images: `batch` frames, frame b starts at
 * d_images + b*frame_pitch, rows `row_stride` bytes apart.  Outputs are
 * [batch][capacity] keypoints, [batch][capacity][32] descriptor bytes and
 * [batch] counts.  d_count[b] is the true number found; when it exceeds
 * `capacity` only the first `capacity` (canonical order) are written. */
int slam_amd_extract_batch_async(slam_amd_extractor* ex, const uint8_t* d_images, int batch,
                                 size_t frame_pitch, int row_stride,
                                 slam_amd_keypoint* d_kp, uint8_t* d_desc, int32_t* d_count,
                                 int capacity, void* stream);

/* Grid distribution of keypoints -- not in the reference, whose cv::ORB keeps the
 * per-level best-N by score only (src/frame/detector.cpp:11).  cell = 8, 16 or 32: per
 * level and after the border filter, only the best NMS survivor of every cell x cell
 * block of pixels (blocks anchored at the level origin; highest FAST score, first in
 * raster order among equals) takes part in the per-level best-N selection.  cell = 0
 * (the state after create) switches it off.  Applies to every later extract call. */
int slam_amd_extractor_set_grid(slam_amd_extractor* ex, int cell);

/* Quadtree distribution of keypoints with an exact per-level count -- not in the reference
 * either (it is ORB-SLAM's DistributeOctTree made deterministic).  on = 1: a level whose NMS
 * survivors (after the border filter) exceed its quota keeps exactly the quota, chosen as
 * follows.  The region [edge, w - edge) x [edge, h - edge) is cut into max(1, round(W / H))
 * root cells side by side; a cell splits at x0 + ceil(w / 2), y0 + ceil(h / 2), empty children
 * are dropped, cells are ordered by path (root, then quadrant 2 * lower + right per split).
 * While splitting every cell with more than one survivor still leaves fewer cells than the
 * quota, all are split; in the last round they are split one by one, most populated first
 * (path order among equals), until the quota is reached.  Every cell keeps its best survivor
 * (highest score, then lowest y, then lowest x); if that is more than the quota, the best
 * quota of them stay.  Levels with no more survivors than the quota, or with more than 8192,
 * are selected as without the option.  Excludes the grid option; levels up to 4096 pixels
 * per side.  on = 0 (the state after create) switches it off. */
int slam_amd_extractor_set_quadtree(slam_amd_extractor* ex, int on);

/* Copies one level of frame `frame` of the last batch to the host (widths[l]
 * columns, heights[l] rows, host_stride bytes per row): the unblurred pyramid
 * level, or the 7x7-blurred level the descriptors are sampled from.  For tests. */
int slam_amd_extractor_debug_level(slam_amd_extractor* ex, int frame, int level,
                                   uint8_t* host_out, int host_stride);
int slam_amd_extractor_debug_blurred_level(slam_amd_extractor* ex, int frame, int level,
                                           uint8_t* host_out, int host_stride);
/* The selection step ranks the kept corners of a level from a cache of 4096 entries in LDS and
 * falls back to ranking against the lists in memory when a level keeps more (long runs of equal
 * scores).  This lowers the cache size in use (0..4096) so that tests can force that path. */
int slam_amd_extractor_debug_set_select_cache(slam_amd_extractor* ex, int entries);

/* ---- matching: replaces cv::BFMatcher::radiusMatch + _filterMatches ------
 * reference: slam::Matcher::frameMatch, src/frame/matcher.cpp:15-55 and
 * Matcher::_filterMatches, src/frame/matcher.cpp:131-161 */

int slam_amd_matcher_create(int device, slam_amd_matcher** out);
int slam_amd_matcher_destroy(slam_amd_matcher* m);
/* Sizes the matcher's device scratch for calls of at most `batch` problems of at most
 * `query_rows` x `train_rows` rows each (train_rows may be 0 when no call will have the
 * window gate on): 4 bytes per query row, 128 more while the ratio option is set, 44.5 per
 * row of both sets for the reordered copies of window-gated calls.  Later _async calls
 * within these sizes allocate nothing.
 * Rule for HIP graphs: an _async call issued while its stream is capturing never
 * allocates -- if the scratch is too small it returns SLAM_AMD_ERR_CAPACITY -- and once a
 * call has been captured the scratch never moves again (a later call that would need more
 * returns SLAM_AMD_ERR_CAPACITY too, because the instantiated graph holds the old
 * address).  So: reserve (and set the options) first, then capture. */
int slam_amd_matcher_reserve(slam_amd_matcher* m, int batch, int query_rows, int train_rows);

/* Two post-filters the reference does not have (it calls radiusMatch and keeps the
 * nearest gated row, src/frame/matcher.cpp:131-161); both are off after create and
 * apply to every later frame_match call of this matcher, after the gates and max_dist:
 *   ratio > 0         keep a match only if best < ratio * second best, the second best
 *                     being the next smallest distance among the rows passing the gates
 *                     (a query with one candidate is kept);
 *   histo_length > 0  rotation consistency: histogram of (angle_query - angle_train)
 *                     mod 360 over the matches of one call in histo_length (<= 64) bins,
 *                     bin = round(diff * histo_length / 360) mod histo_length; matches
 *                     outside the three fullest bins are dropped, and the second / third
 *                     bin counts only if it holds at least a tenth of the first.
 *                     Needs the keypoints of both sets.
 * With ratio > 0 the scratch of slam_amd_matcher_reserve grows to 132 bytes per row.
 * slam_amd_match_pairs_async returns SLAM_AMD_ERR_UNSUPPORTED while either is set. */
int slam_amd_matcher_set_options(slam_amd_matcher* m, float ratio, int histo_length);

/* Test hook.  With the window gate on (area > 0) both sets are first reordered by grid
 * cell (index words and positions in the matcher's scratch), and every group of neighbouring
 * queries is matched only against the blocks of train rows that lie within the window of
 * it; the result is identical to the full sweep.  mode 0 (default): train sets of up to
 * 32768 rows go through the matrix-core kernel for reordered sets, larger ones through the
 * popcount kernel for reordered sets once queries x train rows >= 2^30, the rest through the
 * full sweep; 1 forces the full sweep; 2 forces the popcount kernel on reordered sets; 3 the
 * matrix-core kernel on reordered sets (where the train set fits, else as 0). */
int slam_amd_matcher_debug_set_window_mode(slam_amd_matcher* m, int mode);

/* Test hook.  The full sweep (every query against every train row) forms its distances on the
 * matrix cores -- Hamming distance as popcount(b) + sum t (1 - 2 b) over bytes, the octave and
 * window gates applied to the products before the arg-min -- with or without the gates and the
 * ratio option; the result is identical to the popcount kernel's.  mode 0 = matrix cores
 * (default), 1 = always the popcount kernel. */
int slam_amd_matcher_debug_set_sweep_kernel(slam_amd_matcher* m, int mode);

/* Host, blocking.  For every query row (ascending), the train row of least
 * Hamming distance among those that pass the gates:
 *   distance <= max_dist                                   (matcher.cpp:43-45)
 *   not (max_level != -1 && q.octave > max_level && t.octave > max_level)
 *                                                           (matcher.cpp:144-148)
 *   area == -1 || (|q.x - t.x| < area && |q.y - t.y| < area) (matcher.cpp:149-157)
 * ids (may be null / n_ids 0): rows of set 1 to use as queries, in that order;
 * queryIdx of the result is the id (matcher.cpp:26-36, 51-53).
 * q_xy (may be null): 2 floats per query replacing the query positions in the
 * window gate (projectionMatch, matcher.cpp:114-120).
 * out holds at most one record per query; *n_out = number of matches. */
int slam_amd_frame_match(slam_amd_matcher* m,
                         const uint8_t* desc1, const slam_amd_keypoint* kp1, int n1,
                         const uint8_t* desc2, const slam_amd_keypoint* kp2, int n2,
                         const int32_t* ids, int n_ids, const float* q_xy,
                         float max_dist, float area, int max_level,
                         slam_amd_dmatch* out, int* n_out);

/* Device, asynchronous.  Same semantics on device arrays.  d_n1 / d_n2 (may be
 * null) point to device int32 counts that override n1 / n2 when smaller, so the
 * call can be chained behind slam_amd_extract_batch_async without a host sync.
 * d_out: [max queries] cv::DMatch records, d_n_out: device int32 match count.
 * d_ids and d_q_xy as in the host call (device pointers, may be null). */
int slam_amd_frame_match_async(slam_amd_matcher* m,
                               const uint8_t* d_desc1, const slam_amd_keypoint* d_kp1, int n1,
                               const int32_t* d_n1,
                               const uint8_t* d_desc2, const slam_amd_keypoint* d_kp2, int n2,
                               const int32_t* d_n2,
                               const int32_t* d_ids, int n_ids, const float* d_q_xy,
                               float max_dist, float area, int max_level,
                               slam_amd_dmatch* d_out, int32_t* d_n_out, void* stream);

/* No counterpart in the reference.  For callers that cannot allocate device memory -- code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc: the same
 * call as slam_amd_frame_match_async (same arguments, same results), but its device scratch
 * -- the running best keys, and the (best, second) planes while the ratio option is set --
 * is a fixed array inside the library's code object (4.1 MiB per GPU, held until the process
 * ends) instead of buffers the matcher allocates.  It calls no hipMalloc / hipFree, and always
 * uses the full sweep (never the window paths over reordered sets, which need scratch for them).
 * At most SLAM_AMD_NOALLOC_MAX_QUERIES query rows (max(n1, n_ids)); more is
 * SLAM_AMD_ERR_CAPACITY.  All calls on one GPU share that GPU's array: calls that overlap in
 * time on one GPU (different streams, different matchers) race; serialise them. */
#define SLAM_AMD_NOALLOC_MAX_QUERIES 32768
int slam_amd_frame_match_noalloc_async(slam_amd_matcher* m,
                                       const uint8_t* d_desc1, const slam_amd_keypoint* d_kp1, int n1,
                                       const int32_t* d_n1,
                                       const uint8_t* d_desc2, const slam_amd_keypoint* d_kp2, int n2,
                                       const int32_t* d_n2,
                                       const int32_t* d_ids, int n_ids, const float* d_q_xy,
                                       float max_dist, float area, int max_level,
                                       slam_amd_dmatch* d_out, int32_t* d_n_out, void* stream);

/* Batched matching of `batch` independent (set 1, set 2) problems of equal
 * capacity in one launch: problem b uses rows [b*cap1, b*cap1+n1[b]) of set 1
 * and [b*cap2, b*cap2+n2[b]) of set 2 (counts on the device).  Output
 * d_out[b*cap1 + i], d_n_out[b].  Serves frame-vs-previous-frame matching of a
 * batch of frames. */
int slam_amd_frame_match_batch_async(slam_amd_matcher* m, int batch,
                                     const uint8_t* d_desc1, const slam_amd_keypoint* d_kp1,
                                     int cap1, const int32_t* d_n1,
                                     const uint8_t* d_desc2, const slam_amd_keypoint* d_kp2,
                                     int cap2, const int32_t* d_n2,
                                     float max_dist, float area, int max_level,
                                     slam_amd_dmatch* d_out, int32_t* d_n_out, void* stream);

/* The same for a run of consecutive frames without a copy of their records: frame b of the
 * batch (rows [b*cap, b*cap+n[b]) of d_desc / d_kp) is set 2 of problem b, and set 1 of problem
 * b is the frame before it -- frame b-1 of the same arrays, and for b = 0 the single frame at
 * d_prev_* (the last frame of the batch before; d_prev_n points to its count).  This is the
 * reference's frame-against-the-one-before matching (Matcher::frameMatch(initial, current),
 * src/tracking/mapper.cpp:34; last key frame against the current frame, src/tracking/tracker.cpp:74)
 * over a batch of consecutive frames.  With the window gate on, every frame is reordered by grid
 * cell once and serves as the train set of its own problem and the query set of the next.
 * Results as slam_amd_frame_match_batch_async: d_out[b*cap + i], d_n_out[b]. */
int slam_amd_frame_match_chain_async(slam_amd_matcher* m, int batch,
                                     const uint8_t* d_prev_desc, const slam_amd_keypoint* d_prev_kp,
                                     const int32_t* d_prev_n,
                                     const uint8_t* d_desc, const slam_amd_keypoint* d_kp,
                                     int cap, const int32_t* d_n,
                                     float max_dist, float area, int max_level,
                                     slam_amd_dmatch* d_out, int32_t* d_n_out, void* stream);

/* Keyframe-pair batch: n_keyframes descriptor sets of n_per rows each, resident
 * on the device ([n_keyframes][n_per][32]; d_kp [n_keyframes][n_per] may be null
 * when both gates are off).  d_pairs: n_pairs (query keyframe, train keyframe)
 * index pairs.  Output: one 32-bit word per query row of every pair,
 * (distance << 16) | trainIdx, or 0xFFFFFFFF when no train row passes.
 * n_per <= 65535.  This is frameMatch(kf_i, kf_j, {}, ...) for many pairs, the
 * all-defaults form of src/tracking/mapper.cpp:34.  With the window gate on and n_per <= 32768
 * the keyframes are first reordered by grid cell in the matcher's scratch (44.5 bytes per
 * row; allocated on the first such call, which therefore must not be under stream capture --
 * or reserved with slam_amd_matcher_reserve(m, n_keyframes, 0, n_per)). */
int slam_amd_match_pairs_async(slam_amd_matcher* m,
                               const uint8_t* d_desc, const slam_amd_keypoint* d_kp,
                               int n_keyframes, int n_per,
                               const int32_t* d_pairs, int n_pairs,
                               float max_dist, float area, int max_level,
                               uint32_t* d_out, void* stream);

/* ---- projection: replaces cv::Rodrigues + cv::projectPoints ----------------
 * reference: Matcher::_projectMapPoints, src/frame/matcher.cpp:163-184 */

/* Host, blocking.  Rt: row-major 3x4 pose [R|t] of the target keyframe;
 * K4: fx, fy, cx, cy; dist5: k1 k2 p1 p2 k3 or null; xyz: n world points;
 * uv: n projected pixels.  Arithmetic in double, results rounded to float. */
int slam_amd_project_points(slam_amd_matcher* m, const double* Rt, const double* K4,
                            const double* dist5, const float* xyz, int n, float* uv);

int slam_amd_project_points_async(slam_amd_matcher* m, const double* Rt, const double* K4,
                                  const double* dist5, const float* d_xyz, int n, float* d_uv,
                                  void* stream);

/* ---- duplicate map points: replaces the per-pair BFMatcher::match loop -----
 * reference: Mapper::_isDuplicate / _keyframeDuplicates,
 * src/tracking/mapper.cpp:292-346 */

/* Host, blocking.  flag[i*n2 + j] = 1 when hamming(desc1[i], desc2[j]) <=
 * desc_dist and the Euclidean distance of world points pos1[i], pos2[j] (3
 * floats each) is < point_dist, else 0. */
int slam_amd_duplicate_pairs(slam_amd_matcher* m,
                             const uint8_t* desc1, const float* pos1, int n1,
                             const uint8_t* desc2, const float* pos2, int n2,
                             int desc_dist, double point_dist, uint8_t* flag);

/* ---------------------------------------------------------------------------
 * Input front end: what main.cpp:124-129 does to every video frame before
 * tracker.track() -- cv::cvtColor(BGR2GRAY), then cv::resize by
 * scale = (float)target_width / (float)width (INTER_LINEAR, aspect kept) -- in one
 * kernel.  channels = 3: interleaved BGR bytes; channels = 1: already grey (resize only).
 * ------------------------------------------------------------------------- */
typedef struct slam_amd_frontend slam_amd_frontend;
int slam_amd_frontend_create(int src_width, int src_height, int channels, int target_width,
                             int device, slam_amd_frontend** out);
int slam_amd_frontend_destroy(slam_amd_frontend* fe);
/* Size of the grey image: round(src * scale) per axis, as cv::resize computes it. */
int slam_amd_frontend_output_size(const slam_amd_frontend* fe, int* width, int* height);
/* Device pointers, asynchronous on `stream`; batch frames src_frame_pitch /
 * dst_frame_pitch bytes apart.  The output can be handed to slam_amd_extract_batch_async. */
int slam_amd_frontend_run_async(slam_amd_frontend* fe, const uint8_t* d_src, int batch,
                                size_t src_frame_pitch, int src_row_stride,
                                uint8_t* d_gray, size_t dst_frame_pitch, int dst_row_stride, void* stream);
/* Host pointers, blocking, one frame. */
int slam_amd_frontend_run(slam_amd_frontend* fe, const uint8_t* src, int src_row_stride,
                          uint8_t* gray_out, int dst_row_stride);

/* Host, blocking.  isOutlier of src/map/mappoint.cpp:54-81 (with projectionError,
 * :83-101) for n triangulated points at once, as Mapper calls it per point at
 * src/tracking/mapper.cpp:48 and :121.  Point i belongs to matches[i] (queryIdx into
 * kp_q, trainIdx into kp_t; undistorted keypoints of the two keyframes).  flags[i] = 1
 * when the point is NaN, the cosine of its parallax between the two camera centres is
 * outside [0, 0.999f], it lies within 1e-6 of a centre, or it reprojects more than one
 * pixel from its keypoint in either keyframe.  center_*: 3 floats (-R^T t); Rt_*: pose
 * as row-major 3x4 doubles; K4_*: fx fy cx cy; dist5_*: k1 k2 p1 p2 k3 or NULL. */
int slam_amd_outlier_flags(slam_amd_matcher* m, const float* xyz, const slam_amd_dmatch* matches, int n,
                           const slam_amd_keypoint* kp_q, int nq, const slam_amd_keypoint* kp_t, int nt,
                           const float* center_q, const float* center_t,
                           const double* Rt_q, const double* K4_q, const double* dist5_q,
                           const double* Rt_t, const double* K4_t, const double* dist5_t,
                           uint8_t* flags);

/* ---- stream pipeline: frames in pinned host memory, copies overlapped with compute -------
 * The reference processes its video one frame at a time on the host (main.cpp:110-140:
 * read, cvtColor, resize, tracker.track).  This is that loop's extraction + frame-to-frame
 * matching with the three transfers overlapped: upload of frame i + 1, extract + match of frame
 * i against frame i - 1, download of frame i - 1's results, on three HIP streams over a ring of
 * `depth` (>= 3) slots.  Frames are numbered 0, 1, 2, ... and submitted in order:
 *   slam_amd_pipeline_input(p, i, &buf, &stride)  pinned host buffer to fill with frame i (blocks
 *                                                 until the slot's previous occupant, frame i - depth, is done)
 *   slam_amd_pipeline_submit(p, i)                enqueue everything for frame i, return at once
 *   slam_amd_pipeline_result(p, i, ...)           wait for frame i; pointers into pinned host
 *                                                 buffers, valid until _input(p, i + depth)
 * Matches are frameMatch(frame i - 1, frame i, {}, max_dist, area, max_level): queries are the
 * previous frame's keypoints (none for frame 0).
 * create_batched: a slot collects `batch` (1..1024) frames and queues them as one upload, one extraction
 * launch sequence, one chained matching call and one download when the last of them is submitted -- a
 * single large frame leaves the chip idle between its short launches, a few per launch fill it; the price
 * is the batch's latency.  Frames are still numbered, filled, submitted and read one by one;
 * slam_amd_pipeline_flush queues an incomplete batch (end of the stream, or a producer that stalls), and
 * _result of a frame whose batch is neither complete nor flushed is an error, not a wait.  Results live
 * until depth - 1 more batches are queued.  create = create_batched with batch 1.
 * A submit or flush that fails after part of its work was queued drains the three streams and leaves
 * the pipeline unusable except for _result of earlier frames and _destroy. */
typedef struct slam_amd_pipeline slam_amd_pipeline;
int slam_amd_pipeline_create(const slam_amd_orb_params* params, int width, int height, int capacity,
                             int depth, int device, float max_dist, float area, int max_level,
                             slam_amd_pipeline** out);
int slam_amd_pipeline_create_batched(const slam_amd_orb_params* params, int width, int height, int capacity,
                                     int depth, int batch, int device, float max_dist, float area, int max_level,
                                     slam_amd_pipeline** out);
int slam_amd_pipeline_flush(slam_amd_pipeline* p);
int slam_amd_pipeline_destroy(slam_amd_pipeline* p);
int slam_amd_pipeline_input(slam_amd_pipeline* p, long long frame, uint8_t** host_image, int* row_stride);
int slam_amd_pipeline_submit(slam_amd_pipeline* p, long long frame);
int slam_amd_pipeline_result(slam_amd_pipeline* p, long long frame, const slam_amd_keypoint** kp,
                             const uint8_t** desc, int* n, const slam_amd_dmatch** matches, int* n_matches);

/* ---- per-stage timing (process-wide, off by default) -------------------------
 * With max_marks > 0 every kernel stage launched through this library is
 * bracketed by HIP events on its own stream (up to max_marks stages recorded);
 * 0 switches it off.  collect() waits for the recorded events, adds the elapsed
 * milliseconds and call counts per stage into arrays of n_stages >=
 * SLAM_AMD_PROF_STAGES entries and clears the record.  Not thread-safe. */
#define SLAM_AMD_PROF_STAGES 10
int slam_amd_prof_enable(int max_marks);
int slam_amd_prof_collect(double* total_ms, int* calls, int n_stages);
const char* slam_amd_prof_stage_name(int stage);

/* ---- host-side helpers exported for tests --------------------------------- */

/* cos and sin of a float angle (radians, 0 <= a <= 7) computed in double by the
 * routine the descriptor kernel uses, rounded to float. */
void slam_amd_sincos_f32(float angle_rad, float* cos_out, float* sin_out);
/* The degree-7 polynomial atan2 (degrees) the orientation kernel uses. */
float slam_amd_fast_atan2(float y, float x);
/* INTER_LINEAR_EXACT tables of one axis as the pyramid kernel receives them. */
int slam_amd_linear_exact_coeffs(int src_size, int dst_size, int32_t* offsets, uint16_t* c1);
/* Launch plan of the pyramid kernel for an image size (pure host computation, no GPU needed).
 * *n_groups = number of launches; for `group`: info6 = {first level, last level, tile size of the
 * last level, tiles in x, tiles in y, LDS bytes}; spanx / spany = for every tile and slot (slot 0 =
 * the source level first - 1, slot k = level first + k - 1) the range computed [c0, c1) and the range
 * owned (stored) [o0, o1) as 4 int16 each, [tiles][last - first + 2][4].  Any pointer may be NULL. */
int slam_amd_debug_pyramid_plan(const slam_amd_orb_params* params, int width, int height, int group,
                                int* n_groups, int* info6, int16_t* spanx, int16_t* spany, int span_capacity);
/* Host-side table of cv::resize(INTER_LINEAR) for 8-bit images, one axis: source
 * offset and the two 11-bit taps per destination index (clamp_frac: horizontal rule). */
int slam_amd_linear_taps(int src_size, int dst_size, double inv_scale, int clamp_frac,
                         int32_t* offsets, int16_t* taps);

#ifdef __cplusplus
}
#endif
#endif
