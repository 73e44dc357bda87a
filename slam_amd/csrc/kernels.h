// Argument blocks of the kernels and their host-side launchers.
#pragma once

#include "common.h"

namespace sa {

// Range of one axis that a pyramid tile computes [c0, c1) and owns (stores) [o0, o1) at one level.
struct PyrSpan { short c0, c1, o0, o1; };

struct PyramidArgs {
    const uint8_t* src0;       // caller's level-0 frames
    size_t src0_frame_pitch;
    int src0_pitch;
    int src0_aligned;          // 1: base and pitches are multiples of 4
    uint8_t* pyr;              // levels >= 1
    // the source level of this launch, level first - 1 (set by the launcher from the fields above)
    const uint8_t* src;
    size_t src_frame_pitch;
    int src_pitch, src_aligned, src_w, src_h;
    PyrGeom g;
    int first, last;           // levels this launch produces (first >= 1); the source is level first - 1
    int ntx, nty;              // tiles of level `last`
    const PyrSpan* spanx;      // [ntx][last - first + 2], slot 0 = the source level
    const PyrSpan* spany;      // [nty][last - first + 2]
    const uint32_t* xtab[kMaxLevels];  // per level: source index | second-tap weight << 16, padded
    const uint32_t* ytab[kMaxLevels];
    int tab_bytes;             // LDS bytes reserved for a tile's coefficient slices (multiple of 16)
    int x_entries;             // entries of all x slices together (the y slices follow them)
    int xat[kMaxLevels + 1];   // first entry of slot k's x slice (4-byte entries) and
    int yat[kMaxLevels + 1];   // of its y slice (2-byte entries)
    int buf_even, buf_odd;     // bytes of the LDS buffers of the even / odd slots
};

struct FastArgs {
    const uint8_t* img0;  // caller's level-0 frames
    size_t img0_frame_pitch;
    int img0_pitch;
    int img0_aligned;
    const uint8_t* pyr;   // levels >= 1 (level 0 slot unused)
    uint8_t* blur;        // blurred levels, same geometry as pyr (level 0 included)
    unsigned* hist;       // [batch][nlevels][bands][256]
    unsigned* tilecount;  // [batch][total_tiles] NMS survivors of every FAST tile
    uint32_t* tilelist;   // [batch][total_tiles][kTileListCap] lx | ly << 6 | score << 16, any order
    int grid_shift;       // 0: off; 3, 4, 5: keep the best survivor of every 8, 16, 32 pixel cell
    const uint2* tiledesc;     // [2][total_tiles] {level | tile column << 4 | tile row << 16 | band << 28, tile number} of every FAST tile, in two orders of a frame (device memory)
    int tiles_eighth;     // ceil(total_tiles / 8) (set by the launcher)
    int frames_by_xcd;    // 1: grid = (8 x tile, frame / 8), for a multiple of 8 frames (set by the launcher)
    PyrGeom g;
};

struct SelectCompactArgs {
    const unsigned* hist;
    const unsigned* tilecount;
    const uint32_t* tilelist;
    int keycap;                // kept survivors of a level ranked from LDS (<= 4096; lower in tests)
    int quadtree;              // 1: levels over their quota keep exactly the quota, chosen by the quadtree kernel
    FrameSelect* sel;
    int32_t* d_count;          // may be null
    slam_amd_keypoint* kp;     // [batch][capacity]
    uint32_t* kpxy;            // [batch][capacity]  x | y << 14 | level << 28
    int capacity;
    PyrGeom g;
};

struct DescribeArgs {
    const uint8_t* img0;
    size_t img0_frame_pitch;
    int img0_pitch;
    int img0_aligned;
    const uint8_t* pyr;
    const uint8_t* blur;
    const FrameSelect* sel;
    slam_amd_keypoint* kp;
    const uint32_t* kpxy;
    const int32_t* lanegeom;   // [kMaxLevels][8]: width, height, pitch, offset, scale bits, 0, 0, 0 of every level (device memory)
    uint8_t* desc;       // [batch][capacity][32]
    unsigned* hist_reset; // if set: the score histograms [batch][nlevels][bands][256]; the selection has read them, this kernel zeroes them for the next call
    int capacity;
    int batch;
    PyrGeom g;
};

struct MatchArgs {
    const uint8_t* qdesc;            // set 1 descriptors
    const slam_amd_keypoint* qkp;    // set 1 keypoints (may be null when both gates are off)
    const int32_t* d_n1;             // optional device count of set 1 (per problem)
    int n1;                          // rows of set 1 (upper bound when d_n1 is given)
    const uint8_t* tdesc;
    const slam_amd_keypoint* tkp;
    const int32_t* d_n2;
    int n2;
    const int32_t* ids;              // optional gather of set-1 rows
    int n_ids;
    const float* q_xy;               // optional query positions (2 floats per query)
    int cap1, cap2;                  // rows between consecutive problems of a batch
    float area;
    int max_level;
    int dmax;                        // largest accepted distance (floor of max_dist, <= 256)
    unsigned* best;                  // [batch][cap1] running keys
    int chunk;                       // train rows per split (multiple of kTrainTile)
    // pair mode
    const int32_t* pairs;            // [n_pairs][2] keyframe indices, or null
    int n_per;                       // rows per keyframe
    int qblocks;                     // query blocks per problem
    int splits, batch;               // train splits per problem, problems per launch
    uint32_t* pair_out;              // [n_pairs][n_per]
    // second-best mode (ratio test): every split stores (best key, second key)
    uint2* part;                     // [splits][part_pitch], or null
    size_t part_pitch;               // rows per split plane (batch * cap1)
    int chain;                       // 1: every frame of set 2 against the frame before it (see query_side)
};

// Query side of problem `prob`.  Plain batches: rows prob * cap1 ... of the set-1 arrays.  Chained
// batches (every frame against the one before it): problem 0 takes the set-1 arrays, which hold one
// frame, the one before the batch; problem p >= 1 takes frame p - 1 of the set-2 arrays.  Results are
// indexed by prob * cap1 either way.
struct QuerySide {
    const uint8_t* desc;
    const slam_amd_keypoint* kp;
    size_t base;              // first row of the problem's queries in desc / kp
    const int32_t* count;     // device count of the problem's queries, or null
};
__device__ __forceinline__ QuerySide query_side(int chain, int prob, const uint8_t* qdesc, const slam_amd_keypoint* qkp,
                                                const int32_t* d_n1, int cap1, const uint8_t* tdesc,
                                                const slam_amd_keypoint* tkp, const int32_t* d_n2, int cap2) {
    QuerySide q;
    if (!chain) {
        q.desc = qdesc; q.kp = qkp; q.base = (size_t)prob * cap1; q.count = d_n1 ? d_n1 + prob : nullptr;
    } else if (prob == 0) {
        q.desc = qdesc; q.kp = qkp; q.base = 0; q.count = d_n1;
    } else {
        q.desc = tdesc; q.kp = tkp; q.base = (size_t)(prob - 1) * cap2; q.count = d_n2 ? d_n2 + (prob - 1) : nullptr;
    }
    return q;
}

constexpr int kMaxSecondSplits = 16;

struct MatchPlan {
    int q;        // queries per lane (1 or 2)
    int qblocks;  // query blocks per problem
    int splits;   // train splits per problem
    int chunk;    // train rows per split
};

// Reordering of both sets of a windowed match by grid cell (set 0 = queries, 1 = train).
struct SpatialSortArgs {
    const uint8_t* qdesc; const slam_amd_keypoint* qkp; const int32_t* d_n1; int n1; int cap1;
    const uint8_t* tdesc; const slam_amd_keypoint* tkp; const int32_t* d_n2; int n2; int cap2;
    const int32_t* ids; int n_ids;
    const float* q_xy;
    float area;               // > 0 and finite
    int max_level;
    uint4* sdesc[2];          // [batch][scap][2]  descriptors in cell order
    float2* sxy[2];           // [batch][scap]     positions
    uint32_t* sidx[2];        // [batch][scap]     original index | (octave > max_level) << 31
    float4* tile_box[2];      // [batch][scap/box_rows]  xmin, xmax, ymin, ymax of every tile of box_rows rows (set 1 only)
    int scap[2];              // rows per problem in the sorted arrays, multiples of 64
    int qchunks;              // 4096-row chunks of the query set (set by the launcher)
    int batch_q;              // problems of the query set (the launcher's batch unless set: chained batches reorder one frame)
    int box_rows;             // rows per box of set 1: 64 (popcount kernel) or 32 (matrix-core kernel)
    int copy_desc;            // 1: the descriptors are copied into sdesc (popcount kernel); 0: the matcher fetches them by index
};

struct WindowMatchArgs {
    const uint4* sdesc[2];
    const float2* sxy[2];
    const uint32_t* sidx[2];
    const float4* tile_box;   // of the train set
    int scap[2];
    const int32_t* d_n1; int n1;
    const int32_t* d_n2; int n2;
    int has_ids, n_ids;
    int cap1;                 // rows between consecutive problems in best / part
    float area;
    int max_level;
    int batch;
    int splits;               // train tiles are dealt round-robin to this many waves per query tile
    unsigned* best;           // [batch][cap1], merged with atomicMin
    uint2* part;              // [splits][part_pitch] (best, second) when the ratio test is on, else null
    size_t part_pitch;
    // matrix-core kernel only
    int chain;                // 1 = the queries of problem p are set 1 of problem p - 1 (set 0 for p = 0)
    const uint8_t* qdesc;     // the descriptors where the caller holds them, fetched through the index words (sdesc is not read)
    const uint8_t* tdesc;
    const int32_t* ids;       // the caller's gather of set-1 rows, or null
    int cap2;                 // rows between consecutive problems in tdesc (cap1: in qdesc)
    // pair mode (matrix-core kernel): problem p matches keyframe pairs[2p] (queries) against keyframe pairs[2p + 1],
    // both in set 1 (n2 rows each, no counts); every query's result goes to pair_out, no splits
    const int32_t* pairs;
    uint32_t* pair_out;       // [n_pairs][n2]  distance << 16 | train index, or 0xffffffff
    int dmax;
};

struct FinalizeArgs {
    const unsigned* best;
    const int32_t* d_n1;
    int n1;
    const int32_t* ids;
    int n_ids;
    int cap1;
    int dmax;
    slam_amd_dmatch* out;     // [batch][cap1]
    int32_t* n_out;           // [batch]
    // options beyond the reference
    const uint2* part;        // per-split (best, second) keys, or null: read `best`
    size_t part_pitch;
    int splits;
    float ratio;              // > 0: keep a match only if best < ratio * second best
    int histo_length;         // > 0: rotation-consistency filter with that many bins (<= 64)
    const slam_amd_keypoint* qkp;
    const slam_amd_keypoint* tkp;
    int cap2;
    int chain;                // as in MatchArgs
    const int32_t* d_n2;      // chained batches: the counts of set 2 are the query counts of the next problem
    unsigned* reset;          // if set (= best): every key read is put back to "no match", so the next call needs no fill
    unsigned* ticket;         // [batch] zeros, or null: with it a problem of many queries is spread over workgroups
    int nq_bound;             // host bound on a problem's queries (decides how many workgroups are launched)
    int multi;                // set by the launcher: grid = (workgroups of a problem, problems) instead of (problems)
};
constexpr int kFinalTicketProblems = 16384;   // problems per call the ticket array serves

struct ProjectArgs {
    double Rt[12];
    double K4[4];
    double dist[5];
    const float* xyz;
    float* uv;
    int n;
};

struct Camera {           // pose (row-major 3x4), fx fy cx cy, k1 k2 p1 p2 k3 of one keyframe
    double Rt[12], K4[4], dist[5];
};

struct OutlierArgs {
    const float* xyz;                  // [n][3] triangulated points
    const slam_amd_dmatch* matches;    // [n]
    int n;
    const slam_amd_keypoint* kp_q; int nq;
    const slam_amd_keypoint* kp_t; int nt;
    float cq[3], ct[3];                // camera centres of the two keyframes
    Camera cam_q, cam_t;
    uint8_t* flags;                    // [n] 1 = outlier
};

struct FrontendArgs {
    const uint8_t* src;        // [batch] frames, `channels` interleaved bytes per pixel (3 = BGR, 1 = grey)
    size_t src_frame_pitch;
    int src_pitch, sw, sh, channels;
    uint8_t* dst;              // [batch] grey frames
    size_t dst_frame_pitch;
    int dst_pitch, dw, dh;
    const int32_t* xofs;       // [dw] source column, clamped to [0, sw - 1]
    const int16_t* xtap;       // [dw][2] 11-bit taps
    const int32_t* yofs;       // [dh] source row, may be -1 .. sh - 1 (clamped in the kernel)
    const int16_t* ytap;       // [dh][2]
};

struct DupArgs {
    const uint8_t* d1; const float* p1; int n1;
    const uint8_t* d2; const float* p2; int n2;
    int desc_dist; double point_dist;
    uint8_t* flag;
};

int launch_pyramid(const PyramidArgs& a, int batch, size_t lds_bytes, hipStream_t st);
int launch_fast(const FastArgs& a, int batch, hipStream_t st);
int launch_select_compact(const SelectCompactArgs& a, int batch, hipStream_t st);
int launch_describe(const DescribeArgs& a, int batch, hipStream_t st);
MatchPlan plan_match(int batch, int nq_bound, int nt_bound, int n_cu, bool second);
int launch_match(MatchArgs a, const MatchPlan& p, int batch, hipStream_t st);
int launch_match_pairs(MatchArgs a, int n_pairs, hipStream_t st);
MatchPlan plan_match_mfma(int batch, int nq_bound, int nt_bound, int n_cu, bool second);
int launch_match_mfma(MatchArgs a, const MatchPlan& p, int batch, hipStream_t st);
int launch_match_pairs_mfma(MatchArgs a, int n_pairs, hipStream_t st);
int launch_spatial_sort(SpatialSortArgs a, int batch, int nq_bound, int nt_bound, hipStream_t st);
int window_splits(int batch, int nq_bound, int nt_bound, int n_cu);
int launch_match_window(const WindowMatchArgs& a, int batch, int nq_bound, hipStream_t st);
int window_mfma_max_train();      // largest train set of the kernel below
int window_mfma_splits(int batch, int nq_bound, int nt_bound, int n_cu);
int launch_match_window_mfma(const WindowMatchArgs& a, int batch, int nq_bound, hipStream_t st);   // boxes per 32 train rows
int launch_finalize(const FinalizeArgs& a, int batch, hipStream_t st);
int launch_project(const ProjectArgs& a, hipStream_t st);
int launch_frontend(const FrontendArgs& a, int batch, hipStream_t st);
int launch_outliers(const OutlierArgs& a, hipStream_t st);
int launch_duplicates(const DupArgs& a, hipStream_t st);

}  // namespace sa
