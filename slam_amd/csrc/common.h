// Shared declarations of the gfx950 ORB frontend library (host + device).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/slam_amd.h"

namespace sa {

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define SA_HIP(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return sa::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

#define SA_FAIL(code, msg)      \
    do {                        \
        sa::set_error(msg);     \
        return (code);          \
    } while (0)

constexpr int kMaxLevels = SLAM_AMD_MAX_LEVELS;

// Geometry of one pyramid level, shared by every kernel.  Level 0 is read from
// the caller's image; levels >= 1 live in the context's pyramid buffer.
struct LevelGeom {
    int w, h;           // level size in pixels
    int pitch;          // bytes per row in the pyramid / survivor-map buffers
    int quota;          // per-level keypoint quota (retain-best cut)
    unsigned off;       // byte offset of the level inside one frame's buffer
    int row_base;       // index of the level's first row among all rows of a frame
    int tile_base;      // index of the level's first FAST tile among all tiles of a frame
    int tiles_x;        // FAST tiles per row of tiles
    float scale;        // level scale (float(pow(scale_factor, level)))
    float inv_scale;    // 1.0f / scale
    float size;         // keypoint size at this level: patch_size * scale
};

// The blurred levels are stored in tiles of 16 x 8 pixels (128 bytes, rows of a tile one after the other;
// a level's rows are padded to a multiple of 8 and its pitch is a multiple of 64): the descriptor kernel
// fetches square patches, and a 64-byte piece of the address space then holds 16 x 4 pixels instead of 64 x 1.
constexpr int kBlurTileW = 16, kBlurTileH = 8;
__host__ __device__ inline unsigned blur_tiled_offset(unsigned pitch, unsigned x, unsigned y) {
    return (((y >> 3) * (pitch >> 4) + (x >> 4)) << 7) + ((y & 7u) << 4) + (x & 15u);
}

// Tile rows [first, last) of band k of a level with nty tile rows cut into `bands` bands: the rows ty with
// ty * bands / nty == k.
__host__ __device__ inline int band_first_tile_row(int k, int nty, int bands) { return (k * nty + bands - 1) / bands; }

struct PyrGeom {
    int nlevels;
    int edge;           // edge threshold
    int fast_threshold;
    int total_rows;     // sum of level heights
    int total_tiles;    // FAST tiles per frame
    unsigned frame_bytes;  // bytes of one frame's pyramid (and survivor map)
    int bands;          // horizontal bands a level is cut into for the selection (score histograms per band): 1, or up to 16 for large or single frames
    LevelGeom lv[kMaxLevels];
};

// Result of the per-frame selection step.
struct FrameSelect {
    int cut[kMaxLevels];    // keep survivors with score >= cut (256: keep none)
    int count[kMaxLevels];  // survivors kept per level
    int offset[kMaxLevels]; // index of the level's first keypoint in the frame's output
    int total;
    int pad[3];
};

// Optional per-stage timing with HIP events on the stream the kernels run on.
enum Stage {
    kStageResize = 0, kStageFast, kStageSelect, kStageSpatialSort, kStageMatchWindow, kStageDescribe,
    kStageMatch, kStageFinalize, kStageMatchPairs, kStageMemset, kStageCount
};
void prof_mark(int stage, bool begin, hipStream_t st);
struct ProfScope {
    int stage; hipStream_t st;
    ProfScope(int s, hipStream_t t) : stage(s), st(t) { prof_mark(stage, true, st); }
    ~ProfScope() { prof_mark(stage, false, st); }
};

constexpr int kFastTileW = 64;
constexpr int kFastTileH = 32;
// Survivor list of one FAST tile.  A strict 3x3 maximum excludes its 8 neighbours, so a tile
// holds at most one survivor per 2 x 2 block: the list cannot overflow.
constexpr int kTileListCap = (kFastTileW / 2) * (kFastTileH / 2);

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence over
// every address space: it also waits until the wave's outstanding global stores and atomics have
// completed (s_waitcnt vmcnt(0)), which costs microseconds per barrier in a kernel that streams
// results to memory between LDS phases.  Where only LDS data is handed between the waves, waiting
// for the LDS counter is enough.
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

}  // namespace sa
