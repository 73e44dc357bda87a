// HIP kernels of the ORB extraction path for gfx950 (wave64, LDS-staged tiles).
//
// Per batch of frames, in stream order:
//   pyramid       (pyramid_kernels.hip) chained bit-exact bilinear pyramid
//   fast_nms                      FAST-9/16 score + 3x3 NMS + border filter + per-level score
//                                 histogram + per-row survivor lists + 7x7 blur, all levels, one launch
//   select_compact                per-level cut score from the histogram (retain-best), then
//                                 the kept survivors of every row's sparse list -> keypoints
//                                 in canonical (octave, y, x) order
//   describe                      intensity-centroid angle + rBRIEF on the blurred level,
//                                 one wavefront on one keypoint at a time, patch held in LDS
#include "common.h"
#include "detmath.h"
#include "kernels.h"

// build-time experiment switch (1: describe without its global patch loads); 0 in every shipped build
#ifndef SA_DESC_EXP
#define SA_DESC_EXP 0
#endif
#ifndef SA_FAST_FRAMES_BY_XCD
#define SA_FAST_FRAMES_BY_XCD 1
#endif
#ifndef SA_FAST_EXP
#define SA_FAST_EXP 0     // 1: fast_nms prints the clock at its phase boundaries for a few workgroups (timing experiments)
#endif
#ifndef SA_FAST_SKIP
#define SA_FAST_SKIP 0    // timing experiments only (results are wrong): 1 no blur store, 2 no blur at all, 4 no ring test / score /
#endif                    // suppression (list 1 taken as empty), 8 no compass test either, 16 tile not loaded from memory,
                          // 32 the blur's matrix products replaced by vector additions
#if SA_FAST_EXP
#define SA_STAMP_DECL unsigned long long stamp_[16] = {}
#define SA_STAMP(i) stamp_[i] = __builtin_amdgcn_s_memtime()
#define SA_STAMP_PRINT                                                                                              \
    if ((threadIdx.x & 63) == 0 && b == 3 && (ftile == 75 || ftile == 200 || ftile == 300 || ftile == 420))        \
        printf("fast tile %d lvl %d wave %d n1 %d n2 %d : %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", ftile, l, \
               (int)(threadIdx.x >> 6), n1, n2, stamp_[1] - stamp_[0], stamp_[2] - stamp_[1],                      \
               stamp_[3] - stamp_[2], stamp_[4] - stamp_[3], stamp_[5] - stamp_[4], stamp_[6] - stamp_[5],         \
               stamp_[7] - stamp_[6], stamp_[8] - stamp_[7], stamp_[9] - stamp_[8], stamp_[10] - stamp_[9], stamp_[15] - stamp_[10])
#else
#define SA_STAMP_DECL
#define SA_STAMP(i)
#define SA_STAMP_PRINT
#endif
#ifndef SA_SEL_EXP
#define SA_SEL_EXP 0      // build-time experiment switch; 0 in every shipped build
#endif
#ifndef SA_SEL_STAMPS
#define SA_SEL_STAMPS 0   // 1: select_compact prints the clock at its step boundaries for two frames (timing experiments)
#endif
#ifndef SA_DESC_STAMPS
#define SA_DESC_STAMPS 0  // 1: describe prints the clock at its pass boundaries for a few workgroups (timing experiments)
#endif
#if SA_DESC_STAMPS
#define SA_DSTAMP_DECL unsigned long long dstamp_[8] = {}
#define SA_DSTAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); dstamp_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define SA_DSTAMP_PRINT                                                                                   \
    if (threadIdx.x == 0 && (blockIdx.x == 1000 || blockIdx.x == 5000 || blockIdx.x == 9000 || blockIdx.x == 14000)) \
        printf("describe wg %d nk %d : %llu %llu %llu %llu %llu %llu\n", (int)blockIdx.x, nk, dstamp_[1] - dstamp_[0], \
               dstamp_[2] - dstamp_[1], dstamp_[3] - dstamp_[2], dstamp_[4] - dstamp_[3], dstamp_[5] - dstamp_[4], dstamp_[6] - dstamp_[5])
#else
#define SA_DSTAMP_DECL
#define SA_DSTAMP(i)
#define SA_DSTAMP_PRINT
#endif

namespace sa {

// ---------------------------------------------------------------------------
// FAST-9/16 + NMS + blur.  A workgroup owns a 64 x 32 tile of one level of one
// frame.  The image tile with a 4-pixel halo is staged in LDS twice (reflect-101
// outside the image): as bytes, for the gathers of the sparse phases and the blur; and with every byte
// shifted right by two (six bits per pixel), for the compass test, which handles four
// pixels per 32-bit add / sub / and / or (32-bit add / sub / and / or issue at full rate on this chip; packed 16-bit and
// three-operand instructions take twice as long).  Then
//   phase 1  every position of the tile plus a 1-pixel halo: compass test (a 9-arc
//            always holds one pixel of every opposite pair) on the 6-bit pixels with
//            the threshold rounded so that no pixel the exact test passes is lost;
//            one LDS list entry per passing (pixel, polarity)
//   phase 2  full 16-pixel ring on the exact bytes for the listed positions, one
//            polarity each (brighter arcs are darker arcs of the complemented
//            pixels): the corner score, which also says whether the position is a
//            corner; corners are compacted into a list per wave
//   phase 3  the 7x7 blur of the tile as two banded matrix products on the matrix cores
//            (see blur_tile) -> blurred level
//   phase 4  3x3 strict-maximum suppression + border filter over the corner lists
//            -> survivor list and 256-bin score histogram
// The lists keep the expensive phases dense instead of divergent.
// ---------------------------------------------------------------------------
constexpr int kImgHalo = 4;
constexpr int kImgDw = (kFastTileW + 2 * kImgHalo) / 4;   // 18 staged dwords per row
constexpr int kImgPitch = 80;                             // bytes per staged row: 72 of pixels, padded to a multiple of 16
                                                          // (the blur reads rows in 16-byte pieces)
constexpr int kImgRows = kFastTileH + 2 * kImgHalo;   // 40
constexpr int kScW = kFastTileW + 2;                  // 66 scored columns
constexpr int kScRows = kFastTileH + 2;               // 34 scored rows
constexpr int kScPitch = kScW;                        // score map: scored row r, column c at r * 66 + c (staged row r + 3, column c + 3)
constexpr int kScN = kScW * kScRows;                  // 2244 positions
constexpr int kScBytes = ((kScRows * kScPitch + 15) / 16) * 16;
constexpr int kListBytes = ((kScN * 2 + 15) / 16) * 16;
constexpr int kList1Bytes = 4608;                     // list 1: 2304 entries
constexpr int kQPitch = kImgDw + 1;                   // 6-bit tile: a spare dword before the 18 of a row (read, never used, by the
                                                      // first column group's left neighbour; the last one's right neighbour is the next row's)

__device__ __forceinline__ int div_up_dev(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// reflect101 without a loop for n >= 4 and the indices a tile needs exactly (-4 <= i <= n + 3);
// indices further out (staged but never used) are clamped into the image
__device__ __forceinline__ int reflect101_near(int i, int n) {
    int r = i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
    return min(max(r, 0), n - 1);
}

__device__ __forceinline__ bool has_run9(unsigned m16) {
    unsigned m = m16 | (m16 << 16);
    unsigned r = m & (m >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= m >> 8;
    return (r & 0xffffu) != 0;
}

// x >> s, or x << -s for a negative (compile-time) s
__device__ __forceinline__ unsigned shift_right(unsigned x, int s) { return s >= 0 ? x >> s : x << -s; }

// Inclusive prefix sum over the 64 lanes (DPP row shifts, then the row totals).
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// Append the flagged lanes' values to an LDS list (order within the list is free).
__device__ __forceinline__ void list_append(bool flag, unsigned value, uint16_t* list, unsigned* counter) {
    const unsigned long long m = __ballot(flag);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == __builtin_ctzll(m)) base = atomicAdd(counter, (unsigned)__builtin_popcountll(m));
    base = __shfl(base, __builtin_ctzll(m), 64);
    if (flag) list[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint16_t)value;
}

// Minimum / maximum of the low 16 bits as signed numbers (the 16-bit two-operand forms issue at full rate; the 32-bit
// and three-operand ones at half).  The upper half of the result is not used.
__device__ __forceinline__ short min16(short a, short b) { return a < b ? a : b; }
__device__ __forceinline__ short max16(short a, short b) { return a > b ? a : b; }

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(unsigned v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ unsigned as_u32(us2 v) { return __builtin_bit_cast(unsigned, v); }

// a.lo * b.lo + a.hi * b.hi + c on 16-bit halves, saturating at 2^32 - 1
__device__ __forceinline__ unsigned dot2(unsigned a, unsigned b, unsigned c) {
    return __builtin_amdgcn_udot2(as_us2(a), as_us2(b), c, true);
}

// The ring of radius 3 as byte offsets in the staged tile, in circular order.
#define SA_RING_OFFSETS(P) {3 * P, 3 * P + 1, 2 * P + 2, P + 3, 3, -P + 3, -2 * P + 2, -3 * P + 1, \
                            -3 * P, -3 * P - 1, -2 * P - 2, -P - 3, -3, P - 3, 2 * P - 2, 3 * P - 1}

// Row of a byte offset in the staged tile: code / 80 for code < 40 * 80.
__device__ __forceinline__ unsigned staged_row(unsigned code) { return (code * 3277u) >> 18; }
static_assert(kImgPitch == 80, "staged_row divides by the pitch of the staged tile");

// ---------------------------------------------------------------------------
// 7x7 blur of a 64 x 32 tile on the matrix cores.  The blur is separable with 8-bit taps
// 18 34 49 55 49 34 18 (8.8 fixed point; each is rounded on its own and they add up to 257, not 256), no rounding
// between the passes and one at the end: out = min((sum + 2^15) >> 16, 255).  Each pass is a product with a banded
// matrix of taps, exact in 32-bit integers, on v_mfma_i32_16x16x64_i8 (D = A B + C; lane l holds row l & 15 of A
// and column l & 15 of B, 16 of the 64 k each: k group l >> 4, and D[4 (l >> 4) + i][l & 15] in register i).
// Wave w takes the columns 16 w .. 16 w + 15 of the tile:
//   horizontal  H[r][c] = sum_t tap[t] p[r][c - 3 + t] for the 38 rows the vertical pass needs, as three products
//               of 16 rows: A = 16 bytes of a staged row per lane (staged as pixel - 128: the product is signed), B[k][n] =
//               tap[k - n - 1], C = 128 * 257, so that D = H, 16 bits.  A lane ends up with H of ONE column, rows
//               16 m + 4 (l >> 4) + i in register i of product m.
//   vertical    out[r][c] = sum_t tap[t] H[r + t][c] sums over the row index of H, which lies in the registers of a
//               lane: the lane's H values are the A operand as they stand (k slot (l >> 4, j) = row 16 (j >> 2) +
//               4 (l >> 4) + (j & 3), byte j & 3 of dword j >> 2: the four results of product j >> 2, packed), once
//               their low bytes and once their high bytes (as signed bytes again), B[k][n] = tap[row of k - n] --
//               the result has the image row on the lane and four neighbouring columns in the registers, which is
//               how the blurred level is stored (a dword per lane).  sum = low + 256 high + the constants.
// The k slots of the two operands only have to agree with each other, and both are built here.
// ---------------------------------------------------------------------------
struct BlurTaps { unsigned h[64][4], v[64][4]; };
constexpr BlurTaps make_blur_taps() {
    constexpr int tap[7] = {18, 34, 49, 55, 49, 34, 18};
    BlurTaps t = {};
    for (int l = 0; l < 64; l++) {
        const int n = l & 15, kg = l >> 4;
        for (int j = 0; j < 16; j++) {
            const int th = 16 * kg + j - n - 1;                          // horizontal: k = 16 kg + j is staged column 16 w + k
            const int tv = 16 * (j >> 2) + 4 * kg + (j & 3) - n;         // vertical, rows 0 .. 15 of the tile: row of the k slot - n
            if (th >= 0 && th < 7) t.h[l][j >> 2] |= (unsigned)tap[th] << (8 * (j & 3));
            if (tv >= 0 && tv < 7) t.v[l][j >> 2] |= (unsigned)tap[tv] << (8 * (j & 3));
        }
    }
    return t;
}
__device__ const BlurTaps c_blur_taps = make_blur_taps();
typedef int blur_v4i __attribute__((ext_vector_type(4)));

// th, tv: the lane's taps of the two passes, c_blur_taps.h[lane][0..3] and c_blur_taps.v[lane][0..1] (dwords 2 and 3 of the
// latter are zero) -- loaded by the caller at the start of the kernel, where the tile's own loads hide the latency
__device__ __forceinline__ void blur_tile(const uint8_t* s_img, int wave, int lane, const uint4& th, const uint2& tv,
                                          uint8_t* blur_level, unsigned pitch, int x0, int y0, int h) {
    const int n = lane & 15, kg = lane >> 4;
    const blur_v4i bh = {(int)th.x, (int)th.y, (int)th.z, (int)th.w};
    // ---- horizontal: H row hr = 16 m + n is staged row hr + 1 (rows past the staged tile are never used: clamped) ----
    unsigned lo[3], hi[3];
    const blur_v4i c_h = {128 * 257, 128 * 257, 128 * 257, 128 * 257};
#pragma unroll
    for (int m = 0; m < 3; m++) {
        const int row = min(16 * m + n + 1, kImgRows - 1);
        uint4 p = make_uint4(0u, 0u, 0u, 0u);
        if (kg < 2) p = *reinterpret_cast<const uint4*>(s_img + row * kImgPitch + 16 * wave + 16 * kg);   // (taps of k >= 23 are zero)
        const blur_v4i av = {(int)p.x, (int)p.y, (int)p.z, (int)p.w};       // signed bytes: pixel - 128
        const blur_v4i d = (SA_FAST_SKIP & 32) ? av + bh + c_h : __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bh, c_h, 0, 0, 0);
        // bytes 0 and 1 of the four results -> one dword each, as signed bytes
        const unsigned p01 = __builtin_amdgcn_perm((unsigned)d.y, (unsigned)d.x, 0x05010400u);   // lo0 lo1 hi0 hi1
        const unsigned p23 = __builtin_amdgcn_perm((unsigned)d.w, (unsigned)d.z, 0x05010400u);
        lo[m] = __builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u;
        hi[m] = __builtin_amdgcn_perm(p23, p01, 0x07060302u) ^ 0x80808080u;
    }
    // ---- vertical: rows r0 .. r0 + 15; H row of k slot (kg, j) is 16 (j >> 2) + 4 kg + (j & 3) ----
    const blur_v4i a_lo = {(int)lo[0], (int)lo[1], (int)lo[2], 0}, a_hi = {(int)hi[0], (int)hi[1], (int)hi[2], 0};
    constexpr int kC = 128 * 257 + 256 * 128 * 257 + 32768;      // the two offsets of 128 put back, and the rounding
    const blur_v4i c_v = {kC, kC, kC, kC}, c_0 = {0, 0, 0, 0};
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const blur_v4i bv = half == 0 ? blur_v4i{(int)tv.x, (int)tv.y, 0, 0} : blur_v4i{0, (int)tv.x, (int)tv.y, 0};
        const blur_v4i dl = (SA_FAST_SKIP & 32) ? a_lo + bv + c_v : __builtin_amdgcn_mfma_i32_16x16x64_i8(a_lo, bv, c_v, 0, 0, 0);
        const blur_v4i dh = (SA_FAST_SKIP & 32) ? a_hi + bv : __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hi, bv, c_0, 0, 0, 0);
        unsigned v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = ((unsigned)dh[i] << 8) + (unsigned)dl[i];       // < 2^24 + 2^17
        // (v >> 16) of two results as 16-bit halves, clamped to 255; then the four low bytes
        const us2 m255 = {255, 255};
        const unsigned q01 = as_u32(__builtin_elementwise_min(as_us2(__builtin_amdgcn_perm(v[1], v[0], 0x07060302u)), m255));
        const unsigned q23 = as_u32(__builtin_elementwise_min(as_us2(__builtin_amdgcn_perm(v[3], v[2], 0x07060302u)), m255));
        const unsigned out = __builtin_amdgcn_perm(q23, q01, 0x06040200u);
        const int gy = y0 + 16 * half + n;          // image row on the lane, columns x0 + 16 wave + 4 kg .. + 3 in `out`
        if (gy < h && (!(SA_FAST_SKIP & 1) || out == 0x12345678u))
            *reinterpret_cast<uint32_t*>(blur_level + blur_tiled_offset(pitch, (unsigned)(x0 + 16 * wave + 4 * kg), (unsigned)gy)) = out;
    }
}

__global__ __launch_bounds__(256) void fast_nms_kernel(FastArgs a) {
    // One block of LDS carved by hand (separately declared arrays get padded and ordered by the compiler, and four
    // bytes too many cost a workgroup per CU: 8 workgroups need 160 KB / 8 each; 7 are 4 % slower, 5 27 %).
    //   s_img      byte tile, pixel - 128 as signed bytes      (dead after phase 3; then: grid cells, survivor list)
    //   s_sc       score map
    //   s_list1    list 1 (phases 1-2)
    //   s_list2    the 6-bit tile (phase 1), then the corner lists, one segment per wave (phases 2-4)
    //   s_imgc     byte tile, 127 - pixel as signed bytes      (phase 2: a darker arc is a brighter arc of the complemented pixels)
    //   s_n        [0] length of list 1, [2] of the survivor list, [3] of list 1 when it is rebuilt with one entry per
    //              pixel, [4] set when the entries per (pixel, polarity) did not fit
    constexpr int kWaves = 4;
    constexpr int kSegEntries = ((kScN + 255) / 256) * 64;     // corners one wave can find: its 64 of every 256 entries
    constexpr int kSegBytes = kSegEntries * 2;
    constexpr int kOffImg = 0, kOffSc = kOffImg + kImgRows * kImgPitch, kOffList1 = kOffSc + kScBytes,
                  kOffList2 = kOffList1 + kList1Bytes, kOffImgC = kOffList2 + kWaves * kSegBytes,
                  kOffN = kOffImgC + kImgRows * kImgPitch, kLdsBytes = kOffN + 5 * 4;
    static_assert(kLdsBytes <= 20480, "LDS of fast_nms");
    static_assert(kOffSc % 16 == 0 && kOffList1 % 16 == 0 && kOffList2 % 16 == 0 && kOffImgC % 16 == 0 && kOffN % 4 == 0, "LDS alignment");
    __shared__ __align__(16) uint8_t s_mem[kLdsBytes];
    uint8_t* const s_img = s_mem + kOffImg;
    uint8_t* const s_sc = s_mem + kOffSc;
    uint16_t* const s_list1 = reinterpret_cast<uint16_t*>(s_mem + kOffList1);
    uint16_t* const s_list2 = reinterpret_cast<uint16_t*>(s_mem + kOffList2);
    unsigned* const s_n = reinterpret_cast<unsigned*>(s_mem + kOffN);
    constexpr unsigned kList1Cap = kList1Bytes / 2;
    // the 6-bit tile of phase 1 lies where the corner lists are built from phase 2 on
    uint32_t* s_q = reinterpret_cast<uint32_t*>(s_list2);
    static_assert((kImgRows * kQPitch + 1) * 4 <= kWaves * kSegBytes, "the 6-bit tile must fit the region of the corner lists");
    // grid cells of >= 8 x 8 pixels (grid option) and this tile's survivors, lx | ly << 6 | score << 16, over the byte
    // tile (dead once the blur has read it)
    unsigned* const s_cellbest = reinterpret_cast<unsigned*>(s_img);
    constexpr int kCells = (kFastTileW / 8) * (kFastTileH / 8);
    uint32_t* s_surv = reinterpret_cast<uint32_t*>(s_img) + kCells;
    static_assert((kTileListCap + kCells) * 4 <= kImgRows * kImgPitch, "survivor list and grid cells must fit the byte tile");
    static_assert(kListBytes <= kList1Bytes, "list 1 must hold one entry per scored position");
    constexpr int P = kImgPitch;

    // Grid: x = tile slot of a frame (a multiple of 8 of them), y = frame.  Workgroups are dealt
    // round-robin over the 8 XCDs, each with its own L2, in x-fastest order, so XCD k sees the slots
    // with x % 8 == k: it takes the k-th contiguous eighth of the tile table, whole tile rows of every level
    // (the table's order deals the rows out), and the tiles of a row, which share halo columns, run on the
    // same L2 instead of fetching them once per XCD.  The tile's level and place come from that table: no search over the levels, no division (scalar instructions
    // take issue cycles from the vector ones, and every wave of the workgroup repeats them).
    int b, slot;      // slot: place in the tile table
    if (a.frames_by_xcd) {
        // a multiple of 8 frames: XCD k takes whole frames (k, k + 8, ...): every XCD gets the same mix of
        // heavy (level 0, interior) and light (top levels, mostly border) tiles, and a frame stays in one L2
        b = (int)blockIdx.y * 8 + (int)(blockIdx.x & 7u);
        slot = (int)(blockIdx.x >> 3);
    } else {
        b = blockIdx.y;
        slot = (int)(blockIdx.x & 7u) * a.tiles_eighth + (int)(blockIdx.x >> 3);
    }
    if (slot >= a.g.total_tiles) return;
    const uint2 tde = a.tiledesc[slot];
    const uint32_t td = tde.x;
    const int l = (int)(td & 15u), txi = (int)((td >> 4) & 0xfffu), tyi = (int)((td >> 16) & 0xfffu), band = (int)(td >> 28);
    const LevelGeom& L = a.g.lv[l];
    const int ftile = (int)tde.y;      // the tile's number in the frame (levels, rows, columns)
    const int x0 = txi * kFastTileW, y0 = tyi * kFastTileH;
    const int w = L.w, h = L.h;
    const uint8_t* img;
    int pitch;
    bool aligned;
    if (l == 0) {
        img = a.img0 + (size_t)b * a.img0_frame_pitch;
        pitch = a.img0_pitch;
        aligned = a.img0_aligned != 0;
    } else {
        img = a.pyr + (size_t)b * a.g.frame_bytes + L.off;
        pitch = L.pitch;
        aligned = true;
    }
    const int tid = threadIdx.x;
    SA_STAMP_DECL;
    SA_STAMP(0);
    // the blur's taps of this lane, on their way while the tile is staged
    const uint4 blur_th = *reinterpret_cast<const uint4*>(&c_blur_taps.h[tid & 63][0]);
    const uint2 blur_tv = *reinterpret_cast<const uint2*>(&c_blur_taps.v[tid & 63][0]);
    if (tid < 5) s_n[tid] = 0;
    if (tid < kScBytes / 16) reinterpret_cast<uint4*>(s_sc)[tid] = make_uint4(0u, 0u, 0u, 0u);
    // Thread (column group cg, row rr) of an 18 x 14 arrangement: one dword column
    // of the staged tile, rows rr, rr + 14, rr + 28.  252 of the 256 threads take part.
    const int rr = tid / kImgDw, cg = tid - rr * kImgDw;
    const bool in_grid = tid < kImgDw * 14;
    // ---- stage image tile + halo, reflect-101 outside the image ----
    if (in_grid) {
        const int gx = x0 - kImgHalo + 4 * cg;
        uint32_t v[3];
        // workgroup-uniform cases: a tile whose halo lies inside the level needs no reflection at all
        const bool inside = aligned && x0 >= kImgHalo && x0 + kFastTileW + kImgHalo <= w && y0 >= kImgHalo &&
                            y0 + kFastTileH + kImgHalo <= h;
        if (SA_FAST_SKIP & 16) {
            for (int it = 0; it < 3; it++) v[it] = 0x40404040u + (unsigned)(tid & 3) * 0x01010101u;
        } else if (inside) {
            const uint8_t* p0 = img + (size_t)(y0 - kImgHalo + rr) * pitch + gx;
#pragma unroll
            for (int it = 0; it < 3; it++)
                v[it] = rr + 14 * it < kImgRows ? *reinterpret_cast<const uint32_t*>(p0 + (size_t)(14 * it) * pitch) : 0u;
        } else {
            const bool fast = aligned && gx >= 0 && gx + 3 < w;
            const bool near = w >= 4 && h >= 4;
            int rx[4];
#pragma unroll
            for (int j = 0; j < 4; j++) rx[j] = near ? reflect101_near(gx + j, w) : reflect101(gx + j, w);
#pragma unroll
            for (int it = 0; it < 3; it++) {
                const int r = rr + 14 * it;
                v[it] = 0;
                if (r < kImgRows) {
                    const int gy = near ? reflect101_near(y0 - kImgHalo + r, h) : reflect101(y0 - kImgHalo + r, h);
                    const uint8_t* row = img + (size_t)gy * pitch;
                    if (fast) {
                        v[it] = *reinterpret_cast<const uint32_t*>(row + gx);
                    } else {
                        v[it] = (uint32_t)row[rx[0]] | ((uint32_t)row[rx[1]] << 8) | ((uint32_t)row[rx[2]] << 16) |
                                ((uint32_t)row[rx[3]] << 24);
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int r = rr + 14 * it;
            if (r < kImgRows) {
                // both byte tiles hold signed bytes: pixel - 128 (what the blur's signed products take as they stand)
                // and 127 - pixel
                *reinterpret_cast<uint32_t*>(s_img + r * P + 4 * cg) = v[it] ^ 0x80808080u;
                *reinterpret_cast<uint32_t*>(s_img + (kOffImgC - kOffImg) + r * P + 4 * cg) = v[it] ^ 0x7f7f7f7fu;
                s_q[r * kQPitch + 1 + cg] = (v[it] >> 2) & 0x3f3f3f3fu;
            }
        }
    }
    SA_STAMP(1);
    lds_barrier();
    SA_STAMP(2);
    const int t = a.g.fast_threshold;
    const int edge = a.g.edge;
    const bool level_ok = (w > 2 * edge) && (h > 2 * edge);
    // ---- phase 1: compass test, four pixels per 32-bit operation ----
    // Item (row r of the scored region, column group cg) covers staged columns 4cg .. 4cg+3 of
    // staged row r + 3.  With d = centre - compass pixel for the four compass points: an arc of 9
    // of the 16 ring pixels holds at least one pixel of every opposite pair, so a darker arc needs
    // (d0 > t or d8 > t) and (d4 > t or d12 > t), a brighter one the same with d < -t.
    // The test runs on pixels shifted right by two, c = 4 c' + (0..3): c - x > t implies
    // 4 (c' - x') + 3 >= t + 1, i.e. c' - x' >= k = (t + 1) / 4 rounded down -- every pixel the exact
    // test passes passes this one (5 % more pass it; the ring test of phase 2 is exact).  A byte holds
    // c' - x' + 128 - k, which stays within [65 - k, 191 - k]: no carry or borrow between the bytes of
    // a word, and bit 7 says c' - x' >= k.
    // Only positions that can matter are tested: a keypoint lies at least `edge` pixels from
    // the level border and its 3x3 suppression looks one pixel further, so scores outside
    // [edge - 1, size - edge] are never read (edge >= 31 > 3, the FAST margin).
    unsigned Yp = 0;   // flags of the thread's three items: pixel j of item `it`: bit 8 j + 2 it brighter, + 1 darker
    const unsigned entry0 = ((unsigned)((rr + 3) * P + 4 * cg) >> 2) << 5;   // the thread's first item: staged offset / 4, above the bit number
    {
        // valid pixels of this thread's column group: bits 7 and 6 of the pixel's byte
        unsigned cmask = 0;
        if (in_grid && level_ok) {
            // valid staged columns [clo, chi): the scored region, cut to image columns [edge - 1, w - edge]
            const int clo = max(3, edge - 1 - (x0 - kImgHalo)), chi = min(kFastTileW + 5, w - edge + 1 - (x0 - kImgHalo));
            const unsigned span = (unsigned)max(chi - clo, 0);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((unsigned)(4 * cg + j - clo) < span) cmask |= 0xc0u << (8 * j);
        }
        // held in vector registers: an operation with a scalar-register operand issues at half rate
        unsigned KB = 0x80808080u - (unsigned)((t + 1) >> 2) * 0x01010101u, M7 = 0x80808080u;
        asm volatile("v_mov_b32 %0, %0" : "+v"(KB));
        asm volatile("v_mov_b32 %0, %0" : "+v"(M7));
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int r = rr + 14 * it;
            const int gy = y0 - 1 + r;
            const bool row_ok = gy >= edge - 1 && gy < h - edge + 1;
            // (rows that cannot hold a keypoint or a neighbour of one are skipped: in the tiles along the upper
            // and lower border whole waves fall out; the reads stay inside the tile either way)
            if (!(SA_FAST_SKIP & 8) && r < kScRows && row_ok && cmask != 0u) {     // (cmask is 0 where the level is too small for any keypoint)
                const uint32_t* qc = s_q + (r + 3) * kQPitch + 1 + cg;
                const unsigned W0 = qc[-1], C = qc[0], W2 = qc[1];
                const unsigned U = s_q[r * kQPitch + 1 + cg], D = s_q[(r + 6) * kQPitch + 1 + cg];
                const unsigned L = __builtin_amdgcn_alignbyte(C, W0, 1);     // columns - 3
                const unsigned R = __builtin_amdgcn_alignbyte(W2, C, 3);     // columns + 3
                const unsigned lo = C + KB, hi = KB - C;
                const unsigned dark = ((lo - U) | (lo - D)) & ((lo - L) | (lo - R));
                const unsigned brig = ((U + hi) | (D + hi)) & ((L + hi) | (R + hi));
                const unsigned y = ((dark & M7) | ((brig & M7) >> 1)) & cmask;
                Yp |= y >> (6 - 2 * it);
            }
        }
        // One list entry per passing (pixel, polarity), so that the ring test handles one polarity per entry:
        // (code0 / 4) << 5 | bit number in Yp (odd: darker).  The entry is taken apart by the phase that reads it,
        // where every lane has one, not in the loop below, which runs as long as the wave's busiest lane needs.
        // The list holds kList1Cap entries, more than any image but a contrived one produces; a wave whose
        // entries do not fit raises s_n[4] instead of writing them, and the list is then rebuilt below with one
        // entry per pixel (both flags), which always fits.
        const int cnt = __builtin_popcount(Yp);
        const int incl = wave_incl_scan(cnt);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total) {
            // (one atomic per wave: per-lane atomics on the one counter serialise in the LDS and cost more
            // than the prefix sum)
            unsigned base = 0;
            if ((tid & 63) == 0) base = atomicAdd(&s_n[0], (unsigned)total);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (base + (unsigned)total <= kList1Cap) {
                uint16_t* dst = s_list1 + base + (unsigned)(incl - cnt);
                unsigned E = Yp;
                while (E) {
                    const unsigned i = (unsigned)__builtin_ctz(E);
                    E &= E - 1;
                    *dst++ = (uint16_t)(entry0 | i);
                }
            } else if ((tid & 63) == 0) {
                s_n[4] = 1u;
            }
        }
    }
    SA_STAMP(3);
    lds_barrier();
    SA_STAMP(4);
    // ---- phase 2: full ring, 9 contiguous darker (or, on complemented pixels, brighter) ----
    // The sign of (ring - (v - t)) says "darker"; alignbit shifts it into the mask.
    const bool merged = s_n[4] != 0u;      // (workgroup-uniform)
    if (merged) {
        // the entries per (pixel, polarity) did not fit: one entry per pixel with both flags instead
        unsigned E = (Yp | (Yp >> 1)) & 0x15151515u;
        const int cnt = __builtin_popcount(E);
        const int incl = wave_incl_scan(cnt);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        unsigned base = 0;
        if ((tid & 63) == 0 && total) base = atomicAdd(&s_n[3], (unsigned)total);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        uint16_t* dst = s_list1 + base + (unsigned)(incl - cnt);
        while (E) {
            const unsigned i = (unsigned)__builtin_ctz(E);       // (even: the pixel's brighter flag)
            E &= E - 1;
            const unsigned f = (Yp >> i) & 3u;                   // bit 0 brighter, bit 1 darker
            *dst++ = (uint16_t)(entry0 | i | (f >> 1) | ((f & 1u) << 15));
        }
        lds_barrier();
    }
    const int n1 = (SA_FAST_SKIP & 4) ? 0 : (int)(merged ? s_n[3] : s_n[0]);
    // The score -- the largest threshold that still detects: the best arc's smallest difference, minus one -- says
    // whether the pixel is a corner at all (best > t), so it is computed for every listed position while its ring
    // is in registers: the corners need no second round of gathers.  A darker corner has no brighter arc and vice
    // versa: one polarity per entry.  Minima and maxima commute with subtracting the centre, so the arcs are
    // evaluated on the ring pixels themselves: for a brighter corner, best = (largest over the 16 arcs of the
    // smallest pixel of the arc) - centre; a darker corner is a brighter one of the complemented tile, which
    // lies in LDS too, so a lane only picks the tile it reads from -- no subtraction, no complement per ring pixel.
    // The 16 arcs of 9 pixels share their middles: arc k and arc k + 1 (k even) are pixels k + 1 .. k + 8
    // plus pixel k or pixel k + 9 (the larger of the two decides), and those middles are built from 8 pair minima -- 48 two-operand minima and maxima
    // in all, issued as 16-bit operations, which run at twice the rate of the 32-bit and three-operand forms.
    // Corners go to a list per wave (the wave that found them also suppresses them): its length stays in a scalar
    // register, and appending takes a ballot and a lane count, no atomic.
    const int wave = tid >> 6;
    uint16_t* const my_list2 = s_list2 + wave * kSegEntries;
    int n2 = 0;       // corners in this wave's list (wave-uniform)
    for (int k0 = 0; k0 < n1; k0 += 256) {
        const int k = k0 + tid;
        bool corner = false;
        unsigned code = 0, pol = 0;
        if (k < n1) {
            // entry: (staged offset of the thread's first item / 4) << 5 | bit number: pixel (bit >> 3) of item (bit >> 1) & 3,
            // odd = darker; an entry of a rebuilt list has bit 0 = darker, bit 15 = brighter
            const unsigned e = s_list1[k];
            code = ((e >> 3) & 0xffcu) + ((e >> 1) & 3u) * (unsigned)(14 * P) + ((e >> 3) & 3u);
            constexpr int off[16] = SA_RING_OFFSETS(P);
            auto arc_best = [&](unsigned brighter) {
                const uint8_t* p = s_img + code + (brighter ? 0u : (unsigned)(kOffImgC - kOffImg));
                short d[16];
#pragma unroll
                for (int j = 0; j < 16; j++) d[j] = (short)(int8_t)p[off[j]];
                const int centre = (int8_t)p[0];
                short pm[8], qm[8];
#pragma unroll
                for (int i = 0; i < 8; i++) pm[i] = min16(d[2 * i + 1], d[(2 * i + 2) & 15]);
#pragma unroll
                for (int i = 0; i < 8; i++) qm[i] = min16(pm[i], pm[(i + 1) & 7]);
                short best = -128;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const short core = min16(qm[i], qm[(i + 2) & 7]);        // pixels 2i + 1 .. 2i + 8
                    best = max16(best, min16(core, max16(d[2 * i], d[(2 * i + 9) & 15])));      // the better of arcs 2i and 2i + 1
                }
                return (int)best - centre;
            };
            pol = e & 1u ? 0u : 1u;     // darker where the entry says so
            int best = arc_best(pol);
            if (merged && best <= t && (e & 0x8001u) == 0x8001u) {   // an entry of a rebuilt list with both flags
                pol = 1u;
                best = arc_best(1u);
            }
            corner = best > t;
            if (corner) {
                const unsigned srow = staged_row(code);
                s_sc[code - (unsigned)(P - kScPitch) * srow - (unsigned)(3 * kScPitch + 3)] = (uint8_t)(best - 1);   // (srow - 3) * 66 + (column - 3)
            }
        }
        const unsigned long long cm = __ballot(corner);
        if (corner) my_list2[n2 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u))] = (uint16_t)(code | (pol << 12));
        n2 += __builtin_popcountll(cm);
    }
    SA_STAMP(5);
    // ---- phase 3: blur (no LDS written: no barrier in front of it) ----
    if (!(SA_FAST_SKIP & 2))
        blur_tile(s_img, wave, tid & 63, blur_th, blur_tv, a.blur + (size_t)b * a.g.frame_bytes + L.off, (unsigned)L.pitch, x0, y0, h);
    SA_STAMP(7);
    lds_barrier();      // every wave's scores are in the score map; the byte tile is dead from here on
    SA_STAMP(8);
    // ---- phase 4a: NMS + border filter over the corner list ----
    // A strict 3x3 maximum has no second one among its 8 neighbours, so a 64 x 32 tile holds at
    // most 32 x 16 = kTileListCap of them: the list cannot overflow.  Its order is arbitrary
    // (the selection kernel ranks the kept entries by position).
    {
        const int sh = a.grid_shift;
        // survivor of list entry k: lx | ly << 6 | score << 16, or 0
        auto survivor = [&](int k) -> unsigned {
            if (k >= n2) return 0u;
            const unsigned code = my_list2[k] & 0xfffu;
            const unsigned sr = staged_row(code);
            const int lx = (int)(code - sr * P) - kImgHalo, ly = (int)sr - kImgHalo;
            const int gx = x0 + lx, gy = y0 + ly;
            if ((unsigned)lx >= (unsigned)kFastTileW || (unsigned)ly >= (unsigned)kFastTileH || gx < edge ||
                gx >= w - edge || gy < edge || gy >= h - edge)
                return 0u;
            const uint8_t* q = s_sc + (code - (unsigned)(P - kScPitch) * sr - (unsigned)(3 * kScPitch + 3));
            constexpr int S = kScPitch;
            // all nine reads first: one LDS round trip instead of a chain of eight
            const int s = q[0];
            const int n0 = q[-1], n1 = q[1], n2_ = q[-S - 1], n3 = q[-S], n4 = q[-S + 1], n5 = q[S - 1], n6 = q[S], n7 = q[S + 1];
            const int top = max(max(max(n0, n1), max(n2_, n3)), max(max(n4, n5), max(n6, n7)));
            return (s != 0 && s > top) ? ((unsigned)lx | ((unsigned)ly << 6) | ((unsigned)s << 16)) : 0u;
        };
        // Optional grid distribution (not in the reference): of the survivors of one
        // cell of grid_cell x grid_cell pixels only the best stays -- highest score,
        // first in raster order among equals.  Cells are anchored at the level origin
        // and divide the tile, so the contest is local to the workgroup.
        auto cell_key = [&](unsigned sv, int& cell) -> unsigned {
            const int lx = sv & 63, ly = (sv >> 6) & 31;
            cell = (ly >> sh) * (kFastTileW >> sh) + (lx >> sh);
            return (sv & 0xffff0000u) | (0xffffu - (unsigned)(ly * kFastTileW + lx));
        };
        if (sh) {
            if (tid < kCells) s_cellbest[tid] = 0;
            lds_barrier();      // uniform: grid_shift is a kernel argument
            for (int k0 = 0; k0 < n2; k0 += 64) {
                const unsigned sv = survivor(k0 + (tid & 63));
                if (sv) { int cell; const unsigned key = cell_key(sv, cell); atomicMax(&s_cellbest[cell], key); }
            }
            lds_barrier();      // uniform: grid_shift is a kernel argument
        }
        for (int k0 = 0; k0 < n2; k0 += 64) {
            unsigned sv = survivor(k0 + (tid & 63));
            if (sh && sv) { int cell; const unsigned key = cell_key(sv, cell); if (s_cellbest[cell] != key) sv = 0; }
            const unsigned long long m = __ballot(sv != 0);
            if (m) {
                const int lane = tid & 63;
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&s_n[2], (unsigned)__builtin_popcountll(m));
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                if (sv) {
                    const unsigned slot = base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (slot < (unsigned)kTileListCap) s_surv[slot] = sv;
                }
            }
        }
    }
    SA_STAMP(9);
    SA_STAMP(10);
    lds_barrier();   // the survivor list is complete
    {
        const unsigned ns = min(s_n[2], (unsigned)kTileListCap);
        const size_t gt = (size_t)b * a.g.total_tiles + ftile;
        if (tid == 0) a.tilecount[gt] = ns;
        uint32_t* out = a.tilelist + gt * kTileListCap;
        // the level's score histogram takes the survivors directly (no copy in LDS: a tile adds a dozen), and
        // last of all: atomics on a few hundred hot addresses are slow, and what a wave sends to memory
        // after them queues behind them
        unsigned* hist = a.hist + (((size_t)b * a.g.nlevels + l) * a.g.bands + band) * 256;
        for (unsigned k = tid; k < ns; k += 256) {
            const uint32_t sv = s_surv[k];
            out[k] = sv;
            atomicAdd(&hist[sv >> 16], 1u);
        }
    }
    SA_STAMP(15);
    SA_STAMP_PRINT;
}

// ---------------------------------------------------------------------------
// Selection + compaction, one workgroup of 8 waves per (frame, level) (16 waves measured 5 us slower per launch).
//   1. KeyPointsFilter::retainBest from the score histograms: wave w takes level
//      w (levels up to the workgroup's own), suffix sums over the 256 bins by
//      shuffles -> cut score and kept count.  The level's first output index is
//      the sum of the kept counts of the lower levels.
//   2. The kept survivors (score >= cut) of the level's tile lists are gathered into LDS as
//      keys y << 14 | x.  A key's place in the canonical (octave, y, x) order is the number of
//      kept keys in the rows above it plus those of its own row further left: keys are counted
//      per row, the counts prefixed, the keys' x grouped by row and compared within the row
//      (a handful of keys) -- no sort and no atomics on the output.  A level with more kept
//      survivors than the LDS cache holds (long runs of equal scores) is ranked against the tile
//      lists in memory instead; one with more rows than the prefix array by counting the smaller
//      keys, all against all.
// ---------------------------------------------------------------------------
#ifndef SA_SEL_THREADS
#define SA_SEL_THREADS 512
#endif
constexpr int kSelThreads = SA_SEL_THREADS;        // threads of a select_compact workgroup
constexpr int kSelTileChunk = 4 * kSelThreads;    // tiles whose counts are held in LDS at a time (four per thread)
constexpr int kSelPrefCap = 4096;                 // entries of the prefix array: a chunk of tiles, or the rows of a level
static_assert(kSelTileChunk <= kSelPrefCap && kSelPrefCap % kSelThreads == 0, "prefix array");
constexpr int kSelKeyCap = 4096;      // kept survivors ranked from LDS

constexpr int kQtCap = 8192;          // survivors of one level the quadtree kernel sorts in LDS

// Selection of every level up to my_level of frame b, a wave per level (nwaves of them take turns): cut score and kept
// count of KeyPointsFilter::retainBest from the score histogram (lane owns bins 4 lane ..
// 4 lane + 3, suffix sums over the bins).  With the quadtree option a level that holds more
// survivors than its quota (and no more than the quadtree kernel sorts) keeps exactly the quota
// and is marked mode 1: the quadtree kernel writes it, the plain kernel skips it.
__device__ __forceinline__ void level_selection(const SelectCompactArgs& a, int bands, int b, int my_level, int wave, int nwaves, int lane,
                                                int* s_cut, int* s_cnt, int* s_mode) {
    for (int l = wave; l <= my_level; l += nwaves) {
        // (the level's histogram is the sum of its bands')
        const uint4* h = reinterpret_cast<const uint4*>(a.hist + ((size_t)b * a.g.nlevels + l) * bands * 256) + lane;
        unsigned v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        for (int k = 0; k < bands; k++) {
            const uint4 q = h[k * 64];
            v0 += q.x; v1 += q.y; v2 += q.z; v3 += q.w;
        }
        const unsigned own = v0 + v1 + v2 + v3;
        unsigned suf = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned n = __shfl_down(suf, o, 64);
            if (lane + o < 64) suf += n;
        }
        const unsigned above = suf - own;
        const unsigned s3 = above + v3, s2 = s3 + v2, s1 = s2 + v1, s0 = s1 + v0;
        const unsigned total = __shfl(suf, 0, 64);
        const unsigned quota = (unsigned)a.g.lv[l].quota;
        int cut = 1, cnt = (int)total, mode = 0;
        if (total > quota) {
            if (quota == 0) {
                cut = 256; cnt = 0;
            } else if (a.quadtree && total <= (unsigned)kQtCap) {
                mode = 1; cnt = (int)quota;
            } else {
                // largest bin c with suffix(c) >= quota; ties at the cut are all kept
                int best = -1; unsigned bc = 0;
                if (s3 >= quota) { best = 4 * lane + 3; bc = s3; }
                else if (s2 >= quota) { best = 4 * lane + 2; bc = s2; }
                else if (s1 >= quota) { best = 4 * lane + 1; bc = s1; }
                else if (s0 >= quota) { best = 4 * lane; bc = s0; }
                const unsigned long long m = __ballot(best >= 0);
                const int top = 63 - __builtin_clzll(m);  // m != 0 because total > quota >= 1
                cut = __shfl(best, top, 64);
                cnt = (int)__shfl(bc, top, 64);
            }
        }
        if (lane == 0) { s_cut[l] = cut; s_cnt[l] = cnt; s_mode[l] = mode; }
    }
}

// Exclusive prefix over kSelThreads * R values, thread t holding values R t .. R t + R - 1: returns the
// prefix of the thread's first value; `total` is the sum of all.  Synchronises the workgroup (twice).
template <int R>
__device__ __forceinline__ unsigned select_block_scan(const unsigned (&c)[R], unsigned* s_wave, unsigned& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned sum = 0;
#pragma unroll
    for (int k = 0; k < R; k++) sum += c[k];
    unsigned inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    __syncthreads();              // s_wave of an earlier scan has been read
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned before = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < kSelThreads / 64; w++) {
        const unsigned v = s_wave[w];
        if (w < wave) before += v;
        total += v;
    }
    return before + inc - sum;
}

// kBanded: the level's tile rows are shared out to a.g.bands workgroups (large frames); else one workgroup per level.
template <bool kBanded>
__global__ __launch_bounds__(kSelThreads) void select_compact_kernel(SelectCompactArgs a) {
    __shared__ int s_cut[kMaxLevels], s_cnt[kMaxLevels], s_mode[kMaxLevels], s_bandkept[16];
    __shared__ unsigned s_wave[kSelThreads / 64];
    __shared__ unsigned s_pref[kSelPrefCap + 1];      // tile prefix while gathering, row prefix while ranking
    __shared__ __align__(16) unsigned s_key[kSelKeyCap];
    __shared__ uint16_t s_rowx[kSelKeyCap];           // the kept keys' x, grouped by row
    __shared__ uint8_t s_score[kSelKeyCap];
    __shared__ unsigned s_n;
    // Workgroups go round-robin over the 8 XCDs; the level is rotated by the frame so that every XCD
    // gets all levels (with 8 levels and level = blockIdx % 8, one XCD would get every frame's level 0,
    // the longest lists, and another every frame's top level).
    // One workgroup per level: the lower levels, with the long lists, come first in the grid and the top quarter of the
    // levels last.  Three workgroups fit a CU; when the grid is more than one round of them (128 frames x 8 levels are
    // 1.33), the workgroups that start late are the short ones and the launch ends with them instead of with a second
    // level-0 list (24 -> 22.5 us at 128 frames: the workgroups of a CU share its LDS and issue slots, so most of the time is
    // throughput, not the tail).
    const int bands = kBanded ? a.g.bands : 1;
    const int per_frame = a.g.nlevels * bands;
    int b, band = 0, my_level;
    if (kBanded) {
        b = blockIdx.x / per_frame;
        const int in_frame = (int)blockIdx.x - b * per_frame;
        band = in_frame / a.g.nlevels;          // the band of tile rows this workgroup selects from
        my_level = (in_frame - band * a.g.nlevels + b) % a.g.nlevels;
    } else {
        const int n_low = a.g.nlevels - a.g.nlevels / 4, n_top = a.g.nlevels - n_low;      // 6 + 2 of 8 levels
        const int frames = (int)gridDim.x / a.g.nlevels, low_groups = frames * n_low;
        if ((int)blockIdx.x < low_groups) {
            b = blockIdx.x / n_low;
            my_level = ((int)blockIdx.x - b * n_low + b) % n_low;
        } else {
            const int j = (int)blockIdx.x - low_groups;
            b = j / n_top;
            my_level = n_low + (j - b * n_top + b) % n_top;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if SA_SEL_STAMPS
    unsigned long long sst[6] = {};
    sst[0] = __builtin_amdgcn_s_memtime();
#endif
#if SA_SEL_EXP == 1
    if (a.capacity > 0) return;     // experiment: what launching the grid costs
#endif
    const int l = my_level;
    const LevelGeom& L = a.g.lv[l];
    const int tiles_x = L.tiles_x, nty = div_up_dev(L.h, kFastTileH), ntl = tiles_x * nty;
    // this workgroup's tiles: the tile rows of its band
    const int row_begin = kBanded ? band_first_tile_row(band, nty, bands) : 0, row_end = kBanded ? band_first_tile_row(band + 1, nty, bands) : nty;
    const int t_begin = row_begin * tiles_x, t_end = row_end * tiles_x;
    const unsigned* tcount = a.tilecount + (size_t)b * a.g.total_tiles + L.tile_base;
    const uint32_t* tlist = a.tilelist + ((size_t)b * a.g.total_tiles + L.tile_base) * kTileListCap;
    // survivor counts of four tiles per thread; the first chunk's are fetched together with the histograms
    unsigned c4[4];
    auto load_counts = [&](int chunk0) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int t = chunk0 + tid * 4 + k;
            c4[k] = t < t_end ? min(tcount[t], (unsigned)kTileListCap) : 0u;
        }
    };
    load_counts(t_begin);
    if (tid == 0) s_n = 0;
    level_selection(a, bands, b, my_level, wave, kSelThreads / 64, lane, s_cut, s_cnt, s_mode);
    __syncthreads();
#if SA_SEL_EXP == 2
    if (a.capacity > 0) return;     // experiment: launch + selection
#endif
#if SA_SEL_STAMPS
    sst[1] = __builtin_amdgcn_s_memtime();
#endif

    const unsigned cut = (unsigned)s_cut[l];
    int kept = s_cnt[l];
    int level_base = 0;                   // keypoints of the lower levels
    for (int i = 0; i < l; i++) level_base += s_cnt[i];
    int base = level_base;                // ... and of the level's bands above this one
    if (kBanded) {
        // kept keypoints of every band up to this one: its histogram from the cut score on (a wave per band)
        for (int k = wave; k <= band; k += kSelThreads / 64) {
            const uint4 q = reinterpret_cast<const uint4*>(a.hist + (((size_t)b * a.g.nlevels + l) * bands + k) * 256)[lane];
            const unsigned s0 = 4u * (unsigned)lane;
            const int c = (int)((s0 >= cut ? q.x : 0u) + (s0 + 1 >= cut ? q.y : 0u) + (s0 + 2 >= cut ? q.z : 0u) + (s0 + 3 >= cut ? q.w : 0u));
            const int total = __builtin_amdgcn_readlane(wave_incl_scan(c), 63);
            if (lane == 0) s_bandkept[k] = total;
        }
        __syncthreads();
        for (int k = 0; k < band; k++) base += s_bandkept[k];
        kept = s_bandkept[band];
    }
    slam_amd_keypoint* kp = a.kp + (size_t)b * a.capacity;
    uint32_t* kpxy = a.kpxy + (size_t)b * a.capacity;
    const bool cached = kept <= a.keycap;   // uniform

    auto emit = [&](unsigned key, unsigned sc, int pos) {
        if ((unsigned)pos < (unsigned)a.capacity) {
            const unsigned x = key & 0x3fffu, y = key >> 14;
            slam_amd_keypoint q;
            q.x = (float)x;
            q.y = (float)y;
            q.size = L.size;
            q.angle = -1.f;
            q.response = (float)sc;
            q.octave = l;
            q.class_id = -1;
            kp[pos] = q;
            kpxy[pos] = x | (y << 14) | ((uint32_t)l << 28);
        }
    };
    // key of entry e of tile t of this level
    auto key_of = [&](int t, uint32_t e) -> unsigned {
        const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
        const unsigned x = (unsigned)(txi * kFastTileW) + (e & 63u), y = (unsigned)(tyi * kFastTileH) + ((e >> 6) & 31u);
        return (y << 14) | x;
    };

    if (kept > 0 && cut < 256 && s_mode[l] == 0) {
        for (int chunk0 = t_begin; chunk0 < t_end; chunk0 += kSelTileChunk) {
            const int nt = min(kSelTileChunk, t_end - chunk0);
            if (chunk0 > t_begin) {
                __syncthreads();          // s_pref of the chunk before has been read
                load_counts(chunk0);
            }
            // exclusive prefix of the tiles' survivor counts
            unsigned total;
            unsigned run = select_block_scan<4>(c4, s_wave, total);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (tid * 4 + k < nt) s_pref[tid * 4 + k] = run;
                run += c4[k];
            }
            if (tid == 0) s_pref[nt] = total;
            __syncthreads();
#if SA_SEL_STAMPS
            sst[2] = __builtin_amdgcn_s_memtime();
#endif
            // every entry of the chunk's lists, flat; its tile by binary search in the prefix.  Several
            // entries per thread and trip, so that their loads are in flight together (an entry past
            // the end reads as score 0, below every cut).
#ifndef SA_SEL_GATHER
#define SA_SEL_GATHER 5
#endif
            constexpr int kGather = SA_SEL_GATHER;
            for (unsigned e0 = 0; e0 < total; e0 += kGather * kSelThreads) {
                uint32_t ent[kGather];
                int tile[kGather];
#pragma unroll
                for (int u = 0; u < kGather; u++) {
                    const unsigned e = e0 + (unsigned)(u * kSelThreads + tid);
                    int lo = 0, hi = nt;      // largest t with s_pref[t] <= e
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (s_pref[mid] <= e) lo = mid; else hi = mid;
                    }
                    tile[u] = chunk0 + lo;
                    ent[u] = e < total ? tlist[(size_t)(chunk0 + lo) * kTileListCap + (e - s_pref[lo])] : 0u;
                }
#pragma unroll
                for (int u = 0; u < kGather; u++) {
                    const unsigned sc = ent[u] >> 16;
                    if (sc < cut) continue;
                    const unsigned key = key_of(tile[u], ent[u]);
                    if (cached) {
                        // one atomic per wave: the lanes' atomics on the one counter would be served one by one
                        const unsigned long long m = __ballot(true);
                        const int leader = __builtin_ctzll(m);
                        unsigned slot = 0;
                        if (lane == leader) slot = atomicAdd(&s_n, (unsigned)__builtin_popcountll(m));
                        slot = __shfl(slot, leader, 64) + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                        if (slot < (unsigned)kSelKeyCap) { s_key[slot] = key; s_score[slot] = (uint8_t)sc; }
                    } else {
                        // rank against the lists in memory: every kept entry of the level with a smaller key
                        int rank = 0;
                        for (int t = 0; t < ntl; t++) {
                            const unsigned n = min(tcount[t], (unsigned)kTileListCap);
                            const uint32_t* p = tlist + (size_t)t * kTileListCap;
                            for (unsigned j = 0; j < n; j++) {
                                const uint32_t o = p[j];
                                rank += ((o >> 16) >= cut) && (key_of(t, o) < key);
                            }
                        }
                        emit(key, sc, level_base + rank);
                    }
                }
            }
        }
        if (cached) {
            __syncthreads();
#if SA_SEL_STAMPS
            sst[3] = __builtin_amdgcn_s_memtime();
#endif
            const int n = (int)min(s_n, (unsigned)kSelKeyCap);
            const unsigned y_begin = (unsigned)(row_begin * kFastTileH);        // first image row of the band
            if (min(L.h, row_end * kFastTileH) - (int)y_begin <= kSelPrefCap) {
                // Rank = keys in the rows above + keys of the same row further left: count the keys of
                // every row, prefix over the rows, group the keys' x by row, compare within the row.
                constexpr int R = kSelPrefCap / kSelThreads;
#pragma unroll
                for (int k = 0; k < R; k++) s_pref[tid + k * kSelThreads] = 0;
                __syncthreads();
                for (int i = tid; i < n; i += kSelThreads) atomicAdd(&s_pref[(s_key[i] >> 14) - y_begin], 1u);
                __syncthreads();
                unsigned r[R];
#pragma unroll
                for (int k = 0; k < R; k++) r[k] = s_pref[tid * R + k];
                unsigned rows_total;
                unsigned run = select_block_scan<R>(r, s_wave, rows_total);
#pragma unroll
                for (int k = 0; k < R; k++) {
                    s_pref[tid * R + k] = run;       // first place of row tid R + k ...
                    run += r[k];
                }
                __syncthreads();
                for (int i = tid; i < n; i += kSelThreads) {
                    const unsigned key = s_key[i];
                    s_rowx[atomicAdd(&s_pref[(key >> 14) - y_begin], 1u)] = (uint16_t)(key & 0x3fffu);   // ... and now one past its last
                }
                __syncthreads();
                for (int i = tid; i < n; i += kSelThreads) {
                    const unsigned key = s_key[i];
                    const unsigned y = (key >> 14) - y_begin, x = key & 0x3fffu;
                    const unsigned first = y ? s_pref[y - 1] : 0u, last = s_pref[y];
                    int rank = (int)first;
                    for (unsigned j = first; j < last; j++) rank += s_rowx[j] < x;
                    emit(key, s_score[i], base + rank);
                }
            } else {
                // a level with more rows than the prefix array holds: count the smaller keys, all against all
                if (tid < 4 && n + tid < kSelKeyCap) s_key[n + tid] = 0xffffffffu;   // pad to a multiple of 4
                __syncthreads();
                for (int i = tid; i < n; i += kSelThreads) {
                    const unsigned key = s_key[i];
                    int rank = 0;
                    for (int j = 0; j < n; j += 4) {
                        const uint4 k4 = *reinterpret_cast<const uint4*>(&s_key[j]);
                        rank += (k4.x < key) + (k4.y < key) + (k4.z < key) + (k4.w < key);
                    }
                    emit(key, s_score[i], base + rank);
                }
            }
        }
    }
#if SA_SEL_STAMPS
    sst[4] = __builtin_amdgcn_s_memtime();
    if (tid == 0 && my_level == 0)
        printf("select start %llu end %llu frame %d level %d kept %d entries %u : selection %llu prefix %llu gather %llu rank+emit %llu\n", sst[0] & 0xfffffffull, sst[4] & 0xfffffffull, b, my_level, s_cnt[my_level],
               (unsigned)s_n, sst[1] - sst[0], sst[2] - sst[1], sst[3] - sst[2], sst[4] - sst[3]);
#endif
    // the workgroup of the top level has every level's count: it publishes the totals
    if (threadIdx.x == 0 && my_level == a.g.nlevels - 1 && band == 0) {
        FrameSelect fs;
        int acc = 0;
        for (int i = 0; i < kMaxLevels; i++) {
            const bool on = i < a.g.nlevels;
            fs.cut[i] = on ? s_cut[i] : 256;
            fs.count[i] = on ? s_cnt[i] : 0;
            fs.offset[i] = acc;
            acc += fs.count[i];
        }
        fs.total = acc;
        fs.pad[0] = fs.pad[1] = fs.pad[2] = 0;
        a.sel[b] = fs;
        if (a.d_count) a.d_count[b] = acc;
    }
}

// ---------------------------------------------------------------------------
// Quadtree distribution (option beyond the reference; the CPU oracle walks an explicit tree).
// One workgroup per (frame, level) whose survivors exceed the quota.  Every survivor gets the
// path code of its finest cell: root index, then one quadrant digit (2 lower + right) per split,
// cells split at x0 + ceil(w / 2), y0 + ceil(h / 2).  Sorted by code, the cells of every depth
// are runs, and the number of leading digits a survivor shares with its predecessor says at
// which depth the two separate: cells(d) = 1 + #{shared <= d}.  The tree grows whole levels
// while that stays below the target; in the last round cells are split one by one, most
// populated first (path order among equals), until the target is reached.  Every cell keeps
// its best survivor; if that is more than the target, the best `target` stay.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void quadtree_select_kernel(SelectCompactArgs a) {
    __shared__ int s_cut[kMaxLevels], s_cnt[kMaxLevels], s_mode[kMaxLevels];
    __shared__ unsigned long long s_key[kQtCap];   // code << 32 | score << 24 | y << 12 | x
    __shared__ unsigned s_aux[kQtCap];             // cell candidates, then the winners
    __shared__ uint8_t s_mark[kQtCap];             // bit 0: first of a node; bit 1: first of a split cell
    __shared__ unsigned s_hc[32];
    __shared__ unsigned s_n, s_nc, s_nw;
    const int b = blockIdx.x / a.g.nlevels;
    const int l = blockIdx.x - b * a.g.nlevels;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_n = 0; s_nc = 0; s_nw = 0; }
    if (tid < 32) s_hc[tid] = 0;
    level_selection(a, a.g.bands, b, l, wave, 16, lane, s_cut, s_cnt, s_mode);
    __syncthreads();
    if (s_mode[l] != 1) return;
    const LevelGeom& L = a.g.lv[l];
    const int target = s_cnt[l];
    int base = 0;
    for (int i = 0; i < l; i++) base += s_cnt[i];
    // geometry of the tree
    const int edge = a.g.edge;
    const int rx0 = edge, ry0 = edge, rx1 = L.w - edge, ry1 = L.h - edge;
    const int W = rx1 - rx0, H = ry1 - ry0;
    const int nini = max(1, (2 * W + H) / (2 * H));
    const int rw = (W + nini - 1) / nini;
    int dmax = 0;
    while ((1 << dmax) < max(rw, H)) dmax++;
    // ---- gather the level's survivors with their path codes ----
    const int tiles_x = L.tiles_x, ntl = tiles_x * div_up_dev(L.h, kFastTileH);
    const unsigned* tcount = a.tilecount + (size_t)b * a.g.total_tiles + L.tile_base;
    const uint32_t* tlist = a.tilelist + ((size_t)b * a.g.total_tiles + L.tile_base) * kTileListCap;
    for (int t = wave; t < ntl; t += 16) {
        const unsigned nt = min(tcount[t], (unsigned)kTileListCap);
        const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
        for (unsigned e0 = 0; e0 < nt; e0 += 64) {
            const unsigned e = e0 + lane;
            const bool on = e < nt;
            unsigned long long key = 0;
            if (on) {
                const uint32_t ent = tlist[(size_t)t * kTileListCap + e];
                const int x = txi * kFastTileW + (int)(ent & 63u), y = tyi * kFastTileH + (int)((ent >> 6) & 31u);
                const int root = (x - rx0) / rw;
                int x0 = rx0 + root * rw, x1 = min(x0 + rw, rx1), y0 = ry0, y1 = ry1;
                unsigned code = (unsigned)root;
                for (int d = 0; d < dmax; d++) {
                    const int mx = x0 + (x1 - x0 + 1) / 2, my = y0 + (y1 - y0 + 1) / 2;
                    const bool right = x >= mx, lower = y >= my;
                    code = (code << 2) | ((unsigned)lower << 1) | (unsigned)right;
                    if (right) x0 = mx; else x1 = mx;
                    if (lower) y0 = my; else y1 = my;
                }
                key = ((unsigned long long)code << 32) | ((unsigned long long)(ent >> 16) << 24) | ((unsigned)y << 12) | (unsigned)x;
            }
            const unsigned long long m = __ballot(on);
            unsigned slot0 = 0;
            if (lane == 0) slot0 = atomicAdd(&s_n, (unsigned)__builtin_popcountll(m));
            slot0 = (unsigned)__builtin_amdgcn_readfirstlane((int)slot0);
            if (on && slot0 + lane < (unsigned)kQtCap) s_key[slot0 + lane] = key;
        }
    }
    __syncthreads();
    const int S = (int)min(s_n, (unsigned)kQtCap);
    int n2 = 1;
    while (n2 < S) n2 <<= 1;
    for (int i = S + tid; i < n2; i += 1024) s_key[i] = ~0ull;
    __syncthreads();
    // ---- bitonic sort, ascending ----
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = tid; p < n2 / 2; p += 1024) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));   // bit j of i is 0
                const int q = i | j;
                const unsigned long long ki = s_key[i], kq = s_key[q];
                const bool up = (i & k) == 0;
                if ((ki > kq) == up) { s_key[i] = kq; s_key[q] = ki; }
            }
            __syncthreads();
        }
    }
    // ---- digits shared with the predecessor (digit 0 = root index), cells per depth ----
    auto shared_digits = [&](int i) -> int {
        const unsigned x = (unsigned)(s_key[i] >> 32) ^ (unsigned)(s_key[i - 1] >> 32);
        if (dmax < 16 && (x >> (2 * dmax)) != 0) return 0;
        const int hb = 31 - __builtin_clz(x);         // codes differ: positions are distinct
        return 1 + (dmax - 1 - (hb >> 1));
    };
    for (int i = 1 + tid; i < S; i += 1024) atomicAdd(&s_hc[shared_digits(i)], 1u);
    __syncthreads();
    int dstop = dmax, cells_d = 0, cells_prev = 0;
    {
        int cells = 1;
        for (int d = 0; d <= dmax; d++) {
            cells += (int)s_hc[d];
            if (cells >= target) { dstop = d; cells_d = cells; break; }
            cells_prev = cells;
        }
    }
    (void)cells_d;
    const int D = max(dstop - 1, 0);
    const bool careful = dstop > 0;
    const int need = target - cells_prev;           // cells still missing at depth D (careful round)
    // ---- careful round: cells of depth D that gain by a split, ordered by population ----
    for (int i = tid; i < S; i += 1024) s_mark[i] = 0;
    __syncthreads();
    if (careful) {
        for (int i = tid; i < S; i += 1024) {
            if (i != 0 && shared_digits(i) > D) continue;      // not the first of a cell
            int pop = 1, gain = 0;
            for (int j = i + 1; j < S; j++) {
                const int c = shared_digits(j);
                if (c <= D) break;
                pop++;
                gain += c == D + 1;
            }
            if (gain > 0) {
                const unsigned slot = atomicAdd(&s_nc, 1u);
                // order: population descending, then path order; 13 bits each for index and gain
                s_aux[slot] = ((unsigned)pop << 18) | ((unsigned)(kQtCap - 1 - i) << 5) | (unsigned)min(gain, 3);
            }
        }
    }
    __syncthreads();
    if (careful) {
        const int nc = (int)s_nc;
        for (int c = tid; c < nc; c += 1024) {
            const unsigned me = s_aux[c];
            int before = 0;      // cells gained by the candidates ahead of this one
            for (int o = 0; o < nc; o++) {
                const unsigned v = s_aux[o];
                before += ((v >> 5) > (me >> 5)) ? (int)(v & 31u) : 0;
            }
            if (before < need) s_mark[kQtCap - 1 - (int)((me >> 5) & 8191u)] = 2;
        }
    }
    __syncthreads();
    // ---- nodes: first survivor of every final cell ----
    for (int i = tid; i < S; i += 1024) {
        if (i != 0 && shared_digits(i) > D) continue;
        const bool split = (s_mark[i] & 2) != 0;
        s_mark[i] |= 1;      // every byte of s_mark has one writer per phase
        if (!split) continue;
        for (int j = i + 1; j < S; j++) {
            const int c = shared_digits(j);
            if (c <= D) break;
            if (c == D + 1) s_mark[j] = 1;
        }
    }
    __syncthreads();
    // ---- winner of every node: highest score, then lowest y, then lowest x ----
    for (int i = tid; i < S; i += 1024) {
        if (!(s_mark[i] & 1)) continue;
        unsigned bestw = 0;
        for (int j = i; j < S; j++) {
            if (j > i && (s_mark[j] & 1)) break;
            const unsigned pay = (unsigned)s_key[j];
            const unsigned w = (pay & 0xff000000u) | (0x00ffffffu - (pay & 0x00ffffffu));
            bestw = max(bestw, w);
        }
        s_aux[atomicAdd(&s_nw, 1u)] = bestw;      // the candidate list is dead
    }
    __syncthreads();
    // ---- at most `target` winners, written in (y, x) order ----
    const int M = (int)s_nw;
    slam_amd_keypoint* kp = a.kp + (size_t)b * a.capacity;
    uint32_t* kpxy = a.kpxy + (size_t)b * a.capacity;
    unsigned* s_kept = reinterpret_cast<unsigned*>(s_key);      // the sorted keys are dead
    for (int i = tid; i < M; i += 1024) {
        const unsigned w = s_aux[i];
        int better = 0;
        for (int o = 0; o < M; o++) better += s_aux[o] > w;
        // 0xffffffff sorts after every position
        s_kept[i] = better < target ? (0x00ffffffu - (w & 0x00ffffffu)) : 0xffffffffu;
    }
    __syncthreads();
    for (int i = tid; i < M; i += 1024) {
        const unsigned yx = s_kept[i];
        if (yx == 0xffffffffu) continue;
        int rank = 0;
        for (int o = 0; o < M; o++) rank += s_kept[o] < yx;
        const int pos = base + rank;
        if ((unsigned)pos < (unsigned)a.capacity) {
            const unsigned x = yx & 0xfffu, y = yx >> 12;
            slam_amd_keypoint q;
            q.x = (float)x;
            q.y = (float)y;
            q.size = L.size;
            q.angle = -1.f;
            q.response = (float)(s_aux[i] >> 24);
            q.octave = l;
            q.class_id = -1;
            kp[pos] = q;
            kpxy[pos] = x | (y << 14) | ((uint32_t)l << 28);
        }
    }
}

// ---------------------------------------------------------------------------
// Describe: a single-wave workgroup takes kDescKps consecutive keypoints of one frame.
// Everything that belongs to one keypoint (position, level, patch addresses, later its
// moments, angle, cos / sin and the finished descriptor) lives in the registers of one
// lane, keypoint j in lane j; what the whole wave needs of it is handed out by
// v_readlane, and results go back by v_writelane.  The scalar unit then has next to
// nothing to do per keypoint, which matters: scalar instructions are not free on this
// chip (a wave that issues one per two vector instructions already runs a quarter
// slower than the vector instructions alone).
//   pass 1  intensity centroid of every keypoint, the wave on one keypoint at a time:
//           the 31 x 31 patch of the unblurred level arrives as one 16-byte load per
//           lane (row lane / 2, columns 16 (lane % 2) ...) and never leaves the
//           registers; a lane's 16 pixels meet constant byte weights (column numbers,
//           0 outside the radius-15 disc; ones inside it) in v_dot4_u32_u8
//   pass 2  angle, cos / sin, keypoint record: one keypoint per lane, once
//   pass 3  256 rotated rBRIEF comparisons per keypoint, 4 per lane, assembled by ballot,
//           on the 37 x 37 patch of the blurred level (the rotated pattern reaches at most
//           18 pixels), staged in LDS by three 16-byte loads per lane: rows of 64 bytes
//           from the 16-byte boundary before the patch, (row, chunk) = (lane / 4, lane % 4)
// The unblurred patch is fetched at its own alignment (the memory system handles unaligned
// 16-byte loads, at a cycle per lane in the address unit), and the loads of the next
// keypoints are in flight while the current one is processed.  Keypoints lie >= 31 pixels
// from the level border, so both patches are inside the level; the bytes a row's 64 reach
// beyond the patch are inside the level's rows, which are 64-byte aligned (never used).
// No workgroup barrier is needed.
// ---------------------------------------------------------------------------
constexpr int kURad = 15, kURows = 31;
constexpr int kBRad = 18, kBRows = 37;
#ifndef SA_DESC_PITCH
#define SA_DESC_PITCH 80
#endif
constexpr int kPPitch = SA_DESC_PITCH;   // LDS bytes per row of the blurred patch: a multiple of 16 (16-byte stores) whose rows start on
                                         // 8 different bank offsets (20 r mod 32); with 64 bytes the 37 rows fall on 2 and the samples
                                         // of a wave, which lie within 13 pixels of the centre, on half of the 32 banks
constexpr int kBLoadRows = 48;     // rows fetched at most: from the tile row that holds the patch's first row (up to 7 rows above it)

// rBRIEF sampling pattern as floats: x0, y0, x1, y1 of pair p at [4 * p].
__constant__ __align__(16) float c_pattern[256 * 4] = {
#include "../../include/orb_pattern_31.inc"
};

// Half-width of row |v| of the radius-15 disc, 4 bits per row:
// 15 15 15 15 14 14 14 13 13 12 11 10 9 8 6 3 for |v| = 0 .. 15.  The disc is
// symmetric, so the same table gives the half-height of column |u|.
constexpr int disc_half_width(int av) { return (int)((0x3689ABCDDEEEFFFFull >> (4 * av)) & 15u); }

// Byte weights of pass 1.  Lane l holds pixels (row l / 2, columns 16 (l % 2) + 4 q + e) in byte e of
// register q: w[l][q] the column numbers of the pixels inside the disc (0 outside), w[l][4 + q] ones
// for them.
struct DiscWeights { unsigned w[64][8]; };
constexpr DiscWeights make_disc_weights() {
    DiscWeights t = {};
    for (int l = 0; l < 64; l++)
        for (int q = 0; q < 4; q++)
            for (int e = 0; e < 4; e++) {
                const int r = l / 2, c = 16 * (l % 2) + 4 * q + e;
                if (r >= kURows || c >= kURows) continue;
                const int v = r - kURad, u = c - kURad;
                if ((u < 0 ? -u : u) > disc_half_width(v < 0 ? -v : v)) continue;
                t.w[l][q] |= (unsigned)c << (8 * e);
                t.w[l][4 + q] |= 1u << (8 * e);
            }
    return t;
}
__constant__ __align__(16) DiscWeights c_disc = make_disc_weights();

// Sum over the 64 lanes, left in lane 63: butterfly inside each row of 16 lanes, then the row
// totals are passed up (rows 1 and 3 take the row before, rows 2 and 3 the half before).
__device__ __forceinline__ int wave_sum_to_lane63(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);  // row_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
}

// v_writelane: lane `lane` of every v takes the (wave-uniform) value next to it.  The lane select travels
// in M0 (an instruction reads at most one scalar register besides it), which nothing else in these
// kernels uses; it is set in the statement that reads it.
__device__ __forceinline__ void write_lane2(int& v0, int& v1, int a0, int a1, int lane) {
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                 : "+v"(v0), "+v"(v1) : "s"(a0), "s"(a1), "s"(lane));
}
__device__ __forceinline__ void write_lane8(unsigned (&v)[8], const unsigned long long (&a)[4], int lane) {
    asm volatile("s_mov_b32 m0, %12\n\ts_nop 0\n\t"
                 "v_writelane_b32 %0, %8, m0\n\tv_writelane_b32 %1, %9, m0\n\t"
                 "v_writelane_b32 %2, %10, m0\n\tv_writelane_b32 %3, %11, m0\n\t"
                 "v_writelane_b32 %4, %13, m0\n\tv_writelane_b32 %5, %14, m0\n\t"
                 "v_writelane_b32 %6, %15, m0\n\tv_writelane_b32 %7, %16, m0"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "s"((unsigned)a[0]), "s"((unsigned)(a[0] >> 32)), "s"((unsigned)a[1]), "s"((unsigned)(a[1] >> 32)), "s"(lane),
                   "s"((unsigned)a[2]), "s"((unsigned)(a[2] >> 32)), "s"((unsigned)a[3]), "s"((unsigned)(a[3] >> 32)));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_a1;   // a 16-byte load from any address
typedef const __attribute__((address_space(1))) uint8_t* global_bytes_t;
typedef const __attribute__((address_space(1))) u32x4_a1* global_u4_t;

#ifndef SA_DESC_KPS
#define SA_DESC_KPS 8
#endif
constexpr int kDescKps = SA_DESC_KPS;                   // keypoints per (single-wave) workgroup
#ifndef SA_DESC_AHEAD
#define SA_DESC_AHEAD 3
#endif
constexpr int kDescAhead = SA_DESC_AHEAD < SA_DESC_KPS ? SA_DESC_AHEAD : SA_DESC_KPS;   // blurred patches in flight (3: 90 registers, 5 waves per SIMD; 4: 101, 4 waves, 3 us slower)

__global__ __launch_bounds__(64) void describe_kernel(DescribeArgs a) {
    __shared__ __align__(16) uint8_t s_patch[kBLoadRows * kPPitch];
    // 1-D grid, frame index fastest, so the workgroups beyond the last keypoint of
    // every frame (they exit at once) sit at the end of the grid.
    SA_DSTAMP_DECL;
    SA_DSTAMP(0);
    const int b = blockIdx.x % a.batch;
    const int i0 = (blockIdx.x / a.batch) * kDescKps;
    const int lane = threadIdx.x;
    // The workgroups of a frame leave the frame's score histograms zeroed for the next call, one histogram
    // (1 KB) each, the first workgroups taking a second one where there are more histograms than workgroups
    // (the selection kernels, which read them, ran before this one).
    if (a.hist_reset) {
        const int wg = (int)(blockIdx.x / (unsigned)a.batch), wgs = (int)(gridDim.x / (unsigned)a.batch);
        uint4* hz = reinterpret_cast<uint4*>(a.hist_reset + (size_t)b * a.g.nlevels * a.g.bands * 256);
        for (int h = wg; h < a.g.nlevels * a.g.bands; h += wgs) hz[h * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
    }
    const int n = min(a.sel[b].total, a.capacity);
    if (i0 >= n) return;
    const int nk = min(kDescKps, n - i0);
    SA_DSTAMP(1);      // count known
    auto exchange = [](int from_lane, int v) { return __builtin_amdgcn_ds_bpermute(4 * from_lane, v); };

    // ---- per lane, once: geometry of level `lane`, then everything about keypoint `lane` ----
    // (from a table in device memory: a per-lane read of the kernel arguments themselves is a vector
    // load from the argument buffer, which need not be cached)
    int gw = 0, gh = 0, gpitch = 0, goff = 0, gscale = 0;
    if (lane < a.g.nlevels) {
        const int4 g4 = reinterpret_cast<const int4*>(a.lanegeom)[2 * lane];
        gw = g4.x; gh = g4.y; gpitch = g4.z; goff = g4.w; gscale = a.lanegeom[8 * lane + 4];
    }
    uint32_t xyv = 0xffffffffu;                       // level 15: rejected
    if (lane < nk) xyv = a.kpxy[(size_t)b * a.capacity + i0 + lane];
    const int px = (int)(xyv & 0x3fffu), py = (int)((xyv >> 14) & 0x3fffu), pl = (int)(xyv >> 28);
    const bool lvl_ok = pl < a.g.nlevels;
    const int lsel = lvl_ok ? pl : 0;
    const int lw = exchange(lsel, gw), lh = exchange(lsel, gh), lpitch = exchange(lsel, gpitch);
    const unsigned loff = (unsigned)exchange(lsel, goff);
    const float lscale = __builtin_bit_cast(float, exchange(lsel, gscale));
    // The selection kernel only emits positions at least `edge` (>= 31) pixels from the border, so this
    // rejects nothing it wrote; it keeps a record that was never written (zeros) or a count that ran ahead
    // of the records from becoming a wild read.  (24 = patch radius 18 and some; the 48 bytes of a row reach
    // past the patch and may run past the level's right border into the next row, which then exists.)
    const bool okv = lvl_ok && !(px < 24 || py < 24 || px >= lw - 24 || py >= lh - 24);
    // First byte of either patch and the row pitch there.  A lane without a valid keypoint points at the
    // start of the frame's buffers with a row pitch of 0: the passes below run for it like for any
    // other and its results are dropped at the end.
    const uint8_t* ulevel = pl == 0 ? a.img0 + (size_t)b * a.img0_frame_pitch : a.pyr + (size_t)b * a.g.frame_bytes + loff;
    const int upitch = okv ? (pl == 0 ? a.img0_pitch : lpitch) : 0;
    const int bpitch = okv ? lpitch : 0;
    const unsigned long long uaddr = reinterpret_cast<uintptr_t>(
        okv ? ulevel + (ptrdiff_t)(py - kURad) * upitch + (px - kURad) : a.pyr + (size_t)b * a.g.frame_bytes);
    // The blurred level lies in tiles of 16 x 8 pixels.  The patch is fetched from the tile that holds its
    // first pixel: whole 16-byte tile rows, four tiles across, up to 7 rows above the patch and 15 columns
    // left of it coming along (bshift, yshift: where the patch starts in what was fetched).
    const int bx0 = px - kBRad, by0 = py - kBRad;
    const int bshift = okv ? (bx0 & 15) : 0, yshift = okv ? (by0 & 7) : 0;
    const unsigned long long baddr = reinterpret_cast<uintptr_t>(
        a.blur + (size_t)b * a.g.frame_bytes + (okv ? loff + blur_tiled_offset((unsigned)lpitch, (unsigned)(bx0 & ~15), (unsigned)(by0 & ~7)) : 0u));
    const int ualo = (int)(unsigned)uaddr, uahi = (int)(unsigned)(uaddr >> 32);
    const int balo = (int)(unsigned)baddr, bahi = (int)(unsigned)(baddr >> 32);
    auto patch_base = [&](int alo, int ahi, int j) {
#ifdef SA_DESC_EXP_SAMEPATCH
        j = 0;      // timing experiment: every keypoint of the workgroup fetches the first one's patches
#endif
        return (global_bytes_t)((unsigned long long)(unsigned)__builtin_amdgcn_readlane(alo, j) |
                                ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(ahi, j) << 32));
    };

    SA_DSTAMP(2);      // keypoints and geometry known
    // ---- pass 1: intensity centroid, one keypoint at a time ----
    // m10 = sum of (column - 15) x pixel, m01 = sum of (row - 15) x pixel over the disc.
    int m10v = 0, m01v = 0;     // moments of keypoint `lane`
    {
        const u32x4 wc = *reinterpret_cast<const u32x4*>(&c_disc.w[lane][0]);
        const u32x4 wm = *reinterpret_cast<const u32x4*>(&c_disc.w[lane][4]);
        const int urow = min(lane >> 1, kURows - 1), ucol = 16 * (lane & 1);   // lanes 62, 63: row 30 again, weights 0
        const int vrow = (lane >> 1) - kURad;
        auto load_u = [&](int j) {
            const int pitch = __builtin_amdgcn_readlane(upitch, j);
            return *(global_u4_t)(patch_base(ualo, uahi, j) + (__umul24((unsigned)urow, (unsigned)pitch) + (unsigned)ucol));
        };
        // every keypoint's load goes out before the first one's arithmetic (slots beyond the last keypoint
        // repeat it; their results land in lanes that hold no keypoint)
        u32x4 up[kDescKps];
#pragma unroll
        for (int j = 0; j < kDescKps; j++) up[j] = load_u(min(j, nk - 1));
#pragma unroll
        for (int j = 0; j < kDescKps; j++) {
            const u32x4 p = up[j];
            unsigned wsum = 0, sum = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                wsum = __builtin_amdgcn_udot4(p[q], wc[q], wsum, false);
                sum = __builtin_amdgcn_udot4(p[q], wm[q], sum, false);
            }
            const int m10 = __builtin_amdgcn_readlane(wave_sum_to_lane63((int)wsum + __mul24((int)sum, -kURad)), 63);
            const int m01 = __builtin_amdgcn_readlane(wave_sum_to_lane63(__mul24(vrow, (int)sum)), 63);
            write_lane2(m10v, m01v, m10, m01, j);
        }
    }
    SA_DSTAMP(3);      // pass 1 done
    // ---- pass 2: angle, cos and sin, keypoint record: every keypoint in its own lane ----
    float cav = 0.f, snv = 0.f;
    if (okv) {
        const float angle = fast_atan2_deg((float)m01v, (float)m10v);
        const float rad = SA_FMUL(angle, (float)(3.14159265358979323846 / 180.0));
        sincos_f32(rad, &cav, &snv);
        slam_amd_keypoint* kp = a.kp + (size_t)b * a.capacity + i0 + lane;
        typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
        f32x2_a4 xy = {SA_FMUL((float)px, lscale), SA_FMUL((float)py, lscale)};
        *reinterpret_cast<f32x2_a4*>(&kp->x) = xy;     // x and y are the record's first two fields
        kp->angle = angle;
    }
    SA_DSTAMP(4);      // pass 2 done
    // ---- pass 3: rBRIEF, pair p = 64*k + lane, k = 0..3; ballot k holds bytes 8k..8k+7 ----
    unsigned dsc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // descriptor of keypoint `lane`
    {
        float4 pat[4];
#pragma unroll
        for (int k = 0; k < 4; k++) pat[k] = reinterpret_cast<const float4*>(c_pattern)[64 * k + lane];
        // Rows of 64 bytes from the 16-byte column at or before the patch's first one, 16 rows per load, the
        // third load as far as the patch reaches (what a load costs the address unit is the number of
        // (quad of lanes, 64-byte line) pairs it touches: in the tiled level four rows of a 16-byte column
        // are one aligned line).  The patch starts at the same column of every row of s_patch.
        // lane -> (row of a 16-row load, 16-byte column): a quad of lanes takes four rows of one tile,
        // 64 consecutive bytes, and 16 consecutive lanes fill 256 consecutive bytes of s_patch
        // (eight rows of a column per eight lanes - a whole 128-byte tile - measured slower: those 16 lanes
        // write s_patch with two-way bank conflicts)
        const int brow = 4 * (lane >> 4) + (lane & 3), bchunk = (lane >> 2) & 3;
        uint8_t* const w0 = s_patch + brow * kPPitch + 16 * bchunk;
        struct PatchRegs { u32x4 v0, v1, v2; };
        auto load_b = [&](int j) {
            global_bytes_t base = patch_base(balo, bahi, j);
            const unsigned pitch8 = (unsigned)__builtin_amdgcn_readlane(bpitch, j) * 8u;   // bytes from one tile row to the next
            const int rows = __builtin_amdgcn_readlane(yshift, j) + kBRows;               // rows of the fetch that the patch reaches
            const unsigned at = ((brow & 8) ? pitch8 : 0u) + 128u * (unsigned)bchunk + 16u * (unsigned)(brow & 7);
            // the fourth 16-byte column is only reached by a patch that starts in the last 4 bytes of the first one:
            // otherwise its quads of lanes stay out of the loads (a quarter of what they cost the address unit)
            const bool col = 16 * bchunk < __builtin_amdgcn_readlane(bshift, j) + kBRows;
            PatchRegs r;
            r.v0 = r.v1 = r.v2 = u32x4{0u, 0u, 0u, 0u};
            if (col) {
                r.v0 = *(global_u4_t)(base + at);
                r.v1 = *(global_u4_t)(base + (at + 2u * pitch8));
                if (32 + brow < rows) r.v2 = *(global_u4_t)(base + (at + 4u * pitch8));
            }
            return r;
        };
        // LDS traffic of one wave executes in order; this only keeps the compiler from moving LDS
        // reads and writes across it.
        auto lds_order = [] {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        // Rounding to the nearest integer, ties to even, as the low bits of x + 1.5 * 2^23; the sum's bit
        // pattern is 0x4B400000 + the integer.  Row * pitch + column on the patterns themselves, the two
        // offsets taken off at the end (together with the patch centre's place in s_patch).
        constexpr float kMagic = 12582912.0f;
        constexpr unsigned kMagicBits = 0x4B400000u;
        constexpr unsigned kUnbias = (kMagicBits & 0xffffffu) * (unsigned)kPPitch + kMagicBits - (unsigned)(kBRad * kPPitch + kBRad);
        const unsigned s_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s_patch - kUnbias;
        auto sample_at = [&](float fx, float fy, unsigned origin) -> int {
            const unsigned bx = __builtin_bit_cast(unsigned, SA_FADD(fx, kMagic)), by = __builtin_bit_cast(unsigned, SA_FADD(fy, kMagic));
            const unsigned at = (unsigned)__umul24(by, (unsigned)kPPitch) + bx + origin;    // the row number is the low 24 bits of by
            return *(__attribute__((address_space(3))) const uint8_t*)(uintptr_t)at;
        };
        // kDescAhead patches are in flight ahead of the one being sampled
        PatchRegs ahead[kDescAhead];
#pragma unroll
        for (int d = 0; d < kDescAhead; d++) ahead[d] = load_b(min(d, nk - 1));
#pragma unroll
        for (int j = 0; j < kDescKps; j++) {
            *reinterpret_cast<u32x4*>(w0) = ahead[j % kDescAhead].v0;
            *reinterpret_cast<u32x4*>(w0 + 16 * kPPitch) = ahead[j % kDescAhead].v1;
            *reinterpret_cast<u32x4*>(w0 + 32 * kPPitch) = ahead[j % kDescAhead].v2;
            if (j + kDescAhead < kDescKps) ahead[j % kDescAhead] = load_b(min(j + kDescAhead, nk - 1));
            lds_order();
            // (copies in vector registers: an instruction with a scalar-register operand issues at half rate)
            float ca = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cav), j));
            float sn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, snv), j));
            unsigned origin = s_base + (unsigned)__builtin_amdgcn_readlane(bshift + yshift * kPPitch, j);
            asm volatile("v_mov_b32 %0, %0" : "+v"(ca));
            asm volatile("v_mov_b32 %0, %0" : "+v"(sn));
            asm volatile("v_mov_b32 %0, %0" : "+v"(origin));
            unsigned long long bits[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int t0 = sample_at(SA_FSUB(SA_FMUL(pat[k].x, ca), SA_FMUL(pat[k].y, sn)),
                                         SA_FADD(SA_FMUL(pat[k].x, sn), SA_FMUL(pat[k].y, ca)), origin);
                const int t1 = sample_at(SA_FSUB(SA_FMUL(pat[k].z, ca), SA_FMUL(pat[k].w, sn)),
                                         SA_FADD(SA_FMUL(pat[k].z, sn), SA_FMUL(pat[k].w, ca)), origin);
                bits[k] = __ballot(t0 < t1);
            }
            write_lane8(dsc, bits, j);
            lds_order();   // the next keypoint's patch overwrites s_patch: reads first
        }
    }
    SA_DSTAMP(5);      // pass 3 done
    if (okv) {
        uint4* out = reinterpret_cast<uint4*>(a.desc + ((size_t)b * a.capacity + i0 + lane) * 32);
        out[0] = make_uint4(dsc[0], dsc[1], dsc[2], dsc[3]);
        out[1] = make_uint4(dsc[4], dsc[5], dsc[6], dsc[7]);
    }
    SA_DSTAMP(6);
    SA_DSTAMP_PRINT;
}

// ---------------------------------------------------------------------------
// Launchers (host)
// ---------------------------------------------------------------------------
int launch_fast(const FastArgs& a0, int batch, hipStream_t st) {
    FastArgs a = a0;
    a.tiles_eighth = div_up(a.g.total_tiles, 8);
    if (batch > 65535) SA_FAIL(SLAM_AMD_ERR_ARG, "fast: batch too large for one launch");
    a.frames_by_xcd = SA_FAST_FRAMES_BY_XCD && batch % 8 == 0;
    dim3 grid(a.tiles_eighth * 8, batch);
    if (a.frames_by_xcd) grid = dim3(a.g.total_tiles * 8, batch / 8);
    else a.tiledesc += a.g.total_tiles;      // the table's second order: tile rows dealt over the eighths
    hipLaunchKernelGGL(fast_nms_kernel, grid, dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_select_compact(const SelectCompactArgs& a, int batch, hipStream_t st) {
    if (a.g.bands > 1) hipLaunchKernelGGL(select_compact_kernel<true>, dim3(batch * a.g.nlevels * a.g.bands), dim3(kSelThreads), 0, st, a);
    else hipLaunchKernelGGL(select_compact_kernel<false>, dim3(batch * a.g.nlevels), dim3(kSelThreads), 0, st, a);
    SA_HIP(hipGetLastError());
    if (a.quadtree) {
        hipLaunchKernelGGL(quadtree_select_kernel, dim3(batch * a.g.nlevels), dim3(1024), 0, st, a);
        SA_HIP(hipGetLastError());
    }
    return SLAM_AMD_OK;
}

int launch_describe(const DescribeArgs& a, int batch, hipStream_t st) {
    DescribeArgs d = a;
    d.batch = batch;
    const long long blocks = (long long)div_up(a.capacity, kDescKps) * batch;
    if (blocks > 0x7fffffffLL) SA_FAIL(SLAM_AMD_ERR_ARG, "describe: capacity x batch too large for one launch");
    dim3 grid((unsigned)blocks);
    hipLaunchKernelGGL(describe_kernel, grid, dim3(64), 0, st, d);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

}  // namespace sa
