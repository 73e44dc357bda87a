// HIP kernels of the ORB extraction path for gfx950 (wave64, LDS-staged tiles).
//
// Per batch of frames, in stream order:
//   pyramid       (pyramid_kernels.hip) chained bit-exact bilinear pyramid
//   fast_nms                      FAST-9/16 score + 3x3 NMS + border filter + per-level score
//                                 histogram + per-row survivor lists + 7x7 blur, all levels, one launch
//   select_compact                per-level cut score from the histogram (retain-best), then
//                                 the kept survivors of every row's sparse list -> keypoints
//                                 in canonical (octave, y, x) order
//   describe                      intensity-centroid angle + 7x7 blur + rBRIEF,
//                                 one wavefront per keypoint, patch held in LDS
#include "common.h"
#include "detmath.h"
#include "kernels.h"

namespace sa {

// ---------------------------------------------------------------------------
// FAST-9/16 + NMS + blur.  A workgroup owns a 64 x 32 tile of one level of one
// frame.  The image tile with a 4-pixel halo is staged in LDS (reflect-101
// outside the image), then
//   phase 1  every position of the tile plus a 1-pixel halo: compass test (a
//            9-arc always covers two of the four compass points); survivors are
//            compacted into an LDS list
//   phase 2  full 16-pixel ring masks for the listed positions; corners are
//            compacted into a second list
//   phase 3  corner score for the listed corners only
//   phase 4  3x3 strict-maximum suppression + border filter -> survivor map,
//            256-bin histogram of kept scores; 7-tap blur row pass (packed u16)
//   phase 5  blur column pass -> blurred level
// The lists keep the expensive phases dense instead of divergent.
// ---------------------------------------------------------------------------
constexpr int kImgHalo = 4;
constexpr int kImgPitch = kFastTileW + 2 * kImgHalo + 4;  // 72 staged columns; 19 dwords per row, odd, so that
                                                          // the rows of a column fall into different LDS banks
constexpr int kImgRows = kFastTileH + 2 * kImgHalo;   // 40
constexpr int kScW = kFastTileW + 2;                  // 66 scored columns
constexpr int kScRows = kFastTileH + 2;               // 34 scored rows
constexpr int kScPitch = kImgPitch;                   // LDS column = tile column + 4
constexpr int kScN = kScW * kScRows;                  // 2244 positions
constexpr int kRowRows = kFastTileH + 6;              // 38 rows of the blur row pass
constexpr int kListBytes = ((kScN * 2 + 15) / 16) * 16;

__device__ __forceinline__ int div_up_dev(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

__device__ __forceinline__ bool has_run9(unsigned m16) {
    unsigned m = m16 | (m16 << 16);
    unsigned r = m & (m >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= m >> 8;
    return (r & 0xffffu) != 0;
}

// v - ring[k] for the 16 ring pixels around c (LDS image tile, pitch kImgPitch).
__device__ __forceinline__ void ring_diffs(const uint8_t* c, int (&d)[16]) {
    const int P = kImgPitch;
    const int v = c[0];
    d[0] = v - c[3 * P];       d[1] = v - c[3 * P + 1];   d[2] = v - c[2 * P + 2];   d[3] = v - c[P + 3];
    d[4] = v - c[3];           d[5] = v - c[-P + 3];      d[6] = v - c[-2 * P + 2];  d[7] = v - c[-3 * P + 1];
    d[8] = v - c[-3 * P];      d[9] = v - c[-3 * P - 1];  d[10] = v - c[-2 * P - 2]; d[11] = v - c[-P - 3];
    d[12] = v - c[-3];         d[13] = v - c[P - 3];      d[14] = v - c[2 * P - 2];  d[15] = v - c[3 * P - 1];
}

// Corner score: max(t, best arc of 9 by min |difference|) - 1.  Windows of 3 by
// three-input min / max, windows of 9 from three windows of 3.
__device__ __forceinline__ int fast_corner_score(const int (&d)[16], int t) {
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo3[k] = min(min(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
        hi3[k] = max(max(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
    }
    int a0 = t, b0 = 255;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const int lo9a = min(min(lo3[k], lo3[(k + 3) & 15]), lo3[(k + 6) & 15]);
        const int lo9b = min(min(lo3[k + 1], lo3[(k + 4) & 15]), lo3[(k + 7) & 15]);
        const int hi9a = max(max(hi3[k], hi3[(k + 3) & 15]), hi3[(k + 6) & 15]);
        const int hi9b = max(max(hi3[k + 1], hi3[(k + 4) & 15]), hi3[(k + 7) & 15]);
        a0 = max(max(a0, lo9a), lo9b);
        b0 = min(min(b0, hi9a), hi9b);
    }
    b0 = min(b0, -a0);
    return -b0 - 1;
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

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ss2 as_ss2(unsigned v) { return __builtin_bit_cast(ss2, v); }
__device__ __forceinline__ unsigned as_u32s(ss2 v) { return __builtin_bit_cast(unsigned, v); }

__device__ __forceinline__ us2 as_us2(unsigned v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ unsigned as_u32(us2 v) { return __builtin_bit_cast(unsigned, v); }

__global__ __launch_bounds__(256) void fast_nms_kernel(FastArgs a) {
    __shared__ __align__(16) uint8_t s_img[kImgRows * kImgPitch];
    __shared__ __align__(16) uint8_t s_sc[kScRows * kScPitch];
    // lists (phases 1-3) and the blur row-pass buffer (phases 4-5) share one region
    __shared__ __align__(16) uint8_t s_scratch[2 * kListBytes];
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_n[3];   // lengths of list 1, list 2 and the survivor list
    __shared__ unsigned s_cellbest[(kFastTileW / 8) * (kFastTileH / 8)];   // grid cells of >= 8 x 8 pixels
    __shared__ uint32_t s_surv[kTileListCap];   // this tile's survivors: lx | ly << 6 | score << 16
    uint16_t* s_list1 = reinterpret_cast<uint16_t*>(s_scratch);
    uint16_t* s_list2 = reinterpret_cast<uint16_t*>(s_scratch + kListBytes);
    unsigned* s_row = reinterpret_cast<unsigned*>(s_scratch);  // [38][32] packed u16 pairs
    static_assert(kScPitch == kImgPitch, "score tile and image tile share one pitch");
    static_assert(kRowRows * kFastTileW * 2 <= 2 * kListBytes, "row-pass buffer must fit the scratch region");

    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  XCD k
    // takes the k-th contiguous eighth of all (frame, tile) pairs, so tiles that share
    // halo lines run on the same L2 instead of fetching them once per XCD.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = xcd * a.tiles_per_xcd + slot;
    if (g >= a.tiles_all) return;
    const int b = g / a.g.total_tiles;
    const int ftile = g - b * a.g.total_tiles;
    int l = 0;
#pragma unroll 1
    for (int i = 1; i < a.g.nlevels; i++)
        if (ftile >= a.g.lv[i].tile_base) l = i;
    const LevelGeom& L = a.g.lv[l];
    const int tile = ftile - L.tile_base;
    const int tyi = tile / L.tiles_x, txi = tile - tyi * L.tiles_x;
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
    s_hist[tid] = 0;
    if (tid < 3) s_n[tid] = 0;
    if (tid < (kFastTileW / 8) * (kFastTileH / 8)) s_cellbest[tid] = 0;
    // Thread (column group cg, row rr) of an 18 x 14 arrangement: one dword column
    // of the staged tile, rows rr, rr + 14, rr + 28.  252 of the 256 threads take part.
    const int rr = tid / 18, cg = tid - rr * 18;
    const bool in_grid = tid < 252;
    // ---- stage image tile + halo, reflect-101 outside the image ----
    if (in_grid) {
        const int gx = x0 - kImgHalo + 4 * cg;
        const bool fast = aligned && gx >= 0 && gx + 3 < w;
        int rx[4];
#pragma unroll
        for (int j = 0; j < 4; j++) rx[j] = reflect101(gx + j, w);
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int r = rr + 14 * it;
            if (r < kImgRows) {
                const int gy = reflect101(y0 - kImgHalo + r, h);
                const uint8_t* row = img + (size_t)gy * pitch;
                uint32_t v;
                if (fast) {
                    v = *reinterpret_cast<const uint32_t*>(row + gx);
                } else {
                    v = (uint32_t)row[rx[0]] | ((uint32_t)row[rx[1]] << 8) | ((uint32_t)row[rx[2]] << 16) |
                        ((uint32_t)row[rx[3]] << 24);
                }
                *reinterpret_cast<uint32_t*>(s_img + r * kImgPitch + 4 * cg) = v;
            }
        }
    }
    lds_barrier();
    const int t = a.g.fast_threshold;
    const int edge = a.g.edge;
    const bool level_ok = (w > 2 * edge) && (h > 2 * edge);
    // ---- phase 1: compass test, 4 pixels per item, packed 16-bit arithmetic ----
    // Item (row r of the scored region, column group cg) covers staged columns
    // 4cg .. 4cg+3 of staged row r + 3.  With d = centre - compass pixel for the four
    // compass points: an arc of 9 of the 16 ring pixels holds at least one pixel of every
    // opposite pair, so a darker arc needs max(d0, d8) > t and max(d4, d12) > t, a brighter
    // one min(d0, d8) < -t and min(d4, d12) < -t.
    // Only positions that can matter are tested: a keypoint lies at least `edge` pixels from
    // the level border and its 3x3 suppression looks one pixel further, so scores outside
    // [edge - 1, size - edge] are never read (edge >= 31 > 3, the FAST margin).
#pragma unroll
    for (int it = 0; it < 3; it++) {
        const int r = rr + 14 * it;
        unsigned pass4 = 0;
        if (in_grid && r < kScRows) {
            const uint32_t* rowc = reinterpret_cast<const uint32_t*>(s_img + (r + 3) * kImgPitch);
            const uint32_t C = rowc[cg];
            const uint32_t Wm = cg > 0 ? rowc[cg - 1] : 0u;
            const uint32_t Wp = cg < 17 ? rowc[cg + 1] : 0u;
            const uint32_t U = reinterpret_cast<const uint32_t*>(s_img + r * kImgPitch)[cg];
            const uint32_t D = reinterpret_cast<const uint32_t*>(s_img + (r + 6) * kImgPitch)[cg];
            const uint32_t L = __builtin_amdgcn_alignbyte(C, Wm, 1);   // columns -3
            const uint32_t R = __builtin_amdgcn_alignbyte(Wp, C, 3);   // columns +3
            reinterpret_cast<uint32_t*>(s_sc)[r * (kScPitch / 4) + cg] = 0;  // clears the score tile
            const ss2 T2 = {(short)t, (short)t};
            unsigned f[2];
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                const unsigned sel = hf ? 0x0c030c02u : 0x0c010c00u;
                const ss2 c = as_ss2(__builtin_amdgcn_perm(0u, C, sel));
                const ss2 d0 = c - as_ss2(__builtin_amdgcn_perm(0u, D, sel));
                const ss2 d4 = c - as_ss2(__builtin_amdgcn_perm(0u, R, sel));
                const ss2 d8 = c - as_ss2(__builtin_amdgcn_perm(0u, U, sel));
                const ss2 d12 = c - as_ss2(__builtin_amdgcn_perm(0u, L, sel));
                const ss2 hi1 = __builtin_elementwise_max(d0, d8), lo1 = __builtin_elementwise_min(d0, d8);
                const ss2 hi2 = __builtin_elementwise_max(d4, d12), lo2 = __builtin_elementwise_min(d4, d12);
                // the smaller of the two pair maxima / the larger of the two pair minima
                const ss2 dark = __builtin_elementwise_min(hi1, hi2), bright = __builtin_elementwise_max(lo1, lo2);
                // sign bit of each 16-bit half set <=> that pixel passes
                f[hf] = as_u32s(T2 - dark) | as_u32s(bright + T2);
            }
            pass4 = ((f[0] >> 15) & 1u) | ((f[0] >> 30) & 2u) | ((f[1] >> 13) & 4u) | ((f[1] >> 28) & 8u);
            // positions outside the scored region or within 3 pixels of the image edge
            const int col0 = 4 * cg;                    // staged column of pixel 0
            const int gx0 = x0 - kImgHalo + col0;
            const int gy = y0 - 1 + r;
            int jmin = max(3 - col0, (edge - 1) - gx0), jmax = min(69 - col0, (w - edge + 1) - gx0);
            jmin = max(jmin, 0); jmax = min(jmax, 4);
            unsigned vmask = (level_ok && jmax > jmin && gy >= edge - 1 && gy < h - edge + 1)
                                 ? (((1u << jmax) - 1u) & ~((1u << jmin) - 1u)) : 0u;
            pass4 &= vmask;
        }
        // append the staged offsets of the passing pixels to list 1
        const int lane = tid & 63;
        const unsigned long long b0 = __ballot(pass4 & 1u), b1 = __ballot(pass4 & 2u),
                                 b2 = __ballot(pass4 & 4u), b3 = __ballot(pass4 & 8u);
        const int n0 = __builtin_popcountll(b0), n01 = n0 + __builtin_popcountll(b1),
                  n012 = n01 + __builtin_popcountll(b2), total = n012 + __builtin_popcountll(b3);
        if (total) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_n[0], (unsigned)total);
            base = __shfl(base, 0, 64);
            const unsigned long long lt = (1ull << lane) - 1ull;
            const unsigned code = (unsigned)((r + 3) * kImgPitch + 4 * cg);
            if (pass4 & 1u) s_list1[base + __builtin_popcountll(b0 & lt)] = (uint16_t)code;
            if (pass4 & 2u) s_list1[base + n0 + __builtin_popcountll(b1 & lt)] = (uint16_t)(code + 1);
            if (pass4 & 4u) s_list1[base + n01 + __builtin_popcountll(b2 & lt)] = (uint16_t)(code + 2);
            if (pass4 & 8u) s_list1[base + n012 + __builtin_popcountll(b3 & lt)] = (uint16_t)(code + 3);
        }
    }
    lds_barrier();
    // ---- phase 2: full ring, 9 contiguous darker or brighter ----
    // One subtract and one funnel shift per flag: the sign of (ring - (v - t)) says
    // "darker", the sign of ((v + t) - ring) "brighter"; alignbit shifts it into the mask.
    const int n1 = (int)s_n[0];
    for (int k0 = 0; k0 < n1; k0 += 256) {
        const int k = k0 + tid;
        bool corner = false;
        unsigned code = 0;
        if (k < n1) {
            code = s_list1[k];
            const uint8_t* p = s_img + code;
            const int v = p[0];
            const unsigned lo = (unsigned)(v - t), hi = (unsigned)(v + t);
            unsigned dark = 0, bright = 0;
            constexpr int P = kImgPitch;
            constexpr int off[16] = {3 * P, 3 * P + 1, 2 * P + 2, P + 3, 3, -P + 3, -2 * P + 2, -3 * P + 1,
                                     -3 * P, -3 * P - 1, -2 * P - 2, -P - 3, -3, P - 3, 2 * P - 2, 3 * P - 1};
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const unsigned ring = p[off[j]];
                dark = __builtin_amdgcn_alignbit(dark, ring - lo, 31);
                bright = __builtin_amdgcn_alignbit(bright, hi - ring, 31);
            }
            corner = has_run9(dark) || has_run9(bright);
        }
        list_append(corner, code, s_list2, &s_n[1]);
    }
    lds_barrier();
    // ---- phase 3: score of the corners ----
    const int n2 = (int)s_n[1];
    for (int k = tid; k < n2; k += 256) {
        const unsigned code = s_list2[k];
        int d[16];
        ring_diffs(s_img + code, d);
        s_sc[code - 3 * kImgPitch] = (uint8_t)fast_corner_score(d, t);   // same pitch, 3 rows up
    }
    lds_barrier();
    // ---- phase 4a: NMS + border filter, 4 pixels per thread ----
    // A strict 3x3 maximum has no second one among its 8 neighbours, so a 64 x 32 tile holds at
    // most 32 x 16 = kTileListCap of them: the list cannot overflow.  Its order is arbitrary
    // (the selection kernel ranks the kept entries by position).
    auto keep_survivor = [&](int lx, int ly, unsigned sc) {
        atomicAdd(&s_hist[sc], 1u);
        const unsigned slot = atomicAdd(&s_n[2], 1u);
        if (slot < (unsigned)kTileListCap) s_surv[slot] = (uint32_t)lx | ((uint32_t)ly << 6) | (sc << 16);
    };
    {
        const int tx = tid & 15, tr = tid >> 4;
        uint32_t keep[2] = {0u, 0u};   // NMS survivors of this thread's 4 pixels in rows tr and tr + 16
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int ly = tr + 16 * rr;
            const int gy = y0 + ly;
            if (gy >= h) continue;
            const uint8_t* q0 = s_sc + (ly + 1) * kScPitch + 4 + 4 * tx;
            const uint32_t own = *reinterpret_cast<const uint32_t*>(q0);
            uint32_t packed = 0;
            if (own != 0 && level_ok && gy >= edge && gy < h - edge) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int s = (own >> (8 * j)) & 255;
                    const int gx = x0 + 4 * tx + j;
                    if (s == 0 || gx < edge || gx >= w - edge) continue;
                    const uint8_t* q = q0 + j;
                    if (s > q[-1] && s > q[1] && s > q[-kScPitch - 1] && s > q[-kScPitch] &&
                        s > q[-kScPitch + 1] && s > q[kScPitch - 1] && s > q[kScPitch] && s > q[kScPitch + 1])
                        packed |= (uint32_t)s << (8 * j);
                }
            }
            keep[rr] = packed;
        }
        if (a.grid_shift) {
            // Optional grid distribution (not in the reference): of the survivors of one
            // cell of grid_cell x grid_cell pixels only the best stays -- highest score,
            // first in raster order among equals.  Cells are anchored at the level origin
            // and divide the tile, so the contest is local to the workgroup.
            const int sh = a.grid_shift;
            auto contest = [&](bool write) {
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int ly = tr + 16 * rr;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const unsigned s = (keep[rr] >> (8 * j)) & 255u;
                        if (s == 0) continue;
                        const int lx = 4 * tx + j;
                        const int cell = (ly >> sh) * (kFastTileW >> sh) + (lx >> sh);
                        const unsigned key = (s << 16) | (0xffffu - (unsigned)(ly * kFastTileW + lx));
                        if (write) atomicMax(&s_cellbest[cell], key);
                        else if (s_cellbest[cell] != key) keep[rr] &= ~(255u << (8 * j));
                    }
                }
            };
            contest(true);
            lds_barrier();      // uniform: grid_shift is a kernel argument
            contest(false);
        }
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            if (!keep[rr]) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned s = (keep[rr] >> (8 * j)) & 255u;
                if (s) keep_survivor(4 * tx + j, tr + 16 * rr, s);
            }
        }
    }
    // ---- phase 4b: blur row pass, rows y0-3 .. y0+34, packed 16-bit arithmetic ----
    // (the lists are dead: every thread passed the barrier after phase 3)
    for (int it = tid; it < kRowRows * 16; it += 256) {
        const int r = it >> 4, g = it & 15;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(s_img + (r + 1) * kImgPitch) + g;
        const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];  // bytes 4g .. 4g+11; tap p_k = byte k+1
        // P_k = (p_k, p_k+1) as two 16-bit lanes
        const us2 P0 = as_us2(__builtin_amdgcn_perm(w1, w0, 0x0c020c01));
        const us2 P1 = as_us2(__builtin_amdgcn_perm(w1, w0, 0x0c030c02));
        const us2 P2 = as_us2(__builtin_amdgcn_perm(w1, w0, 0x0c040c03));
        const us2 P3 = as_us2(__builtin_amdgcn_perm(w1, w0, 0x0c050c04));
        const us2 P4 = as_us2(__builtin_amdgcn_perm(w1, w0, 0x0c060c05));
        const us2 P5 = as_us2(__builtin_amdgcn_perm(w1, w0, 0x0c070c06));
        const us2 P6 = as_us2(__builtin_amdgcn_perm(w2, w1, 0x0c040c03));
        const us2 P7 = as_us2(__builtin_amdgcn_perm(w2, w1, 0x0c050c04));
        const us2 P8 = as_us2(__builtin_amdgcn_perm(w2, w1, 0x0c060c05));
        const us2 k18 = {18, 18}, k34 = {34, 34}, k49 = {49, 49}, k55 = {55, 55};
        const us2 o01 = (P0 + P6) * k18 + (P1 + P5) * k34 + (P2 + P4) * k49 + P3 * k55;
        const us2 o23 = (P2 + P8) * k18 + (P3 + P7) * k34 + (P4 + P6) * k49 + P5 * k55;
        s_row[r * 32 + 2 * g] = as_u32(o01);
        s_row[r * 32 + 2 * g + 1] = as_u32(o23);
    }
    lds_barrier();
    // ---- phase 5: blur column pass, 2 rows x 4 columns per thread ----
    {
        uint8_t* blur = a.blur + (size_t)b * a.g.frame_bytes + L.off;
        const int cq = tid & 15, rs = tid >> 4;
        unsigned col[8][4];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned lo = s_row[(2 * rs + r) * 32 + 2 * cq], hi = s_row[(2 * rs + r) * 32 + 2 * cq + 1];
            col[r][0] = lo & 0xffffu; col[r][1] = lo >> 16; col[r][2] = hi & 0xffffu; col[r][3] = hi >> 16;
        }
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int gy = y0 + 2 * rs + rr;
            if (gy >= h) break;
            uint32_t packed = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned acc = 18u * (col[rr][j] + col[rr + 6][j]) + 34u * (col[rr + 1][j] + col[rr + 5][j]) +
                                     49u * (col[rr + 2][j] + col[rr + 4][j]) + 55u * col[rr + 3][j] + 32768u;
                packed |= min(acc >> 16, 255u) << (8 * j);
            }
            *reinterpret_cast<uint32_t*>(blur + (size_t)gy * L.pitch + x0 + 4 * cq) = packed;
        }
    }
    // (the barrier after the blur row pass also completed the survivor list)
    {
        const unsigned ns = min(s_n[2], (unsigned)kTileListCap);
        const size_t gt = (size_t)b * a.g.total_tiles + ftile;
        if (tid == 0) a.tilecount[gt] = ns;
        uint32_t* out = a.tilelist + gt * kTileListCap;
        for (unsigned k = tid; k < ns; k += 256) out[k] = s_surv[k];
    }
    const unsigned hv = s_hist[tid];
    if (hv) atomicAdd(&a.hist[((size_t)b * a.g.nlevels + l) * 256 + tid], hv);
}

// ---------------------------------------------------------------------------
// Selection + compaction, one workgroup of 16 waves per (frame, level).
//   1. KeyPointsFilter::retainBest from the score histograms: wave w takes level
//      w (levels up to the workgroup's own), suffix sums over the 256 bins by
//      shuffles -> cut score and kept count.  The level's first output index is
//      the sum of the kept counts of the lower levels.
//   2. The kept survivors (score >= cut) of the level's tile lists are gathered into LDS as
//      keys y << 14 | x, and every one is ranked by counting the smaller keys: its rank is its
//      place in the canonical (octave, y, x) order.  The kept set of a level is about its quota
//      (a few hundred to a few thousand keys), so the quadratic count is cheap, needs no sort
//      and no atomics on the output.  A level with more kept survivors than the LDS cache
//      holds (long runs of equal scores) is ranked against the tile lists in memory instead.
// ---------------------------------------------------------------------------
constexpr int kSelTileChunk = 4096;   // tiles whose counts are held in LDS at a time
constexpr int kSelKeyCap = 4096;      // kept survivors ranked from LDS

__global__ __launch_bounds__(1024) void select_compact_kernel(SelectCompactArgs a) {
    __shared__ int s_cut[kMaxLevels], s_cnt[kMaxLevels];
    __shared__ unsigned s_wave[16];
    __shared__ unsigned s_pref[kSelTileChunk + 1];
    __shared__ __align__(16) unsigned s_key[kSelKeyCap];
    __shared__ uint8_t s_score[kSelKeyCap];
    __shared__ unsigned s_n;
    const int b = blockIdx.x / a.g.nlevels;
    const int my_level = blockIdx.x - b * a.g.nlevels;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_n = 0;
    if (wave <= my_level) {
        const int l = wave;
        const unsigned* h = a.hist + ((size_t)b * a.g.nlevels + l) * 256;
        // lane owns bins 4*lane .. 4*lane+3; suffix sums over bins
        const unsigned v0 = h[4 * lane], v1 = h[4 * lane + 1], v2 = h[4 * lane + 2], v3 = h[4 * lane + 3];
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
        int cut = 1, cnt = (int)total;
        if (total > quota) {
            if (quota == 0) {
                cut = 256; cnt = 0;
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
        if (lane == 0) { s_cut[l] = cut; s_cnt[l] = cnt; }
    }
    __syncthreads();

    const int l = my_level;
    const LevelGeom& L = a.g.lv[l];
    const unsigned cut = (unsigned)s_cut[l];
    const int kept = s_cnt[l];
    int base = 0;                         // keypoints of the lower levels
    for (int i = 0; i < l; i++) base += s_cnt[i];
    slam_amd_keypoint* kp = a.kp + (size_t)b * a.capacity;
    uint32_t* kpxy = a.kpxy + (size_t)b * a.capacity;
    const int tiles_x = L.tiles_x, ntl = tiles_x * div_up_dev(L.h, kFastTileH);
    const unsigned* tcount = a.tilecount + (size_t)b * a.g.total_tiles + L.tile_base;
    const uint32_t* tlist = a.tilelist + ((size_t)b * a.g.total_tiles + L.tile_base) * kTileListCap;
    const bool cached = kept <= a.keycap;   // uniform

    auto emit = [&](unsigned key, unsigned sc, int rank) {
        const int pos = base + rank;
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

    if (kept > 0 && cut < 256) {
        for (int chunk0 = 0; chunk0 < ntl; chunk0 += kSelTileChunk) {
            const int nt = min(kSelTileChunk, ntl - chunk0);
            // exclusive prefix of the tiles' survivor counts: four tiles per thread
            unsigned c4[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int t = tid * 4 + k;
                c4[k] = t < nt ? min(tcount[chunk0 + t], (unsigned)kTileListCap) : 0u;
                sum += c4[k];
            }
            unsigned inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = __shfl_up(inc, o, 64);
                if (lane >= o) inc += v;
            }
            __syncthreads();              // s_pref / s_wave of the previous chunk have been read
            if (lane == 63) s_wave[wave] = inc;
            __syncthreads();
            unsigned before = 0, total = 0;
            for (int w = 0; w < 16; w++) {
                const unsigned v = s_wave[w];
                if (w < wave) before += v;
                total += v;
            }
            unsigned run = before + inc - sum;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (tid * 4 + k < nt) s_pref[tid * 4 + k] = run;
                run += c4[k];
            }
            if (tid == 0) s_pref[nt] = total;
            __syncthreads();
            // every entry of the chunk's lists, flat; its tile by binary search in the prefix
            for (unsigned e = tid; e < total; e += 1024) {
                int lo = 0, hi = nt;      // largest t with s_pref[t] <= e
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_pref[mid] <= e) lo = mid; else hi = mid;
                }
                const uint32_t ent = tlist[(size_t)(chunk0 + lo) * kTileListCap + (e - s_pref[lo])];
                const unsigned sc = ent >> 16;
                if (sc < cut) continue;
                const unsigned key = key_of(chunk0 + lo, ent);
                if (cached) {
                    const unsigned slot = atomicAdd(&s_n, 1u);
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
                    emit(key, sc, rank);
                }
            }
        }
        if (cached) {
            __syncthreads();
            const int n = (int)min(s_n, (unsigned)kSelKeyCap);
            // pad to a multiple of 4 with keys larger than any real one
            if (tid < 4 && n + tid < kSelKeyCap) s_key[n + tid] = 0xffffffffu;
            __syncthreads();
            for (int i = tid; i < n; i += 1024) {
                const unsigned key = s_key[i];
                int rank = 0;
                for (int j = 0; j < n; j += 4) {
                    const uint4 k4 = *reinterpret_cast<const uint4*>(&s_key[j]);
                    rank += (k4.x < key) + (k4.y < key) + (k4.z < key) + (k4.w < key);
                }
                emit(key, s_score[i], rank);
            }
        }
    }
    // the workgroup of the top level has every level's count: it publishes the totals
    if (threadIdx.x == 0 && my_level == a.g.nlevels - 1) {
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
// Describe: one single-wave workgroup per keypoint, so everything derived from the
// keypoint index is scalar.  Two patches are staged in LDS with aligned 4-byte loads:
// 31 x 31 of the unblurred level (intensity centroid over the radius-15 disc) and
// 37 x 37 of the blurred level (256 rotated rBRIEF comparisons, 4 per lane, assembled
// by ballot; the rotated pattern reaches at most 18 pixels).  Keypoints lie >= 31
// pixels from the level border, so both patches are inside the level.
// ---------------------------------------------------------------------------
constexpr int kURad = 15, kURows = 31, kUPitch = 36;   // 31 columns + up to 3 alignment bytes
constexpr int kBRad = 18, kBRows = 37, kBPitch = 40;   // 37 columns + up to 3 alignment bytes
constexpr int kWaveLds = kURows * kUPitch + kBRows * kBPitch;  // 2596

// rBRIEF sampling pattern as floats: x0, y0, x1, y1 of pair p at [4 * p].
__constant__ __align__(16) float c_pattern[256 * 4] = {
#include "../../include/orb_pattern_31.inc"
};

// Half-width of row |v| of the radius-15 disc, 4 bits per row:
// 15 15 15 15 14 14 14 13 13 12 11 10 9 8 6 3 for |v| = 0 .. 15.  The disc is
// symmetric, so the same table gives the half-height of column |u|.
__device__ __forceinline__ int disc_half_width(int av) {
    const unsigned long long packed = 0x3689ABCDDEEEFFFFull;
    return (int)((packed >> (4 * av)) & 15u);
}

// Copies rows [y - R, y + R] x columns [x - R, x + R] of img into LDS.  Returns
// the column shift: patch column c of row r is at dst[r * PITCH + shift + c].
// A lane keeps one (row, dword) position of a block of 64 / DW rows and walks
// down the patch block by block: the loads of all blocks are issued together.
template <int R, int PITCH>
__device__ __forceinline__ int stage_patch(const uint8_t* img, int pitch, bool aligned, int x, int y,
                                           uint8_t* dst, int lane) {
    constexpr int ROWS = 2 * R + 1;
    const uint8_t* p0 = img + (size_t)(y - R) * pitch + (x - R);
    if (aligned) {
        constexpr int DW = PITCH / 4, RPI = 64 / DW, ITERS = (ROWS + RPI - 1) / RPI;
        const int shift = (int)(reinterpret_cast<uintptr_t>(p0) & 3);
        const uint8_t* base = p0 - shift;
        const int lr = lane / DW, lc = lane - lr * DW;
        const unsigned voff = (unsigned)(lr * pitch + 4 * lc);
        uint8_t* d = dst + lr * PITCH + 4 * lc;
        uint32_t v[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            v[it] = 0;
            if (lr < RPI && lr + it * RPI < ROWS)
                v[it] = *reinterpret_cast<const uint32_t*>(base + (size_t)(it * RPI) * pitch + voff);
        }
#pragma unroll
        for (int it = 0; it < ITERS; it++)
            if (lr < RPI && lr + it * RPI < ROWS)
                *reinterpret_cast<uint32_t*>(d + it * RPI * PITCH) = v[it];
        return shift;
    }
    for (int k = lane; k < ROWS * ROWS; k += 64) {
        const int r = k / ROWS, c = k - r * ROWS;
        dst[r * PITCH + c] = p0[(size_t)r * pitch + c];
    }
    return 0;
}

// Sum over the 64 lanes, returned in a scalar register: butterfly inside each row of
// 16 lanes with DPP, then the four row sums are read out.
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);  // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

__global__ __launch_bounds__(64) void describe_kernel(DescribeArgs a) {
    __shared__ __align__(16) uint8_t s_all[kWaveLds + 64];
    // 1-D grid, frame index fastest, so the workgroups beyond the last keypoint of
    // every frame (they exit at once) sit at the end of the grid.
    const int b = blockIdx.x % a.batch;
    const int i = blockIdx.x / a.batch;
    const int n = min(a.sel[b].total, a.capacity);
    if (i >= n) return;
    const int lane = threadIdx.x;
    uint8_t* s_u = s_all;
    uint8_t* s_b = s_all + kURows * kUPitch;
    // the lane's four pattern rows (pair p = 64*k + lane), fetched ahead of everything else
    float4 pat[4];
#pragma unroll
    for (int k = 0; k < 4; k++) pat[k] = reinterpret_cast<const float4*>(c_pattern)[64 * k + lane];

    slam_amd_keypoint* kp = a.kp + (size_t)b * a.capacity + i;
    const uint32_t xy = a.kpxy[(size_t)b * a.capacity + i];
    const int x = (int)(xy & 0x3fffu), y = (int)((xy >> 14) & 0x3fffu);
    const int l = (int)(xy >> 28);
    const LevelGeom& L = a.g.lv[l];
    // Both patches must lie inside the level.  The selection kernel only emits positions at least
    // `edge` (>= 31) pixels from the border, so this rejects nothing it wrote; it keeps a record that
    // was never written (zeros) or a count that ran ahead of the records from becoming a wild read.
    // (24 = patch radius 18 + the up to 3 + 3 bytes the aligned dword loads reach further.)
    if (l >= a.g.nlevels || x < 24 || y < 24 || x >= L.w - 24 || y >= L.h - 24) return;
    const uint8_t* img;
    int pitch;
    bool aligned = true;
    if (l == 0) {
        img = a.img0 + (size_t)b * a.img0_frame_pitch;
        pitch = a.img0_pitch;
        aligned = a.img0_aligned != 0;
    } else {
        img = a.pyr + (size_t)b * a.g.frame_bytes + L.off;
        pitch = L.pitch;
    }
    const uint8_t* blur = a.blur + (size_t)b * a.g.frame_bytes + L.off;
    const int su = stage_patch<kURad, kUPitch>(img, pitch, aligned, x, y, s_u, lane);
    const int sb = stage_patch<kBRad, kBPitch>(blur, L.pitch, true, x, y, s_b, lane);
    // LDS traffic of one wave executes in order; this only keeps the compiler from
    // moving the reads above the writes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- intensity centroid ----
    // Lanes 0-30 own the columns u = -15..15 for the rows v = -15..0, lanes 32-62 the
    // same columns for v = 1..15.  A lane adds up its column (rows outside the disc
    // masked off) and the row-weighted sum; the column weight is applied once.
    int m10, m01;
    {
        const int col = lane & 31, half = lane >> 5;
        const int u = col - 15;
        const int vlim = col < 31 ? disc_half_width(u < 0 ? -u : u) : -1;  // |v| <= vlim is inside
        const int vbase = half ? 1 : -15;                                  // v = vbase + vv, vv = 0..15
        const int lo = half ? 0 : 15 - vlim, hi = half ? vlim - 1 : 15;    // vv range inside the disc
        const unsigned rows = hi >= lo ? (2u << hi) - (1u << lo) : 0u;
        // column 31 (u = 16) and row v = 16 are masked off; their reads stay inside s_all
        const uint8_t* colp = s_u + (kURad + vbase) * kUPitch + su + kURad + u;
        int colsum = 0, vsum = 0;
#pragma unroll
        for (int vv = 0; vv < 16; vv++) {
            int val = colp[vv * kUPitch];
            val &= ((int)(rows << (31 - vv))) >> 31;
            colsum += val;
            vsum += vv * val;
        }
        m10 = wave_sum_i32(u * colsum);
        m01 = wave_sum_i32(vbase * colsum + vsum);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- rBRIEF: pair p = 64*k + lane, k = 0..3; ballot k holds bytes 8k..8k+7 ----
    const float rad = SA_FMUL(angle, (float)(3.14159265358979323846 / 180.0));
    float ca, sb_;
    sincos_f32(rad, &ca, &sb_);
    const uint8_t* bc = s_b + kBRad * kBPitch + sb + kBRad;
    unsigned long long bits[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ix0 = SA_RINT_I(SA_FSUB(SA_FMUL(pat[k].x, ca), SA_FMUL(pat[k].y, sb_)));
        const int iy0 = SA_RINT_I(SA_FADD(SA_FMUL(pat[k].x, sb_), SA_FMUL(pat[k].y, ca)));
        const int ix1 = SA_RINT_I(SA_FSUB(SA_FMUL(pat[k].z, ca), SA_FMUL(pat[k].w, sb_)));
        const int iy1 = SA_RINT_I(SA_FADD(SA_FMUL(pat[k].z, sb_), SA_FMUL(pat[k].w, ca)));
        const int t0 = bc[iy0 * kBPitch + ix0];
        const int t1 = bc[iy1 * kBPitch + ix1];
        bits[k] = __ballot(t0 < t1);
    }
    if (lane == 0) {
        uint4* out = reinterpret_cast<uint4*>(a.desc + ((size_t)b * a.capacity + i) * 32);
        out[0] = make_uint4((unsigned)bits[0], (unsigned)(bits[0] >> 32), (unsigned)bits[1], (unsigned)(bits[1] >> 32));
        out[1] = make_uint4((unsigned)bits[2], (unsigned)(bits[2] >> 32), (unsigned)bits[3], (unsigned)(bits[3] >> 32));
        kp->angle = angle;
        kp->x = SA_FMUL((float)x, L.scale);
        kp->y = SA_FMUL((float)y, L.scale);
    }
}

// ---------------------------------------------------------------------------
// Launchers (host)
// ---------------------------------------------------------------------------
int launch_fast(const FastArgs& a0, int batch, hipStream_t st) {
    FastArgs a = a0;
    a.tiles_all = a.g.total_tiles * batch;
    a.tiles_per_xcd = div_up(a.tiles_all, 8);
    dim3 grid(a.tiles_per_xcd * 8);
    hipLaunchKernelGGL(fast_nms_kernel, grid, dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_select_compact(const SelectCompactArgs& a, int batch, hipStream_t st) {
    hipLaunchKernelGGL(select_compact_kernel, dim3(batch * a.g.nlevels), dim3(1024), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_describe(const DescribeArgs& a, int batch, hipStream_t st) {
    DescribeArgs d = a;
    d.batch = batch;
    const long long blocks = (long long)a.capacity * batch;
    if (blocks > 0x7fffffffLL) SA_FAIL(SLAM_AMD_ERR_ARG, "describe: capacity x batch too large for one launch");
    dim3 grid((unsigned)blocks);
    hipLaunchKernelGGL(describe_kernel, grid, dim3(64), 0, st, d);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

}  // namespace sa
