// HIP kernels of the ORB extraction path for gfx950 (wave64, LDS-staged tiles).
//
// Per batch of frames, in stream order:
//   resize_level  x (nlevels-1)  chained bit-exact bilinear pyramid
//   fast_nms                      FAST-9/16 score + 3x3 NMS + border filter +
//                                 per-level score histogram, all levels, one launch
//   select_levels                 per-level cut score from the histogram (retain-best)
//   count_rows / compact_rows     deterministic raster-order compaction -> keypoints
//   describe                      intensity-centroid angle + 7x7 blur + rBRIEF,
//                                 one wavefront per keypoint, patch held in LDS
#include "common.h"
#include "detmath.h"
#include "kernels.h"

namespace sa {

// ---------------------------------------------------------------------------
// Pyramid: INTER_LINEAR_EXACT (8.8 fixed-point taps, two passes, round at 2^15).
// One workgroup produces a 128 x 16 destination tile; the source footprint is
// staged in LDS with 4-byte loads, then every thread blends 2 rows x 4 pixels.
// ---------------------------------------------------------------------------
constexpr int kRzTileW = 128;
constexpr int kRzTileH = 16;


__global__ __launch_bounds__(256) void resize_level_kernel(ResizeArgs a) {
    extern __shared__ __align__(16) uint8_t rz_lds[];
    const int b = blockIdx.z;
    const uint8_t* src = a.src + (size_t)b * a.src_frame_pitch;
    uint8_t* dst = a.dst + (size_t)b * a.dst_frame_pitch;
    const int dx0 = blockIdx.x * kRzTileW, dy0 = blockIdx.y * kRzTileH;
    const int dx_last = min(dx0 + kRzTileW, a.dw) - 1;
    const int dy_last = min(dy0 + kRzTileH, a.dh) - 1;
    const int xs0 = a.xoff[dx0] & ~3;
    const int xs1 = min(a.xoff[dx_last] + 1, a.sw - 1);
    const int ys0 = a.yoff[dy0];
    const int ys1 = min(a.yoff[dy_last] + 1, a.sh - 1);
    const int ncols4 = (xs1 - xs0) / 4 + 1;  // dwords per staged row
    const int nrows = ys1 - ys0 + 1;
    // Stage the source footprint in whole dwords (bytes past the image width are
    // read one by one, so nothing outside the image is touched).
    if (a.src_aligned) {
        for (int i = threadIdx.x; i < ncols4 * nrows; i += 256) {
            int r = i / ncols4, c = i - r * ncols4;
            const uint8_t* p = src + (size_t)(ys0 + r) * a.src_pitch + xs0 + 4 * c;
            uint32_t v;
            if (xs0 + 4 * c + 3 < a.sw) {
                v = *reinterpret_cast<const uint32_t*>(p);
            } else {
                v = 0;
                for (int j = 0; j < 4; j++)
                    if (xs0 + 4 * c + j < a.sw) v |= (uint32_t)p[j] << (8 * j);
            }
            *reinterpret_cast<uint32_t*>(rz_lds + r * a.lds_pitch + 4 * c) = v;
        }
    } else {
        const int ncols = xs1 - xs0 + 1;
        for (int i = threadIdx.x; i < ncols * nrows; i += 256) {
            int r = i / ncols, c = i - r * ncols;
            rz_lds[r * a.lds_pitch + c] = src[(size_t)(ys0 + r) * a.src_pitch + xs0 + c];
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x = dx0 + 4 * tx;
    if (x >= a.dw) return;
    int xo[4], w1[4], xn[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int xx = min(x + j, a.dw - 1);
        int o = a.xoff[xx];
        xo[j] = o - xs0;
        xn[j] = min(o + 1, a.sw - 1) - xs0;
        w1[j] = a.xc1[xx];
    }
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int y = dy0 + ty + 8 * rr;
        if (y >= a.dh) break;
        const int yo = a.yoff[y];
        const uint8_t* r0 = rz_lds + (yo - ys0) * a.lds_pitch;
        const uint8_t* r1 = rz_lds + (min(yo + 1, a.sh - 1) - ys0) * a.lds_pitch;
        const uint32_t m1 = a.yc1[y], m0 = 256 - m1;
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t c1 = w1[j], c0 = 256 - c1;
            uint32_t h0 = r0[xo[j]] * c0 + r0[xn[j]] * c1;
            uint32_t h1 = r1[xo[j]] * c0 + r1[xn[j]] * c1;
            uint32_t v = (h0 * m0 + h1 * m1 + 32768u) >> 16;
            packed |= v << (8 * j);
        }
        // The pitch is a multiple of 64, so the full dword is inside the row.
        *reinterpret_cast<uint32_t*>(dst + (size_t)y * a.dst_pitch + x) = packed;
    }
}

// ---------------------------------------------------------------------------
// FAST-9/16 + NMS.  A workgroup owns a 64 x 32 tile of one level of one frame.
// Image tile with a 4-pixel halo -> LDS; corner scores for the tile plus a
// 1-pixel halo -> LDS; then 3x3 strict-maximum suppression, border filter,
// survivor map (score, or 0) to HBM and a 256-bin histogram of kept scores.
// ---------------------------------------------------------------------------
constexpr int kImgHalo = 4;
constexpr int kImgPitch = kFastTileW + 2 * kImgHalo;  // 72
constexpr int kImgRows = kFastTileH + 2 * kImgHalo;   // 40
constexpr int kScPitch = kFastTileW + 4;              // 68 (66 used)
constexpr int kScRows = kFastTileH + 2;               // 34


__device__ __forceinline__ bool has_run9(unsigned m16) {
    unsigned m = m16 | (m16 << 16);
    unsigned r = m & (m >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= m >> 8;
    return (r & 0xffffu) != 0;
}

// Score of one pixel: 0 if not a corner, else max(t, best 9-arc) - 1.
__device__ __forceinline__ int fast_score_px(const uint8_t* c, int t) {
    const int P = kImgPitch;
    const int v = c[0];
    // compass points first: a 9-arc always covers at least two of them
    const int d0 = v - c[3 * P], d4 = v - c[3], d8 = v - c[-3 * P], d12 = v - c[-3];
    const int nd = (d0 > t) + (d4 > t) + (d8 > t) + (d12 > t);
    const int nb = (d0 < -t) + (d4 < -t) + (d8 < -t) + (d12 < -t);
    if (nd < 2 && nb < 2) return 0;
    int d[16];
    d[0] = d0; d[4] = d4; d[8] = d8; d[12] = d12;
    d[1] = v - c[3 * P + 1];   d[2] = v - c[2 * P + 2];   d[3] = v - c[P + 3];
    d[5] = v - c[-P + 3];      d[6] = v - c[-2 * P + 2];  d[7] = v - c[-3 * P + 1];
    d[9] = v - c[-3 * P - 1];  d[10] = v - c[-2 * P - 2]; d[11] = v - c[-P - 3];
    d[13] = v - c[P - 3];      d[14] = v - c[2 * P - 2];  d[15] = v - c[3 * P - 1];
    unsigned dark = 0, bright = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        dark |= (unsigned)(d[k] > t) << k;
        bright |= (unsigned)(d[k] < -t) << k;
    }
    if (!has_run9(dark) && !has_run9(bright)) return 0;
    // sliding minimum / maximum over every arc of 9
    int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo2[k] = min(d[k], d[(k + 1) & 15]);
        hi2[k] = max(d[k], d[(k + 1) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo4[k] = min(lo2[k], lo2[(k + 2) & 15]);
        hi4[k] = max(hi2[k], hi2[(k + 2) & 15]);
    }
    int a0 = t, b0 = 255;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
        int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
        a0 = max(a0, lo9);
        b0 = min(b0, hi9);
    }
    b0 = min(b0, -a0);
    return -b0 - 1;
}

__global__ __launch_bounds__(256) void fast_nms_kernel(FastArgs a) {
    __shared__ __align__(16) uint8_t s_img[kImgRows * kImgPitch];
    __shared__ __align__(16) uint8_t s_sc[kScRows * kScPitch];
    __shared__ unsigned s_hist[256];
    const int b = blockIdx.y;
    int l = 0;
    {
        const int tile = blockIdx.x;
#pragma unroll 1
        for (int i = 1; i < a.g.nlevels; i++)
            if (tile >= a.g.lv[i].tile_base) l = i;
    }
    const LevelGeom& L = a.g.lv[l];
    const int tile = blockIdx.x - L.tile_base;
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
    s_hist[threadIdx.x] = 0;
    // ---- stage image tile + halo (zero outside the image) ----
    if (aligned) {
        constexpr int kDw = kImgPitch / 4;  // 18 dwords per row
        for (int i = threadIdx.x; i < kDw * kImgRows; i += 256) {
            int r = i / kDw, c = i - r * kDw;
            int gy = y0 - kImgHalo + r, gx = x0 - kImgHalo + 4 * c;
            uint32_t v = 0;
            if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
                const uint8_t* p = img + (size_t)gy * pitch + gx;
                if (gx + 3 < w) {
                    v = *reinterpret_cast<const uint32_t*>(p);
                } else {
                    for (int j = 0; j < 4; j++)
                        if (gx + j < w) v |= (uint32_t)p[j] << (8 * j);
                }
            }
            *reinterpret_cast<uint32_t*>(s_img + r * kImgPitch + 4 * c) = v;
        }
    } else {
        for (int i = threadIdx.x; i < kImgPitch * kImgRows; i += 256) {
            int r = i / kImgPitch, c = i - r * kImgPitch;
            int gy = y0 - kImgHalo + r, gx = x0 - kImgHalo + c;
            uint8_t v = 0;
            if (gy >= 0 && gy < h && gx >= 0 && gx < w) v = img[(size_t)gy * pitch + gx];
            s_img[r * kImgPitch + c] = v;
        }
    }
    __syncthreads();
    // ---- scores for the tile plus a 1-pixel halo ----
    const int t = a.g.fast_threshold;
    for (int i = threadIdx.x; i < 66 * kScRows; i += 256) {
        int r = i / 66, c = i - r * 66;
        int gx = x0 - 1 + c, gy = y0 - 1 + r;
        int s = 0;
        if (gx >= 3 && gx < w - 3 && gy >= 3 && gy < h - 3)
            s = fast_score_px(s_img + (r + 3) * kImgPitch + (c + 3), t);
        s_sc[r * kScPitch + c] = (uint8_t)s;
    }
    __syncthreads();
    // ---- NMS + border filter, 4 pixels per thread ----
    uint8_t* smap = a.smap + (size_t)b * a.g.frame_bytes + L.off;
    const int edge = a.g.edge;
    const bool level_ok = (w > 2 * edge) && (h > 2 * edge);
    const int tx = threadIdx.x & 15, tr = threadIdx.x >> 4;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int ly = tr + 16 * rr;
        const int gy = y0 + ly;
        if (gy >= h) continue;
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int lx = 4 * tx + j;
            const int gx = x0 + lx;
            const uint8_t* q = s_sc + (ly + 1) * kScPitch + (lx + 1);
            const int s = q[0];
            bool keep = s > 0 && level_ok && gx >= edge && gx < w - edge && gy >= edge && gy < h - edge;
            if (keep) {
                keep = s > q[-1] && s > q[1] && s > q[-kScPitch - 1] && s > q[-kScPitch] &&
                       s > q[-kScPitch + 1] && s > q[kScPitch - 1] && s > q[kScPitch] &&
                       s > q[kScPitch + 1];
            }
            if (keep) {
                packed |= (uint32_t)s << (8 * j);
                atomicAdd(&s_hist[s], 1u);
            }
        }
        *reinterpret_cast<uint32_t*>(smap + (size_t)gy * L.pitch + x0 + 4 * tx) = packed;
    }
    __syncthreads();
    const unsigned hv = s_hist[threadIdx.x];
    if (hv) atomicAdd(&a.hist[((size_t)b * a.g.nlevels + l) * 256 + threadIdx.x], hv);
}

// ---------------------------------------------------------------------------
// Selection: KeyPointsFilter::retainBest per level, from the score histogram.
// One workgroup per frame; wave k handles levels k, k+4, ...
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void select_levels_kernel(SelectArgs a) {
    __shared__ int s_cut[kMaxLevels], s_cnt[kMaxLevels];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int l = wave; l < a.g.nlevels; l += 4) {
        const unsigned* h = a.hist + ((size_t)b * a.g.nlevels + l) * 256;
        // lane owns bins 4*lane .. 4*lane+3; suffix sums over bins
        unsigned v0 = h[4 * lane], v1 = h[4 * lane + 1], v2 = h[4 * lane + 2], v3 = h[4 * lane + 3];
        unsigned own = v0 + v1 + v2 + v3;
        // inclusive suffix scan over lanes (sum of lanes >= lane)
        unsigned suf = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned n = __shfl_down(suf, o, 64);
            if (lane + o < 64) suf += n;
        }
        const unsigned above = suf - own;  // bins of higher lanes
        const unsigned s3 = above + v3, s2 = s3 + v2, s1 = s2 + v1, s0 = s1 + v0;
        const unsigned total = __shfl(suf, 0, 64);
        const unsigned quota = (unsigned)a.g.lv[l].quota;
        int cut = 1, cnt = (int)total;
        if (total > quota) {
            if (quota == 0) {
                cut = 256; cnt = 0;
            } else {
                // largest bin c with suffix(c) >= quota
                int best = -1; unsigned bc = 0;
                if (s3 >= quota) { best = 4 * lane + 3; bc = s3; }
                else if (s2 >= quota) { best = 4 * lane + 2; bc = s2; }
                else if (s1 >= quota) { best = 4 * lane + 1; bc = s1; }
                else if (s0 >= quota) { best = 4 * lane; bc = s0; }
                unsigned long long m = __ballot(best >= 0);
                int top = 63 - __builtin_clzll(m);  // m != 0 because total > quota >= 1
                cut = __shfl(best, top, 64);
                cnt = (int)__shfl(bc, top, 64);
            }
        }
        if (lane == 0) { s_cut[l] = cut; s_cnt[l] = cnt; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        FrameSelect fs;
        int acc = 0;
        for (int l = 0; l < kMaxLevels; l++) {
            bool on = l < a.g.nlevels;
            fs.cut[l] = on ? s_cut[l] : 256;
            fs.count[l] = on ? s_cnt[l] : 0;
            fs.offset[l] = acc;
            acc += fs.count[l];
        }
        fs.total = acc;
        fs.pad[0] = fs.pad[1] = fs.pad[2] = 0;
        a.sel[b] = fs;
        if (a.d_count) a.d_count[b] = acc;
    }
}

// ---------------------------------------------------------------------------
// Raster-order compaction of the survivor map, one wavefront per image row.
// ---------------------------------------------------------------------------

__device__ __forceinline__ int level_of_row(const PyrGeom& g, int row) {
    int l = 0;
#pragma unroll 1
    for (int i = 1; i < g.nlevels; i++)
        if (row >= g.lv[i].row_base) l = i;
    return l;
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void count_rows_kernel(CompactArgs a) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.g.total_rows) return;
    const int l = level_of_row(a.g, row);
    const LevelGeom& L = a.g.lv[l];
    const int y = row - L.row_base;
    const unsigned cut = (unsigned)a.sel[b].cut[l];
    unsigned cnt = 0;
    const int edge = a.g.edge;
    if (cut < 256 && y >= edge && y < L.h - edge) {
        const uint8_t* r = a.smap + (size_t)b * a.g.frame_bytes + L.off + (size_t)y * L.pitch;
        for (int x = 4 * lane; x < L.pitch; x += 256) {
            uint32_t v = *reinterpret_cast<const uint32_t*>(r + x);
#pragma unroll
            for (int j = 0; j < 4; j++) cnt += ((v >> (8 * j)) & 255u) >= cut;
        }
    }
    cnt = wave_sum(cnt);
    if (lane == 0) a.rowcnt[(size_t)b * a.g.total_rows + row] = cnt;
}

__global__ __launch_bounds__(256) void compact_rows_kernel(CompactArgs a) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.g.total_rows) return;
    const unsigned* rc = a.rowcnt + (size_t)b * a.g.total_rows;
    if (rc[row] == 0) return;
    const int l = level_of_row(a.g, row);
    const LevelGeom& L = a.g.lv[l];
    const int y = row - L.row_base;
    const unsigned cut = (unsigned)a.sel[b].cut[l];
    // keypoints of earlier rows of this level
    unsigned before = 0;
    for (int r = lane; r < y; r += 64) before += rc[L.row_base + r];
    before = wave_sum(before);
    unsigned base = (unsigned)a.sel[b].offset[l] + before;
    const uint8_t* r = a.smap + (size_t)b * a.g.frame_bytes + L.off + (size_t)y * L.pitch;
    slam_amd_keypoint* kp = a.kp + (size_t)b * a.capacity;
    uint32_t* kpxy = a.kpxy + (size_t)b * a.capacity;
    for (int xb = 0; xb < L.pitch; xb += 256) {
        const int x = xb + 4 * lane;
        uint32_t v = 0;
        if (x < L.pitch) v = *reinterpret_cast<const uint32_t*>(r + x);
        unsigned c = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) c += ((v >> (8 * j)) & 255u) >= cut;
        // exclusive prefix of c over lanes
        unsigned inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned n = __shfl_up(inc, o, 64);
            if (lane >= o) inc += n;
        }
        unsigned idx = base + inc - c;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned s = (v >> (8 * j)) & 255u;
            if (s >= cut) {
                if (idx < (unsigned)a.capacity) {
                    slam_amd_keypoint k;
                    k.x = (float)(x + j);
                    k.y = (float)y;
                    k.size = L.size;
                    k.angle = -1.f;
                    k.response = (float)s;
                    k.octave = l;
                    k.class_id = -1;
                    kp[idx] = k;
                    kpxy[idx] = (uint32_t)(x + j) | ((uint32_t)y << 16);
                }
                idx++;
            }
        }
        base += __shfl(inc, 63, 64);
    }
}

// ---------------------------------------------------------------------------
// Describe: one wavefront per keypoint.  The 43 x 43 unblurred patch around the
// keypoint is staged in the wave's LDS slice; from it
//   - the intensity-centroid moments over the radius-15 disc (wave reduction),
//   - the 7x7 fixed-point Gaussian of the inner 37 x 37 (row pass, column pass),
//   - the 256 rotated rBRIEF comparisons (4 per lane, assembled by ballot).
// Keypoints lie >= 31 pixels from the level border, so the patch is always
// inside the level and the blur never touches the image border.
// ---------------------------------------------------------------------------
constexpr int kPatchR = 21;                 // 18 (max rotated pattern reach) + 3 (blur taps)
constexpr int kPatchD = 2 * kPatchR + 1;    // 43
constexpr int kPatchPitch = 44;
constexpr int kBlurR = 18;
constexpr int kBlurD = 2 * kBlurR + 1;      // 37
constexpr int kRowPitch = 38;               // uint16 row-pass buffer: 43 rows x 37 (+1) columns
constexpr int kBlurPitch = 40;
constexpr int kWaveLds = kPatchD * kPatchPitch + kPatchD * kRowPitch * 2 + kBlurD * kBlurPitch + 12;  // 6652 -> keep 4-aligned


__constant__ int8_t c_pattern[256 * 4] = {
#include "orb_pattern_31.inc"
};

__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__global__ __launch_bounds__(256) void describe_kernel(DescribeArgs a) {
    __shared__ __align__(16) uint8_t s_all[4 * kWaveLds];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    const int n = min(a.sel[b].total, a.capacity);
    const bool active = i < n;
    uint8_t* s_patch = s_all + wave * kWaveLds;
    uint16_t* s_row = reinterpret_cast<uint16_t*>(s_patch + kPatchD * kPatchPitch);
    uint8_t* s_blur = reinterpret_cast<uint8_t*>(s_row + kPatchD * kRowPitch);

    slam_amd_keypoint* kp = a.kp + (size_t)b * a.capacity + i;
    int x = 0, y = 0, l = 0;
    const uint8_t* img = nullptr;
    int pitch = 0;
    if (active) {
        const uint32_t xy = a.kpxy[(size_t)b * a.capacity + i];
        x = (int)(xy & 0xffffu); y = (int)(xy >> 16);
        l = kp->octave;
        if (l == 0) {
            img = a.img0 + (size_t)b * a.img0_frame_pitch;
            pitch = a.img0_pitch;
        } else {
            img = a.pyr + (size_t)b * a.g.frame_bytes + a.g.lv[l].off;
            pitch = a.g.lv[l].pitch;
        }
        // ---- stage the patch ----
        const uint8_t* p0 = img + (size_t)(y - kPatchR) * pitch + (x - kPatchR);
        for (int k = lane; k < kPatchD * kPatchD; k += 64) {
            int r = k / kPatchD, c = k - r * kPatchD;
            s_patch[r * kPatchPitch + c] = p0[(size_t)r * pitch + c];
        }
    }
    __syncthreads();
    float angle = 0.f;
    if (active) {
        // ---- intensity centroid: two disc rows per step (lanes 0-30 / 32-62) ----
        const uint8_t* ctr = s_patch + kPatchR * kPatchPitch + kPatchR;
        int m10 = 0, m01 = 0;
        const int half = lane >> 5, u = (lane & 31) - 15;
#pragma unroll 1
        for (int vv = 0; vv < 16; vv++) {
            // rows: half 0 -> v = -15 + vv (v in [-15, 0]); half 1 -> v = 1 + vv (v in [1, 16))
            const int v = half ? (1 + vv) : (-15 + vv);
            const int av = v < 0 ? -v : v;
            if (av <= 15 && (lane & 31) < 31) {
                if (u >= -c_umax[av] && u <= c_umax[av]) {
                    int val = ctr[v * kPatchPitch + u];
                    m10 += u * val;
                    m01 += v * val;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m10 += __shfl_xor(m10, o, 64);
            m01 += __shfl_xor(m01, o, 64);
        }
        angle = fast_atan2_deg((float)m01, (float)m10);
        // ---- blur, row pass: 43 rows x 37 columns ----
        for (int k = lane; k < kPatchD * kBlurD; k += 64) {
            int r = k / kBlurD, c = k - r * kBlurD;
            const uint8_t* q = s_patch + r * kPatchPitch + c;  // columns c .. c+6
            unsigned acc = 18u * (q[0] + q[6]) + 34u * (q[1] + q[5]) + 49u * (q[2] + q[4]) + 55u * q[3];
            s_row[r * kRowPitch + c] = (uint16_t)acc;
        }
    }
    __syncthreads();
    if (active) {
        // ---- blur, column pass: 37 x 37 ----
        for (int k = lane; k < kBlurD * kBlurD; k += 64) {
            int r = k / kBlurD, c = k - r * kBlurD;
            const uint16_t* q = s_row + r * kRowPitch + c;  // rows r .. r+6
            unsigned acc = 18u * (q[0] + q[6 * kRowPitch]) + 34u * (q[kRowPitch] + q[5 * kRowPitch]) +
                           49u * (q[2 * kRowPitch] + q[4 * kRowPitch]) + 55u * q[3 * kRowPitch];
            unsigned v = (acc + 32768u) >> 16;
            s_blur[r * kBlurPitch + c] = (uint8_t)min(v, 255u);
        }
    }
    __syncthreads();
    if (active) {
        // ---- rBRIEF: pair p = 64*k + lane, k = 0..3; ballot k holds bytes 8k..8k+7 ----
        const float rad = SA_FMUL(angle, (float)(3.14159265358979323846 / 180.0));
        float ca, sb;
        sincos_f32(rad, &ca, &sb);
        const uint8_t* bc = s_blur + kBlurR * kBlurPitch + kBlurR;
        unsigned long long* out = reinterpret_cast<unsigned long long*>(
            a.desc + ((size_t)b * a.capacity + i) * 32);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int8_t* pp = c_pattern + (64 * k + lane) * 4;
            const float px0 = (float)pp[0], py0 = (float)pp[1], px1 = (float)pp[2], py1 = (float)pp[3];
            const int ix0 = SA_RINT_I(SA_FSUB(SA_FMUL(px0, ca), SA_FMUL(py0, sb)));
            const int iy0 = SA_RINT_I(SA_FADD(SA_FMUL(px0, sb), SA_FMUL(py0, ca)));
            const int ix1 = SA_RINT_I(SA_FSUB(SA_FMUL(px1, ca), SA_FMUL(py1, sb)));
            const int iy1 = SA_RINT_I(SA_FADD(SA_FMUL(px1, sb), SA_FMUL(py1, ca)));
            const int t0 = bc[iy0 * kBlurPitch + ix0];
            const int t1 = bc[iy1 * kBlurPitch + ix1];
            const unsigned long long bits = __ballot(t0 < t1);
            if (lane == 0) out[k] = bits;
        }
        if (lane == 0) {
            const float sc = a.g.lv[l].scale;
            kp->angle = angle;
            kp->x = SA_FMUL((float)x, sc);
            kp->y = SA_FMUL((float)y, sc);
        }
    }
}

// ---------------------------------------------------------------------------
// Launchers (host)
// ---------------------------------------------------------------------------
int launch_resize(const ResizeArgs& a, int batch, hipStream_t st) {
    dim3 grid(div_up(a.dw, kRzTileW), div_up(a.dh, kRzTileH), batch);
    size_t lds = (size_t)a.lds_pitch * a.lds_rows;
    hipLaunchKernelGGL(resize_level_kernel, grid, dim3(256), lds, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_fast(const FastArgs& a, int batch, hipStream_t st) {
    dim3 grid(a.g.total_tiles, batch);
    hipLaunchKernelGGL(fast_nms_kernel, grid, dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_select(const SelectArgs& a, int batch, hipStream_t st) {
    hipLaunchKernelGGL(select_levels_kernel, dim3(batch), dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_count_rows(const CompactArgs& a, int batch, hipStream_t st) {
    dim3 grid(div_up(a.g.total_rows, 4), batch);
    hipLaunchKernelGGL(count_rows_kernel, grid, dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_compact_rows(const CompactArgs& a, int batch, hipStream_t st) {
    dim3 grid(div_up(a.g.total_rows, 4), batch);
    hipLaunchKernelGGL(compact_rows_kernel, grid, dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_describe(const DescribeArgs& a, int batch, hipStream_t st) {
    dim3 grid(div_up(a.capacity, 4), batch);
    hipLaunchKernelGGL(describe_kernel, grid, dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

}  // namespace sa
