// Image pyramid for gfx950: chained INTER_LINEAR_EXACT levels, several levels per launch.
//
// Level l is the bit-exact bilinear resize of level l - 1 (8.8 fixed-point taps, two passes,
// one rounding at 2^15), so the levels form a dependency chain.  Instead of one launch per
// level, a workgroup owns one tile of the LAST level of a group and produces, for every level
// of the group, the part of that level it owns plus the few extra rows and columns the next
// level's part needs (recomputed by the neighbours too).  The chain lives in two LDS buffers
// that alternate as source and destination; the source image is read from memory once, every
// level is written once.  All per-axis resize coefficients are computed on the host in double
// (as OpenCV's soft-float does) and arrive as tables of (source index | weight << 16).
#include "common.h"
#include "kernels.h"

namespace sa {

// floor(x / d) for 0 <= x < 256, 1 <= d < 256, by one multiplication.
__device__ __forceinline__ int small_div(int x, int d_inv16) { return (x * d_inv16) >> 16; }

__global__ __launch_bounds__(256) void pyramid_kernel(PyramidArgs a) {
    extern __shared__ __align__(16) uint8_t py_lds[];
    const int slots = a.last - a.first + 2;          // slot 0 = source level (first - 1), slot k = level first + k - 1
    const int ntiles = a.ntx * a.nty;
    const int b = blockIdx.x / ntiles, tile = blockIdx.x - b * ntiles;
    const int tj = tile / a.ntx, ti = tile - tj * a.ntx;
    const PyrSpan* sx = a.spanx + ti * slots;
    const PyrSpan* sy = a.spany + tj * slots;
    uint32_t* s_xt = reinterpret_cast<uint32_t*>(py_lds);                 // [slots - 1][tabcap]
    uint32_t* s_yt = s_xt + (slots - 1) * a.tabcap;                       // [slots - 1][tabcap]
    uint8_t* buf0 = reinterpret_cast<uint8_t*>(s_yt + (slots - 1) * a.tabcap);
    uint8_t* buf1 = buf0 + a.buf_even;
    const int tid = threadIdx.x;

    // ---- coefficient slices of every level of the group.  A slice covers the computed range,
    //      which is the needed range widened to whole dwords; the extra positions are never used,
    //      so their source indices are clamped into the source range instead of widening that too ----
    //      (one wave per level, so that the levels' loads are in flight together)
    for (int k = 1 + (tid >> 6); k < slots; k += 4) {
        const int l = a.first + k - 1;
        const int nx = sx[k].c1 - sx[k].c0, ny = sy[k].c1 - sy[k].c0;
        const unsigned xlo = sx[k - 1].c0, xhi = sx[k - 1].c1 - 2, ylo = sy[k - 1].c0, yhi = sy[k - 1].c1 - 2;
        for (int i = tid & 63; i < nx; i += 64) {
            const uint32_t e = a.xtab[l][sx[k].c0 + i];
            s_xt[(k - 1) * a.tabcap + i] = (e & 0xffff0000u) | min(max(e & 0xffffu, xlo), xhi);
        }
        for (int i = tid & 63; i < ny; i += 64) {
            const uint32_t e = a.ytab[l][sy[k].c0 + i];
            s_yt[(k - 1) * a.tabcap + i] = (e & 0xffff0000u) | min(max(e & 0xffffu, ylo), yhi);
        }
    }
    // ---- stage the source region (slot 0); columns / rows past the image stay unwritten or zero:
    //      they are only ever read with weight 0 ----
    {
        const int ls = a.first - 1;
        const uint8_t* src;
        int pitch, aligned;
        if (ls == 0) {
            src = a.src0 + (size_t)b * a.src0_frame_pitch; pitch = a.src0_pitch; aligned = a.src0_aligned;
        } else {
            src = a.pyr + (size_t)b * a.g.frame_bytes + a.g.lv[ls].off; pitch = a.g.lv[ls].pitch; aligned = 1;
        }
        const int w = a.g.lv[ls].w, h = a.g.lv[ls].h;
        const int x0 = sx[0].c0, n4 = (sx[0].c1 - sx[0].c0 + 3) >> 2;
        const int y0 = sy[0].c0, rows = min(sy[0].c1, h) - y0;
        const int inv = 65536 / n4 + 1;
        const int q = 256 / n4, r = 256 - q * n4;
        int row = small_div(tid, inv), c = tid - row * n4;
        while (row < rows) {
            // four dwords per trip, loads first
            constexpr int kUnroll = 4;
            int li[kUnroll];
            uint32_t v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                li[u] = row < rows ? row * n4 + c : -1;
                v[u] = 0;
                if (row < rows) {
                    const int x = x0 + 4 * c;
                    const uint8_t* p = src + (size_t)(y0 + row) * pitch + x;
                    if (aligned && x + 3 < w) {
                        v[u] = *reinterpret_cast<const uint32_t*>(p);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (x + j < w) v[u] |= (uint32_t)p[j] << (8 * j);
                    }
                }
                row += q; c += r;
                if (c >= n4) { c -= n4; row++; }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++)
                if (li[u] >= 0) *reinterpret_cast<uint32_t*>(buf0 + li[u] * 4) = v[u];
        }
    }
    __syncthreads();

    // ---- the chain ----
    for (int k = 1; k < slots; k++) {
        const int l = a.first + k - 1;
        const LevelGeom& L = a.g.lv[l];
        const uint8_t* sbuf = (k & 1) ? buf0 : buf1;
        uint8_t* dbuf = (k & 1) ? buf1 : buf0;
        const int spitch = ((sx[k - 1].c1 - sx[k - 1].c0 + 3) >> 2) << 2;
        const int sx0 = sx[k - 1].c0, sy0 = sy[k - 1].c0;
        const int cx0 = sx[k].c0, cy0 = sy[k].c0;
        const int n4 = (sx[k].c1 - cx0 + 3) >> 2, rows = sy[k].c1 - cy0;
        const int ox0 = sx[k].o0, ox1 = sx[k].o1, oy0 = sy[k].o0, oy1 = sy[k].o1;
        uint8_t* gdst = a.pyr + (size_t)b * a.g.frame_bytes + L.off;
        const uint32_t* xt = s_xt + (k - 1) * a.tabcap;
        const uint32_t* yt = s_yt + (k - 1) * a.tabcap;
        const int inv = 65536 / n4 + 1;
        const int q = 256 / n4, r = 256 - q * n4;
        int row = small_div(tid, inv), c = tid - row * n4;
        // kUnroll items per trip: every LDS read of the trip is issued before the first result is
        // stored, so the reads of one item overlap the arithmetic of another
        constexpr int kUnroll = 4;
        const unsigned gbase = (unsigned)cy0 * (unsigned)L.pitch + (unsigned)cx0;   // a level is < 2^28 bytes
        while (row < rows) {
            int irow[kUnroll], icol[kUnroll];
            uint4 xe[kUnroll];
            uint32_t ye[kUnroll];
            uint32_t t0[kUnroll][4][2], t1[kUnroll][4][2];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                irow[u] = row; icol[u] = c;
                row += q; c += r;
                if (c >= n4) { c -= n4; row++; }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const bool on = irow[u] < rows;
                xe[u] = *reinterpret_cast<const uint4*>(xt + 4 * (on ? icol[u] : 0));
                ye[u] = yt[on ? irow[u] : 0];
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint8_t* r0 = sbuf + __umul24((ye[u] & 0xffffu) - (unsigned)sy0, (unsigned)spitch) - sx0;
                const uint8_t* r1 = r0 + spitch;
                const uint32_t e[4] = {xe[u].x, xe[u].y, xe[u].z, xe[u].w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // (byte reads: a 2-byte read at an odd address is legal on gfx950 but several times slower)
                    const uint32_t xo = e[j] & 0xffffu;
                    t0[u][j][0] = r0[xo]; t0[u][j][1] = r0[xo + 1];
                    t1[u][j][0] = r1[xo]; t1[u][j][1] = r1[xo + 1];
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                if (irow[u] >= rows) continue;
                const uint32_t m1 = ye[u] >> 16, m0 = 256u - m1;
                const uint32_t e[4] = {xe[u].x, xe[u].y, xe[u].z, xe[u].w};
                uint32_t packed = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t c1 = e[j] >> 16, c0 = 256u - c1;
                    const uint32_t h0 = __umul24(t0[u][j][0], c0) + __umul24(t0[u][j][1], c1);
                    const uint32_t h1 = __umul24(t1[u][j][0], c0) + __umul24(t1[u][j][1], c1);
                    const uint32_t v = (__umul24(h0, m0) + __umul24(h1, m1) + 32768u) >> 16;
                    packed |= v << (8 * j);
                }
                const unsigned li = __umul24((unsigned)irow[u], (unsigned)n4) + (unsigned)icol[u];
                *reinterpret_cast<uint32_t*>(dbuf + li * 4) = packed;
                const int gy = cy0 + irow[u], gx = cx0 + 4 * icol[u];
                if (gy >= oy0 && gy < oy1 && gx >= ox0 && gx < ox1)
                    *reinterpret_cast<uint32_t*>(gdst + (gbase + __umul24((unsigned)irow[u], (unsigned)L.pitch) + 4u * (unsigned)icol[u])) = packed;
            }
        }
        lds_barrier();
    }
}

int launch_pyramid(const PyramidArgs& a, int batch, size_t lds_bytes, hipStream_t st) {
    const long long blocks = (long long)a.ntx * a.nty * batch;
    if (blocks <= 0) return SLAM_AMD_OK;
    if (blocks > 0x7fffffffLL) SA_FAIL(SLAM_AMD_ERR_ARG, "pyramid: too many tiles for one launch");
    hipLaunchKernelGGL(pyramid_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

}  // namespace sa
