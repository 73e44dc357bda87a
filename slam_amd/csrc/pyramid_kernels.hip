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

typedef unsigned short py_us2 __attribute__((ext_vector_type(2)));
// a.lo * b.lo + a.hi * b.hi + c on 16-bit halves
__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(py_us2, a), __builtin_bit_cast(py_us2, b), c, false);
}

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
    // Thread (band of destination rows, group of four destination columns) walks down the source rows
    // of its band.  Horizontal step per source row: the three dwords that hold the group's source
    // bytes, the eight bytes from the first source column on moved into two registers (v_alignbyte),
    // each pixel's two taps picked as a pair of 16-bit halves (v_perm, selector constant per thread),
    // c0 a + c1 b as one 16-bit dot product.  The exact 16-bit sums of the last two source rows stay
    // in registers; whenever they are the two rows a destination row reads, the vertical step forms
    // (m0 h0 + m1 h1 + 2^15) >> 16, again one dot product per pixel, byte 2 of which is the result.
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
        const int par = min(256 / n4, rows);            // bands side by side
        const int rr = small_div(tid, inv), gi = tid - rr * n4;
        if (rr < par) {
            // destination rows of the band
            const int ra = (rr * rows) / par, rb = ((rr + 1) * rows) / par;
            const uint4 xe = *reinterpret_cast<const uint4*>(xt + 4 * gi);
            const uint32_t e[4] = {xe.x, xe.y, xe.z, xe.w};
            const unsigned first = (e[0] & 0xffffu) - (unsigned)sx0;     // byte of the first tap in a staged row
            const unsigned shift = first & 3u;
            unsigned sel[4], wgt[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned o = (e[j] & 0xffffu) - (e[0] & 0xffffu);  // <= 6: checked when the plan is made
                sel[j] = o | 0x0c000c00u | ((o + 1u) << 16);
                const unsigned c1 = e[j] >> 16;
                wgt[j] = (256u - c1) | (c1 << 16);
            }
            const uint8_t* src = sbuf + (first & ~3u);
            const unsigned gbase = (unsigned)cy0 * (unsigned)L.pitch + (unsigned)cx0 + 4u * (unsigned)gi;   // a level is < 2^28 bytes
            const int gx = cx0 + 4 * gi;
            const bool own_x = gx >= ox0 && gx < ox1;
            int r = ra;                                                  // next destination row to write
            uint32_t ye = yt[min(r, rows - 1)];
            int want = (int)(ye & 0xffffu) - sy0 + 1;                    // it is complete with this source row
            const int s_first = want - 1;
            const int s_last = (int)(yt[max(rb, 1) - 1] & 0xffffu) - sy0 + 1;
            uint32_t hp[4] = {0u, 0u, 0u, 0u}, hc[4] = {0u, 0u, 0u, 0u};
            for (int sr = s_first; sr <= s_last && r < rb; sr++) {
                const uint32_t* p = reinterpret_cast<const uint32_t*>(src + sr * spitch);
                const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
                const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, shift), hi = __builtin_amdgcn_alignbyte(d2, d1, shift);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    hp[j] = hc[j];
                    hc[j] = dot2_u16(__builtin_amdgcn_perm(hi, lo, sel[j]), wgt[j], 0u);
                }
                // destination rows whose second source row this is (one at most while the scale is >= 1,
                // several only among the clamped rows at the edge of the region)
                while (r < rb && want == sr) {
                    const unsigned m1 = ye >> 16, mw = (256u - m1) | (m1 << 16);
                    uint32_t v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        v[j] = dot2_u16(__builtin_amdgcn_perm(hc[j], hp[j], 0x05040100u), mw, 32768u);
                    const uint32_t packed = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u) | __builtin_amdgcn_perm(v[3], v[2], 0x06020c0cu);
                    *reinterpret_cast<uint32_t*>(dbuf + (r * n4 + gi) * 4) = packed;
                    const int gy = cy0 + r;
                    if (own_x && gy >= oy0 && gy < oy1)
                        *reinterpret_cast<uint32_t*>(gdst + (gbase + __umul24((unsigned)r, (unsigned)L.pitch))) = packed;
                    r++;
                    ye = yt[min(r, rows - 1)];
                    want = (int)(ye & 0xffffu) - sy0 + 1;
                }
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
