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

// build-time experiment switch (1: staging only, 2: chain without global stores); 0 in every shipped build
#ifndef SA_PYR_EXP
#define SA_PYR_EXP 0
#endif

namespace sa {

typedef unsigned short py_us2 __attribute__((ext_vector_type(2)));
// a.lo * b.lo + a.hi * b.hi + c on 16-bit halves
__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(py_us2, a), __builtin_bit_cast(py_us2, b), c, false);
}

// floor(x / d) for 0 <= x < 256, 1 <= d < 256, by one multiplication.
__device__ __forceinline__ int small_div(int x, int d_inv16) { return (x * d_inv16) >> 16; }

#ifndef SA_PYR_STAMPS
#define SA_PYR_STAMPS 0   // 1: prints the clock at the kernel's step boundaries for a few workgroups (timing experiments)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void pyramid_kernel(PyramidArgs a) {
    extern __shared__ __align__(16) uint8_t py_lds[];
#if SA_PYR_STAMPS
    unsigned long long pst[12] = {};
    int psn = 0;
    pst[psn++] = __builtin_amdgcn_s_memtime();
#endif
    const int slots = a.last - a.first + 2;          // slot 0 = source level (first - 1), slot k = level first + k - 1
    const int ntiles = a.ntx * a.nty;
    const int b = blockIdx.x / ntiles, tile = blockIdx.x - b * ntiles;
    const int tj = tile / a.ntx, ti = tile - tj * a.ntx;
    const PyrSpan* sx = a.spanx + ti * slots;
    const PyrSpan* sy = a.spany + tj * slots;
    // coefficient slices, packed: the x slice of every level (source column | weight << 16, multiples of
    // four entries), then the y slices (weight | source row within the staged region << 9, 16 bits each)
    uint32_t* s_xt = reinterpret_cast<uint32_t*>(py_lds);
    int x_entries = 0;
    for (int k = 1; k < slots; k++) x_entries += sx[k].c1 - sx[k].c0;
    uint16_t* s_yt = reinterpret_cast<uint16_t*>(s_xt + x_entries);
    uint8_t* buf0 = py_lds + a.tab_bytes;
    uint8_t* buf1 = buf0 + a.buf_even;
    const int tid = threadIdx.x;

    // ---- stage the source region (slot 0); columns / rows past the image stay unwritten or zero:
    //      they are only ever read with weight 0.  All loads of a thread are issued first (up to
    //      kStage dwords per trip; a 32-pixel tile needs one trip), the coefficient slices are
    //      fetched while they are in flight, then the dwords go to LDS ----
#ifndef SA_PYR_STAGE
#define SA_PYR_STAGE 16
#endif
    constexpr int kStage = SA_PYR_STAGE;
    const int ls0 = a.first - 1;
    const uint8_t* src_img;
    int src_pitch, src_aligned;
    if (ls0 == 0) {
        src_img = a.src0 + (size_t)b * a.src0_frame_pitch; src_pitch = a.src0_pitch; src_aligned = a.src0_aligned;
    } else {
        src_img = a.pyr + (size_t)b * a.g.frame_bytes + a.g.lv[ls0].off; src_pitch = a.g.lv[ls0].pitch; src_aligned = 1;
    }
    const int w0 = a.g.lv[ls0].w, h0 = a.g.lv[ls0].h;
    const int st_x0 = sx[0].c0, st_n4 = (sx[0].c1 - sx[0].c0 + 3) >> 2;
    const int st_y0 = sy[0].c0, st_rows = min(sy[0].c1, h0) - st_y0;
    const int st_inv = 65536 / st_n4 + 1;
    const int st_q = 256 / st_n4, st_r = 256 - st_q * st_n4;
    int st_row = small_div(tid, st_inv), st_c = tid - st_row * st_n4;
    int li[kStage];
    uint32_t sv[kStage];
    auto stage_loads = [&]() {
#pragma unroll
        for (int u = 0; u < kStage; u++) {
            li[u] = st_row < st_rows ? st_row * st_n4 + st_c : -1;
            sv[u] = 0;
            if (st_row < st_rows) {
                const int x = st_x0 + 4 * st_c;
                const uint8_t* p = src_img + (size_t)(st_y0 + st_row) * src_pitch + x;
                if (src_aligned && x + 3 < w0) {
                    sv[u] = *reinterpret_cast<const uint32_t*>(p);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (x + j < w0) sv[u] |= (uint32_t)p[j] << (8 * j);
                }
            }
            st_row += st_q; st_c += st_r;
            if (st_c >= st_n4) { st_c -= st_n4; st_row++; }
        }
    };
    auto stage_stores = [&]() {
#pragma unroll
        for (int u = 0; u < kStage; u++)
            if (li[u] >= 0) *reinterpret_cast<uint32_t*>(buf0 + li[u] * 4) = sv[u];
    };
    stage_loads();
#if SA_PYR_STAMPS
    unsigned long long sst[4] = {};
    sst[0] = __builtin_amdgcn_s_memtime();
#endif
    // ---- coefficient slices of every level of the group.  A slice covers the computed range,
    //      which is the needed range widened to whole dwords; the extra positions are never used,
    //      so their source indices are clamped into the source range instead of widening that too ----
    //      (one wave per level, so that the levels' loads are in flight together)
    for (int k = 1 + (tid >> 6); k < slots; k += 4) {
        const int l = a.first + k - 1;
        const int nx = sx[k].c1 - sx[k].c0, ny = sy[k].c1 - sy[k].c0;
        int xat = 0, yat = 0;
        for (int j = 1; j < k; j++) { xat += sx[j].c1 - sx[j].c0; yat += sy[j].c1 - sy[j].c0; }
        const unsigned xlo = sx[k - 1].c0, xhi = sx[k - 1].c1 - 2, ylo = sy[k - 1].c0, yhi = sy[k - 1].c1 - 2;
        for (int i = tid & 63; i < nx; i += 64) {
            const uint32_t e = a.xtab[l][sx[k].c0 + i];
            s_xt[xat + i] = (e & 0xffff0000u) | min(max(e & 0xffffu, xlo), xhi);
        }
        for (int i = tid & 63; i < ny; i += 64) {
            const uint32_t e = a.ytab[l][sy[k].c0 + i];
            s_yt[yat + i] = (uint16_t)((e >> 16) | ((min(max(e & 0xffffu, ylo), yhi) - ylo) << 9));
        }
    }
#if SA_PYR_STAMPS
    sst[1] = __builtin_amdgcn_s_memtime();
#endif
    stage_stores();
#if SA_PYR_STAMPS
    sst[2] = __builtin_amdgcn_s_memtime();
#endif
    while (st_row < st_rows) {     // larger source regions: further trips
        stage_loads();
        stage_stores();
    }
#if SA_PYR_STAMPS
    sst[3] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#if SA_PYR_EXP == 1
    return;
#endif
#if SA_PYR_STAMPS
    pst[psn++] = __builtin_amdgcn_s_memtime();
#endif

    // ---- the chain ----
    // Thread (destination row, group of four destination columns); the group is fixed per thread and
    // level, the rows are taken four at a time so that all LDS reads of a trip are in flight before
    // the first result is needed.  Horizontal step for each of the row's two source rows: the three
    // dwords that hold the group's source bytes, the eight bytes from the first source column on moved
    // into two registers (v_alignbyte), each pixel's two taps picked as a pair of 16-bit halves (v_perm,
    // selector constant per thread), c0 a + c1 b as one 16-bit dot product -- exact in 16 bits.
    // Vertical step: (m0 h0 + m1 h1 + 2^15) >> 16, again one dot product per pixel, byte 2 of which
    // is the result.
    const uint32_t* xt = s_xt;
    const uint16_t* yt = s_yt;
    for (int k = 1; k < slots; k++) {
        const int l = a.first + k - 1;
        const LevelGeom& L = a.g.lv[l];
        const uint8_t* sbuf = (k & 1) ? buf0 : buf1;
        uint8_t* dbuf = (k & 1) ? buf1 : buf0;
        const int spitch = ((sx[k - 1].c1 - sx[k - 1].c0 + 3) >> 2) << 2;
        const int sx0 = sx[k - 1].c0;
        const int cx0 = sx[k].c0, cy0 = sy[k].c0;
        const int n4 = (sx[k].c1 - cx0 + 3) >> 2, rows = sy[k].c1 - cy0;
        const int ox0 = sx[k].o0, ox1 = sx[k].o1, oy0 = sy[k].o0, oy1 = sy[k].o1;
        uint8_t* gdst = a.pyr + (size_t)b * a.g.frame_bytes + L.off;
        const int inv = 65536 / n4 + 1;
        const int par = 256 / n4;                       // rows side by side
        const int rr = small_div(tid, inv), gi = tid - rr * n4;
        if (rr < par) {
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
            constexpr int kUnroll = 4;
            for (int r = rr; r < rows; r += kUnroll * par) {
                uint32_t ye[kUnroll], d[kUnroll][2][3];
#pragma unroll
                for (int u = 0; u < kUnroll; u++) ye[u] = yt[min(r + u * par, rows - 1)];
#pragma unroll
                for (int u = 0; u < kUnroll; u++) {
                    const uint32_t* p = reinterpret_cast<const uint32_t*>(src + (ye[u] >> 9) * spitch);
                    const uint32_t* q = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(p) + spitch);
                    d[u][0][0] = p[0]; d[u][0][1] = p[1]; d[u][0][2] = p[2];
                    d[u][1][0] = q[0]; d[u][1][1] = q[1]; d[u][1][2] = q[2];
                }
#pragma unroll
                for (int u = 0; u < kUnroll; u++) {
                    const int ru = r + u * par;
                    if (ru >= rows) break;
                    uint32_t h[2][4];
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const uint32_t lo = __builtin_amdgcn_alignbyte(d[u][t][1], d[u][t][0], shift);
                        const uint32_t hi = __builtin_amdgcn_alignbyte(d[u][t][2], d[u][t][1], shift);
#pragma unroll
                        for (int j = 0; j < 4; j++) h[t][j] = dot2_u16(__builtin_amdgcn_perm(hi, lo, sel[j]), wgt[j], 0u);
                    }
                    const unsigned m1 = ye[u] & 0x1ffu, mw = (256u - m1) | (m1 << 16);
                    uint32_t v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        v[j] = dot2_u16(__builtin_amdgcn_perm(h[1][j], h[0][j], 0x05040100u), mw, 32768u);
                    const uint32_t packed = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u) | __builtin_amdgcn_perm(v[3], v[2], 0x06020c0cu);
                    *reinterpret_cast<uint32_t*>(dbuf + (ru * n4 + gi) * 4) = packed;
                    const int gy = cy0 + ru;
                    if (SA_PYR_EXP != 2 && own_x && gy >= oy0 && gy < oy1)
                        *reinterpret_cast<uint32_t*>(gdst + (gbase + __umul24((unsigned)ru, (unsigned)L.pitch))) = packed;
                }
            }
        }
        xt += sx[k].c1 - cx0;
        yt += rows;
        lds_barrier();
#if SA_PYR_STAMPS
        pst[psn++] = __builtin_amdgcn_s_memtime();
#endif
    }
#if SA_PYR_STAMPS
    if (threadIdx.x == 0 && (blockIdx.x == 100 || blockIdx.x == 900 || blockIdx.x == 1100 || blockIdx.x == 1800))
        printf("pyramid wg %d start %llu : loads issued %llu tables %llu stores %llu trips %llu | stage %llu levels %llu %llu %llu %llu %llu %llu %llu\n", (int)blockIdx.x, pst[0] & 0xffffffull, sst[0] - pst[0], sst[1] - sst[0], sst[2] - sst[1], sst[3] - sst[2], pst[1] - pst[0],
               pst[2] - pst[1], pst[3] - pst[2], pst[4] - pst[3], pst[5] - pst[4], pst[6] - pst[5], pst[7] - pst[6], pst[8] - pst[7]);
#endif
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
