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

constexpr int kPyrSpanBytes = 2 * kMaxLevels * (int)sizeof(PyrSpan);   // head of the kernel's LDS: the tile's spans

typedef unsigned short py_us2 __attribute__((ext_vector_type(2)));
typedef unsigned py_u32x4 __attribute__((ext_vector_type(4)));
typedef py_u32x4 __attribute__((aligned(1))) py_u32x4_a1;   // a 16-byte load from any address
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
    // grid: x, y = tile of the group's last level, z = frame
    const int ti = blockIdx.x, tj = blockIdx.y, b = blockIdx.z;
    const PyrSpan* sx = a.spanx + ti * slots;
    const PyrSpan* sy = a.spany + tj * slots;
    // coefficient slices, packed: the x slice of every level (source column | weight << 16, multiples of
    // four entries), then the y slices (weight | source row within the staged region << 9, 16 bits each)
    uint32_t* s_xt = reinterpret_cast<uint32_t*>(py_lds + kPyrSpanBytes);
    uint16_t* s_yt = reinterpret_cast<uint16_t*>(s_xt + a.x_entries);
    uint8_t* buf0 = py_lds + a.tab_bytes;
    uint8_t* buf1 = buf0 + a.buf_even;
    const int tid = threadIdx.x;
    // The tile's spans, copied to LDS for the chain: once the chain has stored to memory, reading them from
    // memory again is a vector load with its full latency at the head of every level.
    uint2* s_span = reinterpret_cast<uint2*>(py_lds);          // x spans at [0, slots), y spans at [kMaxLevels, ...)
    if (tid < slots) s_span[tid] = *reinterpret_cast<const uint2*>(sx + tid);
    else if (tid >= kMaxLevels && tid < kMaxLevels + slots) s_span[tid] = *reinterpret_cast<const uint2*>(sy + (tid - kMaxLevels));

    // ---- stage the source region (slot 0); columns / rows past the image stay unwritten or zero:
    //      they are only ever read with weight 0.  A row of the region is taken as 16-byte pieces
    //      (from any address: the region starts on a multiple of 4 pixels, the image anywhere) while
    //      they lie inside the image, and single dwords (bytes at the image's edge) for what is left
    //      of it.  All loads of a thread are issued first - kStage pieces and one dword per trip, a
    //      32-pixel tile needs one trip - then the coefficient slices' loads, then everything goes to LDS.
#ifndef SA_PYR_STAGE
#define SA_PYR_STAGE 4
#endif
    constexpr int kStage = SA_PYR_STAGE;
    const uint8_t* src_img = a.src + (size_t)b * a.src_frame_pitch;
    const int src_pitch = a.src_pitch, src_aligned = a.src_aligned;
    const int w0 = a.src_w, h0 = a.src_h;
    const int st_x0 = sx[0].c0, st_n4 = (sx[0].c1 - sx[0].c0 + 3) >> 2;
    const int st_y0 = sy[0].c0, st_rows = min(sy[0].c1, h0) - st_y0;
    const uint8_t* st_src = src_img + (size_t)st_y0 * src_pitch + st_x0;
    // pieces: q per row, over the dwords that lie wholly inside the image
    const int st_q = min(st_n4, max(w0 - st_x0, 0) >> 2) >> 2;
    const int st_rem0 = 4 * st_q, st_rem = st_n4 - st_rem0;       // left-over dwords of a row
    // (row, column) of a thread's next item advance by 256 items per step
    const int pq = max(st_q, 1), pr = max(st_rem, 1);
    const int p_dq = 256 / pq, p_dr = 256 - p_dq * pq;
    int p_row = small_div(tid, 65536 / pq + 1), p_c = tid - p_row * pq;
    if (st_q == 0) p_row = st_rows;
    const int d_dq = 256 / pr, d_dr = 256 - d_dq * pr;
    int d_row = small_div(tid, 65536 / pr + 1), d_c = tid - d_row * pr;
    if (st_rem == 0) d_row = st_rows;
    int pl[kStage], dl;
    py_u32x4 pv[kStage];
    uint32_t dv;
    auto piece_loads = [&]() {
#pragma unroll
        for (int u = 0; u < kStage; u++) {
            pl[u] = -1;
            if (p_row < st_rows) {
                pl[u] = p_row * st_n4 + 4 * p_c;
                pv[u] = *reinterpret_cast<const py_u32x4_a1*>(st_src + (size_t)p_row * src_pitch + 16 * p_c);
            }
            p_row += p_dq; p_c += p_dr;
            if (p_c >= pq) { p_c -= pq; p_row++; }
        }
    };
    auto piece_stores = [&]() {
#pragma unroll
        for (int u = 0; u < kStage; u++)
            if (pl[u] >= 0) {
                uint32_t* d = reinterpret_cast<uint32_t*>(buf0) + pl[u];
                d[0] = pv[u].x; d[1] = pv[u].y; d[2] = pv[u].z; d[3] = pv[u].w;
            }
    };
    auto dword_load = [&]() {
        dl = -1;
        dv = 0;
        if (d_row < st_rows) {
            const int c = st_rem0 + d_c, x = st_x0 + 4 * c;
            dl = d_row * st_n4 + c;
            const uint8_t* p = st_src + (size_t)d_row * src_pitch + 4 * c;
            if (src_aligned && x + 3 < w0) {
                dv = *reinterpret_cast<const uint32_t*>(p);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (x + j < w0) dv |= (uint32_t)p[j] << (8 * j);
            }
        }
        d_row += d_dq; d_c += d_dr;
        if (d_c >= pr) { d_c -= pr; d_row++; }
    };
    auto dword_store = [&]() {
        if (dl >= 0) reinterpret_cast<uint32_t*>(buf0)[dl] = dv;
    };
    piece_loads();
    dword_load();
#if SA_PYR_STAMPS
    unsigned long long sst[4] = {};
    sst[0] = __builtin_amdgcn_s_memtime();
#endif
    // ---- coefficient slices of every level of the group.  A slice covers the computed range,
    //      which is the needed range widened to whole dwords; the extra positions are never used,
    //      so their source indices are clamped into the source range instead of widening that too.
    //      Wave w takes slots 1 + w, 5 + w, ...; the first 128 entries of the first two of its slices
    //      (all there is for 8 levels and tiles of up to 32 pixels) are loaded before any is stored ----
    const int lane = tid & 63;
    uint32_t ce[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
        const int k = 1 + (tid >> 6) + 4 * kk;
        ce[kk][0] = ce[kk][1] = ce[kk][2] = ce[kk][3] = 0;
        if (k < slots) {
            const int l = a.first + k - 1;
            const int nx = sx[k].c1 - sx[k].c0, ny = sy[k].c1 - sy[k].c0;
            const uint32_t* gx = a.xtab[l] + sx[k].c0;
            const uint32_t* gy = a.ytab[l] + sy[k].c0;
            if (lane < nx) ce[kk][0] = gx[lane];
            if (lane + 64 < nx) ce[kk][1] = gx[lane + 64];
            if (lane < ny) ce[kk][2] = gy[lane];
            if (lane + 64 < ny) ce[kk][3] = gy[lane + 64];
        }
    }
#if SA_PYR_STAMPS
    sst[1] = __builtin_amdgcn_s_memtime();
#endif
    piece_stores();
    dword_store();
    // entries [from, ...) of slot k's slices, one after the other
    auto slices_from = [&](int k, int from) {
        const int l = a.first + k - 1;
        const int nx = sx[k].c1 - sx[k].c0, ny = sy[k].c1 - sy[k].c0;
        const unsigned xlo = sx[k - 1].c0, xhi = sx[k - 1].c1 - 2, ylo = sy[k - 1].c0, yhi = sy[k - 1].c1 - 2;
        uint32_t* xs = s_xt + a.xat[k];
        uint16_t* ys = s_yt + a.yat[k];
        for (int i = lane + from; i < nx; i += 64) {
            const uint32_t e = a.xtab[l][sx[k].c0 + i];
            xs[i] = (e & 0xffff0000u) | min(max(e & 0xffffu, xlo), xhi);
        }
        for (int i = lane + from; i < ny; i += 64) {
            const uint32_t e = a.ytab[l][sy[k].c0 + i];
            ys[i] = (uint16_t)((e >> 16) | ((min(max(e & 0xffffu, ylo), yhi) - ylo) << 9));
        }
    };
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
        const int k = 1 + (tid >> 6) + 4 * kk;
        if (k < slots) {
            const int nx = sx[k].c1 - sx[k].c0, ny = sy[k].c1 - sy[k].c0;
            const unsigned xlo = sx[k - 1].c0, xhi = sx[k - 1].c1 - 2, ylo = sy[k - 1].c0, yhi = sy[k - 1].c1 - 2;
            uint32_t* xs = s_xt + a.xat[k];
            uint16_t* ys = s_yt + a.yat[k];
            auto xentry = [&](uint32_t e) { return (e & 0xffff0000u) | min(max(e & 0xffffu, xlo), xhi); };
            auto yentry = [&](uint32_t e) { return (uint16_t)((e >> 16) | ((min(max(e & 0xffffu, ylo), yhi) - ylo) << 9)); };
            if (lane < nx) xs[lane] = xentry(ce[kk][0]);
            if (lane + 64 < nx) xs[lane + 64] = xentry(ce[kk][1]);
            if (lane < ny) ys[lane] = yentry(ce[kk][2]);
            if (lane + 64 < ny) ys[lane + 64] = yentry(ce[kk][3]);
            if (max(nx, ny) > 128) slices_from(k, 128);
        }
    }
    for (int k = 9 + (tid >> 6); k < slots; k += 4) slices_from(k, 0);
#if SA_PYR_STAMPS
    sst[2] = __builtin_amdgcn_s_memtime();
#endif
    while (p_row < st_rows) {     // larger source regions: further trips
        piece_loads();
        piece_stores();
    }
    while (d_row < st_rows) {
        dword_load();
        dword_store();
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
    const unsigned span0 = s_span[0].x;
    int prev_x0 = __builtin_amdgcn_readfirstlane((int)(span0 & 0xffffu)), prev_x1 = __builtin_amdgcn_readfirstlane((int)(span0 >> 16));   // span of the slot before, in x
    for (int k = 1; k < slots; k++) {
        const int l = a.first + k - 1;
        const LevelGeom& L = a.g.lv[l];
        const uint8_t* sbuf = (k & 1) ? buf0 : buf1;
        uint8_t* dbuf = (k & 1) ? buf1 : buf0;
        const uint2 cur_x = s_span[k], cur_y = s_span[kMaxLevels + k];
        const int cx0 = __builtin_amdgcn_readfirstlane((int)(cur_x.x & 0xffffu)), cx1 = __builtin_amdgcn_readfirstlane((int)(cur_x.x >> 16));
        const int ox0 = __builtin_amdgcn_readfirstlane((int)(cur_x.y & 0xffffu)), ox1 = __builtin_amdgcn_readfirstlane((int)(cur_x.y >> 16));
        const int cy0 = __builtin_amdgcn_readfirstlane((int)(cur_y.x & 0xffffu)), cy1 = __builtin_amdgcn_readfirstlane((int)(cur_y.x >> 16));
        const int oy0 = __builtin_amdgcn_readfirstlane((int)(cur_y.y & 0xffffu)), oy1 = __builtin_amdgcn_readfirstlane((int)(cur_y.y >> 16));
        const int spitch = ((prev_x1 - prev_x0 + 3) >> 2) << 2;
        const int sx0 = prev_x0;
        const int n4 = (cx1 - cx0 + 3) >> 2, rows = cy1 - cy0;
        uint8_t* gdst = a.pyr + (size_t)b * a.g.frame_bytes + L.off;
        const uint32_t* xt = s_xt + a.xat[k];
        const uint16_t* yt = s_yt + a.yat[k];
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
        prev_x0 = cx0; prev_x1 = cx1;
        lds_barrier();
#if SA_PYR_STAMPS
        pst[psn++] = __builtin_amdgcn_s_memtime();
#endif
    }
#if SA_PYR_STAMPS
    if (threadIdx.x == 0 && (blockIdx.x == 2 && blockIdx.y == 3 && (blockIdx.z == 3 || blockIdx.z == 30 || blockIdx.z == 36 || blockIdx.z == 60)))
        printf("pyramid frame %d start %llu : loads issued %llu tables %llu stores %llu trips %llu | stage %llu levels %llu %llu %llu %llu %llu %llu %llu\n", (int)blockIdx.z, pst[0] & 0xffffffull, sst[0] - pst[0], sst[1] - sst[0], sst[2] - sst[1], sst[3] - sst[2], pst[1] - pst[0],
               pst[2] - pst[1], pst[3] - pst[2], pst[4] - pst[3], pst[5] - pst[4], pst[6] - pst[5], pst[7] - pst[6], pst[8] - pst[7]);
#endif
}

int launch_pyramid(const PyramidArgs& args, int batch, size_t lds_bytes, hipStream_t st) {
    if (args.ntx <= 0 || args.nty <= 0 || batch <= 0) return SLAM_AMD_OK;
    if (args.nty > 65535 || batch > 65535) SA_FAIL(SLAM_AMD_ERR_ARG, "pyramid: too many tiles or frames for one launch");
    // the source level of the group: the caller's frames, or a level an earlier group produced
    PyramidArgs a = args;
    const int ls0 = a.first - 1;
    if (ls0 == 0) {
        a.src = a.src0; a.src_frame_pitch = a.src0_frame_pitch; a.src_pitch = a.src0_pitch; a.src_aligned = a.src0_aligned;
    } else {
        a.src = a.pyr + a.g.lv[ls0].off; a.src_frame_pitch = a.g.frame_bytes; a.src_pitch = a.g.lv[ls0].pitch; a.src_aligned = 1;
    }
    a.src_w = a.g.lv[ls0].w; a.src_h = a.g.lv[ls0].h;
    hipLaunchKernelGGL(pyramid_kernel, dim3((unsigned)a.ntx, (unsigned)a.nty, (unsigned)batch), dim3(256), lds_bytes, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

}  // namespace sa
