// Gated brute-force Hamming matching for gfx950.
//
// radiusMatch with a radius that never prunes, followed by _filterMatches, is a
// gated arg-min: per query, the train row of least Hamming distance among the
// rows that pass the octave and window gates, ties to the lowest train index.
// Query descriptors sit in VGPRs (Q per lane), train descriptors are staged in
// LDS tiles and read wave-uniformly (broadcast); a distance is 8 xor + 8
// popcount-accumulate; the running best is one unsigned minimum over the key
// (distance << 20 | train index).  No sort and no Nq x Nt intermediate exist.
#include "common.h"
#include "kernels.h"

namespace sa {

constexpr int kMatchBlock = 256;
constexpr int kTrainTile = 256;      // train rows per LDS tile
constexpr unsigned kNoMatch = 0xffffffffu;
constexpr int kIdxBits = 20;         // key = distance << 20 | train index


// popcount(x) + acc in one instruction.  Written as asm because the compiler
// otherwise splits the eight-term sum into popcounts plus separate adds.
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) {
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

__device__ __forceinline__ int hamming256(const uint32_t (&q)[8], const uint4& a, const uint4& b) {
    unsigned d = bcnt_acc(q[0] ^ a.x, 0u);
    d = bcnt_acc(q[1] ^ a.y, d);
    d = bcnt_acc(q[2] ^ a.z, d);
    d = bcnt_acc(q[3] ^ a.w, d);
    d = bcnt_acc(q[4] ^ b.x, d);
    d = bcnt_acc(q[5] ^ b.y, d);
    d = bcnt_acc(q[6] ^ b.z, d);
    d = bcnt_acc(q[7] ^ b.w, d);
    return (int)d;
}

// Q queries per lane; AREA / LEVEL switch the window and octave gates on.
// PAIRS: problem = keyframe pair, one split, result stored as (d << 16 | idx).
// SECOND: also keep the second smallest key (for the ratio test); every split then
// stores its (best, second) pair and the finalize kernel merges the splits.
template <int Q, bool AREA, bool LEVEL, bool PAIRS, bool SECOND>
__global__ __launch_bounds__(kMatchBlock) void match_kernel(MatchArgs a) {
    __shared__ uint4 s_desc[kTrainTile * 2];
    __shared__ float s_tx[AREA ? kTrainTile : 1];
    __shared__ float s_ty[AREA ? kTrainTile : 1];
    __shared__ int s_toct[LEVEL ? kTrainTile : 1];

    int prob, qblock, split;
    size_t qbase, tbase;  // first row of the problem's query / train set
    int nq, nt;
    QuerySide qs = {a.qdesc, a.qkp, 0, nullptr};      // where the queries are read (qbase: where results go)
    if (PAIRS) {
        prob = blockIdx.x / a.qblocks;
        qblock = blockIdx.x - prob * a.qblocks;
        split = 0;
        qbase = (size_t)a.pairs[2 * prob] * a.n_per;
        qs.base = qbase;
        tbase = (size_t)a.pairs[2 * prob + 1] * a.n_per;
        nq = nt = a.n_per;
    } else {
        // 1-D grid, problem index fastest: consecutive workgroups (dealt round-robin
        // over the XCDs) carry different problems, and the query blocks beyond the
        // true counts, which exit at once, sit at the end of the grid.
        const int id = blockIdx.x;
        prob = id % a.batch;
        const int rest = id / a.batch;
        split = rest % a.splits;
        qblock = rest / a.splits;
        qbase = (size_t)prob * a.cap1;
        tbase = (size_t)prob * a.cap2;
        qs = query_side(a.chain, prob, a.qdesc, a.qkp, a.d_n1, a.cap1, a.tdesc, a.tkp, a.d_n2, a.cap2);
        int n1 = a.n1, n2 = a.n2;
        if (qs.count) n1 = min(n1, *qs.count);
        if (a.d_n2) n2 = min(n2, a.d_n2[prob]);
        nq = a.ids ? a.n_ids : n1;
        nt = n2;
    }
    const int t_begin = split * a.chunk;
    const int t_end = PAIRS ? nt : min(nt, t_begin + a.chunk);
    if (qblock * (kMatchBlock * Q) >= nq || t_begin >= t_end) return;

    uint32_t q[Q][8];
    float qx[Q], qy[Q];
    bool qhigh[Q];
    unsigned best[Q], second[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int qi = qblock * (kMatchBlock * Q) + k * kMatchBlock + threadIdx.x;
        best[k] = kNoMatch;
        second[k] = kNoMatch;
        qx[k] = qy[k] = 0.f;
        qhigh[k] = false;
        if (qi < nq) {
            const size_t row = qs.base + (size_t)((!PAIRS && a.ids) ? a.ids[qi] : qi);
            const uint4* p = reinterpret_cast<const uint4*>(qs.desc + row * 32);
            const uint4 lo = p[0], hi = p[1];
            q[k][0] = lo.x; q[k][1] = lo.y; q[k][2] = lo.z; q[k][3] = lo.w;
            q[k][4] = hi.x; q[k][5] = hi.y; q[k][6] = hi.z; q[k][7] = hi.w;
            if (AREA) {
                if (!PAIRS && a.q_xy) {
                    qx[k] = a.q_xy[2 * (qbase + qi)];
                    qy[k] = a.q_xy[2 * (qbase + qi) + 1];
                } else {
                    qx[k] = qs.kp[row].x;
                    qy[k] = qs.kp[row].y;
                }
            }
            if (LEVEL) qhigh[k] = qs.kp[row].octave > a.max_level;
        } else {
#pragma unroll
            for (int w = 0; w < 8; w++) q[k][w] = 0;
        }
    }

    for (int t0 = t_begin; t0 < t_end; t0 += kTrainTile) {
        const int tn = min(kTrainTile, t_end - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < tn * 2; i += kMatchBlock)
            s_desc[i] = reinterpret_cast<const uint4*>(a.tdesc + (tbase + t0) * 32)[i];
        if (AREA || LEVEL) {
            for (int i = threadIdx.x; i < tn; i += kMatchBlock) {
                const slam_amd_keypoint& kp = a.tkp[tbase + t0 + i];
                if (AREA) { s_tx[i] = kp.x; s_ty[i] = kp.y; }
                if (LEVEL) s_toct[i] = kp.octave;
            }
        }
        __syncthreads();
        // one train row against the Q queries of this lane
        auto key_of = [&](int j, int k) -> unsigned {
            const uint4 ta = s_desc[2 * j], tb = s_desc[2 * j + 1];
            const int d = hamming256(q[k], ta, tb);
            unsigned key = ((unsigned)d << kIdxBits) | (unsigned)(t0 + j);
            if (LEVEL) { if (qhigh[k] && s_toct[j] > a.max_level) key = kNoMatch; }
            if (AREA) {
                const float dx = qx[k] - s_tx[j], dy = qy[k] - s_ty[j];
                if (!(fabsf(dx) < a.area && fabsf(dy) < a.area)) key = kNoMatch;
            }
            return key;
        };
        int j = 0;
        if (SECOND) {
            // best <= second always holds, so the two smallest of {best, second, key}
            // are min(best, key) and the median of the three
            for (; j < tn; j++) {
#pragma unroll
                for (int k = 0; k < Q; k++) {
                    const unsigned key = key_of(j, k);
                    second[k] = min(max(best[k], key), second[k]);
                    best[k] = min(best[k], key);
                }
            }
        }
        // two train rows per trip: independent popcount chains and one 3-way minimum
        for (; j + 1 < tn; j += 2) {
#pragma unroll
            for (int k = 0; k < Q; k++) {
                const unsigned k0 = key_of(j, k), k1 = key_of(j + 1, k);
                best[k] = min(min(best[k], k0), k1);
            }
        }
        if (j < tn) {
#pragma unroll
            for (int k = 0; k < Q; k++) best[k] = min(best[k], key_of(j, k));
        }
    }

#pragma unroll
    for (int k = 0; k < Q; k++) {
        const int qi = qblock * (kMatchBlock * Q) + k * kMatchBlock + threadIdx.x;
        if (qi >= nq) continue;
        if (PAIRS) {
            unsigned out = kNoMatch;
            if (best[k] != kNoMatch) {
                const unsigned d = best[k] >> kIdxBits, t = best[k] & ((1u << kIdxBits) - 1);
                if ((int)d <= a.dmax) out = (d << 16) | t;
            }
            a.pair_out[(size_t)prob * a.n_per + qi] = out;
        } else if (SECOND) {
            a.part[(size_t)split * a.part_pitch + qbase + qi] = make_uint2(best[k], second[k]);
        } else {
            if (best[k] != kNoMatch) atomicMin(&a.best[qbase + qi], best[k]);
        }
    }
}

// Keys -> cv::DMatch records in ascending query order, one workgroup per problem.
// EX adds the two options the reference does not have: the ratio test on the
// (best, second) pairs merged over the train splits, and the rotation-consistency
// filter (histogram of angle differences, matches of the three fullest bins kept).

__device__ __forceinline__ int rotation_bin(float angle_q, float angle_t, int len) {
    float rot = __fsub_rn(angle_q, angle_t);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    const float factor = __fdiv_rn((float)len, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));
    if (bin >= len) bin = 0;
    return bin;
}

constexpr int kFinalRows = 8;                      // rows of 64 queries per wave and chunk
constexpr int kFinalChunk = 16 * kFinalRows * 64;  // queries per chunk of the 16 waves
constexpr int kFinalRowsMulti = 2;                 // the same where a problem is spread over workgroups, one chunk each
constexpr int kFinalChunkMulti = 16 * kFinalRowsMulti * 64;

template <bool EX>
__global__ __launch_bounds__(1024) void finalize_matches_kernel(FinalizeArgs a) {
    __shared__ int s_wave[16];
    __shared__ int s_hist[EX ? 64 : 1];
    const int prob = a.multi ? blockIdx.y : blockIdx.x;
    const QuerySide qs = query_side(a.chain, prob, nullptr, a.qkp, a.d_n1, a.cap1, nullptr, a.tkp, a.d_n2, a.cap2);
    int n1 = a.n1;
    if (qs.count) n1 = min(n1, *qs.count);
    const int nq = a.ids ? a.n_ids : n1;
    const unsigned* best = a.best + (size_t)prob * a.cap1;
    slam_amd_dmatch* out = a.out + (size_t)prob * a.cap1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool rot_on = EX && a.histo_length > 0;
    // key of query q after the distance bound and the ratio test, kNoMatch if rejected
    auto accepted = [&](int q) -> unsigned {
        unsigned b, s2 = kNoMatch;
        if (EX && a.part) {
            b = kNoMatch;
            for (int sp = 0; sp < a.splits; sp++) {
                const uint2 p = a.part[(size_t)sp * a.part_pitch + (size_t)prob * a.cap1 + q];
                s2 = min(min(s2, p.y), max(b, p.x));
                b = min(b, p.x);
            }
        } else {
            b = best[q];
        }
        if (b == kNoMatch || (int)(b >> kIdxBits) > a.dmax) return kNoMatch;
        if (EX && a.ratio > 0.f && s2 != kNoMatch) {
            const float d1 = (float)(b >> kIdxBits), d2 = (float)(s2 >> kIdxBits);
            if (!(d1 < __fmul_rn(a.ratio, d2))) return kNoMatch;
        }
        return b;
    };
    auto bin_of = [&](int q, unsigned key) -> int {
        const size_t qrow = qs.base + (a.ids ? a.ids[q] : q);
        const size_t trow = (size_t)prob * a.cap2 + (key & ((1u << kIdxBits) - 1));
        return rotation_bin(qs.kp[qrow].angle, a.tkp[trow].angle, a.histo_length);
    };
    int i1 = -1, i2 = -1, i3 = -1;
    if (rot_on) {
        if (threadIdx.x < 64) s_hist[threadIdx.x] = 0;
        __syncthreads();
        for (int q = threadIdx.x; q < nq; q += 1024) {
            const unsigned key = accepted(q);
            if (key != kNoMatch) atomicAdd(&s_hist[bin_of(q, key)], 1);
        }
        __syncthreads();
        // the three fullest bins, scanned in ascending order with strict comparisons
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < a.histo_length; i++) {
            const int s = s_hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
            else if (s > max3) { max3 = s; i3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
    }
    auto kept = [&](int q) -> unsigned {
        const unsigned key = accepted(q);
        if (rot_on && key != kNoMatch) {
            const int b = bin_of(q, key);
            if (b != i1 && b != i2 && b != i3) return kNoMatch;
        }
        return key;
    };
    // A problem spread over gridDim.x workgroups (plain keys only): workgroup c owns queries
    // [c, c + 1) * kFinalChunkMulti.  Its first output slot is the number of matches kept before its
    // chunk, which it counts for itself from the keys (a few coalesced loads per thread, from L2) --
    // no workgroup waits for another.  Nobody may put keys back while others still count them: the
    // workgroups draw a ticket when they are done reading, and the last one puts the problem's keys
    // back to "no match" and the ticket counter to zero.
    if (!EX && a.multi) {
        const int c = blockIdx.x;
        const int chunks = max(1, (nq + kFinalChunkMulti - 1) / kFinalChunkMulti);
        if (c >= chunks) return;                 // (uniform; the host's bound on the queries was not tight)
        // (per-lane sums and one reduction at the end: nothing between the loads of different trips, so they overlap)
        int before = 0;
#pragma unroll 8
        for (int q = threadIdx.x; q < c * kFinalChunkMulti; q += 1024) before += accepted(q) != kNoMatch;
#pragma unroll
        for (int o = 32; o; o >>= 1) before += __shfl_xor(before, o, 64);
        unsigned key[kFinalRowsMulti];
        unsigned long long mask[kFinalRowsMulti];
        int wcount = 0;
        const int base = c * kFinalChunkMulti;
#pragma unroll
        for (int k = 0; k < kFinalRowsMulti; k++) {
            const int q = base + (wave * kFinalRowsMulti + k) * 64 + lane;
            key[k] = q < nq ? accepted(q) : kNoMatch;
            mask[k] = __ballot(key[k] != kNoMatch);
            wcount += __popcll(mask[k]);
        }
        __shared__ int s_pre[16];
        if (lane == 0) { s_wave[wave] = wcount; s_pre[wave] = before; }
        __syncthreads();
        int pos = 0, total = 0;
        for (int w = 0; w < 16; w++) {
            const int v = s_wave[w];
            pos += s_pre[w] + (w < wave ? v : 0);
            total += s_pre[w] + v;
        }
        const unsigned long long below_ = (1ull << lane) - 1ull;
#pragma unroll
        for (int k = 0; k < kFinalRowsMulti; k++) {
            if (key[k] != kNoMatch) {
                const int q = base + (wave * kFinalRowsMulti + k) * 64 + lane;
                slam_amd_dmatch r;
                r.queryIdx = a.ids ? a.ids[q] : q;
                r.trainIdx = (int)(key[k] & ((1u << kIdxBits) - 1));
                r.imgIdx = 0;
                r.distance = (float)(key[k] >> kIdxBits);
                out[pos + __popcll(mask[k] & below_)] = r;
            }
            pos += __popcll(mask[k]);
        }
        if (c == chunks - 1 && threadIdx.x == 0) a.n_out[prob] = total;
        __shared__ int s_last;
        __syncthreads();                          // every key this workgroup reads has been read
        if (threadIdx.x == 0) {
            __threadfence();
            s_last = atomicAdd(&a.ticket[prob], 1u) == (unsigned)(chunks - 1);
        }
        __syncthreads();
        if (s_last) {
            __threadfence();
            if (a.reset)
                for (int q = threadIdx.x; q < nq; q += 1024) a.reset[(size_t)prob * a.cap1 + q] = kNoMatch;
            if (threadIdx.x == 0) a.ticket[prob] = 0u;
        }
        return;
    }
    // Chunks of kFinalChunk queries, kFinalRows rows of 64 consecutive queries per wave:
    // keys are fetched once with coalesced loads and stay in registers; a ballot per
    // row gives the counts, one pass over the 16 wave totals gives the wave's first
    // output slot, and a lane's slot inside a row is the number of set bits below it,
    // so the records of a row are written side by side.
    int written = 0;  // matches of the chunks before this one (same in every thread)
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = 0; base < nq; base += kFinalChunk) {
        unsigned key[kFinalRows];
        unsigned long long mask[kFinalRows];
        int wcount = 0;
#pragma unroll
        for (int k = 0; k < kFinalRows; k++) {
            const int q = base + (wave * kFinalRows + k) * 64 + lane;
            key[k] = q < nq ? kept(q) : kNoMatch;
            // (the last read of this query's running key: the matching kernels touch exactly the rows below nq)
            if (a.reset && q < nq) a.reset[(size_t)prob * a.cap1 + q] = kNoMatch;
            mask[k] = __ballot(key[k] != kNoMatch);
            wcount += __popcll(mask[k]);
        }
        __syncthreads();
        if (lane == 0) s_wave[wave] = wcount;
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < 16; w++) {
            const int v = s_wave[w];
            if (w < wave) before += v;
            total += v;
        }
        int pos = written + before;
#pragma unroll
        for (int k = 0; k < kFinalRows; k++) {
            if (key[k] != kNoMatch) {
                const int q = base + (wave * kFinalRows + k) * 64 + lane;
                slam_amd_dmatch r;
                r.queryIdx = a.ids ? a.ids[q] : q;
                r.trainIdx = (int)(key[k] & ((1u << kIdxBits) - 1));
                r.imgIdx = 0;
                r.distance = (float)(key[k] >> kIdxBits);
                out[pos + __popcll(mask[k] & below)] = r;
            }
            pos += __popcll(mask[k]);
        }
        written += total;
    }
    if (threadIdx.x == 0) a.n_out[prob] = written;
}

// Pinhole projection with radial/tangential distortion, double arithmetic.

__device__ __forceinline__ float2 project_one(const double* Rt, const double* K4, const double* dist,
                                              double X, double Y, double Z) {
    double x = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(Rt[0], X), __dmul_rn(Rt[1], Y)), __dmul_rn(Rt[2], Z)), Rt[3]);
    double y = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(Rt[4], X), __dmul_rn(Rt[5], Y)), __dmul_rn(Rt[6], Z)), Rt[7]);
    double z = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(Rt[8], X), __dmul_rn(Rt[9], Y)), __dmul_rn(Rt[10], Z)), Rt[11]);
    z = z != 0.0 ? __ddiv_rn(1.0, z) : 1.0;
    x = __dmul_rn(x, z);
    y = __dmul_rn(y, z);
    const double r2 = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
    const double r4 = __dmul_rn(r2, r2), r6 = __dmul_rn(r4, r2);
    const double a1 = __dmul_rn(__dmul_rn(2.0, x), y);
    const double a2 = __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, x), x));
    const double a3 = __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, y), y));
    const double cd = __dadd_rn(__dadd_rn(__dadd_rn(1.0, __dmul_rn(dist[0], r2)), __dmul_rn(dist[1], r4)), __dmul_rn(dist[4], r6));
    const double xd = __dadd_rn(__dadd_rn(__dmul_rn(x, cd), __dmul_rn(dist[2], a1)), __dmul_rn(dist[3], a2));
    const double yd = __dadd_rn(__dadd_rn(__dmul_rn(y, cd), __dmul_rn(dist[2], a3)), __dmul_rn(dist[3], a1));
    return make_float2((float)__dadd_rn(__dmul_rn(xd, K4[0]), K4[2]), (float)__dadd_rn(__dmul_rn(yd, K4[1]), K4[3]));
}

__global__ __launch_bounds__(256) void project_kernel(ProjectArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const float2 uv = project_one(a.Rt, a.K4, a.dist, a.xyz[3 * i], a.xyz[3 * i + 1], a.xyz[3 * i + 2]);
    a.uv[2 * i] = uv.x;
    a.uv[2 * i + 1] = uv.y;
}

// isOutlier of the reference for every triangulated point: NaN, parallax cosine outside
// [0, 0.999f], within 1e-6 of a camera centre, or reprojected more than one pixel away
// from its keypoint in either keyframe.  Float differences, double sums, as OpenCV's
// Mat arithmetic on three-element CV_32F vectors does.
__global__ __launch_bounds__(256) void outlier_kernel(OutlierArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const float X = a.xyz[3 * i], Y = a.xyz[3 * i + 1], Z = a.xyz[3 * i + 2];
    const slam_amd_dmatch mt = a.matches[i];
    uint8_t out = 1;
    // indices were validated on the host for the blocking call; guard device callers too
    const bool idx_ok = mt.queryIdx >= 0 && mt.queryIdx < a.nq && mt.trainIdx >= 0 && mt.trainIdx < a.nt;
    if (idx_ok && !(X != X)) {
        const float ax = __fsub_rn(X, a.cq[0]), ay = __fsub_rn(Y, a.cq[1]), az = __fsub_rn(Z, a.cq[2]);
        const float bx = __fsub_rn(X, a.ct[0]), by = __fsub_rn(Y, a.ct[1]), bz = __fsub_rn(Z, a.ct[2]);
        auto dot3 = [](float x0, float y0, float z0, float x1, float y1, float z1) {
            return __dadd_rn(__dadd_rn(__dmul_rn((double)x0, (double)x1), __dmul_rn((double)y0, (double)y1)),
                             __dmul_rn((double)z0, (double)z1));
        };
        const double dot = dot3(ax, ay, az, bx, by, bz);
        const double na = __dsqrt_rn(dot3(ax, ay, az, ax, ay, az)), nb = __dsqrt_rn(dot3(bx, by, bz, bx, by, bz));
        const double par = __ddiv_rn(dot, __dmul_rn(na, nb));
        const bool par_bad = par < (double)0.0f || par > (double)0.999f;
        const bool near = fabs(na) < 1E-6 || fabs(nb) < 1E-6;
        if (!par_bad && !near) {
            auto perr = [&](const Camera& c, float tx, float ty) {
                const float2 uv = project_one(c.Rt, c.K4, c.dist, X, Y, Z);
                const float dx = __fsub_rn(tx, uv.x), dy = __fsub_rn(ty, uv.y);
                return __dsqrt_rn(__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)));
            };
            const slam_amd_keypoint kq = a.kp_q[mt.queryIdx], kt = a.kp_t[mt.trainIdx];
            if (!(perr(a.cam_q, kq.x, kq.y) > 1.0) && !(perr(a.cam_t, kt.x, kt.y) > 1.0)) out = 0;
        }
    }
    a.flags[i] = out;
}

// Duplicate test of every (i, j): Hamming <= desc_dist and world distance < point_dist.

__global__ __launch_bounds__(256) void duplicate_kernel(DupArgs a) {
    __shared__ uint4 s_desc[kTrainTile * 2];
    __shared__ float s_p[kTrainTile * 3];
    const int i = blockIdx.x * 256 + threadIdx.x;
    uint32_t q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float px = 0, py = 0, pz = 0;
    if (i < a.n1) {
        const uint4* p = reinterpret_cast<const uint4*>(a.d1 + (size_t)i * 32);
        const uint4 lo = p[0], hi = p[1];
        q[0] = lo.x; q[1] = lo.y; q[2] = lo.z; q[3] = lo.w;
        q[4] = hi.x; q[5] = hi.y; q[6] = hi.z; q[7] = hi.w;
        px = a.p1[3 * i]; py = a.p1[3 * i + 1]; pz = a.p1[3 * i + 2];
    }
    for (int t0 = 0; t0 < a.n2; t0 += kTrainTile) {
        const int tn = min(kTrainTile, a.n2 - t0);
        __syncthreads();
        for (int k = threadIdx.x; k < tn * 2; k += 256)
            s_desc[k] = reinterpret_cast<const uint4*>(a.d2 + (size_t)t0 * 32)[k];
        for (int k = threadIdx.x; k < tn * 3; k += 256) s_p[k] = a.p2[(size_t)t0 * 3 + k];
        __syncthreads();
        if (i < a.n1) {
            for (int j = 0; j < tn; j++) {
                const int d = hamming256(q, s_desc[2 * j], s_desc[2 * j + 1]);
                uint8_t dup = 0;
                if (d <= a.desc_dist) {
                    const double dx = (double)__fsub_rn(px, s_p[3 * j]);
                    const double dy = (double)__fsub_rn(py, s_p[3 * j + 1]);
                    const double dz = (double)__fsub_rn(pz, s_p[3 * j + 2]);
                    const double s = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                    dup = __dsqrt_rn(s) < a.point_dist;
                }
                a.flag[(size_t)i * a.n2 + t0 + j] = dup;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
template <int Q, bool PAIRS, bool SECOND>
static void launch_match_t(const MatchArgs& a, dim3 grid, hipStream_t st) {
    const bool area = a.area != -1.f, level = a.max_level != -1;
    if (area && level) hipLaunchKernelGGL((match_kernel<Q, true, true, PAIRS, SECOND>), grid, dim3(kMatchBlock), 0, st, a);
    else if (area) hipLaunchKernelGGL((match_kernel<Q, true, false, PAIRS, SECOND>), grid, dim3(kMatchBlock), 0, st, a);
    else if (level) hipLaunchKernelGGL((match_kernel<Q, false, true, PAIRS, SECOND>), grid, dim3(kMatchBlock), 0, st, a);
    else hipLaunchKernelGGL((match_kernel<Q, false, false, PAIRS, SECOND>), grid, dim3(kMatchBlock), 0, st, a);
}

// nq_bound / nt_bound: host-side upper bounds used to size the grid.
MatchPlan plan_match(int batch, int nq_bound, int nt_bound, int n_cu, bool second) {
    MatchPlan p = {1, 0, 1, 0};
    if (nq_bound <= 0 || nt_bound <= 0 || batch <= 0) return p;
    // two queries per lane once that still leaves enough workgroups to fill the chip
    const long long tiles = (long long)batch * div_up(nq_bound, kMatchBlock * 2) * div_up(nt_bound, kTrainTile);
    p.q = tiles >= (long long)n_cu * 8 ? 2 : 1;
    p.qblocks = div_up(nq_bound, kMatchBlock * p.q);
    // The bounds are capacities: the true counts live on the device and are often
    // half of them, so aim for many small workgroups (those beyond the true counts
    // exit at once) and merge the train splits with an atomic minimum -- or, when
    // second-best keys are kept, in the finalize kernel over at most 16 splits.
    constexpr int kGrain = 64;
    int splits = div_up(n_cu * 16, p.qblocks * batch);
    splits = max(1, min(splits, div_up(nt_bound, kGrain)));
    if (second) splits = min(splits, kMaxSecondSplits);
    p.chunk = div_up(div_up(nt_bound, splits), kGrain) * kGrain;
    p.splits = div_up(nt_bound, p.chunk);
    return p;
}

int launch_match(MatchArgs a, const MatchPlan& p, int batch, hipStream_t st) {
    if (p.qblocks <= 0 || batch <= 0) return SLAM_AMD_OK;
    a.chunk = p.chunk;
    a.qblocks = p.qblocks;
    a.splits = p.splits;
    a.batch = batch;
    const long long blocks = (long long)p.qblocks * p.splits * batch;
    if (blocks > 0x7fffffffLL) SA_FAIL(SLAM_AMD_ERR_ARG, "match: problem too large for one launch");
    dim3 grid((unsigned)blocks);
    const bool second = a.part != nullptr;
    if (p.q == 2) { if (second) launch_match_t<2, false, true>(a, grid, st); else launch_match_t<2, false, false>(a, grid, st); }
    else { if (second) launch_match_t<1, false, true>(a, grid, st); else launch_match_t<1, false, false>(a, grid, st); }
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_match_pairs(MatchArgs a, int n_pairs, hipStream_t st) {
    if (n_pairs <= 0 || a.n_per <= 0) return SLAM_AMD_OK;
    constexpr int Q = 2;
    a.qblocks = div_up(a.n_per, kMatchBlock * Q);
    a.chunk = 0;
    const long long blocks = (long long)a.qblocks * n_pairs;
    if (blocks > 0x7fffffffLL) SA_FAIL(SLAM_AMD_ERR_ARG, "match_pairs: too many pairs for one launch");
    dim3 grid((unsigned)blocks);
    launch_match_t<Q, true, false>(a, grid, st);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_finalize(const FinalizeArgs& a, int batch, hipStream_t st) {
    const bool ex = a.part != nullptr || a.histo_length > 0;
    // plain keys, many queries: one workgroup per kFinalChunkMulti queries (blockIdx.y = problem)
    const int chunks = (!ex && a.ticket && batch <= kFinalTicketProblems) ? div_up(std::max(a.nq_bound, 1), kFinalChunkMulti) : 1;
    if (ex) hipLaunchKernelGGL(finalize_matches_kernel<true>, dim3(batch), dim3(1024), 0, st, a);
    else if (chunks > 2 && batch <= 65535) {
        FinalizeArgs b = a;
        b.multi = 1;
        hipLaunchKernelGGL(finalize_matches_kernel<false>, dim3(chunks, batch), dim3(1024), 0, st, b);
    } else hipLaunchKernelGGL(finalize_matches_kernel<false>, dim3(batch), dim3(1024), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_project(const ProjectArgs& a, hipStream_t st) {
    if (a.n <= 0) return SLAM_AMD_OK;
    hipLaunchKernelGGL(project_kernel, dim3(div_up(a.n, 256)), dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_outliers(const OutlierArgs& a, hipStream_t st) {
    if (a.n <= 0) return SLAM_AMD_OK;
    hipLaunchKernelGGL(outlier_kernel, dim3(div_up(a.n, 256)), dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_duplicates(const DupArgs& a, hipStream_t st) {
    if (a.n1 <= 0 || a.n2 <= 0) return SLAM_AMD_OK;
    hipLaunchKernelGGL(duplicate_kernel, dim3(div_up(a.n1, 256)), dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

}  // namespace sa
