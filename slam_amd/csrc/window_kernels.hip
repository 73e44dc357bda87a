// Window-gated matching without the full Nq x Nt sweep: the reordering of both sets by grid cell
// (spatial_sort_kernel, which serves both window kernels) and the popcount window kernel.
//
// With the window gate on (|dx| < area and |dy| < area, src/frame/matcher.cpp:148-153)
// only train rows near a query can win.  Both sets are first reordered by grid cell
// (cell edge >= area): index word (original index | octave-above-max_level flag) and position
// per row, a bounding box per tile of 32 or 64 train rows, and for the popcount kernel a copy of
// the descriptors.  The popcount kernel (match_window_kernel, train sets beyond what the
// matrix-core kernel of match_mfma_kernels.hip indexes): a wave of 64 spatially close queries
// visits only the train tiles whose box lies within `area` of the wave's own box; inside a
// visited tile every row is still put through the exact gate and distance of the brute-force
// kernel, and the key carries the row's original index, so the result is bit-identical to the
// full sweep.
#include <math.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace sa {

constexpr unsigned kNoMatch = 0xffffffffu;
constexpr int kIdxBits = 20;
constexpr int kSortBlock = 1024;
constexpr int kMaxCellsAxis = 64;
constexpr int kMaxCells = kMaxCellsAxis * kMaxCellsAxis;
constexpr unsigned kHighFlag = 0x80000000u;

// False for NaN and infinities: such a position fails the window test against
// everything, so it may be left out of every bounding box.
__device__ __forceinline__ bool finite2(float x, float y) {
    return fabsf(x) <= 3.402823466e38f && fabsf(y) <= 3.402823466e38f;
}

// Positions clamped for the cell computation only, so that extents stay finite.
__device__ __forceinline__ float cell_coord(float v) { return fminf(fmaxf(v, -1.0e18f), 1.0e18f); }

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// One workgroup per (chunk of 4096 rows, problem); the chunks of the query set come
// first.  A chunk is reordered within itself: waves and tiles only need rows that lie
// close together, not a global order, and chunks sort in parallel.
constexpr int kChunkRows = 4096;
constexpr int kRowsPerThread = kChunkRows / kSortBlock;
// Row k of thread `tid` inside its chunk: neighbouring lanes take neighbouring rows.  (They usually lie in the same
// grid cell -- an extractor leaves its rows ordered by position -- and their additions to the cell's counter are
// served one after the other, 2.5 us of the kernel's 12 for 64 frames of 2000 rows; dealing the rows to lanes far
// apart removes that and costs 4 us in the loads, which then touch 64 cache lines per instruction.)
__device__ __forceinline__ int local_row(int k, int tid) { return k * kSortBlock + tid; }

__global__ __launch_bounds__(kSortBlock) void spatial_sort_kernel(SpatialSortArgs a) {
    __shared__ int s_cell[kMaxCells];      // counts, then write cursors
    __shared__ float s_red[4][16];
    __shared__ int s_scan[16];
    __shared__ float2 s_sxy[kChunkRows];   // positions and index words in their new order: written out row after row, and for the boxes
    __shared__ unsigned s_sidx[kChunkRows];
    const int prob = blockIdx.y;
    const int set = (int)blockIdx.x >= a.qchunks ? 1 : 0;
    const int chunk = set == 0 ? blockIdx.x : blockIdx.x - a.qchunks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (set == 0 && prob >= a.batch_q) return;
    // Everything a thread needs of its rows -- position, octave, descriptor -- is requested at once, by the host's
    // bound on the set (rows past the bound repeat the last one below it); the set's true count, in device memory,
    // is looked at after the requests are out, so that the kernel waits for memory once and not three times.
    const int nb = set == 0 ? (a.ids ? a.n_ids : a.n1) : a.n2;
    const int c0 = chunk * kChunkRows;
    if (c0 >= nb) return;
    const int32_t* const d_n = set == 0 ? a.d_n1 : a.d_n2;
    const size_t base = (size_t)prob * (set == 0 ? a.cap1 : a.cap2);  // first row of the problem's set
    const uint8_t* desc = set == 0 ? a.qdesc : a.tdesc;
    const slam_amd_keypoint* kp = set == 0 ? a.qkp : a.tkp;
    const bool gather = set == 0 && a.ids;
    const bool own_xy = set == 0 && a.q_xy;

    float2 p[kRowsPerThread];
    // (the descriptors as two vectors of 16 dwords, four per row: an array of uint4 ends up in scratch memory here)
    typedef unsigned u16v __attribute__((ext_vector_type(16)));
    static_assert(kRowsPerThread == 4, "four rows per thread");
    u16v d0 = 0u, d1 = 0u;
    int oct[kRowsPerThread];
    // (one loop per kind of load, the conditions outside: inside one loop the compiler waits for every row's loads
    // before it asks for the next row's)
    int src[kRowsPerThread];
    size_t row[kRowsPerThread];
#pragma unroll
    for (int k = 0; k < kRowsPerThread; k++) src[k] = min(c0 + local_row(k, tid), nb - 1);
    if (own_xy) {
#pragma unroll
        for (int k = 0; k < kRowsPerThread; k++) p[k] = reinterpret_cast<const float2*>(a.q_xy)[base + src[k]];
    }
    if (gather) {
#pragma unroll
        for (int k = 0; k < kRowsPerThread; k++) src[k] = a.ids[src[k]];
    }
#pragma unroll
    for (int k = 0; k < kRowsPerThread; k++) row[k] = base + (size_t)src[k];
    if (a.copy_desc) {
#pragma unroll
      for (int k = 0; k < kRowsPerThread; k++) {
        const uint4* d = reinterpret_cast<const uint4*>(desc + row[k] * 32);
        const uint4 lo = d[0], hi = d[1];
        d0[4 * k] = lo.x; d0[4 * k + 1] = lo.y; d0[4 * k + 2] = lo.z; d0[4 * k + 3] = lo.w;
        d1[4 * k] = hi.x; d1[4 * k + 1] = hi.y; d1[4 * k + 2] = hi.z; d1[4 * k + 3] = hi.w;
      }
    }
    if (!own_xy) {
#pragma unroll
        for (int k = 0; k < kRowsPerThread; k++) p[k] = make_float2(kp[row[k]].x, kp[row[k]].y);
    }
#pragma unroll
    for (int k = 0; k < kRowsPerThread; k++) oct[k] = 0;
    if (a.max_level != -1) {
#pragma unroll
        for (int k = 0; k < kRowsPerThread; k++) oct[k] = kp[row[k]].octave;
    }
    int n = nb;
    if (d_n) n = set == 0 && a.ids ? nb : min(nb, d_n[prob]);      // (with ids, the count of set 1 does not bound the queries)
    if (c0 >= n) return;
    const int cn = min(kChunkRows, n - c0);   // rows of this chunk
    const int scap = a.scap[set];
    uint4* sdesc = a.sdesc[set] + ((size_t)prob * scap + c0) * 2;
    float2* sxy = a.sxy[set] + (size_t)prob * scap + c0;
    uint32_t* sidx = a.sidx[set] + (size_t)prob * scap + c0;
    const int box_rows = a.box_rows;      // 64 or 32
    float4* box = a.tile_box[set] + (size_t)prob * (scap / box_rows) + c0 / box_rows;

    unsigned idx[kRowsPerThread];
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
#pragma unroll
    for (int k = 0; k < kRowsPerThread; k++) {
        const int li = local_row(k, tid);
        idx[k] = 0;
        if (li < cn) {
            idx[k] = (unsigned)(c0 + li);
            if (a.max_level != -1 && oct[k] > a.max_level) idx[k] |= kHighFlag;
            if (finite2(p[k].x, p[k].y)) {
                const float cx = cell_coord(p[k].x), cy = cell_coord(p[k].y);
                xmin = fminf(xmin, cx); xmax = fmaxf(xmax, cx);
                ymin = fminf(ymin, cy); ymax = fmaxf(ymax, cy);
            }
        } else {
            p[k] = make_float2(NAN, NAN);
        }
    }
    xmin = wave_min(xmin); xmax = wave_max(xmax); ymin = wave_min(ymin); ymax = wave_max(ymax);
    if (lane == 0) { s_red[0][wave] = xmin; s_red[1][wave] = xmax; s_red[2][wave] = ymin; s_red[3][wave] = ymax; }
    for (int c = tid; c < kMaxCells; c += kSortBlock) s_cell[c] = 0;
    __syncthreads();
    for (int w = 0; w < 16; w++) {
        xmin = fminf(xmin, s_red[0][w]); xmax = fmaxf(xmax, s_red[1][w]);
        ymin = fminf(ymin, s_red[2][w]); ymax = fmaxf(ymax, s_red[3][w]);
    }
    if (!(xmin <= xmax)) { xmin = xmax = 0.f; ymin = ymax = 0.f; }  // no finite position at all
    const float ex = xmax - xmin, ey = ymax - ymin;
    const float cw = fmaxf(a.area, ex * (1.0f / kMaxCellsAxis));
    const float ch = fmaxf(a.area, ey * (1.0f / kMaxCellsAxis));
    const int ncx = min(kMaxCellsAxis, (int)(ex / cw) + 1);
    const int ncy = min(kMaxCellsAxis, (int)(ey / ch) + 1);
    const float icw = 1.0f / cw, ich = 1.0f / ch;
    // Any mapping position -> cell gives a correct result; this one keeps neighbours
    // together: cells in rows, every other row of cells walked right to left so that
    // consecutive cells always touch.
    int cell[kRowsPerThread], rank[kRowsPerThread] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < kRowsPerThread; k++) {
        cell[k] = 0;
        if (local_row(k, tid) < cn && finite2(p[k].x, p[k].y)) {
            int cx = (int)fminf(fmaxf((cell_coord(p[k].x) - xmin) * icw, 0.f), (float)(ncx - 1));
            const int cy = (int)fminf(fmaxf((cell_coord(p[k].y) - ymin) * ich, 0.f), (float)(ncy - 1));
            if (cy & 1) cx = ncx - 1 - cx;
            cell[k] = cy * ncx + cx;
        }
        // the row's rank among the rows of its cell (whichever order the additions land in)
        if (local_row(k, tid) < cn) rank[k] = atomicAdd(&s_cell[cell[k]], 1);
    }
    __syncthreads();
    // exclusive scan of the cell counts: four consecutive cells per thread
    int c4[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { c4[k] = s_cell[tid * 4 + k]; sum += c4[k]; }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) s_scan[wave] = inc;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; w++) before += s_scan[w];
    int run = before + inc - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) { s_cell[tid * 4 + k] = run; run += c4[k]; }
    __syncthreads();

    // scatter; the order inside a cell is arbitrary, which does not change any result
#pragma unroll
    for (int k = 0; k < kRowsPerThread; k++) {
        if (local_row(k, tid) >= cn) continue;
        const int pos = s_cell[cell[k]] + rank[k];
        if (a.copy_desc) {
            sdesc[2 * pos] = make_uint4(d0[4 * k], d0[4 * k + 1], d0[4 * k + 2], d0[4 * k + 3]);
            sdesc[2 * pos + 1] = make_uint4(d1[4 * k], d1[4 * k + 1], d1[4 * k + 2], d1[4 * k + 3]);
        }
        s_sxy[pos] = p[k];
        s_sidx[pos] = idx[k];
    }
    __syncthreads();
    for (int i = tid; i < cn; i += kSortBlock) { sxy[i] = s_sxy[i]; sidx[i] = s_sidx[i]; }
    if (set == 0) return;  // query waves compute their own box

    const int ntiles = (cn + 63) / 64;
    for (int t = wave; t < ntiles; t += kSortBlock / 64) {
        const int i = t * 64 + lane;
        float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY;
        if (i < cn) {
            const float2 q = s_sxy[i];
            if (finite2(q.x, q.y)) { bx0 = bx1 = q.x; by0 = by1 = q.y; }
        }
        for (int o = box_rows >> 1; o > 0; o >>= 1) {
            bx0 = fminf(bx0, __shfl_xor(bx0, o, 64)); bx1 = fmaxf(bx1, __shfl_xor(bx1, o, 64));
            by0 = fminf(by0, __shfl_xor(by0, o, 64)); by1 = fmaxf(by1, __shfl_xor(by1, o, 64));
        }
        if ((lane & (box_rows - 1)) == 0 && t * 64 + lane < cn) box[(t * 64 + lane) / box_rows] = make_float4(bx0, bx1, by0, by1);
    }
}

__device__ __forceinline__ unsigned bcnt_acc_w(unsigned x, unsigned acc) {
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

constexpr int kWinWaves = 4;
constexpr int kAhead = 4;  // train tiles tested and fetched together

// Each wave: 64 consecutive sorted queries against the train tiles near them.
template <bool LEVEL, bool SECOND>
__global__ __launch_bounds__(kWinWaves * 64) void match_window_kernel(WindowMatchArgs a) {
    __shared__ uint4 s_desc[kWinWaves][128];
    __shared__ float2 s_xy[kWinWaves][64];
    __shared__ unsigned s_idx[kWinWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // problem fastest, then the train split, then the block of four query tiles
    const int prob = blockIdx.x % a.batch;
    const int rest = blockIdx.x / a.batch;
    const int split = rest % a.splits;
    const int qtile = (rest / a.splits) * kWinWaves + wave;

    int n1 = a.n1, n2 = a.n2;
    if (a.d_n1) n1 = min(n1, a.d_n1[prob]);
    if (a.d_n2) n2 = min(n2, a.d_n2[prob]);
    const int nq = a.has_ids ? a.n_ids : n1;
    const int nt = n2;
    if (qtile * 64 >= nq) return;  // no workgroup barrier below: waves are independent

    const uint4* qd = a.sdesc[0] + (size_t)prob * a.scap[0] * 2;
    const float2* qxy = a.sxy[0] + (size_t)prob * a.scap[0];
    const uint32_t* qidx = a.sidx[0] + (size_t)prob * a.scap[0];
    const uint4* td = a.sdesc[1] + (size_t)prob * a.scap[1] * 2;
    const float2* txy = a.sxy[1] + (size_t)prob * a.scap[1];
    const uint32_t* tidx = a.sidx[1] + (size_t)prob * a.scap[1];
    const float4* tbox = a.tile_box + (size_t)prob * (a.scap[1] / 64);      // (launched with boxes per 64 rows)

    const int qi = qtile * 64 + lane;
    const bool qvalid = qi < nq;
    uint32_t q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float qx = NAN, qy = NAN;
    unsigned qid = 0;
    if (qvalid) {
        const uint4 lo = qd[2 * qi], hi = qd[2 * qi + 1];
        q[0] = lo.x; q[1] = lo.y; q[2] = lo.z; q[3] = lo.w;
        q[4] = hi.x; q[5] = hi.y; q[6] = hi.z; q[7] = hi.w;
        const float2 p = qxy[qi];
        qx = p.x; qy = p.y;
        qid = qidx[qi];
    }
    const bool qhigh = (qid & kHighFlag) != 0;
    const bool qfin = qvalid && finite2(qx, qy);
    const float wx0 = wave_min(qfin ? qx : INFINITY), wx1 = wave_max(qfin ? qx : -INFINITY);
    const float wy0 = wave_min(qfin ? qy : INFINITY), wy1 = wave_max(qfin ? qy : -INFINITY);

    unsigned best = kNoMatch, second = kNoMatch;
    const int ntiles = (nt + 63) / 64;
    // The splits take the train tiles in turn: tiles near a wave are neighbours in the
    // sorted order, so each split gets its share of them.  Four tiles are tested at a
    // time and the rows of those that pass are fetched together, ahead of their use.
    for (int t0 = split; t0 < ntiles; t0 += kAhead * a.splits) {
        bool hit[kAhead];
        uint4 fa[kAhead], fb[kAhead];
        float2 fxy[kAhead];
        unsigned fid[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            const int t = t0 + u * a.splits;
            hit[u] = false;
            if (t < ntiles) {
                const float4 b = tbox[t];  // xmin, xmax, ymin, ymax
                // Rounding is monotone, so when the boxes are at least `area` apart on one axis
                // every pair of their points fails the strict window test on that axis.
                hit[u] = !(b.x - wx1 >= a.area || wx0 - b.y >= a.area || b.z - wy1 >= a.area || wy0 - b.w >= a.area);
            }
            const int r = t * 64 + lane;
            if (hit[u] && r < nt) {
                fa[u] = td[2 * r]; fb[u] = td[2 * r + 1];
                fxy[u] = txy[r];
                fid[u] = tidx[r];
            }
        }
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
        if (!hit[u]) continue;
        const int t = t0 + u * a.splits;
        const int tn = min(64, nt - t * 64);
        __builtin_amdgcn_wave_barrier();
        if (lane < tn) {
            s_desc[wave][2 * lane] = fa[u];
            s_desc[wave][2 * lane + 1] = fb[u];
            s_xy[wave][lane] = fxy[u];
            s_idx[wave][lane] = fid[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto key_of = [&](int j) -> unsigned {
            const uint4 ta = s_desc[wave][2 * j], tb = s_desc[wave][2 * j + 1];
            unsigned d = bcnt_acc_w(q[0] ^ ta.x, 0u);
            d = bcnt_acc_w(q[1] ^ ta.y, d);
            d = bcnt_acc_w(q[2] ^ ta.z, d);
            d = bcnt_acc_w(q[3] ^ ta.w, d);
            d = bcnt_acc_w(q[4] ^ tb.x, d);
            d = bcnt_acc_w(q[5] ^ tb.y, d);
            d = bcnt_acc_w(q[6] ^ tb.z, d);
            d = bcnt_acc_w(q[7] ^ tb.w, d);
            const unsigned ti = s_idx[wave][j];
            unsigned key = (d << kIdxBits) | (ti & ~kHighFlag);
            if (LEVEL) { if (qhigh && (ti & kHighFlag)) key = kNoMatch; }
            const float2 tp = s_xy[wave][j];
            const float dx = qx - tp.x, dy = qy - tp.y;
            if (!(fabsf(dx) < a.area && fabsf(dy) < a.area)) key = kNoMatch;
            return key;
        };
        int j = 0;
        for (; j + 1 < tn; j += 2) {   // two independent popcount chains per trip
            const unsigned k0 = key_of(j), k1 = key_of(j + 1);
            if (SECOND) {
                second = min(max(best, k0), second);
                best = min(best, k0);
                second = min(max(best, k1), second);
                best = min(best, k1);
            } else {
                best = min(min(best, k0), k1);
            }
        }
        if (j < tn) {
            const unsigned k0 = key_of(j);
            if (SECOND) second = min(max(best, k0), second);
            best = min(best, k0);
        }
        }
    }
    if (!qvalid) return;
    const size_t out = (size_t)prob * a.cap1 + (qid & ~kHighFlag);
    if (SECOND) a.part[(size_t)split * a.part_pitch + out] = make_uint2(best, second);
    else if (best != kNoMatch) atomicMin(&a.best[out], best);
}

int launch_spatial_sort(SpatialSortArgs a, int batch, int nq_bound, int nt_bound, hipStream_t st) {
    a.qchunks = div_up(nq_bound, kChunkRows);
    if (a.batch_q <= 0) a.batch_q = batch;
    if (a.box_rows != 32) a.box_rows = 64;
    const int chunks = a.qchunks + div_up(nt_bound, kChunkRows);
    hipLaunchKernelGGL(spatial_sort_kernel, dim3(chunks, batch), dim3(kSortBlock), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

// Train splits of a windowed match: enough waves to hide latency, at most
// kMaxSecondSplits (the planes of the ratio test) and the number of train tiles.
int window_splits(int batch, int nq_bound, int nt_bound, int n_cu) {
    const long long qwaves = (long long)div_up(nq_bound, 64) * batch;
    if (qwaves <= 0) return 1;
    const long long want = div_up(n_cu * 32, (int)std::min<long long>(qwaves, 1 << 30));
    return (int)std::max<long long>(1, std::min<long long>(std::min<long long>(want, kMaxSecondSplits), div_up(nt_bound, 64)));
}

int launch_match_window(const WindowMatchArgs& a, int batch, int nq_bound, hipStream_t st) {
    const long long blocks = (long long)div_up(div_up(nq_bound, 64), kWinWaves) * a.splits * batch;
    if (blocks <= 0) return SLAM_AMD_OK;
    if (blocks > 0x7fffffffLL) SA_FAIL(SLAM_AMD_ERR_ARG, "match: problem too large for one launch");
    const dim3 grid((unsigned)blocks), block(kWinWaves * 64);
    const bool level = a.max_level != -1, second = a.part != nullptr;
    if (level && second) hipLaunchKernelGGL((match_window_kernel<true, true>), grid, block, 0, st, a);
    else if (level) hipLaunchKernelGGL((match_window_kernel<true, false>), grid, block, 0, st, a);
    else if (second) hipLaunchKernelGGL((match_window_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((match_window_kernel<false, false>), grid, block, 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

}  // namespace sa
