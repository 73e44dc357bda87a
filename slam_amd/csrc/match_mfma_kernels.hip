// Brute-force Hamming matching on the matrix cores of gfx950 (gates off).
//
// All-pairs Hamming distance is a matrix product: with the 256 descriptor bits of a train row
// as a 0/1 vector a and those of a query as b,
//     H(a, b) = popcount(a) + popcount(b) - 2 a.b
// v_mfma_i32_32x32x32_i8 forms a.b for a 32 x 32 block of (train, query) pairs.  The train bits
// are expanded to bytes {0, 1}, the query bits to bytes {0, -64}, and a ninth K-step adds
// 32 popcount(a) + (row of the train inside its block of 32): the train side carries its four
// per-dword popcounts and its row number, the query side 32, 32, 32, 32, 1.  An accumulator then
// holds 32 (H - popcount(b)) + row exactly: the order of (distance, train index) inside the
// block, so the arg-min over a block is a plain signed minimum over accumulators.  Across
// blocks the running best is the minimum of (H - popcount(b)) << 20 | train index, rebuilt from
// the block's minimum with three instructions.  popcount(b) is constant
// per query and is added after the arg-min.  Ties go to the lowest train index, as in the
// popcount kernel.
//
// Layout of one product: train rows are the M side (A operand), queries the N side (B operand).
// An accumulator register of lane l belongs to query l & 31, so the arg-min over the train rows
// runs inside the lane; the two lane halves hold different train rows and are merged once at
// the end.  Queries stay in registers for the whole kernel (expanded once); train tiles of 64
// rows are expanded cooperatively into LDS, double buffered.
//
// The K order inside a step does not matter as long as both operands use the same one: lane
// half h of either operand supplies the bits of bytes 16h .. 16h + 15 of its descriptor.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace sa {

constexpr int kMmBlock = 256;                 // 4 waves
#ifndef SA_MM_NB
#define SA_MM_NB 2
#endif
constexpr int kMmNB = SA_MM_NB;               // query blocks of 32 per wave
constexpr int kMmQueries = 4 * kMmNB * 32;    // queries per workgroup
constexpr int kMmSub = 32;                    // train rows of one product
constexpr int kMmTile = 64;                   // train rows per LDS tile (two products)
constexpr int kMmSteps = 9;                   // 8 steps of 32 descriptor bits per lane half + the popcount step
constexpr int kMmSubU4 = kMmSteps * 64;       // uint4 slots of one expanded block of 32 rows: [step][half][row]
constexpr unsigned kMmNoMatch = 0xffffffffu;
constexpr int kMmIdxBits = 20;
constexpr int kMmBlockBits = 15;              // kMmIdxBits minus the 5 row bits an accumulator carries

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// 4 bits -> 4 bytes of 0 / 1 (bit j in byte j).
__device__ __forceinline__ unsigned spread4(unsigned x4) { return (x4 * 0x00204081u) & 0x01010101u; }

// 16 bits -> 16 bytes, each `scale` where the bit is set.
__device__ __forceinline__ v4i expand16(unsigned bits16, unsigned scale) {
    v4i r;
    r.x = (int)(spread4(bits16 & 15u) * scale);
    r.y = (int)(spread4((bits16 >> 4) & 15u) * scale);
    r.z = (int)(spread4((bits16 >> 8) & 15u) * scale);
    r.w = (int)(spread4(bits16 >> 12) * scale);
    return r;
}

// SECOND: also keep the second smallest key (for the ratio test); every split then stores its
// (best, second) pair and the finalize kernel merges the splits.
template <bool PAIRS, bool SECOND>
__global__ __launch_bounds__(kMmBlock) void match_mfma_kernel(MatchArgs a) {
    __shared__ uint4 s_tile[2][2 * kMmSubU4];

    int prob, qblock, split;
    size_t qbase, tbase;
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
        // 1-D grid, problem index fastest (see match_kernel)
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
    const int t_begin = split * a.chunk;      // a multiple of kMmTile
    const int t_end = PAIRS ? nt : min(nt, t_begin + a.chunk);
    if (qblock * kMmQueries >= nq || t_begin >= t_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // ---- queries of this wave: expanded once, {0, -64} bytes ----
    v4i bq[kMmNB][8];
    int pq[kMmNB];      // popcount of the whole query descriptor
    int qidx[kMmNB];
#pragma unroll
    for (int b = 0; b < kMmNB; b++) {
        const int qi = qblock * kMmQueries + (wave * kMmNB + b) * 32 + r;
        qidx[b] = qi;
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (qi < nq) {
            const size_t row = qs.base + (size_t)((!PAIRS && a.ids) ? a.ids[qi] : qi);
            q = reinterpret_cast<const uint4*>(qs.desc + row * 32)[h];
        }
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
        int pc = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            pc += __builtin_popcount(w[d]);
            bq[b][2 * d] = expand16(w[d] & 0xffffu, 0xc0u);
            bq[b][2 * d + 1] = expand16(w[d] >> 16, 0xc0u);
        }
        pq[b] = pc + __shfl_xor(pc, 32, 64);
    }
    // ninth step, query side: 32 x the four per-dword popcounts of the train side, 1 x its row number.
    // Both lane halves add their popcounts; only half 0 adds the row.
    const v4i b_tail = {0x20202020, h == 0 ? 1 : 0, 0, 0};

    // ---- train tile staging: thread (d, tr) expands dword d of train rows tr and tr + 32 ----
    const int sd = tid >> 5, tr = tid & 31;
    const int sh = sd >> 2, sdd = sd & 3;
    auto load_tile = [&](int t0, unsigned (&w)[2]) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int t = t0 + k * kMmSub + tr;
            w[k] = t < t_end ? reinterpret_cast<const unsigned*>(a.tdesc + (tbase + t) * 32)[sd] : 0u;
        }
    };
    auto stage_tile = [&](int buf, int t0, const unsigned (&w)[2]) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            // rows beyond the set: all bits 0 and a popcount of 127 per dword, so that the row
            // loses against every real one
            const bool valid = t0 + k * kMmSub + tr < t_end;
            uint4* dst = s_tile[buf] + k * kMmSubU4;
            const v4i lo = expand16(w[k] & 0xffffu, 1u), hi = expand16(w[k] >> 16, 1u);
            dst[((2 * sdd) * 2 + sh) * 32 + tr] = make_uint4((unsigned)lo.x, (unsigned)lo.y, (unsigned)lo.z, (unsigned)lo.w);
            dst[((2 * sdd + 1) * 2 + sh) * 32 + tr] = make_uint4((unsigned)hi.x, (unsigned)hi.y, (unsigned)hi.z, (unsigned)hi.w);
            reinterpret_cast<uint8_t*>(dst + (8 * 2 + sh) * 32 + tr)[sdd] = (uint8_t)(valid ? __builtin_popcount(w[k]) : 127);
        }
    };

    int best[kMmNB], second[kMmNB];
#pragma unroll
    for (int b = 0; b < kMmNB; b++) best[b] = second[b] = 0x7fffffff;
    // 32 Hrel + row  ->  Hrel << 20 | block * 32 + row: order (distance, train index)
    auto block_key = [](int m, int blockno) { return m * (1 << kMmBlockBits) - (m & 31) * ((1 << kMmBlockBits) - 1) + blockno; };

    // ninth step, train side: bytes 0-3 per-dword popcounts (written per tile), byte 4 the row number,
    // bytes 5-15 zero (they meet zeros of the query side; they only have to be defined)
    if (tid < 128) {
        const uint4 init = make_uint4(0u, (unsigned)(tid & 31), 0u, 0u);
#pragma unroll
        for (int bu = 0; bu < 2; bu++) {
            s_tile[bu][8 * 64 + (tid & 63)] = init;            // rows 0-31 of the tile (tid & 63 = half * 32 + row)
            s_tile[bu][kMmSubU4 + 8 * 64 + (tid & 63)] = init; // rows 32-63
        }
    }
    lds_barrier();
    {
        unsigned w0[2];
        load_tile(t_begin, w0);
        stage_tile(0, t_begin, w0);
    }
    lds_barrier();

    int buf = 0;
    int blockno = 0;
    // LDS byte address of this lane's operand slice of buffer 0, block 0, step 0
    const unsigned tile_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint4*)&s_tile[0][0] + (unsigned)((h * 32 + r) * 16);
    for (int t0 = t_begin; t0 < t_end; t0 += kMmTile) {
        const bool more = t0 + kMmTile < t_end;
        unsigned wnext[2] = {0u, 0u};
        if (more) load_tile(t0 + kMmTile, wnext);
        // The nine train-side operands of a block of 32 rows are read ahead of its products: those of
        // the tile's first block before any product, those of the second block one by one as the first
        // block's registers fall free.  (Read just before use, every pair of products waits for the
        // LDS round trip and the matrix pipe runs dry in between.)  The reads and the waits for them
        // are written out because the compiler would pair each read with its use again; LDS operations
        // complete in order, so "at most n still outstanding" covers the one a product needs, and
        // any LDS operation the compiler puts in between only makes a wait stricter.
        v4i af[kMmSteps];
        const unsigned a_addr = tile_addr + (unsigned)buf * (unsigned)(2 * kMmSubU4 * 16);
#pragma unroll
        for (int s = 0; s < kMmSteps; s++)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[s]) : "v"(a_addr), "n"(s * 1024));
#pragma unroll
        for (int k = 0; k < 2; k++) {
            v16i acc[kMmNB];
#pragma unroll
            for (int b = 0; b < kMmNB; b++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[b][i] = 0;
#pragma unroll
            for (int s = 0; s < kMmSteps; s++) {
                if (k == 0) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(af[s]));
                else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af[s]) : "n"(kMmSteps - 1 - s));
#pragma unroll
                for (int b = 0; b < kMmNB; b++)
                    acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s], s < 8 ? bq[b][s] : b_tail, acc[b], 0, 0, 0);
                if (k == 0) {
                    // the same step of the second block
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[s]) : "v"(a_addr), "n"(kMmSubU4 * 16 + s * 1024));
                }
            }
#pragma unroll
            for (int b = 0; b < kMmNB; b++) {
                if (SECOND) {
                    // the two smallest accumulators of the lane: sorted pairs, merged pairwise
                    int lo[8], hi[8];
#pragma unroll
                    for (int p = 0; p < 8; p++) { lo[p] = min(acc[b][2 * p], acc[b][2 * p + 1]); hi[p] = max(acc[b][2 * p], acc[b][2 * p + 1]); }
#pragma unroll
                    for (int w = 4; w >= 1; w >>= 1)
#pragma unroll
                        for (int p = 0; p < w; p++) {
                            const int l2 = min(lo[p], lo[p + w]);
                            hi[p] = min(max(lo[p], lo[p + w]), min(hi[p], hi[p + w]));
                            lo[p] = l2;
                        }
                    const int k1 = block_key(lo[0], blockno), k2 = block_key(hi[0], blockno);
                    // best <= second and k1 < k2: the two smallest of the four
                    second[b] = min(max(best[b], k1), min(second[b], k2));
                    best[b] = min(best[b], k1);
                } else {
                    int m = acc[b][0];
#pragma unroll
                    for (int i = 1; i < 16; i += 2) m = min(min(m, acc[b][i]), acc[b][i + 1 < 16 ? i + 1 : i]);
                    best[b] = min(best[b], block_key(m, blockno));
                }
            }
            blockno += kMmSub;
        }
        if (more) stage_tile(buf ^ 1, t0 + kMmTile, wnext);
        lds_barrier();
        buf ^= 1;
    }

#pragma unroll
    for (int b = 0; b < kMmNB; b++) {
        const int ob = __shfl_xor(best[b], 32, 64);
        const int m = min(best[b], ob);
        // m = (H - popcount(query)) << 20 | train row relative to t_begin; rows beyond the set carry distances above 256
        auto final_key = [&](int v) -> unsigned {
            const unsigned d = (unsigned)((v >> kMmIdxBits) + pq[b]);   // arithmetic shift: floor
            const unsigned t = (unsigned)(t_begin + (v & ((1 << kMmIdxBits) - 1)));
            return d <= 256u ? ((d << kMmIdxBits) | t) : kMmNoMatch;
        };
        if (SECOND) {
            const int os = __shfl_xor(second[b], 32, 64);
            const int m2 = min(max(best[b], ob), min(second[b], os));
            if (h == 0 && qidx[b] < nq)
                a.part[(size_t)split * a.part_pitch + qbase + qidx[b]] = make_uint2(final_key(m), final_key(m2));
            continue;
        }
        if (h != 0 || qidx[b] >= nq) continue;
        const unsigned d = (unsigned)((m >> kMmIdxBits) + pq[b]);
        const unsigned t = (unsigned)(t_begin + (m & ((1 << kMmIdxBits) - 1)));
        if (PAIRS) {
            a.pair_out[(size_t)prob * a.n_per + qidx[b]] = (d <= 256u && (int)d <= a.dmax) ? ((d << 16) | t) : kMmNoMatch;
        } else if (d <= 256u) {
            atomicMin(&a.best[qbase + qidx[b]], (d << kMmIdxBits) | t);
        }
    }
}

// Workgroups of 256 queries; the train set is split so that the grid fills the chip several
// times over (capacities bound the grid, the true counts live on the device; workgroups beyond
// them exit at once).  Splits are merged by atomicMin on the key -- or, when second-best keys are
// kept, in the finalize kernel over at most kMaxSecondSplits planes.
MatchPlan plan_match_mfma(int batch, int nq_bound, int nt_bound, int n_cu, bool second) {
    MatchPlan p = {0, 0, 1, 0};
    if (nq_bound <= 0 || nt_bound <= 0 || batch <= 0) return p;
    p.qblocks = div_up(nq_bound, kMmQueries);
    // Train rows are split over `splits` workgroups per query block.  Two workgroups fit a CU: a launch whose
    // query blocks alone fill the chip is not split (every split expands the block's queries again: 64 frames
    // x 8 blocks measured 55 us unsplit, 56.6 in two, 58.4 in three); a smaller one is split until there are
    // about three workgroups per CU (one 4K frame, 40 blocks: 38 us in 20 splits, 51 in 12 or 13, 40 in 39).
    const int wgs = p.qblocks * batch;
    int splits = wgs >= 2 * n_cu ? 1 : div_up(3 * n_cu, wgs);
    splits = std::max(1, std::min(splits, div_up(nt_bound, 4 * kMmTile)));
    if (second) splits = std::min(splits, kMaxSecondSplits);
    p.chunk = div_up(div_up(nt_bound, splits), kMmTile) * kMmTile;
    p.splits = div_up(nt_bound, p.chunk);
    return p;
}

int launch_match_mfma(MatchArgs a, const MatchPlan& p, int batch, hipStream_t st) {
    if (p.qblocks <= 0 || batch <= 0) return SLAM_AMD_OK;
    a.chunk = p.chunk;
    a.splits = p.splits;
    a.qblocks = p.qblocks;
    a.batch = batch;
    const long long blocks = (long long)p.qblocks * p.splits * batch;
    if (blocks > 0x7fffffffLL) SA_FAIL(SLAM_AMD_ERR_ARG, "match: problem too large for one launch");
    if (a.part) hipLaunchKernelGGL((match_mfma_kernel<false, true>), dim3((unsigned)blocks), dim3(kMmBlock), 0, st, a);
    else hipLaunchKernelGGL((match_mfma_kernel<false, false>), dim3((unsigned)blocks), dim3(kMmBlock), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

int launch_match_pairs_mfma(MatchArgs a, int n_pairs, hipStream_t st) {
    if (n_pairs <= 0 || a.n_per <= 0) return SLAM_AMD_OK;
    a.qblocks = div_up(a.n_per, kMmQueries);
    a.chunk = 0;
    a.splits = 1;
    const long long blocks = (long long)a.qblocks * n_pairs;
    if (blocks > 0x7fffffffLL) SA_FAIL(SLAM_AMD_ERR_ARG, "match_pairs: too many pairs for one launch");
    hipLaunchKernelGGL((match_mfma_kernel<true, false>), dim3((unsigned)blocks), dim3(kMmBlock), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

}  // namespace sa
