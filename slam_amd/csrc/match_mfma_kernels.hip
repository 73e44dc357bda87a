// Brute-force Hamming matching on the matrix cores of gfx950, with the reference's gates.
//
// All-pairs Hamming distance is a matrix product: with the 256 descriptor bits of a train row
// as a 0/1 vector t and those of a query as b,
//     H(t, b) = popcount(b) + sum_i t_i (1 - 2 b_i)
// v_mfma_i32_32x32x32_i8 forms the sum for a 32 x 32 block of (train, query) pairs in eight
// steps of 32 bits: the train bits are expanded to bytes {0, 1}, the query bits to bytes
// {+32 (bit clear), -32 (bit set)}.  The first product of a block takes a constant accumulator
// operand, 32 * 256 + (row of the train inside its block of 32), so that an accumulator ends as
//     32 (H - popcount(b) + 256) + row  >= 0:
// the order of (distance, train index) inside the block, and the arg-min over a block is a plain
// unsigned minimum over accumulators.  Across blocks the running best is the minimum of
// (H - popcount(b) + 256) << 20 | train index, rebuilt from the block's minimum with three
// instructions.  popcount(b) is constant per query and is added after the arg-min.  Ties go to
// the lowest train index, as in the popcount kernel.
//
// Gates (_filterMatches of the reference: /root/reference/src/frame/matcher.cpp:144-157) act on
// the accumulators before the minimum: a pair that fails sets bit 15 of its accumulator, which
// puts its distance above 256 -- it loses against every pair that passes, and a query whose best
// pair carries the bit has no match.  The octave gate costs nothing for a block of train rows
// without a row above max_level or a block of queries without one (a 32-bit mask per block of
// train rows, tested by the whole wave); the window gate compares positions per pair (train
// positions of the block from LDS, the query's own in a register).  Rows past the end of the
// train set are marked the same way.
//
// Layout of one product: train rows are the M side (A operand), queries the N side (B operand).
// An accumulator register of lane l belongs to query l & 31, so the arg-min over the train rows
// runs inside the lane; the two lane halves hold different train rows and are merged once at
// the end.  Queries stay in registers for the whole kernel (expanded once); train tiles of 64
// rows are expanded cooperatively into LDS, double buffered.
//
// The K order inside a step does not matter as long as both operands use the same one: lane
// half h of either operand supplies the bits of bytes 16h .. 16h + 15 of its descriptor (expand32).
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
constexpr int kMmSteps = 8;                   // steps of 32 descriptor bits per lane half
constexpr int kMmSubU4 = kMmSteps * 64;       // uint4 slots of one expanded block of 32 rows: [step][half][row]
constexpr unsigned kMmNoMatch = 0xffffffffu;
constexpr int kMmIdxBits = 20;
constexpr int kMmBlockBits = 15;              // kMmIdxBits minus the 5 row bits an accumulator carries
constexpr int kMmBias = 32 * 256;             // keeps an accumulator non-negative: H - popcount(b) >= -256
constexpr unsigned kMmFail = 1u << 15;        // set in the accumulator of a pair that fails a gate (distance + 1024)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// One descriptor dword -> the 32 bytes of its bits, as two operand slices of 16 bytes (steps 2d and 2d + 1 of
// the lane half that owns dword d): byte j of output dword k is bit k + 8 j of the input, which takes one
// shift and one mask per output dword.  A byte is `zero` where the bit is clear and `zero + scale` (mod 256)
// where it is set.  Both operands use this order, so the products pair equal bits.
template <unsigned SCALE, unsigned ZERO>
__device__ __forceinline__ void expand32(unsigned w, v4i& lo, v4i& hi) {
    unsigned o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        o[k] = (w >> k) & 0x01010101u;
        if (SCALE != 1u || ZERO != 0u) o[k] = o[k] * SCALE + ZERO;
    }
    lo.x = (int)o[0]; lo.y = (int)o[1]; lo.z = (int)o[2]; lo.w = (int)o[3];
    hi.x = (int)o[4]; hi.y = (int)o[5]; hi.z = (int)o[6]; hi.w = (int)o[7];
}

#ifdef SA_MM_STAMPS
// Diagnostic build only: per wave, the shader clock at four points of the kernel, the hardware id and the real-time clock.
__device__ unsigned long long g_mm_stamps[8192][6];
#define SA_STAMP(slot) do { if (lane == 0) g_mm_stamps[(blockIdx.x * 4 + wave) & 8191][slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SA_STAMP(slot) do { } while (0)
#endif

// Train row inside its block of 32 that accumulator register i of lane half h belongs to.
__device__ __forceinline__ constexpr int acc_row(int i, int h) { return 8 * (i >> 2) + 4 * h + (i & 3); }

// SECOND: also keep the second smallest key (for the ratio test); every split then stores its
// (best, second) pair and the finalize kernel merges the splits.
// AREA / LEVEL: the window and octave gates.
template <bool PAIRS, bool SECOND, bool AREA, bool LEVEL>
__global__ __launch_bounds__(kMmBlock, 2) void match_mfma_kernel(MatchArgs a) {
    __shared__ uint4 s_tile[3][2 * kMmSubU4];
    // per tile, four deep: one keypoint dword per (staging dword index, row) -- x, y, octave, octave ... -- and per
    // block of 32 rows the mask of the rows with octave > max_level (as wave 1 sees it; the other slots are unused)
    __shared__ __align__(16) unsigned s_meta[(AREA || LEVEL) ? 4 : 1][(AREA || LEVEL) ? 8 : 1][(AREA || LEVEL) ? kMmTile : 4];
    __shared__ unsigned s_high[4][2][4];

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
    SA_STAMP(0);
#ifdef SA_MM_STAMPS
    if (lane == 0) {
        g_mm_stamps[(blockIdx.x * 4 + wave) & 8191][4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        g_mm_stamps[(blockIdx.x * 4 + wave) & 8191][5] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    // ---- queries of this wave: expanded once, bytes +32 / -32 ----
    v4i bq[kMmNB][8];
    int pq[kMmNB];      // popcount of the whole query descriptor
    int qidx[kMmNB];
    float qx[kMmNB], qy[kMmNB];
    bool qhigh[kMmNB];
    bool any_qhigh[kMmNB];
#pragma unroll
    for (int b = 0; b < kMmNB; b++) {
        const int qi = qblock * kMmQueries + (wave * kMmNB + b) * 32 + r;
        qidx[b] = qi;
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        qx[b] = qy[b] = 0.f;
        qhigh[b] = false;
        if (qi < nq) {
            const size_t row = qs.base + (size_t)((!PAIRS && a.ids) ? a.ids[qi] : qi);
            q = reinterpret_cast<const uint4*>(qs.desc + row * 32)[h];
            if (AREA) {
                if (!PAIRS && a.q_xy) {
                    qx[b] = a.q_xy[2 * (qbase + qi)];
                    qy[b] = a.q_xy[2 * (qbase + qi) + 1];
                } else {
                    qx[b] = qs.kp[row].x;
                    qy[b] = qs.kp[row].y;
                }
            }
            if (LEVEL) qhigh[b] = qs.kp[row].octave > a.max_level;
        }
        any_qhigh[b] = LEVEL && __ballot(qhigh[b]) != 0ull;
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
        int pc = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            pc += __builtin_popcount(w[d]);
            expand32<0xc0u, 0x20202020u>(w[d], bq[b][2 * d], bq[b][2 * d + 1]);
        }
        pq[b] = pc + __shfl_xor(pc, 32, 64);
    }
    // accumulator operand of a block's first product: the bias and the row an accumulator belongs to
    v16i c_init;
#pragma unroll
    for (int i = 0; i < 16; i++) c_init[i] = kMmBias + acc_row(i, h);

    // ---- train tile staging: thread (d, tr) expands dword d of train rows tr and tr + 32 of a tile.  With a gate on
    //      it also fetches one dword of either row's keypoint: x (d = 0), y (d = 1) or the octave (d >= 2) ----
    const int sd = tid >> 5, tr = tid & 31;
    const int sh = sd >> 2, sdd = sd & 3;
    constexpr bool kMeta = AREA || LEVEL;
    const int meta_dword = sd < 2 ? sd : 5;        // offset of x, y, octave in a keypoint record, in dwords
    struct Staged { unsigned w[2]; unsigned meta[2]; bool valid[2]; };
    // (no branches: rows past the end read the set's last row and are zeroed; 32-bit offsets from bases that are
    // the same in every lane)
    const uint8_t* const tdesc_base = a.tdesc + tbase * 32;
    const uint8_t* const tkp_base = reinterpret_cast<const uint8_t*>(a.tkp + tbase);
    const unsigned last_row = (unsigned)(t_end - 1);
    auto load_row = [&](int t0, int k, Staged& s) {
        const unsigned t = (unsigned)(t0 + k * kMmSub + tr);
        s.valid[k] = t <= last_row;
        const unsigned row = min(t, last_row);
        s.w[k] = *reinterpret_cast<const unsigned*>(tdesc_base + (row * 32u + (unsigned)sd * 4u));
        s.meta[k] = kMeta ? *reinterpret_cast<const unsigned*>(tkp_base + (row * (unsigned)sizeof(slam_amd_keypoint) + (unsigned)meta_dword * 4u)) : 0u;
    };
    auto load_tile = [&](int t0, Staged& s) {
        load_row(t0, 0, s);
        load_row(t0, 1, s);
    };
    // expanded rows: [block][step][half][row] of 16 bytes; this thread's two slices of block k
    auto stage_slot = [&](int buf, int k, int hi) { return &s_tile[buf][k * kMmSubU4 + ((2 * sdd + hi) * 2 + sh) * 32 + tr]; };
    auto stage_meta = [&](int mb, const Staged& s) {
        if (!kMeta) return;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            s_meta[mb][sd][k * kMmSub + tr] = s.meta[k];
            if (LEVEL) {
                // lanes 0-31 of wave 1 hold the octaves of the block's rows (d = 2)
                const unsigned long long mask = __ballot(sd == 2 && s.valid[k] && (int)s.meta[k] > a.max_level);
                s_high[mb][k][wave] = (unsigned)mask;
            }
        }
    };
    auto stage_tile = [&](int buf, const Staged& s) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            v4i lo, hi;
            expand32<1u, 0u>(s.valid[k] ? s.w[k] : 0u, lo, hi);
            *stage_slot(buf, k, 0) = make_uint4((unsigned)lo.x, (unsigned)lo.y, (unsigned)lo.z, (unsigned)lo.w);
            *stage_slot(buf, k, 1) = make_uint4((unsigned)hi.x, (unsigned)hi.y, (unsigned)hi.z, (unsigned)hi.w);
        }
    };
    // this lane's operand slice of block k, step s of a buffer
    auto operand = [&](int buf, int k, int s) { return *reinterpret_cast<const v4i*>(&s_tile[buf][k * kMmSubU4 + s * 64 + h * 32 + r]); };

    unsigned best[kMmNB], second[kMmNB];
#pragma unroll
    for (int b = 0; b < kMmNB; b++) best[b] = second[b] = kMmNoMatch;
    // ---- the arg-min over a block's accumulators, cut into sixteen parts so that it can be issued between the
    //      products of the next block: parts 0-7 take accumulators 2n and 2n + 1 of both query blocks, parts
    //      8-11 and 12-15 fold the minimum of query block 0 and 1 (two minima each with SECOND) into the
    //      running keys: 32 Hb + row  ->  Hb << 20 | block * 32 + row, the order (distance, train index).
    //      An accumulator is below 2^16. ----
    unsigned run_lo[kMmNB], run_hi[kMmNB], key_lo[kMmNB], key_hi[kMmNB];
    auto argmin_part = [&](const v16i (&acc)[kMmNB], int n, int blockno) {
        if (n < 8) {
#pragma unroll
            for (int b = 0; b < kMmNB; b++) {
                const unsigned x = (unsigned)acc[b][2 * n], y = (unsigned)acc[b][2 * n + 1];
                if (SECOND) {
                    // the two smallest so far (run_lo <= run_hi) merged with the sorted pair
                    const unsigned lo = min(x, y), hi = max(x, y);
                    if (n == 0) { run_lo[b] = lo; run_hi[b] = hi; }
                    else {
                        run_hi[b] = min(max(run_lo[b], lo), min(run_hi[b], hi));
                        run_lo[b] = min(run_lo[b], lo);
                    }
                } else {
                    run_lo[b] = n == 0 ? min(x, y) : min(min(run_lo[b], x), y);
                }
            }
            return;
        }
        const int b = (n - 8) >> 2, part = (n - 8) & 3;
        if (part == 0) {
            key_lo[b] = run_lo[b] & 31u;
            if (SECOND) key_hi[b] = run_hi[b] & 31u;
        } else if (part == 1) {
            key_lo[b] = (unsigned)blockno - key_lo[b] * ((1u << kMmBlockBits) - 1u);
            if (SECOND) key_hi[b] = (unsigned)blockno - key_hi[b] * ((1u << kMmBlockBits) - 1u);
        } else if (part == 2) {
            key_lo[b] += run_lo[b] << kMmBlockBits;
            if (SECOND) key_hi[b] += run_hi[b] << kMmBlockBits;
        } else {
            // best <= second and key_lo < key_hi: the two smallest of the four
            if (SECOND) second[b] = min(max(best[b], key_lo[b]), min(second[b], key_hi[b]));
            best[b] = min(best[b], key_lo[b]);
        }
    };

    // ---- gates: mark the accumulators of the pairs that fail.  `rows_left`: rows of the block inside the set ----
    auto apply_gates = [&](v16i (&acc)[kMmNB], int mb, int k, int rows_left) {
        if (rows_left < kMmSub) {
#pragma unroll
            for (int b = 0; b < kMmNB; b++)
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (acc_row(i, h) >= rows_left) acc[b][i] |= (int)kMmFail;
        }
        if (LEVEL) {
            const unsigned high = (unsigned)__builtin_amdgcn_readfirstlane((int)s_high[mb][k][1]);
#pragma unroll
            for (int b = 0; b < kMmNB; b++) {
                if (high != 0u && any_qhigh[b]) {
                    // both octaves above max_level: bit acc_row(i, h) of the block's mask, for the queries that are high
                    const unsigned mine = qhigh[b] ? high >> (4 * h) : 0u;
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        acc[b][i] |= (int)(((mine >> acc_row(i, 0)) & 1u) << 15);
                }
            }
        }
        if (AREA) {
            const float4* px = reinterpret_cast<const float4*>(&s_meta[mb][0][k * kMmSub + 4 * h]);
            const float4* py = reinterpret_cast<const float4*>(&s_meta[mb][1][k * kMmSub + 4 * h]);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 tx4 = px[2 * j], ty4 = py[2 * j];
                const float tx[4] = {tx4.x, tx4.y, tx4.z, tx4.w}, ty[4] = {ty4.x, ty4.y, ty4.z, ty4.w};
#pragma unroll
                for (int b = 0; b < kMmNB; b++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float dx = qx[b] - tx[c], dy = qy[b] - ty[c];
                        if (!(fabsf(dx) < a.area && fabsf(dy) < a.area)) acc[b][4 * j + c] |= (int)kMmFail;
                    }
            }
        }
    };

    // ---- one block: 8 x kMmNB products, with a slice of other work issued behind each of them.  The slices
    //      must not depend on this block's products; fenced so that the compiler keeps them where they are:
    //      a wave issues in order, and about four vector instructions fit under one product ----
    auto products = [&](v16i (&acc)[kMmNB], const v4i (&af)[kMmSteps], auto&& slice) {
#pragma unroll
        for (int s = 0; s < kMmSteps; s++)
#pragma unroll
            for (int b = 0; b < kMmNB; b++) {
                acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s], bq[b][s], s == 0 ? c_init : acc[b], 0, 0, 0);
                slice(s * kMmNB + b);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    constexpr int kSlots = kMmSteps * kMmNB;          // slices per block
    static_assert(kMmNB == 2, "the slices are laid out for two query blocks per wave");

    // Tile T lies in operand buffer T % 3 and metadata buffer T % 4, and is expanded two tiles ahead: while the
    // products of tile T are issued, its two blocks' worth of slices also read the operands of the next block
    // (block 1 of tile T, then block 0 of tile T + 1, both staged at least one barrier ago), finish the arg-min of
    // the block before, and expand one row each of tile T + 2 (loaded from memory a tile ago) into the third
    // buffer.  One barrier per tile, at its end, and nothing behind it waits for LDS: the next block's operands are
    // in registers already.  What a block's gates need of its tile's metadata is read up to one tile later, hence
    // four metadata buffers.
    const int n_tiles = (t_end - t_begin + kMmTile - 1) / kMmTile;
    Staged g;
    load_tile(t_begin, g);
    stage_tile(0, g);
    stage_meta(0, g);
    load_tile(t_begin + kMmTile, g);            // tiles past the end are zeros
    stage_tile(1, g);
    stage_meta(1, g);
    load_tile(t_begin + 2 * kMmTile, g);
    lds_barrier();

    v4i af0[kMmSteps], af1[kMmSteps];
#pragma unroll
    for (int s = 0; s < kMmSteps; s++) af0[s] = operand(0, 0, s);
    v16i acc0[kMmNB], acc1[kMmNB];
#pragma unroll
    for (int b = 0; b < kMmNB; b++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc1[b][i] = 0xffff;     // "the block before the first": every pair failed

    SA_STAMP(1);
    int cur = 0, nxt = 1, nx2 = 2;            // operand buffers of tiles T, T + 1, T + 2
    int mb = 0, mb_prev = 0, mb_nx2 = 2;      // metadata buffers of tiles T, T - 1, T + 2
    int blockno = 0;
    int rows_left = t_end - t_begin;
    for (int T = 0; T + 1 < n_tiles; T++) {
        if (kMeta) apply_gates(acc1, mb_prev, 1, kMmSub);
        stage_meta(mb_nx2, g);
        unsigned o[4];
        // output dword n - 8 of this thread's row of a block of tile T + 2 (the empty statement keeps the two
        // instructions in their slice: the compiler would otherwise gather them in front of the LDS write)
        auto expand_part = [&](unsigned gw, int n, uint4* w0, uint4* w1) {
            if (n < 8) return;
            o[n & 3] = (gw >> (n - 8)) & 0x01010101u;
            asm volatile("" : "+v"(o[n & 3]));
            if ((n & 3) == 3) *(n < 12 ? w0 : w1) = make_uint4(o[0], o[1], o[2], o[3]);
        };
        const unsigned gw0 = g.valid[0] ? g.w[0] : 0u, gw1 = g.valid[1] ? g.w[1] : 0u;
        {
            uint4* const w0 = stage_slot(nx2, 0, 0); uint4* const w1 = stage_slot(nx2, 0, 1);
            products(acc0, af0, [&](int n) {
                if (n < kMmSteps) af1[n] = operand(cur, 1, n);
                argmin_part(acc1, n, blockno - kMmSub);
                expand_part(gw0, n, w0, w1);
                if (n == 15) load_row(t_begin + (T + 3) * kMmTile, 0, g);
            });
        }
        if (kMeta) apply_gates(acc0, mb, 0, kMmSub);
        {
            uint4* const w0 = stage_slot(nx2, 1, 0); uint4* const w1 = stage_slot(nx2, 1, 1);
            products(acc1, af1, [&](int n) {
                if (n < kMmSteps) af0[n] = operand(nxt, 0, n);
                argmin_part(acc0, n, blockno);
                expand_part(gw1, n, w0, w1);
                if (n == 15) load_row(t_begin + (T + 3) * kMmTile, 1, g);
            });
        }
        lds_barrier();
        blockno += kMmTile;
        rows_left -= kMmTile;
        { const int t = cur; cur = nxt; nxt = nx2; nx2 = t; }
        mb_prev = mb;
        mb = (mb + 1) & 3;
        mb_nx2 = (mb_nx2 + 1) & 3;
    }
    SA_STAMP(2);
    // the last tile: nothing to stage; its blocks may be cut short by the end of the set
    {
        if (kMeta) apply_gates(acc1, mb_prev, 1, kMmSub);
        products(acc0, af0, [&](int n) {
            if (n < kMmSteps) af1[n] = operand(cur, 1, n);
            argmin_part(acc1, n, blockno - kMmSub);
        });
        apply_gates(acc0, mb, 0, rows_left);
        products(acc1, af1, [&](int n) { argmin_part(acc0, n, blockno); });
        apply_gates(acc1, mb, 1, rows_left - kMmSub);
#pragma unroll
        for (int n = 0; n < kSlots; n++) argmin_part(acc1, n, blockno + kMmSub);
    }
#pragma unroll
    for (int b = 0; b < kMmNB; b++) {
        const unsigned ob = (unsigned)__shfl_xor((int)best[b], 32, 64);
        const unsigned m = min(best[b], ob);
        // m = (H - popcount(query) + 256) << 20 | train row relative to t_begin; pairs that failed a gate and rows
        // beyond the set carry distances above 256
        auto final_key = [&](unsigned v) -> unsigned {
            const unsigned d = (v >> kMmIdxBits) + (unsigned)pq[b] - 256u;
            const unsigned t = (unsigned)t_begin + (v & ((1u << kMmIdxBits) - 1u));
            return d <= 256u ? ((d << kMmIdxBits) | t) : kMmNoMatch;
        };
        if (SECOND) {
            const unsigned os = (unsigned)__shfl_xor((int)second[b], 32, 64);
            const unsigned m2 = min(max(best[b], ob), min(second[b], os));
            if (h == 0 && qidx[b] < nq)
                a.part[(size_t)split * a.part_pitch + qbase + qidx[b]] = make_uint2(final_key(m), final_key(m2));
            continue;
        }
        if (h != 0 || qidx[b] >= nq) continue;
        const unsigned key = final_key(m);
        if (PAIRS) {
            const unsigned d = key >> kMmIdxBits;
            a.pair_out[(size_t)prob * a.n_per + qidx[b]] =
                (key != kMmNoMatch && (int)d <= a.dmax) ? ((d << 16) | (key & ((1u << kMmIdxBits) - 1u))) : kMmNoMatch;
        } else if (key != kMmNoMatch) {
            atomicMin(&a.best[qbase + qidx[b]], key);
        }
    }
    SA_STAMP(3);
}

#ifdef SA_MM_STAMPS
extern "C" int slam_amd_debug_mm_stamps(unsigned long long* out, int n_waves) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mm_stamps), (size_t)n_waves * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

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

template <bool PAIRS, bool SECOND>
static void launch_mfma_t(const MatchArgs& a, dim3 grid, hipStream_t st) {
    const bool area = a.area != -1.f, level = a.max_level != -1;
    if (area && level) hipLaunchKernelGGL((match_mfma_kernel<PAIRS, SECOND, true, true>), grid, dim3(kMmBlock), 0, st, a);
    else if (area) hipLaunchKernelGGL((match_mfma_kernel<PAIRS, SECOND, true, false>), grid, dim3(kMmBlock), 0, st, a);
    else if (level) hipLaunchKernelGGL((match_mfma_kernel<PAIRS, SECOND, false, true>), grid, dim3(kMmBlock), 0, st, a);
    else hipLaunchKernelGGL((match_mfma_kernel<PAIRS, SECOND, false, false>), grid, dim3(kMmBlock), 0, st, a);
}

int launch_match_mfma(MatchArgs a, const MatchPlan& p, int batch, hipStream_t st) {
    if (p.qblocks <= 0 || batch <= 0) return SLAM_AMD_OK;
    a.chunk = p.chunk;
    a.splits = p.splits;
    a.qblocks = p.qblocks;
    a.batch = batch;
    const long long blocks = (long long)p.qblocks * p.splits * batch;
    if (blocks > 0x7fffffffLL) SA_FAIL(SLAM_AMD_ERR_ARG, "match: problem too large for one launch");
    if (a.part) launch_mfma_t<false, true>(a, dim3((unsigned)blocks), st);
    else launch_mfma_t<false, false>(a, dim3((unsigned)blocks), st);
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
    launch_mfma_t<true, false>(a, dim3((unsigned)blocks), st);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

}  // namespace sa
