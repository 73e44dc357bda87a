// Brute-force Hamming matching on the matrix cores of gfx950, with the reference's gates.
//
// All-pairs Hamming distance is a matrix product: with the 256 descriptor bits of a train row
// as a 0/1 vector t and those of a query as b,
//     H(t, b) = popcount(b) + sum_i t_i (1 - 2 b_i)
// v_mfma_scale_f32_32x32x64_f8f6f4 on 4-bit operands (FP4, E2M1) forms the sum for a 32 x 32 block of
// (train, query) pairs in four products of 64 bits -- twice the rate of the 8-bit integer product, and
// half the operand bytes.  A train bit becomes the nibble 0b0001 (0.5) or 0, a query bit 0b0010 (+1) where
// it is clear and 0b1010 (-1) where it is set, and the query side carries the block scale 2^6, so every
// term is 0 or +-32 and the fp32 accumulation is exact (integers far below 2^24).  The first product of a
// block takes a constant accumulator operand, 2^15 + 32 * 256 + (row of the train inside its block of 32),
// so that an accumulator ends as the float
//     2^15 + 32 Hb + row,   Hb = H - popcount(b) + 256 in [0, 512]:
// all of them in [2^15, 2^16), where a float's bit pattern is 0x47000000 + (32 Hb + row) << 8.  The bit
// patterns therefore order like (distance, train row inside the block), the arg-min over a block is a plain
// unsigned minimum over accumulators, and the fields are read off the winner's bits without a conversion.
// Across blocks, visited in ascending order, a block's minimum replaces the running one only if its Hb is
// smaller.  popcount(b) is constant per query and is added after the arg-min.  Ties go to the lowest train
// index, as in the popcount kernel.
//
// Gates (_filterMatches of the reference: /root/reference/src/frame/matcher.cpp:144-157) act on
// the accumulators before the minimum: a pair that fails gets the sign bit of its accumulator set (as an
// unsigned pattern above every accumulator of a pair that passes), which puts its distance above 256 -- it loses against every pair that passes, and a query
// whose best pair carries the mark has no match.  The octave gate costs nothing for a block of train rows
// without a row above max_level or a block of queries without one (a 32-bit mask per block of
// train rows, tested by the whole wave); the window gate compares positions per pair (train
// positions of the block from LDS, the query's own in a register).  Rows past the end of the
// train set are marked the same way.
//
// Layout of one product: train rows are the M side (A operand), queries the N side (B operand).
// An accumulator register of lane l belongs to query l & 31, so the arg-min over the train rows
// runs inside the lane; the two lane halves hold different train rows and are merged once at
// the end.  Queries stay in registers for the whole kernel (expanded once); train tiles of 128
// rows are expanded cooperatively into LDS, three buffers deep.
//
// The K order inside a product does not matter as long as both operands use the same one: product s
// takes descriptor dwords 2s and 2s + 1, lane half h of either operand supplies dword 2s + h as the 32
// nibbles of expand_train / expand_query.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace sa {

constexpr int kMmBlock = 256;                 // 4 waves
constexpr int kMmNB = 2;                      // query blocks of 32 per wave
constexpr int kMmQueries = 4 * kMmNB * 32;    // queries per workgroup
constexpr int kMmSub = 32;                    // train rows of one product
#ifndef SA_MM_BT
#define SA_MM_BT 4
#endif
constexpr int kMmBT = SA_MM_BT;               // blocks of 32 train rows per LDS tile (even)
constexpr int kMmTile = kMmSub * kMmBT;       // train rows per LDS tile
constexpr int kMmSteps = 4;                   // products per block: 64 descriptor bits each, 32 per lane half
constexpr int kMmSubU4 = kMmSteps * 64;       // uint4 slots of one expanded block of 32 rows: [step][half][row]
constexpr unsigned kMmNoMatch = 0xffffffffu;
constexpr int kMmIdxBits = 20;
constexpr int kMmBias = 32768 + 32 * 256;     // accumulators in [2^15, 2^16): one binade, H - popcount(b) >= -256
constexpr unsigned kMmBinade = 0x47000000u;   // bit pattern of 32768.f; an accumulator's is kMmBinade + (32 Hb + row) << 8
constexpr unsigned kMmRowBits = 31u << 8;
constexpr unsigned kMmFail = 1u << 31;        // set in the bits of the accumulator of a pair that fails a gate: as unsigned patterns, above all that pass
constexpr int kMmScaleOne = 127, kMmScale64 = 133;   // E8M0 block scales 2^0 and 2^6

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// One descriptor dword -> the 32 nibbles of its bits (one product's operand of one lane): nibble j of output
// dword k is bit k + 4 j of the input, one shift and one mask per output dword.  Train: 0b0001 = 0.5 where the
// bit is set; query: 0b0010 = +1 where it is clear, 0b1010 = -1 where it is set.
__device__ __forceinline__ uint4 expand_train(unsigned w) {
    return make_uint4(w & 0x11111111u, (w >> 1) & 0x11111111u, (w >> 2) & 0x11111111u, (w >> 3) & 0x11111111u);
}
__device__ __forceinline__ v4i expand_query(unsigned w) {
    v4i r;
    r.x = (int)(0x22222222u | ((w & 0x11111111u) << 3));
    r.y = (int)(0x22222222u | (((w >> 1) & 0x11111111u) << 3));
    r.z = (int)(0x22222222u | (((w >> 2) & 0x11111111u) << 3));
    r.w = (int)(0x22222222u | (((w >> 3) & 0x11111111u) << 3));
    return r;
}

// train (A, 0 / 0.5) x query (B, +-1, block scale 2^6), FP4 both: acc + 32 * (set train bits over clear query bits
// minus set train bits over set query bits)
__device__ __forceinline__ v16f product(const v4i& ta, const v4i& qb, const v16f& acc, int scale_a, int scale_b) {
    const v8i A = {ta.x, ta.y, ta.z, ta.w, 0, 0, 0, 0}, B = {qb.x, qb.y, qb.z, qb.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, scale_a, 0, scale_b);
}

// bit pattern of accumulator i (the element is copied first: a bit cast of a vector element in place reads element 0)
__device__ __forceinline__ unsigned acc_bits(const v16f& acc, int i) { const float f = acc[i]; return __float_as_uint(f); }
__device__ __forceinline__ void acc_fail(v16f& acc, int i) { const float f = acc[i]; acc[i] = __uint_as_float(__float_as_uint(f) | kMmFail); }

#ifdef SA_MM_STAMPS
// Diagnostic build only: per wave, the shader clock at four points of the kernel, the hardware id and the real-time clock.
__device__ unsigned long long g_mm_stamps[8192][10];      // 0-3: start, loop, last tile, end; 4: hardware id; 5: real time; 6-9: steps of the prologue
#define SA_STAMP(slot) do { if (lane == 0) g_mm_stamps[(blockIdx.x * 4 + wave) & 8191][slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SA_STAMP(slot) do { } while (0)
#endif

// Train row inside its block of 32 that accumulator register i of lane half h belongs to.
__device__ __forceinline__ constexpr int acc_row(int i, int h) { return 8 * (i >> 2) + 4 * h + (i & 3); }

// SECOND: also keep the second smallest key (for the ratio test); every split then stores its
// (best, second) pair and the finalize kernel merges the splits.
// AREA / LEVEL: the window and octave gates.
#ifndef SA_MM_UNGATED_WGS
#define SA_MM_UNGATED_WGS 3     // workgroups per CU of the variants without a gate (timing experiments: 2)
#endif
template <bool PAIRS, bool SECOND, bool AREA, bool LEVEL>
__global__ __launch_bounds__(kMmBlock, (AREA || LEVEL) ? 2 : SA_MM_UNGATED_WGS) void match_mfma_kernel(MatchArgs a) {
    __shared__ uint4 s_tile[3][kMmBT * kMmSubU4];
    // per tile, four deep: the rows' positions (x, then y) and per block of 32 rows the mask of the rows with
    // octave > max_level
    __shared__ __align__(16) unsigned s_meta[AREA ? 4 : 1][2][AREA ? kMmTile : 4];
    __shared__ unsigned s_high[4][kMmBT];
    __shared__ __align__(16) float s_box[AREA ? 4 : 1][kMmBT][4];      // per block of 32 rows: x min, x max, y min, y max

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
    // (nq and t_end come from counts in device memory; the exit that needs them stands behind the first loads of
    // train rows, which are clamped to the host's row bound instead, so that one memory round trip serves both)
    const int nt_bound = PAIRS ? a.n_per : a.n2;
    if (nt_bound <= 0) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    SA_STAMP(0);
#ifdef SA_MM_STAMPS
    if (lane == 0) {
        g_mm_stamps[(blockIdx.x * 4 + wave) & 8191][4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        g_mm_stamps[(blockIdx.x * 4 + wave) & 8191][5] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    // ---- train tile staging, two shapes that fill the same LDS image.  In the tile loop thread (d, tr) expands dword d of
    //      train rows tr, tr + 32 ... of a tile: four independent dwords, each fetched right after the one before it was
    //      expanded.  The first two tiles, which the prologue fetches and expands in one go, are taken as 16 bytes per
    //      thread instead -- thread (block w = wave, row r, half h) gets dwords 4 h .. 4 h + 3 of row 32 w + r, a wave's
    //      load covers 1 KB without a gap -- because every workgroup of a launch starts at the same time and their loads
    //      queue up in the address units (a quarter of the instructions for the same bytes).  With a gate on, wave w also
    //      fetches what the gates need of block w: lane l the x (l < 32) or y of row l % 32, and its octave ----
    static_assert(kMmBT == 4, "one wave per block of a tile stages the block's keypoint data");
    const int sd = tid >> 5, tr = tid & 31;
    constexpr bool kMeta = AREA || LEVEL;
    struct Staged { unsigned w[kMmBT]; unsigned pos, octave, trow; };
    // (no branches: rows past the bound read the last row below it; rows past the end of the set -- the count in device
    // memory -- are marked as failed where the set ends inside a tile; 32-bit offsets from bases that are the same in
    // every lane)
    const uint8_t* const tdesc_base = a.tdesc + tbase * 32;
    const uint8_t* const tkp_base = reinterpret_cast<const uint8_t*>(a.tkp + tbase);
    const unsigned bound_row = (unsigned)(nt_bound - 1);
    const unsigned my_off = (unsigned)tr * 32u + (unsigned)sd * 4u, bound_off = bound_row * 32u + (unsigned)sd * 4u;
    auto load_row = [&](int t0, int k, Staged& s) {
        s.w[k] = *reinterpret_cast<const unsigned*>(tdesc_base + min(my_off + (unsigned)(t0 + k * kMmSub) * 32u, bound_off));
    };
    const unsigned my_row = (unsigned)(wave * kMmSub + r);
    auto load_rows16 = [&](int t0, Staged& s) {      // w[j]: dword 4 h + j of row 32 wave + r
        const uint4 v = *reinterpret_cast<const uint4*>(tdesc_base + (min((unsigned)t0 + my_row, bound_row) * 32u + (unsigned)h * 16u));
        s.w[0] = v.x; s.w[1] = v.y; s.w[2] = v.z; s.w[3] = v.w;
    };
    auto load_meta = [&](int t0, Staged& s) {
        if (!kMeta) return;
        const unsigned t = (unsigned)(t0 + wave * kMmSub + r);
        const unsigned rec = min(t, bound_row) * (unsigned)sizeof(slam_amd_keypoint);
        s.trow = t;
        if (AREA) s.pos = *reinterpret_cast<const unsigned*>(tkp_base + (rec + (unsigned)h * 4u));
        if (LEVEL) s.octave = *reinterpret_cast<const unsigned*>(tkp_base + (rec + 20u));
    };
    auto load_tile = [&](int t0, Staged& s) {
#pragma unroll
        for (int k = 0; k < kMmBT; k++) load_row(t0, k, s);
        load_meta(t0, s);
    };
    // expanded rows: [block][step][half][row] of 16 bytes; dword d belongs to step d / 2, half d % 2
    auto stage_slot = [&](int buf, int k) { return &s_tile[buf][k * kMmSubU4 + sd * 32 + tr]; };
    auto stage_meta = [&](int mb, const Staged& s) {
        if (AREA) {
            // a position that is NaN or infinite fails the window with every query: stored as +infinity, and a query's as
            // -infinity, so that no difference is ever a NaN (the window test takes a maximum, which would drop one)
            const bool odd = (s.pos & 0x7f800000u) == 0x7f800000u;
            const unsigned pos = odd ? 0x7f800000u : s.pos;
            s_meta[mb][h][wave * kMmSub + r] = pos;
            // the block's box: minimum and maximum over the 32 lanes of either half (x in one, y in the other)
            float lo = __uint_as_float(pos), hi = lo;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) { lo = fminf(lo, __shfl_xor(lo, d, 64)); hi = fmaxf(hi, __shfl_xor(hi, d, 64)); }
            s_box[mb][wave][2 * h] = lo;
            s_box[mb][wave][2 * h + 1] = hi;
        }
        if (LEVEL) {
            // both lane halves hold the block's 32 rows: either half of the ballot is the block's mask
            const unsigned long long mask = __ballot((int)s.trow < t_end && (int)s.octave > a.max_level);
            s_high[mb][wave] = (unsigned)mask;
        }
    };
    auto stage_tile16 = [&](int buf, const Staged& s) {       // a tile fetched by load_rows16
#pragma unroll
        for (int j = 0; j < 4; j++) s_tile[buf][wave * kMmSubU4 + (4 * h + j) * 32 + r] = expand_train(s.w[j]);
    };
    // this lane's operand of block k, product s of a buffer
    auto operand = [&](int buf, int k, int s) { return *reinterpret_cast<const v4i*>(&s_tile[buf][k * kMmSubU4 + s * 64 + h * 32 + r]); };

    // the queries and the first three tiles' rows, requested before anything waits for the counts (query rows past
    // the host's bound repeat the last one below it; rows past the count are dropped at the end)
    const int nq_bound = PAIRS ? a.n_per : (a.ids ? a.n_ids : a.n1);
    if (nq_bound <= 0) return;
    unsigned qw[kMmNB][kMmSteps];      // dword 2s + h of the descriptor for product s
    int qidx[kMmNB];
    float qx[kMmNB], qy[kMmNB];
    int qoct[kMmNB];
#pragma unroll
    for (int b = 0; b < kMmNB; b++) {
        const int qi = qblock * kMmQueries + (wave * kMmNB + b) * 32 + r;
        qidx[b] = qi;
        const int qc = min(qi, nq_bound - 1);
        const size_t row = qs.base + (size_t)((!PAIRS && a.ids) ? a.ids[qc] : qc);
        // 16 bytes of the row per lane half (dwords 4 h .. 4 h + 3); the two dwords the other half needs are swapped below
        const uint4 own = *reinterpret_cast<const uint4*>(qs.desc + row * 32 + 16 * h);
        qw[b][0] = own.x; qw[b][1] = own.y; qw[b][2] = own.z; qw[b][3] = own.w;
        qx[b] = qy[b] = 0.f;
        qoct[b] = 0;
        if (AREA) {
            if (!PAIRS && a.q_xy) {
                qx[b] = a.q_xy[2 * (qbase + qc)];
                qy[b] = a.q_xy[2 * (qbase + qc) + 1];
            } else {
                qx[b] = qs.kp[row].x;
                qy[b] = qs.kp[row].y;
            }
        }
        if (LEVEL) qoct[b] = qs.kp[row].octave;
    }
    // (after the queries: a gather through ids waits for every load before it)
    Staged g, g0, g1;
    load_rows16(t_begin, g0);
    load_meta(t_begin, g0);
    load_rows16(t_begin + kMmTile, g1);
    load_meta(t_begin + kMmTile, g1);
    load_tile(t_begin + 2 * kMmTile, g);
    SA_STAMP(6);      // every load of the prologue is on its way
    if (qblock * kMmQueries >= nq || t_begin >= t_end) return;
    SA_STAMP(7);      // the counts have arrived

    // product s takes dword 2 s + h: half 0 holds dwords 0 .. 3 and needs 0, 2 (its own) and 4, 6 (the other half's
    // first and third); half 1 holds 4 .. 7 and needs 1, 3 (the other half's second and fourth) and 5, 7 (its own)
#pragma unroll
    for (int b = 0; b < kMmNB; b++) {
        const unsigned give0 = h ? qw[b][0] : qw[b][1], give1 = h ? qw[b][2] : qw[b][3];
        const unsigned take0 = (unsigned)__shfl_xor((int)give0, 32, 64), take1 = (unsigned)__shfl_xor((int)give1, 32, 64);
        const unsigned o0 = qw[b][0], o1 = qw[b][1], o2 = qw[b][2], o3 = qw[b][3];
        qw[b][0] = h ? take0 : o0;
        qw[b][1] = h ? take1 : o2;
        qw[b][2] = h ? o1 : take0;
        qw[b][3] = h ? o3 : take1;
    }
    // ---- queries of this wave: expanded once ----
    v4i bq[kMmNB][kMmSteps];
    int pq[kMmNB];      // popcount of the whole query descriptor
    bool qhigh[kMmNB];
    bool any_qhigh[kMmNB];
#pragma unroll
    for (int b = 0; b < kMmNB; b++) {
        qhigh[b] = LEVEL && qidx[b] < nq && qoct[b] > a.max_level;
        if (AREA) {     // see stage_meta
            if ((__float_as_uint(qx[b]) & 0x7f800000u) == 0x7f800000u) qx[b] = __uint_as_float(0xff800000u);
            if ((__float_as_uint(qy[b]) & 0x7f800000u) == 0x7f800000u) qy[b] = __uint_as_float(0xff800000u);
        }
        any_qhigh[b] = LEVEL && __ballot(qhigh[b]) != 0ull;
        int pc = 0;
#pragma unroll
        for (int s = 0; s < kMmSteps; s++) {
            pc += __builtin_popcount(qw[b][s]);
            bq[b][s] = expand_query(qw[b][s]);
        }
        pq[b] = pc + __shfl_xor(pc, 32, 64);
    }
    SA_STAMP(8);      // queries expanded
    // bounding box of this wave's queries (both lane halves hold the same 32 per query block): a block of train rows
    // whose own box lies a window or more away in x or in y cannot pass the window gate with any of them
    float qbox[4] = {0.f, 0.f, 0.f, 0.f};      // x min, x max, y min, y max
    if (AREA) {
        const float inf = __uint_as_float(0x7f800000u);
        qbox[0] = qbox[2] = inf; qbox[1] = qbox[3] = -inf;
#pragma unroll
        for (int b = 0; b < kMmNB; b++)
            if (qidx[b] < nq) {
                qbox[0] = fminf(qbox[0], qx[b]); qbox[1] = fmaxf(qbox[1], qx[b]);
                qbox[2] = fminf(qbox[2], qy[b]); qbox[3] = fmaxf(qbox[3], qy[b]);
            }
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            qbox[0] = fminf(qbox[0], __shfl_xor(qbox[0], d, 64)); qbox[1] = fmaxf(qbox[1], __shfl_xor(qbox[1], d, 64));
            qbox[2] = fminf(qbox[2], __shfl_xor(qbox[2], d, 64)); qbox[3] = fmaxf(qbox[3], __shfl_xor(qbox[3], d, 64));
        }
    }
    // accumulator operand of a block's first product: the bias and the row an accumulator belongs to
    v16f c_init;
#pragma unroll
    for (int i = 0; i < 16; i++) c_init[i] = (float)(kMmBias + acc_row(i, h));
    // block scales of the two operands (byte 0 of a register each)
    int scale_a = kMmScaleOne, scale_b = kMmScale64;
    asm volatile("" : "+v"(scale_a), "+v"(scale_b));     // held in registers, not rebuilt per product

    unsigned best[kMmNB], second[kMmNB];
#pragma unroll
    for (int b = 0; b < kMmNB; b++) best[b] = second[b] = kMmNoMatch;

    // ---- the arg-min over a block's accumulators, cut into sixteen parts so that it can be issued between the
    //      products of the next block: parts 0-7 take accumulators 2n and 2n + 1 of both query blocks (as bit
    //      patterns), parts 8-11 and 12-15 fold the minimum of query block 0 and 1 into the running best.
    //      Without SECOND that is (bit pattern, block number): blocks come in ascending order, so a later block
    //      wins only with a smaller Hb -- its pattern with all row bits set is still below the running one.
    //      With SECOND the two smallest accumulators become keys Hb << 20 | block * 32 + row, the order
    //      (distance, train index), and are merged into the running pair.  A pattern with the failure mark is
    //      clamped to Hb = 1023 there. ----
    unsigned run_lo[kMmNB], run_hi[kMmNB], key_lo[kMmNB], key_hi[kMmNB];
    unsigned best_blk[kMmNB];
#pragma unroll
    for (int b = 0; b < kMmNB; b++) best_blk[b] = 0u;
    auto bits_key = [](unsigned bits, int blockno) {
        const unsigned u = min(bits - kMmBinade, 0x7fffffu);      // Hb << 13 | row << 8
        return ((u & ~0x1fffu) << 7) + ((u >> 8) & 31u) + (unsigned)blockno;
    };
    auto argmin_part = [&](const v16f (&acc)[kMmNB], int n, int blockno) {
        if (n < 8) {
#pragma unroll
            for (int b = 0; b < kMmNB; b++) {
                const unsigned x = acc_bits(acc[b], 2 * n), y = acc_bits(acc[b], 2 * n + 1);
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
        if (!SECOND) {
            if (part == 0) {
                const bool take = (run_lo[b] | kMmRowBits) < best[b];
                best[b] = take ? run_lo[b] : best[b];
                best_blk[b] = take ? (unsigned)blockno : best_blk[b];
            }
            return;
        }
        if (part == 0) key_lo[b] = bits_key(run_lo[b], blockno);
        else if (part == 1) key_hi[b] = bits_key(run_hi[b], blockno);
        else if (part == 2) second[b] = min(max(best[b], key_lo[b]), min(second[b], key_hi[b]));   // best <= second, key_lo < key_hi
        else best[b] = min(best[b], key_lo[b]);
    };

    // ---- gates: mark the accumulators of the pairs that fail ----
    // rows past the end of the set (`rows_left`: rows of the block inside it)
    auto gate_tail = [&](v16f (&acc)[kMmNB], int rows_left) {
        if (rows_left >= kMmSub) return;
#pragma unroll
        for (int b = 0; b < kMmNB; b++)
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (acc_row(i, h) >= rows_left) acc_fail(acc[b], i);
    };
    // The octave and window gates of one block, in four parts (part n: accumulators 4n .. 4n + 3 of both query
    // blocks, the train rows 8n + 4h .. + 3) that go in front of the arg-min parts reading the same accumulators.
    // Window: with m = max(|dx|, |dy|) >= 0 (or NaN), "not (m < area)" is the sign of (bits(area) - 1) - bits(m)
    // in integer arithmetic -- float patterns of non-negative values order like integers, a NaN lies above
    // infinity, a negative area (pattern with the sign bit) fails every pair, as the comparison does.  Octave:
    // bit `row` of the block's mask of high rows, for queries that are high themselves.  Either sign lands in
    // the accumulator's sign bit with one and-or.
    const int area_bits_m1 = (int)__float_as_uint(a.area) - 1;
    typedef const __attribute__((address_space(3))) v4f* LdsRows;
    struct BlockGate { unsigned mine[kMmNB]; LdsRows px, py; v4f x4, y4; };
    // 0: no pair of the block can pass the window (nothing to mark, nothing to fold); 1: gates without octave marks;
    // 2: gates with octave marks.  The same in every lane.
    auto gate_begin = [&](BlockGate& bg, int mbp, int kp) {
        int mode = 1;
        if (LEVEL) {
            const unsigned high = (unsigned)__builtin_amdgcn_readfirstlane((int)s_high[mbp][kp]);
#pragma unroll
            for (int b = 0; b < kMmNB; b++) {
                bg.mine[b] = qhigh[b] ? high >> (4 * h) : 0u;
                if (high != 0u && any_qhigh[b]) mode = 2;
            }
        }
        if (AREA) {
            // fl(a - b) is monotone in a and in b: if the nearest corners are a window apart, so is every pair
            const v4f box = *reinterpret_cast<const v4f*>(&s_box[mbp][kp][0]);
            const bool apart = box.x - qbox[1] >= a.area || qbox[0] - box.y >= a.area ||
                               box.z - qbox[3] >= a.area || qbox[2] - box.w >= a.area;
            if (__ballot(apart) != 0ull) return 0;
            bg.px = (LdsRows)&s_meta[mbp][0][kp * kMmSub + 4 * h];
            bg.py = (LdsRows)&s_meta[mbp][1][kp * kMmSub + 4 * h];
            bg.x4 = bg.px[0]; bg.y4 = bg.py[0];
        }
        return mode;
    };
    auto gate_part = [&](v16f (&acc)[kMmNB], BlockGate& bg, int n, auto level_on) {
        constexpr bool kLevelOn = LEVEL && decltype(level_on)::value;
        if (!AREA && !kLevelOn) return;
        float tx[4] = {0.f, 0.f, 0.f, 0.f}, ty[4] = {0.f, 0.f, 0.f, 0.f};
        if (AREA) {
            tx[0] = bg.x4.x; tx[1] = bg.x4.y; tx[2] = bg.x4.z; tx[3] = bg.x4.w;
            ty[0] = bg.y4.x; ty[1] = bg.y4.y; ty[2] = bg.y4.z; ty[3] = bg.y4.w;
            if (n < 3) {
                // the next part's rows, a slice ahead (and no earlier: the addresses are made opaque here, or the
                // compiler reads all four parts' rows up front and runs out of registers)
                LdsRows nx = bg.px + 2 * (n + 1);
                LdsRows ny = bg.py + 2 * (n + 1);
                asm volatile("" : "+v"(nx), "+v"(ny));
                bg.x4 = *nx; bg.y4 = *ny;
            }
        }
#pragma unroll
        for (int b = 0; b < kMmNB; b++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int i = 4 * n + c;
                unsigned sign = 0u;
                if (AREA) {
                    const float dx = qx[b] - tx[c], dy = qy[b] - ty[c];
                    sign = (unsigned)(area_bits_m1 - (int)__float_as_uint(fmaxf(fabsf(dx), fabsf(dy))));
                }
                if (kLevelOn) {
                    const unsigned lv = bg.mine[b] << (31 - acc_row(i, 0));
                    sign = AREA ? (sign | lv) : lv;
                }
                const float f = acc[b][i];
                acc[b][i] = __uint_as_float((sign & kMmFail) | __float_as_uint(f));
            }
    };

    // ---- one block: kMmSteps x kMmNB products, with a slice of other work issued behind each of them.  The
    //      slices must not depend on this block's products; fenced so that the compiler keeps them where they are:
    //      a wave issues in order, and about five vector instructions fit under one product ----
    auto products = [&](v16f (&acc)[kMmNB], const v4i (&af)[kMmSteps], auto&& slice) {
#pragma unroll
        for (int s = 0; s < kMmSteps; s++)
#pragma unroll
            for (int b = 0; b < kMmNB; b++) {
                acc[b] = product(af[s], bq[b][s], s == 0 ? c_init : acc[b], scale_a, scale_b);
                slice(s * kMmNB + b);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    constexpr int kSlots = kMmSteps * kMmNB;          // slices per block
    static_assert(kMmNB == 2 && kMmSteps == 4 && kMmBT % 2 == 0, "the slices are laid out for 8 products per block and an even number of blocks per tile");

    // Tile T lies in operand buffer T % 3 and metadata buffer T % 4, and is expanded two tiles ahead.  While the
    // products of block k of tile T are issued, the slices behind them refill the operand registers for the next block (of
    // tile T + 1 for the last one; staged at least one barrier ago), finish the arg-min of the block before, and
    // expand this thread's row of block k of tile T + 2 (loaded from memory a tile ago) into the third buffer.  One
    // barrier per tile, at its end, and nothing behind it waits for LDS: the next block's operands are in
    // registers already.  What a block's gates need of its tile's metadata is read up to one tile later, hence
    // four metadata buffers.
    const int n_tiles = (t_end - t_begin + kMmTile - 1) / kMmTile;
    stage_tile16(0, g0);
    stage_meta(0, g0);
    stage_tile16(1, g1);
    stage_meta(1, g1);
    lds_barrier();

    // The operands of a block.  Without gates one set of registers, refilled for the next block product by product as
    // they fall free (slot 2s + 1 is the last to use operand s), which leaves room for a third wave per SIMD; with
    // gates two sets in turn, the next block's read in the first slots (the gate code in the later slots wants the
    // LDS to itself, and two waves per SIMD is all the registers allow there anyway).
    SA_STAMP(9);      // two tiles staged, first barrier passed
    v4i af0[kMmSteps], af1[kMmSteps];
#pragma unroll
    for (int s = 0; s < kMmSteps; s++) af0[s] = operand(0, 0, s);
    auto refill = [&](v4i (&afn)[kMmSteps], int n, int buf, int k) {
        if (kMeta) { if (n < kMmSteps) afn[n] = operand(buf, k, n); }
        else if (n & 1) afn[n >> 1] = operand(buf, k, n >> 1);
    };
    v16f acc0[kMmNB], acc1[kMmNB];
#pragma unroll
    for (int b = 0; b < kMmNB; b++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc1[b][i] = 65535.f;     // "the block before the first": every pair failed

    // products of one block into `acc`; behind them the gates and the arg-min of the block before (`prev`, block kp
    // of the tile with metadata buffer mbp, first train row blockno_prev) and whatever `extra` adds
    auto run_block = [&](v16f (&acc)[kMmNB], v16f (&prev)[kMmNB], const v4i (&af)[kMmSteps], int mbp, int kp, int blockno_prev, auto&& extra) {
        BlockGate bg;
        const int mode = kMeta ? gate_begin(bg, mbp, kp) : 1;
        auto slices = [&](auto level_on) {
            products(acc, af, [&](int n) {
                if (kMeta && n < 4) gate_part(prev, bg, n, level_on);
                argmin_part(prev, 2 * n, blockno_prev);
                argmin_part(prev, 2 * n + 1, blockno_prev);
                extra(n);
            });
        };
        if (AREA && mode == 0) products(acc, af, [&](int n) { extra(n); });
        else if (LEVEL && mode == 2) slices(std::true_type());
        else slices(std::false_type());
    };
    // the same without products: the gates and the arg-min of a block that no other block follows
    auto finish_block = [&](v16f (&prev)[kMmNB], int mbp, int kp, int blockno_prev) {
        BlockGate bg;
        const int mode = kMeta ? gate_begin(bg, mbp, kp) : 1;
        if (AREA && mode == 0) return;
#pragma unroll
        for (int n = 0; n < kSlots; n++) {
            if (kMeta && n < 4) { if (LEVEL && mode == 2) gate_part(prev, bg, n, std::true_type()); else gate_part(prev, bg, n, std::false_type()); }
            argmin_part(prev, 2 * n, blockno_prev);
            argmin_part(prev, 2 * n + 1, blockno_prev);
        }
    };

    SA_STAMP(1);
    int cur = 0, nxt = 1, nx2 = 2;            // operand buffers of tiles T, T + 1, T + 2
    int mb = 0, mb_prev = 0, mb_nx2 = 2;      // metadata buffers of tiles T, T - 1, T + 2
    int blockno = 0;
    int rows_left = t_end - t_begin;
    for (int T = 0; T + 1 < n_tiles; T++) {
        stage_meta(mb_nx2, g);
#pragma unroll
        for (int k = 0; k < kMmBT; k++) {
            v16f (&acc)[kMmNB] = (k & 1) ? acc1 : acc0;
            v16f (&prev)[kMmNB] = (k & 1) ? acc0 : acc1;
            const v4i (&af)[kMmSteps] = (kMeta && (k & 1)) ? af1 : af0;
            v4i (&afn)[kMmSteps] = (kMeta && !(k & 1)) ? af1 : af0;
            uint4* const wslot = stage_slot(nx2, k);
            unsigned o[4];
            run_block(acc, prev, af, k == 0 ? mb_prev : mb, k == 0 ? kMmBT - 1 : k - 1, blockno + (k - 1) * kMmSub, [&](int n) {
                if (k + 1 < kMmBT) refill(afn, n, cur, k + 1); else refill(afn, n, nxt, 0);
                if (n >= 4) {
                    // output dword n - 4 of this thread's row of block k of tile T + 2 (the empty statement keeps the
                    // two instructions in their slice: the compiler would otherwise gather them in front of the write)
                    o[n - 4] = (g.w[k] >> (n - 4)) & 0x11111111u;
                    asm volatile("" : "+v"(o[n - 4]));
                    if (n == 7) {
                        *wslot = make_uint4(o[0], o[1], o[2], o[3]);
                        load_row(t_begin + (T + 3) * kMmTile, k, g);
                        if (k == kMmBT - 1) load_meta(t_begin + (T + 3) * kMmTile, g);
                    }
                }
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
    // the last tile: nothing to stage; the end of the set may cut its blocks short or leave them empty
#pragma unroll
    for (int k = 0; k < kMmBT; k++) {
        v16f (&acc)[kMmNB] = (k & 1) ? acc1 : acc0;
        v16f (&prev)[kMmNB] = (k & 1) ? acc0 : acc1;
        const v4i (&af)[kMmSteps] = (kMeta && (k & 1)) ? af1 : af0;
        v4i (&afn)[kMmSteps] = (kMeta && !(k & 1)) ? af1 : af0;
        if (k > 0) gate_tail(prev, rows_left - (k - 1) * kMmSub);
        const int mbp = k == 0 ? mb_prev : mb, kp = k == 0 ? kMmBT - 1 : k - 1, bprev = blockno + (k - 1) * kMmSub;
        if (k == 0 || rows_left > k * kMmSub) {
            run_block(acc, prev, af, mbp, kp, bprev, [&](int n) {
                if (k + 1 < kMmBT) refill(afn, n, cur, k + 1);
            });
        } else {
            // an empty block: only the block before it is left to finish; its own accumulators all fail
            finish_block(prev, mbp, kp, bprev);
#pragma unroll
            for (int b = 0; b < kMmNB; b++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[b][i] = 65535.f;
        }
    }
    {
        v16f (&lastacc)[kMmNB] = ((kMmBT - 1) & 1) ? acc1 : acc0;
        gate_tail(lastacc, rows_left - (kMmBT - 1) * kMmSub);
        finish_block(lastacc, mb, kMmBT - 1, blockno + (kMmBT - 1) * kMmSub);
    }

#pragma unroll
    for (int b = 0; b < kMmNB; b++) {
        // this lane half's best as a key (H - popcount(query) + 256) << 20 | train row relative to t_begin
        const unsigned mine = SECOND ? best[b] : bits_key(best[b], (int)best_blk[b]);
        const unsigned ob = (unsigned)__shfl_xor((int)mine, 32, 64);
        const unsigned m = min(mine, ob);
        // pairs that failed a gate and rows beyond the set carry distances above 256
        auto final_key = [&](unsigned v) -> unsigned {
            const unsigned d = (v >> kMmIdxBits) + (unsigned)pq[b] - 256u;
            const unsigned t = (unsigned)t_begin + (v & ((1u << kMmIdxBits) - 1u));
            return d <= 256u ? ((d << kMmIdxBits) | t) : kMmNoMatch;
        };
        if (SECOND) {
            const unsigned os = (unsigned)__shfl_xor((int)second[b], 32, 64);
            const unsigned m2 = min(max(mine, ob), min(second[b], os));
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
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mm_stamps), (size_t)n_waves * 10 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// =============================================================================
// Window-gated matching of sets reordered by grid cell, wave by wave.
//
// With the window gate on, a query can only match train rows within `area` of it.  spatial_sort_kernel has put both
// sets in grid-cell order -- index words (original index | octave-above-max_level flag) and positions; the
// descriptors stay where the caller holds them -- so that 32 consecutive queries and 32 consecutive train rows each
// cover a small patch of the image, and has left the bounding box of every block of 32 train rows.  A wave takes 32
// queries (expanded once into registers) and first tests every train block's box against its own, one box per lane:
// the blocks that lie a window or more away -- four in five at a 50-pixel window on 640 x 480 -- are never fetched.
// The others it works through one by one: lane (r, h) fetches the block's row r through its index word (two blocks
// ahead of its use) and expands the dwords its lane half supplies in registers -- the operand of a product is the
// lane's own train row, so the rows never pass through LDS --, then the same four products, accumulator encoding
// and gates as in match_mfma_kernel.  Only the positions and accumulator operands of the block's rows go through a
// wave-private slice of LDS (an accumulator register belongs to another lane's train row).  No barrier, nothing
// shared with the other waves of the workgroup: a wave's time follows its own list of blocks, not the longest
// list among its neighbours (the tiled kernel fed with reordered sets lost most of the gain to that).
//
// A row's place in the reordered set says nothing about which row it was, and ties must go to the lowest original
// train index: the accumulator operand of a block's first product carries the row's original index (13 bits: the 5
// bits that hold the row inside its block in match_mfma_kernel plus the 8 fraction bits that a float in
// [2^15, 2^16) has left), so the bit patterns order like (distance, original index) across all blocks and the
// running minimum is a plain unsigned minimum.  Larger train sets: the reordering works inside chunks of 4096
// consecutive rows, so a block's rows all come from the same aligned group of 2^13 original indices; the accumulators
// carry the index inside the group, and a block's minimum gets its group's base added before it meets the running
// one (full keys, distance << 20 | original index).  Up to 2^15 train rows (the length of a wave's block list).
// =============================================================================
#ifndef SA_WM_WAVES
#define SA_WM_WAVES 4
#endif
#ifndef SA_WM_NB
#define SA_WM_NB 1
#endif
#ifndef SA_WM_SKIP
#define SA_WM_SKIP 0      // timing experiments (wrong results): 1 no gates, 2 no products, 4 empty block lists, 8 no minimum
#endif
constexpr int kWmWaves = SA_WM_WAVES;         // waves per workgroup (they share nothing)
constexpr int kWmNB = SA_WM_NB;               // query blocks of 32 per wave
constexpr int kWmQueries = 32 * kWmNB;        // queries per wave
constexpr int kWmIdxBits = 13;                // index bits an accumulator carries: a row's original index inside its group of 2^13
constexpr int kWmIdxGroup = 1 << kWmIdxBits;
constexpr int kWmMaxTrain = 1 << 15;          // train rows per problem (a wave's block list holds 1024 entries)

#ifdef SA_WM_STAMPS
// Diagnostic build only, per wave: shader clock at the start, at the first block and at the end, blocks worked on,
// cycles spent waiting for the fetches at the top of the block loop, hardware id.
__device__ unsigned long long g_wm_stamps[8192][6];
extern "C" int slam_amd_debug_wm_stamps(unsigned long long* out, int n_waves) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wm_stamps), (size_t)n_waves * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef SA_WM_COUNT
// Diagnostic build only: waves that ran, train blocks they worked on, train blocks they could have, the longest list.
__device__ unsigned long long g_wm_count[4];
extern "C" int slam_amd_debug_wm_count(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wm_count), sizeof g_wm_count) != hipSuccess) return -1;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wm_count), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

// Minimum / maximum over the 32 lanes of a lane half, left in every lane of the wave (both halves hold the same
// values here): four steps inside the rows of 16 lanes on the data-parallel path (lane ^ 1, lane ^ 2, mirrored
// halves of 8, mirrored rows), then the two rows of a half through scalar registers.
template <bool MAX>
__device__ __forceinline__ float half_reduce(float v) {
    auto step = [&](float o) { v = MAX ? fmaxf(v, o) : fminf(v, o); };
    auto dpp = [&](auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    step(dpp(std::integral_constant<int, 0xB1>()));      // quad_perm [1, 0, 3, 2]
    step(dpp(std::integral_constant<int, 0x4E>()));      // quad_perm [2, 3, 0, 1]
    step(dpp(std::integral_constant<int, 0x141>()));     // row_half_mirror
    step(dpp(std::integral_constant<int, 0x140>()));     // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    return MAX ? fmaxf(r0, r1) : fminf(r0, r1);
}

template <bool LEVEL, bool SECOND>
__global__ __launch_bounds__(kWmWaves * 64) void match_window_mfma_kernel(WindowMatchArgs a) {
    __shared__ __align__(16) float s_pos[kWmWaves][2][3][kMmSub];         // its rows' (x, y) pairs (2 x 32 floats) and accumulator operands
    __shared__ unsigned short s_list[kWmWaves][kWmMaxTrain / kMmSub];     // the train blocks a wave works on
    // (the wave's number as a scalar: everything derived from it -- problem, query tile, split -- then stays in scalar
    // registers, divisions included)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
#ifdef SA_WM_STAMPS
    const unsigned long long stamp0 = __builtin_amdgcn_s_memtime();
    unsigned long long stamp_wait = 0;
#endif
    // waves in the order (query tile, problem, split of the train blocks), the split fastest: the waves of a workgroup
    // share their queries when there are splits, and finish together
    const unsigned wid = blockIdx.x * (unsigned)kWmWaves + (unsigned)wave;
    const int split = (int)(wid % (unsigned)a.splits);
    const unsigned rest = wid / (unsigned)a.splits;
    const int prob = (int)(rest % (unsigned)a.batch);
    const int qtile = (int)(rest / (unsigned)a.batch);

    // the query side: the problem's own rows of set 0, or, in a chained batch, the frame before (set 0 holds the one
    // before the batch)
    int qset = 0, qprob = prob, tprob = prob, nqb = a.n1;
    const int32_t* qcount = a.d_n1 ? a.d_n1 + prob : nullptr;
    const int32_t* tcount = a.d_n2 ? a.d_n2 + prob : nullptr;
    if (a.chain) {
        if (prob == 0) { qcount = a.d_n1; }
        else { qset = 1; qprob = prob - 1; nqb = a.n2; qcount = a.d_n2 ? a.d_n2 + (prob - 1) : nullptr; }
    } else if (a.pairs) {
        qset = 1; qprob = a.pairs[2 * prob]; tprob = a.pairs[2 * prob + 1]; nqb = a.n2; qcount = tcount = nullptr;
    }
    const int nq_bound = a.has_ids ? a.n_ids : nqb;
    if (qtile * kWmQueries >= nq_bound || a.n2 <= 0) return;      // (no workgroup barrier below: the waves are on their own)

    // the descriptors stay where the caller holds them and are fetched through the index words
    const uint4* const qd = reinterpret_cast<const uint4*>(qset == 0 ? a.qdesc : a.tdesc) +
                            (qset == 1 || a.chain ? (size_t)qprob * a.cap2 : (size_t)qprob * a.cap1) * 2;
    const float2* const qxy = a.sxy[qset] + (size_t)qprob * a.scap[qset];
    const uint32_t* const qidx = a.sidx[qset] + (size_t)qprob * a.scap[qset];
    const uint4* const td = reinterpret_cast<const uint4*>(a.tdesc) + (size_t)tprob * a.cap2 * 2;
    const unsigned* const txy = reinterpret_cast<const unsigned*>(a.sxy[1] + (size_t)tprob * a.scap[1]);
    const uint32_t* const tidx = a.sidx[1] + (size_t)tprob * a.scap[1];
    const float4* const tbox = a.tile_box + (size_t)tprob * (a.scap[1] / kMmSub);

    // ---- first round of loads, all by the host's bounds (the counts in device memory are among them): the queries'
    //      index words and positions, the boxes of the train blocks.  What lies between a count and its bound has
    //      never been written: an index word from there is brought into range before it is used as an index ----
    unsigned qword[kWmNB];
    float2 qpos[kWmNB];
#pragma unroll
    for (int b = 0; b < kWmNB; b++) {
        const int qc = min(qtile * kWmQueries + b * 32 + r, nq_bound - 1);
        qword[b] = qidx[qc];
        qpos[b] = qxy[qc];
    }
    // (lane l of group g looks at the block (64 g + l) * splits + split: this split's blocks only; the first group's
    // boxes here, the others' one group ahead in the loop below)
    const int nblocks_bound = (a.n2 + kMmSub - 1) / kMmSub;
    float4 bb = tbox[min(lane * a.splits + split, nblocks_bound - 1)];      // x min, x max, y min, y max
    int n1 = nqb, nt = a.n2;
    if (qcount) n1 = min(n1, *qcount);
    if (tcount) nt = min(nt, *tcount);
    const int nq = a.has_ids ? a.n_ids : n1;

    // ---- second round: the queries' descriptors.  Both lane halves hold the same 32 queries per block, half h 16
    //      bytes of the row (dwords 4 h .. 4 h + 3) ----
    uint4 own[kWmNB];
#pragma unroll
    for (int b = 0; b < kWmNB; b++) {
        int qsrc = (int)min(qword[b] & 0x7fffffffu, (unsigned)(nq_bound - 1));
        if (a.ids) qsrc = a.ids[qsrc];
        own[b] = qd[2 * qsrc + h];
    }
    if (qtile * kWmQueries >= nq || nt <= 0) return;

    // ---- the train blocks this wave has to look at, while the descriptors are on their way: lane l tests block
    //      64 g + l.  fl(a - b) is monotone in a and in b, so when the nearest edges of the two boxes are a window
    //      apart, every pair of their points is ----
    float qx[kWmNB], qy[kWmNB];
    bool qvalid[kWmNB], qhigh[kWmNB];
    bool any_qhigh = false;
    const float inf = __uint_as_float(0x7f800000u);
    float qbox[4] = {inf, -inf, inf, -inf};      // x min, x max, y min, y max of the queries with finite positions
#pragma unroll
    for (int b = 0; b < kWmNB; b++) {
        qvalid[b] = qtile * kWmQueries + b * 32 + r < nq;
        qhigh[b] = LEVEL && qvalid[b] && (qword[b] >> 31) != 0u;
        qword[b] &= 0x7fffffffu;
        any_qhigh = any_qhigh || (LEVEL && __ballot(qhigh[b]) != 0ull);
        const float2 p = qpos[b];
        qx[b] = p.x; qy[b] = p.y;
        const bool oddx = (__float_as_uint(p.x) & 0x7f800000u) == 0x7f800000u, oddy = (__float_as_uint(p.y) & 0x7f800000u) == 0x7f800000u;
        // a position that is NaN or infinite fails the window with every train row: -infinity here, +infinity on the
        // train side, so that no difference is ever a NaN (the window test takes a maximum, which would drop one)
        if (oddx) qx[b] = -inf;
        if (oddy) qy[b] = -inf;
        if (!oddx && !oddy && qvalid[b]) {
            qbox[0] = fminf(qbox[0], p.x); qbox[1] = fmaxf(qbox[1], p.x);
            qbox[2] = fminf(qbox[2], p.y); qbox[3] = fmaxf(qbox[3], p.y);
        }
    }
    qbox[0] = half_reduce<false>(qbox[0]); qbox[1] = half_reduce<true>(qbox[1]);
    qbox[2] = half_reduce<false>(qbox[2]); qbox[3] = half_reduce<true>(qbox[3]);
    const int nblocks = (nt + kMmSub - 1) / kMmSub;
    int n_live = 0;      // the list goes through LDS: walking bit masks costs more instructions than the list has entries
    const int ngroups = ((nblocks - split + a.splits - 1) / a.splits + 63) >> 6;      // groups of 64 of this split's blocks
    for (int g = 0; g < ngroups; g++) {
        float4 nb = bb;
        if (g + 1 < ngroups) nb = tbox[min(((g + 1) * 64 + lane) * a.splits + split, nblocks_bound - 1)];
        const int blk = (g * 64 + lane) * a.splits + split;
        const bool on = blk < nblocks &&
                        !(bb.x - qbox[1] >= a.area || qbox[0] - bb.y >= a.area || bb.z - qbox[3] >= a.area || qbox[2] - bb.w >= a.area);
        const unsigned long long mask = __ballot(on);
        const int before = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (on) s_list[wave][n_live + before] = (unsigned short)blk;
        n_live += __builtin_popcountll(mask);
        bb = nb;
    }
#ifdef SA_WM_COUNT
    if (lane == 0) {
        atomicAdd(&g_wm_count[0], 1ull);
        atomicAdd(&g_wm_count[1], (unsigned long long)n_live);
        atomicAdd(&g_wm_count[2], (unsigned long long)((nblocks - split + a.splits - 1) / a.splits));
        atomicMax(&g_wm_count[3], (unsigned long long)n_live);
    }
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // entry i of the list (-1 past its end), the same in every lane
    if (SA_WM_SKIP & 4) n_live = 0;
    auto block_at = [&](int i) { return i < n_live ? __builtin_amdgcn_readfirstlane((int)s_list[wave][i]) : -1; };

    struct Fetched { uint4 lo, hi; unsigned pos, word; };
    // lane (r, h), in two steps: the x (h = 0) or y of the block's row r and its index word, then the row that the
    // word names (all of it in either half: product s takes dword 2 s + h, picked below -- the operand of a product is
    // this lane's own row, so the rows never pass through LDS).  Rows past the end of the set repeat the last one;
    // their position becomes +infinity, which fails the window with every query
    auto fetch_meta = [&](int blk, Fetched& f) {
        const unsigned t = min((unsigned)(blk * kMmSub + r), (unsigned)(nt - 1));
        f.pos = txy[2 * t + h];
        f.word = tidx[t];
    };
    auto fetch_rows = [&](Fetched& f) {
        const unsigned t = f.word & 0x7fffffffu;
        f.lo = td[2 * t];
        f.hi = td[2 * t + 1];
    };
    // two blocks ahead: the index words of the block after the next are on their way while the next one's rows are
    Fetched f, fn;
    f.lo = f.hi = make_uint4(0u, 0u, 0u, 0u);
    int at = 2;
    int blk = block_at(0), blk_n = block_at(1);
    if (blk >= 0) fetch_meta(blk, f);
    if (blk_n >= 0) fetch_meta(blk_n, fn);

    // ---- the queries, expanded once: product s takes dword 2 s + h, so either half hands two of its dwords to the other ----
    v4i bq[kWmNB][kMmSteps];
    int pq[kWmNB];
#pragma unroll
    for (int b = 0; b < kWmNB; b++) {
        const unsigned give0 = h ? own[b].x : own[b].y, give1 = h ? own[b].z : own[b].w;
        const unsigned take0 = (unsigned)__shfl_xor((int)give0, 32, 64), take1 = (unsigned)__shfl_xor((int)give1, 32, 64);
        const unsigned qw[kMmSteps] = {h ? take0 : own[b].x, h ? take1 : own[b].z, h ? own[b].y : take0, h ? own[b].w : take1};
        int pc = 0;
#pragma unroll
        for (int s = 0; s < kMmSteps; s++) {
            pc += __builtin_popcount(qw[s]);
            bq[b][s] = expand_query(qw[s]);
        }
        pq[b] = pc + __shfl_xor(pc, 32, 64);
    }
    if (blk >= 0) fetch_rows(f);

    // into LDS slice p, for the lanes that hold the accumulators of other rows: positions, accumulator operands
    // 2^15 + 32 * 256 + index / 256 (exact in a float); returns the mask of rows above max_level
    auto stage = [&](int p, const Fetched& f, int rows_left) {
        const bool odd = (f.pos & 0x7f800000u) == 0x7f800000u || r >= rows_left;
        s_pos[wave][p][0][2 * r + h] = __uint_as_float(odd ? 0x7f800000u : f.pos);      // (x, y) of row r side by side
        if (h == 0) s_pos[wave][p][2][r] = (float)kMmBias + (float)(f.word & (unsigned)(kWmIdxGroup - 1)) * (1.f / 256.f);
        return LEVEL ? (unsigned)__ballot((f.word >> 31) != 0u) : 0u;
    };

    int scale_a = kMmScaleOne, scale_b = kMmScale64;
    asm volatile("" : "+v"(scale_a), "+v"(scale_b));
    const int area_bits_m1 = (int)__float_as_uint(a.area) - 1;
    unsigned best[kWmNB], second[kWmNB];
#pragma unroll
    for (int b = 0; b < kWmNB; b++) best[b] = second[b] = kMmNoMatch;
    // pattern -> key Hb << 20 | index inside the group (a pattern with the failure mark: Hb = 1023)
    auto bits_key = [](unsigned bits) {
        const unsigned u = min(bits - kMmBinade, 0x7fffffu);
        return ((u & ~0x1fffu) << 7) + (u & 0x1fffu);
    };
    // train sets of more than one index group: the running minimum is kept as full keys (see the head of the kernel)
    const bool groups = a.n2 > kWmIdxGroup;

    int p = 0;
#ifdef SA_WM_STAMPS
    const unsigned long long stamp1 = __builtin_amdgcn_s_memtime();
#endif
    while (blk >= 0) {
#ifdef SA_WM_STAMPS
        { const unsigned long long ta = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          stamp_wait += __builtin_amdgcn_s_memtime() - ta; }
#endif
        const unsigned high = stage(p, f, nt - blk * kMmSub);
        const unsigned base = (unsigned)(blk * kMmSub) & ~(unsigned)(kWmIdxGroup - 1);      // of the block's index group
        v4i af[kMmSteps];
        {
            const uint4 e0 = expand_train(h ? f.lo.y : f.lo.x), e1 = expand_train(h ? f.lo.w : f.lo.z);
            const uint4 e2 = expand_train(h ? f.hi.y : f.hi.x), e3 = expand_train(h ? f.hi.w : f.hi.z);
            af[0] = v4i{(int)e0.x, (int)e0.y, (int)e0.z, (int)e0.w}; af[1] = v4i{(int)e1.x, (int)e1.y, (int)e1.z, (int)e1.w};
            af[2] = v4i{(int)e2.x, (int)e2.y, (int)e2.z, (int)e2.w}; af[3] = v4i{(int)e3.x, (int)e3.y, (int)e3.z, (int)e3.w};
        }
        blk = blk_n;
        f.pos = fn.pos; f.word = fn.word;
        blk_n = block_at(at++);
        if (blk_n >= 0) fetch_meta(blk_n, fn);
        if (blk >= 0) fetch_rows(f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        v16f acc[kWmNB];
        {
            v16f cin;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const v4f v = *reinterpret_cast<const v4f*>(&s_pos[wave][p][2][8 * j + 4 * h]);
                cin[4 * j] = v.x; cin[4 * j + 1] = v.y; cin[4 * j + 2] = v.z; cin[4 * j + 3] = v.w;
            }
#pragma unroll
            for (int s = 0; s < kMmSteps; s++)
#pragma unroll
                for (int b = 0; b < kWmNB; b++) {
                    if (SA_WM_SKIP & 2) { if (s == 0) acc[b] = cin; else acc[b][s] += __int_as_float(af[s].x ^ bq[b][s].y); }
                    else acc[b] = product(af[s], bq[b][s], s == 0 ? cin : acc[b], scale_a, scale_b);
                }
        }
        // ---- gates, as in match_mfma_kernel: the sign of (bits(area) - 1) - bits(max(|dx|, |dy|)) is "not inside
        //      the window"; bit `row` of the block's mask of high rows for queries that are high themselves ----
        auto gates = [&](auto level_on) {
            constexpr bool kLevelOn = LEVEL && decltype(level_on)::value;
            unsigned mine[kWmNB];
#pragma unroll
            for (int b = 0; b < kWmNB; b++) mine[b] = (kLevelOn && qhigh[b]) ? high >> (4 * h) : 0u;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // rows 8 j + 4 h .. + 3 as (x, y) pairs: one packed subtraction per pair of a query and a train row
                const v4f xy01 = *reinterpret_cast<const v4f*>(&s_pos[wave][p][0][2 * (8 * j + 4 * h)]);
                const v4f xy23 = *reinterpret_cast<const v4f*>(&s_pos[wave][p][0][2 * (8 * j + 4 * h) + 4]);
                const v2f txy4[4] = {{xy01.x, xy01.y}, {xy01.z, xy01.w}, {xy23.x, xy23.y}, {xy23.z, xy23.w}};
#pragma unroll
                for (int b = 0; b < kWmNB; b++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int i = 4 * j + c;
                        const v2f qv = {qx[b], qy[b]};
                        const v2f d = qv - txy4[c];
                        unsigned sign = (unsigned)(area_bits_m1 - (int)__float_as_uint(fmaxf(fabsf(d.x), fabsf(d.y))));
                        if (kLevelOn) sign |= mine[b] << (31 - acc_row(i, 0));
                        const float v = acc[b][i];
                        acc[b][i] = __uint_as_float((sign & kMmFail) | __float_as_uint(v));
                    }
            }
        };
        if (SA_WM_SKIP & 1) { }
        else if (LEVEL && high != 0u && any_qhigh) gates(std::true_type());
        else gates(std::false_type());
        // ---- the block's minimum (two smallest with SECOND) into the running one ----
#pragma unroll
        for (int b = 0; b < kWmNB; b++) {
            if (SA_WM_SKIP & 8) {
                best[b] = min(best[b], acc_bits(acc[b], 0) + acc_bits(acc[b], 5) + acc_bits(acc[b], 10) + acc_bits(acc[b], 15));
            } else if (!SECOND && !groups) {
                unsigned m = best[b];
#pragma unroll
                for (int i = 0; i < 16; i += 2) m = min(min(m, acc_bits(acc[b], i)), acc_bits(acc[b], i + 1));
                best[b] = m;
            } else if (!SECOND) {
                unsigned m = min(acc_bits(acc[b], 0), acc_bits(acc[b], 1));
#pragma unroll
                for (int i = 2; i < 16; i += 2) m = min(min(m, acc_bits(acc[b], i)), acc_bits(acc[b], i + 1));
                best[b] = min(best[b], bits_key(m) + base);
            } else {
                unsigned lo = kMmNoMatch, hi = kMmNoMatch;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const unsigned x = acc_bits(acc[b], i), y = acc_bits(acc[b], i + 1);
                    const unsigned l2 = min(x, y), h2 = max(x, y);
                    hi = min(max(lo, l2), min(hi, h2));
                    lo = min(lo, l2);
                }
                const unsigned klo = bits_key(lo) + base, khi = bits_key(hi) + base;
                second[b] = min(max(best[b], klo), min(second[b], khi));
                best[b] = min(best[b], klo);
            }
        }
        p ^= 1;
    }
#ifdef SA_WM_STAMPS
    if (lane == 0) {
        unsigned long long* o = g_wm_stamps[wid & 8191];
        o[0] = stamp0; o[1] = stamp1; o[2] = __builtin_amdgcn_s_memtime(); o[3] = (unsigned long long)n_live; o[4] = stamp_wait;
        o[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
#endif

#pragma unroll
    for (int b = 0; b < kWmNB; b++) {
        const unsigned mine = (SECOND || groups) ? best[b] : bits_key(best[b]);
        const unsigned ob = (unsigned)__shfl_xor((int)mine, 32, 64);
        const unsigned m = min(mine, ob);
        // pairs that failed a gate and rows beyond the set carry distances above 256
        auto final_key = [&](unsigned v) -> unsigned {
            const unsigned d = (v >> kMmIdxBits) + (unsigned)pq[b] - 256u;
            return d <= 256u ? ((d << kMmIdxBits) | (v & ((1u << kMmIdxBits) - 1u))) : kMmNoMatch;
        };
        const size_t out = (size_t)prob * a.cap1 + qword[b];
        if (SECOND) {
            const unsigned os = (unsigned)__shfl_xor((int)second[b], 32, 64);
            const unsigned m2 = min(max(mine, ob), min(second[b], os));
            if (h == 0 && qvalid[b]) a.part[(size_t)split * a.part_pitch + out] = make_uint2(final_key(m), final_key(m2));
            continue;
        }
        if (h != 0 || !qvalid[b]) continue;
        const unsigned key = final_key(m);
        if (a.pairs) {
            const unsigned d = key >> kMmIdxBits;
            a.pair_out[(size_t)prob * a.n2 + qword[b]] =
                (key != kMmNoMatch && (int)d <= a.dmax) ? ((d << 16) | (key & ((1u << kMmIdxBits) - 1u))) : kMmNoMatch;
        } else if (key != kMmNoMatch) {
            atomicMin(&a.best[out], key);
        }
    }
}

int window_mfma_max_train() { return kWmMaxTrain; }

// A wave and its queries are the unit of work; the train blocks are dealt round-robin to `splits` waves per query
// tile only when the query tiles alone leave the SIMDs short of waves (a split expands its queries again).
int window_mfma_splits(int batch, int nq_bound, int nt_bound, int n_cu) {
    const long long qwaves = (long long)div_up(nq_bound, kWmQueries) * batch;
    if (qwaves <= 0) return 1;
    long long want = div_up(n_cu * 16, (int)std::min<long long>(qwaves, 1 << 30));
#ifdef SA_WM_SPLITS
    want = SA_WM_SPLITS;      // timing experiments
#endif
    return (int)std::max<long long>(1, std::min<long long>(std::min<long long>(want, kMaxSecondSplits), div_up(nt_bound, 4 * kMmSub)));
}

int launch_match_window_mfma(const WindowMatchArgs& a, int batch, int nq_bound, hipStream_t st) {
    if (a.n2 > kWmMaxTrain || !(a.area > 0.f) || a.splits < 1 || (a.pairs && (a.splits != 1 || a.part)))
        SA_FAIL(SLAM_AMD_ERR_ARG, "match: the reordered matrix-core kernel takes a window and at most 32768 train rows");
    const long long waves = (long long)div_up(nq_bound, kWmQueries) * a.splits * batch;
    const long long blocks = (waves + kWmWaves - 1) / kWmWaves;
    if (blocks <= 0) return SLAM_AMD_OK;
    if (waves + kWmWaves > 0x7fffffffLL) SA_FAIL(SLAM_AMD_ERR_ARG, "match: problem too large for one launch");
    const dim3 grid((unsigned)blocks), block(kWmWaves * 64);
    const bool level = a.max_level != -1, second = a.part != nullptr;
    if (level && second) hipLaunchKernelGGL((match_window_mfma_kernel<true, true>), grid, block, 0, st, a);
    else if (level) hipLaunchKernelGGL((match_window_mfma_kernel<true, false>), grid, block, 0, st, a);
    else if (second) hipLaunchKernelGGL((match_window_mfma_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((match_window_mfma_kernel<false, false>), grid, block, 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
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
#ifdef SA_MM_SPLITS
    splits = SA_MM_SPLITS;      // timing experiments
#endif
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
