// Keyframe-pair matching sharded over one process per GPU, results gathered to rank 0 over RCCL.
//
// Pairs are independent: the pair list is cut into equal contiguous shares, every rank matches
// its share with slam_amd_match_pairs_async against its own replica of the descriptor store, and
// nothing is exchanged while computing.  The only communication is the gather of the fixed-size
// result rows (one 32-bit word per query row) to rank 0.  A rank's share is cut into chunks; as
// soon as a chunk is computed its rows are sent on a second stream (ncclSend / ncclRecv inside one
// group, rank 0 receiving straight into the gathered table), overlapping the next chunk's
// compute.  xGMI is point to point, so the seven senders reach rank 0 over seven different links.
//
// Built as libslam_amd_dist.so (links librccl and libslam_amd): the core library stays free of
// the RCCL dependency.
#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "common.h"
#include "../../include/slam_amd_dist.h"

using namespace sa;

static_assert(SLAM_AMD_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

struct slam_amd_pair_shard {
    slam_amd_matcher* m = nullptr;
    int device = 0, world = 1, rank = 0, n_pairs = 0, n_per = 0, chunks = 1;
    int lo = 0, hi = 0, per = 0;             // this rank's share [lo, hi) of the pair list, rows reserved per rank
    ncclComm_t comm = nullptr;
    hipStream_t s_comm = nullptr;
    std::vector<hipEvent_t> ev;              // chunk computed
    hipEvent_t ev_sent = nullptr;            // everything queued on s_comm so far (the previous run's sends and receives)
    uint32_t* rows = nullptr;                // [per][n_per] this rank's results (rank 0 with world 1: the result itself)
    uint32_t* gathered = nullptr;            // rank 0, world > 1: [world][per][n_per]
};

static int nccl_fail(ncclResult_t r, const char* what) {
    set_error(std::string(what) + " failed: " + ncclGetErrorString(r));
    return SLAM_AMD_ERR_HIP;
}
#define SA_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return nccl_fail(r_, #call); } while (0)

extern "C" int slam_amd_pair_shard_range(int n_pairs, int world, int rank, int* lo, int* hi, int* per) {
    if (n_pairs < 0 || world <= 0 || rank < 0 || rank >= world) SA_FAIL(SLAM_AMD_ERR_ARG, "bad argument");
    const int p = (n_pairs + world - 1) / world;
    if (lo) *lo = std::min(rank * p, n_pairs);
    if (hi) *hi = std::min((rank + 1) * p, n_pairs);
    if (per) *per = p;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pair_shard_chunk(int per, int chunks, int chunk, int* row0, int* row1) {
    if (per < 0 || chunks <= 0 || chunk < 0 || chunk >= chunks) SA_FAIL(SLAM_AMD_ERR_ARG, "bad argument");
    if (row0) *row0 = (int)((long long)per * chunk / chunks);
    if (row1) *row1 = (int)((long long)per * (chunk + 1) / chunks);
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pair_shard_unique_id(char* id) {
    if (!id) SA_FAIL(SLAM_AMD_ERR_ARG, "id is null");
    ncclUniqueId u;
    SA_NCCL(ncclGetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pair_shard_destroy(slam_amd_pair_shard* s) {
    if (!s) return SLAM_AMD_OK;
    (void)hipSetDevice(s->device);
    if (s->s_comm) (void)hipStreamSynchronize(s->s_comm);
    if (s->comm) (void)ncclCommDestroy(s->comm);
    for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
    if (s->ev_sent) (void)hipEventDestroy(s->ev_sent);
    if (s->s_comm) (void)hipStreamDestroy(s->s_comm);
    if (s->rows) (void)hipFree(s->rows);
    if (s->gathered) (void)hipFree(s->gathered);
    delete s;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pair_shard_create(slam_amd_matcher* m, int device, const char* unique_id, int world, int rank,
                                          int n_pairs, int n_per, int chunks, slam_amd_pair_shard** out) {
    if (!out) SA_FAIL(SLAM_AMD_ERR_ARG, "out is null");
    *out = nullptr;
    if (!m || n_pairs < 0 || n_per <= 0 || chunks <= 0 || world <= 0 || rank < 0 || rank >= world)
        SA_FAIL(SLAM_AMD_ERR_ARG, "bad argument");
    if (world > 1 && !unique_id) SA_FAIL(SLAM_AMD_ERR_ARG, "a unique id is needed for more than one rank");
    slam_amd_pair_shard* s = new (std::nothrow) slam_amd_pair_shard();
    if (!s) SA_FAIL(SLAM_AMD_ERR_NOMEM, "out of host memory");
    s->m = m; s->device = device; s->world = world; s->rank = rank; s->n_pairs = n_pairs; s->n_per = n_per; s->chunks = chunks;
    slam_amd_pair_shard_range(n_pairs, world, rank, &s->lo, &s->hi, &s->per);
    auto fail = [&](int code) { slam_amd_pair_shard_destroy(s); return code; };
    auto hip = [&](hipError_t e, const char* what) { return e == hipSuccess ? 0 : hip_fail(e, what, __FILE__, __LINE__); };
    int rc;
    if ((rc = hip(hipSetDevice(device), "hipSetDevice"))) return fail(rc);
    const size_t share = (size_t)std::max(s->per, 1) * n_per * sizeof(uint32_t);
    if ((rc = hip(hipMalloc((void**)&s->rows, share), "hipMalloc"))) return fail(rc);
    if ((rc = hip(hipMemset(s->rows, 0xff, share), "hipMemset"))) return fail(rc);
    if (world > 1) {
        if (rank == 0 && (rc = hip(hipMalloc((void**)&s->gathered, share * world), "hipMalloc"))) return fail(rc);
        if ((rc = hip(hipStreamCreateWithFlags(&s->s_comm, hipStreamNonBlocking), "hipStreamCreateWithFlags"))) return fail(rc);
        s->ev.resize((size_t)chunks);
        for (auto& e : s->ev)
            if ((rc = hip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"))) return fail(rc);
        if ((rc = hip(hipEventCreateWithFlags(&s->ev_sent, hipEventDisableTiming), "hipEventCreate"))) return fail(rc);
        ncclUniqueId u;
        memcpy(u.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
        ncclResult_t r = ncclCommInitRank(&s->comm, world, u, rank);
        if (r != ncclSuccess) return fail(nccl_fail(r, "ncclCommInitRank"));
    }
    *out = s;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pair_shard_run(slam_amd_pair_shard* s, const uint8_t* d_desc, const slam_amd_keypoint* d_kp,
                                       int n_keyframes, const int32_t* d_pairs_local, int limit_rows,
                                       float max_dist, float area, int max_level, void* stream) {
    if (!s || !d_desc) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    const int mine = s->hi - s->lo;
    if (mine > 0 && !d_pairs_local) SA_FAIL(SLAM_AMD_ERR_ARG, "d_pairs_local is null");
    SA_HIP(hipSetDevice(s->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool exchange = s->world > 1 && limit_rows < 0;   // a limited run is a warm-up: nothing is sent
    // this run overwrites the rows the run before may still be sending (or, on rank 0, copying into the
    // gathered table): the compute stream waits for whatever the communication stream holds
    if (s->s_comm) {
        SA_HIP(hipEventRecord(s->ev_sent, s->s_comm));
        SA_HIP(hipStreamWaitEvent(st, s->ev_sent, 0));
    }
    for (int c = 0; c < s->chunks; c++) {
        int r0, r1;
        slam_amd_pair_shard_chunk(s->per, s->chunks, c, &r0, &r1);
        const int a = std::min(r0, mine);
        int b = std::min(r1, mine);
        if (limit_rows >= 0) b = std::min(b, a + limit_rows);
        if (b > a) {
            int rc = slam_amd_match_pairs_async(s->m, d_desc, d_kp, n_keyframes, s->n_per, d_pairs_local + 2 * (size_t)a, b - a,
                                                max_dist, area, max_level, s->rows + (size_t)a * s->n_per, st);
            if (rc) return rc;
        }
        if (!exchange || r1 <= r0) continue;
        // rows [r0, r1) of every rank's share travel to rank 0 (equal counts on every rank: shares are
        // padded to `per` rows, so sends and receives always pair up)
        SA_HIP(hipEventRecord(s->ev[(size_t)c], st));
        SA_HIP(hipStreamWaitEvent(s->s_comm, s->ev[(size_t)c], 0));
        const size_t count = (size_t)(r1 - r0) * s->n_per, off = (size_t)r0 * s->n_per;
        const size_t share = (size_t)s->per * s->n_per;
        if (s->rank == 0) {
            SA_HIP(hipMemcpyAsync(s->gathered + off, s->rows + off, count * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->s_comm));
            SA_NCCL(ncclGroupStart());
            for (int r = 1; r < s->world; r++) SA_NCCL(ncclRecv(s->gathered + share * r + off, count, ncclUint32, r, s->comm, s->s_comm));
            SA_NCCL(ncclGroupEnd());
        } else {
            SA_NCCL(ncclSend(s->rows + off, count, ncclUint32, 0, s->comm, s->s_comm));
        }
    }
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pair_shard_wait(slam_amd_pair_shard* s, void* stream) {
    if (!s) SA_FAIL(SLAM_AMD_ERR_ARG, "shard is null");
    SA_HIP(hipSetDevice(s->device));
    SA_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    if (s->s_comm) SA_HIP(hipStreamSynchronize(s->s_comm));
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pair_shard_result(slam_amd_pair_shard* s, const uint32_t** d_rows, int* n_rows) {
    if (!s || !d_rows) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    if (s->rank != 0) { *d_rows = nullptr; if (n_rows) *n_rows = 0; return SLAM_AMD_OK; }
    // shares are contiguous ranges of the pair list padded to `per` rows: rank r's rows start at r * per,
    // which is also pair r * per, so the table is in pair-list order and its first n_pairs rows are the result
    *d_rows = s->world > 1 ? s->gathered : s->rows;
    if (n_rows) *n_rows = s->n_pairs;
    return SLAM_AMD_OK;
}
