/*
 * slam_amd_dist.h -- C ABI of the multi-GPU part: keyframe-pair matching sharded over one process
 * per GPU with the results gathered to rank 0 over RCCL (libslam_amd_dist.so; links librccl and
 * libslam_amd.so).
 *
 * No counterpart in the reference, which is one process and one thread; the work is
 * Matcher::frameMatch(kf_i, kf_j, {}, ...) (src/frame/matcher.cpp:15-55, the all-defaults call of
 * src/tracking/mapper.cpp:34) over many keyframe pairs, as relocalisation and loop-candidate
 * search need it.  Pairs are independent, so nothing is exchanged while matching; the result rows
 * (one 32-bit word per query row: distance << 16 | train index, 0xFFFFFFFF = none) are gathered
 * to rank 0 chunk by chunk on a second stream, overlapped with the next chunk's compute.
 */
#ifndef SLAM_AMD_DIST_H
#define SLAM_AMD_DIST_H

#include "slam_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_AMD_UNIQUE_ID_BYTES 128

typedef struct slam_amd_pair_shard slam_amd_pair_shard;

/* Pure host arithmetic (no GPU, no RCCL): share [lo, hi) of rank `rank` in a list of n_pairs pairs
 * -- equal contiguous ranges -- and `per`, the rows reserved per rank (the last share may be shorter). */
int slam_amd_pair_shard_range(int n_pairs, int world, int rank, int* lo, int* hi, int* per);
/* Rows [row0, row1) of a rank's (padded) share that chunk `chunk` of `chunks` covers; equal on every rank. */
int slam_amd_pair_shard_chunk(int per, int chunks, int chunk, int* row0, int* row1);

/* Rank 0 creates the RCCL unique id (SLAM_AMD_UNIQUE_ID_BYTES bytes) and hands it to the other
 * ranks by whatever channel the launcher has (a file, a socket, a torch.distributed store). */
int slam_amd_pair_shard_unique_id(char* id);

/* One per process.  `m` is a matcher on `device`; unique_id may be NULL when world == 1 (then no
 * communicator is created).  Collective over the ranks when world > 1. */
int slam_amd_pair_shard_create(slam_amd_matcher* m, int device, const char* unique_id, int world, int rank,
                               int n_pairs, int n_per, int chunks, slam_amd_pair_shard** out);
int slam_amd_pair_shard_destroy(slam_amd_pair_shard* s);

/* Matches this rank's share and sends it to rank 0.  d_desc / d_kp: the replicated store
 * ([n_keyframes][n_per][32] / [n_keyframes][n_per], d_kp may be NULL with both gates off);
 * d_pairs_local: this rank's pairs, (hi - lo) x 2 int32 on the device.  Everything is enqueued on
 * `stream` (compute) and on the shard's own communication stream; returns without waiting.
 * limit_rows >= 0: match at most that many pairs per chunk and send nothing (warm-up). */
int slam_amd_pair_shard_run(slam_amd_pair_shard* s, const uint8_t* d_desc, const slam_amd_keypoint* d_kp,
                            int n_keyframes, const int32_t* d_pairs_local, int limit_rows,
                            float max_dist, float area, int max_level, void* stream);
/* Blocks until `stream` and the communication stream have drained. */
int slam_amd_pair_shard_wait(slam_amd_pair_shard* s, void* stream);
/* Rank 0: device pointer to the gathered table, n_rows = n_pairs rows of n_per words in pair-list
 * order.  Other ranks: NULL. */
int slam_amd_pair_shard_result(slam_amd_pair_shard* s, const uint32_t** d_rows, int* n_rows);

#ifdef __cplusplus
}

#include <stdexcept>
#include <string>

namespace slam_amd {

// C++ owner of a pair shard (one per process / GPU); throws std::runtime_error on failure.
class PairShard {
public:
    PairShard(slam_amd_matcher* matcher, int device, const char* uniqueId, int world, int rank, int nPairs, int nPer, int chunks) {
        ok(slam_amd_pair_shard_create(matcher, device, uniqueId, world, rank, nPairs, nPer, chunks, &s_));
        ok(slam_amd_pair_shard_range(nPairs, world, rank, &lo_, &hi_, &per_));
    }
    PairShard(const PairShard&) = delete;
    PairShard& operator=(const PairShard&) = delete;
    ~PairShard() { slam_amd_pair_shard_destroy(s_); }
    int lo() const { return lo_; }            // this rank's share of the pair list: [lo, hi)
    int hi() const { return hi_; }
    void run(const uint8_t* dDesc, const slam_amd_keypoint* dKp, int nKeyframes, const int32_t* dPairsLocal,
             float maximumDistance = 300, float areaSize = -1, int maxLevel = -1, void* stream = nullptr, int limitRows = -1) {
        ok(slam_amd_pair_shard_run(s_, dDesc, dKp, nKeyframes, dPairsLocal, limitRows, maximumDistance, areaSize, maxLevel, stream));
    }
    void wait(void* stream = nullptr) { ok(slam_amd_pair_shard_wait(s_, stream)); }
    // rank 0: [nPairs][nPer] device words in pair-list order; other ranks: nullptr
    const uint32_t* result() const {
        const uint32_t* p = nullptr;
        ok(slam_amd_pair_shard_result(s_, &p, nullptr));
        return p;
    }
private:
    static void ok(int status) {
        if (status != SLAM_AMD_OK) throw std::runtime_error(std::string(slam_amd_status_string(status)) + ": " + slam_amd_last_error());
    }
    slam_amd_pair_shard* s_ = nullptr;
    int lo_ = 0, hi_ = 0, per_ = 0;
};

}  // namespace slam_amd
#endif
#endif
