// C++ host components of the streaming and the multi-GPU configuration, on one GPU:
//   slam_amd::FramePipeline  three HIP streams over a ring of pinned slots; every frame's
//                            keypoints, descriptors and matches against the previous frame must
//                            equal the sequential CPU oracle's -- at 320 x 240 one frame at a time with the
//                            gates on, and at 3840 x 2160 / 10000 features two frames per launch sequence
//   slam_amd::PairShard      the RCCL pair shard with a world of one rank (no communicator): all
//                            pairs of a small keyframe store against the oracle, and the share /
//                            chunk arithmetic for larger worlds
// Needs a GPU.  Exit code 0 = all checks passed.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/slam_amd.hpp"
#include "../../include/slam_amd_dist.h"
extern "C" {
#include "../../oracle/orb_oracle.h"
}

static int g_fail = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { g_fail++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); fflush(stdout); _Exit(3); } } while (0)

static unsigned rng_state = 12345;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static std::vector<uint8_t> make_scene(int w, int h, int shift, int blobs = 500) {
    std::vector<uint8_t> img((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            img[(size_t)y * w + x] = (uint8_t)(60 + 40 * std::sin((x + shift) * 0.05) + 30 * std::cos(y * 0.07));
    rng_state = 777;
    for (int i = 0; i < blobs; i++) {
        int cx = (int)(rnd() % w) + shift, cy = (int)(rnd() % h), r = 3 + (int)(rnd() % 12);
        uint8_t v = (uint8_t)(rnd() % 256);
        for (int y = std::max(0, cy - r); y < std::min(h, cy + r); y++)
            for (int x = std::max(0, cx - r); x < std::min(w, cx + r); x++) img[(size_t)y * w + x] = v;
    }
    return img;
}

static void test_pipeline() {
    const int W = 320, H = 240, NF = 400, CAP = 2048, N = 9;
    slam_amd_orb_params p;
    slam_amd_default_orb_params(&p);
    p.nfeatures = NF;
    slam_amd::FramePipeline pipe(p, W, H, CAP, 3, 300, 50, 4);
    orc_params op; orc_default_params(&op); op.nfeatures = NF;
    std::vector<std::vector<uint8_t>> frames;
    for (int i = 0; i < N; i++) frames.push_back(make_scene(W, H, 3 * i));
    std::vector<orc_keypoint> pk, ck(CAP);
    std::vector<uint8_t> pd, cd((size_t)CAP * 32);
    // two frames in flight ahead of the one being read back
    for (int i = 0; i < N + 2; i++) {
        if (i < N) {
            slam_amd::Image8 v; v.data = frames[i].data(); v.cols = W; v.rows = H; v.step = W;
            EXPECT(pipe.push(v) == i, "frame index");
        }
        const int j = i - 2;
        if (j < 0) continue;
        const slam_amd::FramePipeline::Result r = pipe.result(j);
        int cn = 0;
        orc_detect_and_compute(&op, frames[j].data(), W, H, W, ck.data(), cd.data(), CAP, &cn);
        EXPECT(r.n == cn && cn > 100, "frame %d: %d keypoints vs %d", j, r.n, cn);
        EXPECT(std::memcmp(r.keypoints, ck.data(), (size_t)cn * 28) == 0, "frame %d keypoints", j);
        EXPECT(std::memcmp(r.descriptors, cd.data(), (size_t)cn * 32) == 0, "frame %d descriptors", j);
        if (j > 0) {
            std::vector<orc_dmatch> ref(pk.size());
            int rn = 0;
            orc_match_gated(pd.data(), pk.data(), (int)pk.size(), cd.data(), ck.data(), cn, 300.f, 50.f, 4, ref.data(), &rn);
            EXPECT(r.nMatches == rn && rn > 50, "frame %d: %d matches vs %d", j, r.nMatches, rn);
            EXPECT(std::memcmp(r.matches, ref.data(), (size_t)rn * 16) == 0, "frame %d matches", j);
        } else {
            EXPECT(r.nMatches == 0, "frame 0 has no previous frame");
        }
        pk.assign(ck.begin(), ck.begin() + cn);
        pd.assign(cd.begin(), cd.begin() + (size_t)cn * 32);
    }
    bool threw = false;
    try { pipe.result(0); } catch (const slam_amd::Error&) { threw = true; }
    EXPECT(threw, "a slot that was reused must not hand out stale results");
}

// The streaming configuration at its real size: 3840 x 2160 frames, 10000 features, two frames per launch
// sequence, five frames (the last batch is incomplete and flushed), gates off as the stream benchmark runs it.
static void test_pipeline_4k() {
    const int W = 3840, H = 2160, NF = 10000, CAP = 12288, N = 5;
    slam_amd_orb_params p;
    slam_amd_default_orb_params(&p);
    p.nfeatures = NF;
    slam_amd::FramePipeline pipe(p, W, H, CAP, 3, 300, -1, -1, 0, 2);
    orc_params op; orc_default_params(&op); op.nfeatures = NF;
    std::vector<std::vector<uint8_t>> frames;
    for (int i = 0; i < N; i++) frames.push_back(make_scene(W, H, 5 * i, 40000));
    for (int i = 0; i < N; i++) {
        slam_amd::Image8 v; v.data = frames[i].data(); v.cols = W; v.rows = H; v.step = W;
        EXPECT(pipe.push(v) == i, "frame index");
    }
    bool threw = false;
    try { pipe.result(N - 1); } catch (const slam_amd::Error&) { threw = true; }
    EXPECT(threw, "a frame of an incomplete batch has no result before flush()");
    pipe.flush();
    std::vector<orc_keypoint> pk, ck(CAP);
    std::vector<uint8_t> pd, cd((size_t)CAP * 32);
    for (int j = 0; j < N; j++) {
        const slam_amd::FramePipeline::Result r = pipe.result(j);
        int cn = 0;
        orc_detect_and_compute(&op, frames[j].data(), W, H, W, ck.data(), cd.data(), CAP, &cn);
        EXPECT(r.n == cn && cn > 3000, "4K frame %d: %d keypoints vs %d", j, r.n, cn);
        EXPECT(std::memcmp(r.keypoints, ck.data(), (size_t)cn * 28) == 0, "4K frame %d keypoints", j);
        EXPECT(std::memcmp(r.descriptors, cd.data(), (size_t)cn * 32) == 0, "4K frame %d descriptors", j);
        if (j > 0) {
            std::vector<orc_dmatch> ref(pk.size());
            int rn = 0;
            orc_match_gated(pd.data(), pk.data(), (int)pk.size(), cd.data(), ck.data(), cn, 300.f, -1.f, -1, ref.data(), &rn);
            EXPECT(r.nMatches == rn && rn > 1000, "4K frame %d: %d matches vs %d", j, r.nMatches, rn);
            EXPECT(std::memcmp(r.matches, ref.data(), (size_t)rn * 16) == 0, "4K frame %d matches", j);
        } else {
            EXPECT(r.nMatches == 0, "frame 0 has no previous frame");
        }
        pk.assign(ck.begin(), ck.begin() + cn);
        pd.assign(cd.begin(), cd.begin() + (size_t)cn * 32);
    }
}

static void test_pair_shard() {
    // share / chunk arithmetic for larger worlds
    for (int world : {1, 2, 3, 8}) {
        const int n_pairs = 523776;
        int covered = 0, per0 = -1;
        for (int r = 0; r < world; r++) {
            int lo, hi, per;
            EXPECT(slam_amd_pair_shard_range(n_pairs, world, r, &lo, &hi, &per) == 0, "range");
            EXPECT(lo == covered && hi >= lo && hi - lo <= per, "contiguous shares");
            EXPECT(lo == std::min(r * per, n_pairs), "rank r's share starts at row r * per of the gathered table");
            covered = hi;
            if (per0 < 0) per0 = per;
            EXPECT(per == per0, "equal rows reserved per rank");
            int prev = 0;
            for (int c = 0; c < 16; c++) {
                int a, b;
                slam_amd_pair_shard_chunk(per, 16, c, &a, &b);
                EXPECT(a == prev && b >= a, "chunks tile the share");
                prev = b;
            }
            EXPECT(prev == per, "chunks cover the share");
        }
        EXPECT(covered == n_pairs, "shares cover the list");
    }
    // one rank on the GPU against the oracle
    const int n_kf = 6, n_per = 300;
    std::vector<uint8_t> desc((size_t)n_kf * n_per * 32);
    rng_state = 4242;
    for (auto& b : desc) b = (uint8_t)(rnd() & 255);
    std::memcpy(&desc[(size_t)3 * n_per * 32], &desc[(size_t)1 * n_per * 32 + 50 * 32], 100 * 32);   // exact matches across keyframes
    std::vector<int32_t> pairs;
    for (int i = 0; i < n_kf; i++) for (int j = i + 1; j < n_kf; j++) { pairs.push_back(i); pairs.push_back(j); }
    const int n_pairs = (int)pairs.size() / 2;
    uint8_t* d_desc; int32_t* d_pairs;
    HIP(hipMalloc((void**)&d_desc, desc.size()));
    HIP(hipMalloc((void**)&d_pairs, pairs.size() * 4));
    HIP(hipMemcpy(d_desc, desc.data(), desc.size(), hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_pairs, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
    slam_amd_matcher* m = nullptr;
    slam_amd::check(slam_amd_matcher_create(0, &m));
    {
        slam_amd::PairShard shard(m, 0, nullptr, 1, 0, n_pairs, n_per, 4);
        EXPECT(shard.lo() == 0 && shard.hi() == n_pairs, "a world of one owns every pair");
        shard.run(d_desc, nullptr, n_kf, d_pairs, 300, -1, -1, nullptr, 2);    // warm-up: two pairs per chunk
        shard.run(d_desc, nullptr, n_kf, d_pairs);
        shard.wait();
        std::vector<uint32_t> got((size_t)n_pairs * n_per);
        HIP(hipMemcpy(got.data(), shard.result(), got.size() * 4, hipMemcpyDeviceToHost));
        std::vector<orc_keypoint> kp(n_per);
        std::memset(kp.data(), 0, kp.size() * sizeof(orc_keypoint));
        std::vector<orc_dmatch> ref(n_per);
        int bad = 0, exact = 0;
        for (int r = 0; r < n_pairs; r++) {
            int rn = 0;
            orc_match_gated(&desc[(size_t)pairs[2 * r] * n_per * 32], kp.data(), n_per, &desc[(size_t)pairs[2 * r + 1] * n_per * 32],
                            kp.data(), n_per, 300.f, -1.f, -1, ref.data(), &rn);
            std::vector<uint32_t> row(n_per, 0xffffffffu);
            for (int k = 0; k < rn; k++) {
                row[ref[k].queryIdx] = ((uint32_t)ref[k].distance << 16) | (uint32_t)ref[k].trainIdx;
                exact += ref[k].distance == 0.f;
            }
            bad += std::memcmp(row.data(), &got[(size_t)r * n_per], (size_t)n_per * 4) != 0;
        }
        EXPECT(bad == 0, "%d of %d pairs differ from the oracle", bad, n_pairs);
        EXPECT(exact >= 100, "the planted exact matches were found (%d)", exact);
    }
    slam_amd_matcher_destroy(m);
    HIP(hipFree(d_desc)); HIP(hipFree(d_pairs));
}

int main() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { printf("no GPU\n"); return 77; }
    try {
        test_pipeline();
        test_pipeline_4k();
        test_pair_shard();
    } catch (const std::exception& e) {
        printf("exception: %s\n", e.what());
        return 2;
    }
    printf(g_fail ? "%d check(s) failed\n" : "pipeline and pair shard ok (%d failures)\n", g_fail);
    return g_fail ? 1 : 0;
}
