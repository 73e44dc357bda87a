// HIP graph capture through the C ABI, driven with the raw HIP graph API (no PyTorch).
//
// 1. Two graphs -- a batch of frames (extract + match against the previous frame) and a
//    single frame -- are captured AND instantiated back to back before either is launched,
//    then replayed after eager work in between.  Every replay must reproduce the eager
//    results byte for byte.
// 2. The allocation rule of the matcher under capture: a captured call never allocates
//    (SLAM_AMD_ERR_CAPACITY when its scratch is too small), and after a call has been
//    captured the scratch never moves (a larger call is refused instead of freeing a
//    pointer an instantiated graph still holds).
// 3. The scratch that the kernels leave clean for the next call (score histograms, running match keys): a
//    recording made by the very first call of fresh contexts, replayed after eager calls and after another
//    recording; and an eager call that fails half way between recording and replay.
// Needs a GPU.  Exit code 0 = all checks passed.  Every HIP failure ends the program at
// once: nothing is retried on a device that reported an error.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/slam_amd.h"

static int g_fail = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { g_fail++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d: %s\n", hipGetErrorString(e_), __FILE__, __LINE__, #call); fflush(stdout); _Exit(3); } } while (0)
#define OK(call) do { int rc_ = (call); if (rc_ != SLAM_AMD_OK) { printf("slam_amd error %d (%s) at %s:%d: %s\n", rc_, slam_amd_last_error(), __FILE__, __LINE__, #call); fflush(stdout); _Exit(4); } } while (0)

static unsigned rng_state = 12345;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static void make_scene(uint8_t* img, int w, int h, int shift) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            img[(size_t)y * w + x] = (uint8_t)(60 + 40 * std::sin((x + shift) * 0.05) + 30 * std::cos(y * 0.07));
    rng_state = 777;
    for (int i = 0; i < 500; i++) {
        int cx = (int)(rnd() % w) + shift, cy = (int)(rnd() % h), r = 3 + (int)(rnd() % 12);
        uint8_t v = (uint8_t)(rnd() % 256);
        for (int y = std::max(0, cy - r); y < std::min(h, cy + r); y++)
            for (int x = std::max(0, cx - r); x < std::min(w, cx + r); x++) img[(size_t)y * w + x] = v;
    }
}

constexpr int W = 320, H = 240, B = 4, CAP = 2048, NF = 500, KP = 28, DM = 16;

struct Buffers {
    uint8_t *img, *kp, *desc, *out;
    int32_t *cnt, *nout;
};

struct Snapshot {
    std::vector<uint8_t> kp, desc, out;
    std::vector<int32_t> cnt, nout;
    bool operator==(const Snapshot& o) const { return kp == o.kp && desc == o.desc && out == o.out && cnt == o.cnt && nout == o.nout; }
};

// Outputs of slots 1..batch (slot 0 is the "previous frame" and stays as it is).
static void clear_outputs(const Buffers& b, hipStream_t s) {
    HIP(hipMemsetAsync(b.kp + (size_t)CAP * KP, 0, (size_t)B * CAP * KP, s));
    HIP(hipMemsetAsync(b.desc + (size_t)CAP * 32, 0, (size_t)B * CAP * 32, s));
    HIP(hipMemsetAsync(b.cnt + 1, 0, B * sizeof(int32_t), s));
    HIP(hipMemsetAsync(b.out, 0, (size_t)B * CAP * DM, s));
    HIP(hipMemsetAsync(b.nout, 0, B * sizeof(int32_t), s));
}

static Snapshot snapshot(const Buffers& b, int batch, hipStream_t s) {
    HIP(hipStreamSynchronize(s));
    Snapshot r;
    r.cnt.resize(batch); r.nout.resize(batch);
    HIP(hipMemcpy(r.cnt.data(), b.cnt + 1, batch * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(r.nout.data(), b.nout, batch * sizeof(int32_t), hipMemcpyDeviceToHost));
    r.kp.resize((size_t)batch * CAP * KP); r.desc.resize((size_t)batch * CAP * 32); r.out.resize((size_t)batch * CAP * DM);
    HIP(hipMemcpy(r.kp.data(), b.kp + (size_t)CAP * KP, r.kp.size(), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(r.desc.data(), b.desc + (size_t)CAP * 32, r.desc.size(), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(r.out.data(), b.out, r.out.size(), hipMemcpyDeviceToHost));
    return r;
}

int main() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { printf("no GPU\n"); return 77; }
    HIP(hipSetDevice(0));
    slam_amd_orb_params p;
    slam_amd_default_orb_params(&p);
    p.nfeatures = NF;
    slam_amd_extractor* ex = nullptr;
    slam_amd_matcher* m = nullptr;
    OK(slam_amd_extractor_create(&p, W, H, B, CAP, 0, &ex));
    OK(slam_amd_matcher_create(0, &m));
    OK(slam_amd_matcher_reserve(m, B, CAP, CAP));

    Buffers b;
    HIP(hipMalloc((void**)&b.img, (size_t)B * W * H));
    HIP(hipMalloc((void**)&b.kp, (size_t)(B + 1) * CAP * KP));
    HIP(hipMalloc((void**)&b.desc, (size_t)(B + 1) * CAP * 32));
    HIP(hipMalloc((void**)&b.cnt, (B + 1) * sizeof(int32_t)));
    HIP(hipMalloc((void**)&b.out, (size_t)B * CAP * DM));
    HIP(hipMalloc((void**)&b.nout, B * sizeof(int32_t)));
    HIP(hipMemset(b.kp, 0, (size_t)(B + 1) * CAP * KP));
    HIP(hipMemset(b.desc, 0, (size_t)(B + 1) * CAP * 32));
    HIP(hipMemset(b.cnt, 0, (B + 1) * sizeof(int32_t)));
    {
        std::vector<uint8_t> host((size_t)B * W * H);
        for (int f = 0; f < B; f++) make_scene(host.data() + (size_t)f * W * H, W, H, 5 * f);
        HIP(hipMemcpy(b.img, host.data(), host.size(), hipMemcpyHostToDevice));
    }
    hipStream_t s;
    HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

    auto step = [&](int batch) {
        OK(slam_amd_extract_batch_async(ex, b.img, batch, (size_t)W * H, W,
                                        reinterpret_cast<slam_amd_keypoint*>(b.kp + (size_t)CAP * KP), b.desc + (size_t)CAP * 32,
                                        b.cnt + 1, CAP, s));
        OK(slam_amd_frame_match_batch_async(m, batch, b.desc, reinterpret_cast<slam_amd_keypoint*>(b.kp), CAP, b.cnt,
                                            b.desc + (size_t)CAP * 32, reinterpret_cast<slam_amd_keypoint*>(b.kp + (size_t)CAP * KP),
                                            CAP, b.cnt + 1, 300.f, -1.f, -1,
                                            reinterpret_cast<slam_amd_dmatch*>(b.out), b.nout, s));
    };

    // slot 0: the last frame as "previous frame"
    OK(slam_amd_extract_batch_async(ex, b.img + (size_t)(B - 1) * W * H, 1, 0, W, reinterpret_cast<slam_amd_keypoint*>(b.kp),
                                    b.desc, b.cnt, CAP, s));
    HIP(hipStreamSynchronize(s));

    // eager references
    clear_outputs(b, s); step(B);
    const Snapshot ref_batch = snapshot(b, B, s);
    clear_outputs(b, s); step(1);
    const Snapshot ref_one = snapshot(b, 1, s);
    EXPECT(ref_batch.cnt[0] > 100 && ref_batch.nout[1] > 100, "too few keypoints / matches: %d %d", ref_batch.cnt[0], ref_batch.nout[1]);
    EXPECT(ref_one.cnt[0] == ref_batch.cnt[0], "batch-1 count differs from the batch's first frame");

    // both graphs captured and instantiated before either is launched
    hipGraph_t g_batch = nullptr, g_one = nullptr;
    hipGraphExec_t e_batch = nullptr, e_one = nullptr;
    HIP(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    step(B);
    HIP(hipStreamEndCapture(s, &g_batch));
    HIP(hipGraphInstantiate(&e_batch, g_batch, nullptr, nullptr, 0));
    HIP(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    step(1);
    HIP(hipStreamEndCapture(s, &g_one));
    HIP(hipGraphInstantiate(&e_one, g_one, nullptr, nullptr, 0));

    for (int i = 0; i < 3; i++) step(B);                 // eager work between instantiation and replay
    clear_outputs(b, s);
    for (int i = 0; i < 3; i++) HIP(hipGraphLaunch(e_batch, s));
    EXPECT(snapshot(b, B, s) == ref_batch, "replay of the batch graph differs from the eager result");
    for (int i = 0; i < 3; i++) step(1);
    clear_outputs(b, s);
    HIP(hipGraphLaunch(e_one, s));                       // first launch of the graph instantiated second
    EXPECT(snapshot(b, 1, s) == ref_one, "first replay of the single-frame graph differs from the eager result");
    for (int i = 0; i < 3; i++) HIP(hipGraphLaunch(e_one, s));
    for (int i = 0; i < 2; i++) HIP(hipGraphLaunch(e_batch, s));
    EXPECT(snapshot(b, B, s) == ref_batch, "batch graph after the single-frame graph");
    printf("graph replays reproduce the eager results (%d keypoints in frame 0, %d matches in frame 1)\n",
           ref_batch.cnt[0], ref_batch.nout[1]);

    // ---- allocation rule under capture ----
    {
        slam_amd_matcher* m2 = nullptr;
        OK(slam_amd_matcher_create(0, &m2));
        auto call = [&](int n) {
            return slam_amd_frame_match_async(m2, b.desc, reinterpret_cast<slam_amd_keypoint*>(b.kp), n, nullptr,
                                              b.desc + (size_t)CAP * 32, reinterpret_cast<slam_amd_keypoint*>(b.kp + (size_t)CAP * KP),
                                              n, nullptr, nullptr, 0, nullptr, 300.f, -1.f, -1,
                                              reinterpret_cast<slam_amd_dmatch*>(b.out), b.nout, s);
        };
        hipGraph_t g = nullptr;
        HIP(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        const int rc_unreserved = call(256);             // no scratch yet: must refuse, not allocate
        HIP(hipStreamEndCapture(s, &g));
        if (g) HIP(hipGraphDestroy(g));
        EXPECT(rc_unreserved == SLAM_AMD_ERR_CAPACITY, "captured call without reserve returned %d", rc_unreserved);
        OK(slam_amd_matcher_reserve(m2, 1, 256, 256));
        hipGraphExec_t e = nullptr;
        g = nullptr;
        HIP(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        const int rc_reserved = call(256);
        HIP(hipStreamEndCapture(s, &g));
        EXPECT(rc_reserved == SLAM_AMD_OK, "captured call within the reserve returned %d", rc_reserved);
        HIP(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
        HIP(hipGraphLaunch(e, s));
        HIP(hipStreamSynchronize(s));
        int32_t n_graph = -1;
        HIP(hipMemcpy(&n_graph, b.nout, sizeof n_graph, hipMemcpyDeviceToHost));
        EXPECT(n_graph == 256, "256 queries with no gate must all match, got %d", n_graph);
        EXPECT(call(CAP) == SLAM_AMD_ERR_CAPACITY, "a larger call after capture must be refused (the graph holds the scratch)");
        EXPECT(slam_amd_matcher_reserve(m2, 1, CAP, CAP) == SLAM_AMD_ERR_CAPACITY, "reserve must not move captured scratch");
        EXPECT(call(200) == SLAM_AMD_OK, "a smaller call still works");
        HIP(hipGraphLaunch(e, s));                       // and the graph is still valid
        HIP(hipStreamSynchronize(s));
        HIP(hipGraphExecDestroy(e));
        HIP(hipGraphDestroy(g));
        OK(slam_amd_matcher_destroy(m2));
    }

    // ---- self-cleaning scratch: first call recorded, eager calls and a failing call before the replay ----
    {
        slam_amd_extractor* ex3 = nullptr;
        slam_amd_matcher* m3 = nullptr;
        OK(slam_amd_extractor_create(&p, W, H, B, CAP, 0, &ex3));
        OK(slam_amd_matcher_create(0, &m3));
        OK(slam_amd_matcher_reserve(m3, B, CAP, 0));      // no room for the cell-sorted copies: the windowed call below must fail
        auto step3 = [&](int batch) {
            OK(slam_amd_extract_batch_async(ex3, b.img, batch, (size_t)W * H, W,
                                            reinterpret_cast<slam_amd_keypoint*>(b.kp + (size_t)CAP * KP), b.desc + (size_t)CAP * 32,
                                            b.cnt + 1, CAP, s));
            OK(slam_amd_frame_match_batch_async(m3, batch, b.desc, reinterpret_cast<slam_amd_keypoint*>(b.kp), CAP, b.cnt,
                                                b.desc + (size_t)CAP * 32, reinterpret_cast<slam_amd_keypoint*>(b.kp + (size_t)CAP * KP),
                                                CAP, b.cnt + 1, 300.f, -1.f, -1,
                                                reinterpret_cast<slam_amd_dmatch*>(b.out), b.nout, s));
        };
        hipGraph_t g1 = nullptr, g2 = nullptr;
        hipGraphExec_t e1 = nullptr, e2 = nullptr;
        HIP(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        step3(B);                                         // the contexts' first call is a recording: nothing runs
        HIP(hipStreamEndCapture(s, &g1));
        HIP(hipGraphInstantiate(&e1, g1, nullptr, nullptr, 0));
        clear_outputs(b, s); step3(B);                    // an eager call before the first replay
        EXPECT(snapshot(b, B, s) == ref_batch, "eager call after a recorded first call differs from the reference");
        HIP(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        step3(B);
        HIP(hipStreamEndCapture(s, &g2));
        HIP(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
        clear_outputs(b, s);
        HIP(hipGraphLaunch(e2, s));                       // the second recording runs first
        EXPECT(snapshot(b, B, s) == ref_batch, "second recording differs from the reference");
        clear_outputs(b, s);
        HIP(hipGraphLaunch(e1, s));
        EXPECT(snapshot(b, B, s) == ref_batch, "first recording differs from the reference");
        // an eager windowed call that fails after it has started (its reordered copies cannot be allocated: recordings
        // hold the scratch), then the replays again
        OK(slam_amd_matcher_debug_set_window_mode(m3, 2));
        const int rc_fail = slam_amd_frame_match_batch_async(m3, B, b.desc, reinterpret_cast<slam_amd_keypoint*>(b.kp), CAP, b.cnt,
                                                             b.desc + (size_t)CAP * 32, reinterpret_cast<slam_amd_keypoint*>(b.kp + (size_t)CAP * KP),
                                                             CAP, b.cnt + 1, 300.f, 40.f, -1,
                                                             reinterpret_cast<slam_amd_dmatch*>(b.out), b.nout, s);
        EXPECT(rc_fail == SLAM_AMD_ERR_CAPACITY, "the windowed call without room for its copies returned %d", rc_fail);
        OK(slam_amd_matcher_debug_set_window_mode(m3, 0));
        clear_outputs(b, s);
        HIP(hipGraphLaunch(e1, s));
        EXPECT(snapshot(b, B, s) == ref_batch, "replay after a failed eager call differs from the reference");
        clear_outputs(b, s); step3(B);
        EXPECT(snapshot(b, B, s) == ref_batch, "eager call after a failed eager call differs from the reference");
        HIP(hipGraphExecDestroy(e1)); HIP(hipGraphExecDestroy(e2));
        HIP(hipGraphDestroy(g1)); HIP(hipGraphDestroy(g2));
        OK(slam_amd_extractor_destroy(ex3));
        OK(slam_amd_matcher_destroy(m3));
        printf("recordings made before any call ran, and replays after a failed call, reproduce the eager results\n");
    }

    HIP(hipGraphExecDestroy(e_batch)); HIP(hipGraphExecDestroy(e_one));
    HIP(hipGraphDestroy(g_batch)); HIP(hipGraphDestroy(g_one));
    HIP(hipStreamDestroy(s));
    OK(slam_amd_extractor_destroy(ex));
    OK(slam_amd_matcher_destroy(m));
    for (void* q : {(void*)b.img, (void*)b.kp, (void*)b.desc, (void*)b.cnt, (void*)b.out, (void*)b.nout}) HIP(hipFree(q));
    printf(g_fail ? "%d check(s) failed\n" : "graph capture ok (%d failures)\n", g_fail);
    return g_fail ? 1 : 0;
}
