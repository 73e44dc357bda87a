// Host side of the streaming configuration: frames arrive in pinned host memory, and the copy
// to the device, extraction + matching against the previous frame, and the copy of the results
// back run on three HIP streams over a ring of slots, so that frame i + 1 is uploaded and frame
// i - 1 downloaded while frame i is computed.  A slot holds a batch of frames (one by default):
// large frames leave the chip idle between the short launches of a single frame, a few per launch
// fill it, at the price of the batch's latency.  Plain C++ over the HIP runtime; no kernel here.
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "common.h"

using namespace sa;

struct slam_amd_pipeline {
    int device = 0, width = 0, height = 0, capacity = 0, depth = 0, batch = 1;
    float max_dist = 300.f, area = -1.f;
    int max_level = -1;
    slam_amd_extractor* ex = nullptr;
    slam_amd_matcher* m = nullptr;
    hipStream_t s_copy = nullptr, s_comp = nullptr, s_back = nullptr;
    struct Slot {                            // every buffer holds `batch` frames
        uint8_t* h_img = nullptr;            // pinned
        uint8_t* d_img = nullptr;
        slam_amd_keypoint* d_kp = nullptr; uint8_t* d_desc = nullptr; int32_t* d_cnt = nullptr;
        slam_amd_dmatch* d_match = nullptr; int32_t* d_nmatch = nullptr;
        slam_amd_keypoint* h_kp = nullptr; uint8_t* h_desc = nullptr; int32_t* h_cnt = nullptr;   // pinned
        slam_amd_dmatch* h_match = nullptr; int32_t* h_nmatch = nullptr;                          // pinned
        hipEvent_t ev_h2d = nullptr, ev_done = nullptr, ev_out = nullptr;
        long long first = -1;                // frames [first, first + count) are the slot's occupants: their results
        int count = 0;                       // are, or will be, in the host buffers
    };
    std::vector<Slot> slots;
    long long submitted = 0;                 // frames submitted so far (frame indices are 0, 1, 2, ...)
    long long open_first = 0;                // first frame of the batch being collected (not queued yet)
    long long batches = 0;                   // batches queued so far: batch k lives in slot k % depth
    bool failed = false;                     // a batch failed half-way: its slot's events no longer describe it
};

#define SA_PIPE_USABLE(p) \
    do { if ((p)->failed) SA_FAIL(SLAM_AMD_ERR_ARG, "an earlier submit failed after part of it was queued: destroy the pipeline"); } while (0)

static void free_slot(slam_amd_pipeline::Slot& s) {
    for (void* h : {(void*)s.h_img, (void*)s.h_kp, (void*)s.h_desc, (void*)s.h_cnt, (void*)s.h_match, (void*)s.h_nmatch})
        if (h) (void)hipHostFree(h);
    for (void* d : {(void*)s.d_img, (void*)s.d_kp, (void*)s.d_desc, (void*)s.d_cnt, (void*)s.d_match, (void*)s.d_nmatch})
        if (d) (void)hipFree(d);
    for (hipEvent_t e : {s.ev_h2d, s.ev_done, s.ev_out})
        if (e) (void)hipEventDestroy(e);
    s = slam_amd_pipeline::Slot();
}

extern "C" int slam_amd_pipeline_destroy(slam_amd_pipeline* p) {
    if (!p) return SLAM_AMD_OK;
    (void)hipSetDevice(p->device);
    for (hipStream_t st : {p->s_copy, p->s_comp, p->s_back})
        if (st) (void)hipStreamSynchronize(st);
    for (auto& s : p->slots) free_slot(s);
    for (hipStream_t st : {p->s_copy, p->s_comp, p->s_back})
        if (st) (void)hipStreamDestroy(st);
    slam_amd_extractor_destroy(p->ex);
    slam_amd_matcher_destroy(p->m);
    delete p;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pipeline_create_batched(const slam_amd_orb_params* params, int width, int height, int capacity,
                                                int depth, int batch, int device, float max_dist, float area, int max_level,
                                                slam_amd_pipeline** out) {
    if (!out) SA_FAIL(SLAM_AMD_ERR_ARG, "out is null");
    *out = nullptr;
    if (depth < 3 || depth > 64) SA_FAIL(SLAM_AMD_ERR_ARG, "depth must be 3..64 (the previous batch's slot must outlive the match)");
    if (batch < 1 || batch > 1024) SA_FAIL(SLAM_AMD_ERR_ARG, "batch must be 1..1024");
    if (capacity <= 0) SA_FAIL(SLAM_AMD_ERR_ARG, "capacity must be positive");
    slam_amd_pipeline* p = new (std::nothrow) slam_amd_pipeline();
    if (!p) SA_FAIL(SLAM_AMD_ERR_NOMEM, "out of host memory");
    p->device = device; p->width = width; p->height = height; p->capacity = capacity; p->depth = depth; p->batch = batch;
    p->max_dist = max_dist; p->area = area; p->max_level = max_level;
    auto fail = [&](int code) { slam_amd_pipeline_destroy(p); return code; };
    int rc;
    if ((rc = slam_amd_extractor_create(params, width, height, batch, capacity, device, &p->ex))) return fail(rc);
    if ((rc = slam_amd_matcher_create(device, &p->m))) return fail(rc);
    if ((rc = slam_amd_matcher_reserve(p->m, batch, capacity, capacity))) return fail(rc);
    auto hip = [&](hipError_t e, const char* what) { return e == hipSuccess ? 0 : hip_fail(e, what, __FILE__, __LINE__); };
    if ((rc = hip(hipSetDevice(device), "hipSetDevice"))) return fail(rc);
    for (hipStream_t* st : {&p->s_copy, &p->s_comp, &p->s_back})
        if ((rc = hip(hipStreamCreateWithFlags(st, hipStreamNonBlocking), "hipStreamCreateWithFlags"))) return fail(rc);
    p->slots.resize((size_t)depth);
    const size_t img = (size_t)width * height * batch, cap = (size_t)capacity * batch;
    const size_t cnt = ((size_t)batch * sizeof(int) + 255) / 256 * 256;      // a count per frame of the slot
    for (auto& s : p->slots) {
        if ((rc = hip(hipHostMalloc((void**)&s.h_img, img, hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_img, img), "hipMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_kp, cap * sizeof(slam_amd_keypoint)), "hipMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_desc, cap * 32), "hipMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_cnt, cnt), "hipMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_match, cap * sizeof(slam_amd_dmatch)), "hipMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_nmatch, cnt), "hipMalloc")) ||
            (rc = hip(hipHostMalloc((void**)&s.h_kp, cap * sizeof(slam_amd_keypoint), hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipHostMalloc((void**)&s.h_desc, cap * 32, hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipHostMalloc((void**)&s.h_cnt, cnt, hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipHostMalloc((void**)&s.h_match, cap * sizeof(slam_amd_dmatch), hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipHostMalloc((void**)&s.h_nmatch, cnt, hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipEventCreateWithFlags(&s.ev_h2d, hipEventDisableTiming), "hipEventCreate")) ||
            (rc = hip(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming), "hipEventCreate")) ||
            (rc = hip(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming), "hipEventCreate")) ||
            (rc = hip(hipMemset(s.d_cnt, 0, cnt), "hipMemset")) || (rc = hip(hipMemset(s.d_nmatch, 0, cnt), "hipMemset")))
            return fail(rc);
        memset(s.h_cnt, 0, cnt); memset(s.h_nmatch, 0, cnt);
    }
    *out = p;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pipeline_create(const slam_amd_orb_params* params, int width, int height, int capacity,
                                        int depth, int device, float max_dist, float area, int max_level,
                                        slam_amd_pipeline** out) {
    return slam_amd_pipeline_create_batched(params, width, height, capacity, depth, 1, device, max_dist, area, max_level, out);
}

extern "C" int slam_amd_pipeline_input(slam_amd_pipeline* p, long long frame, uint8_t** host_image, int* row_stride) {
    if (!p || !host_image) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    SA_PIPE_USABLE(p);
    if (frame != p->submitted) SA_FAIL(SLAM_AMD_ERR_ARG, "frames are submitted in order: ask for the input buffer of the next frame");
    auto& s = p->slots[(size_t)(p->batches % p->depth)];
    SA_HIP(hipSetDevice(p->device));
    // the slot's previous occupants (the batch `depth` batches back): their images have been uploaded and their
    // results are out
    if (frame == p->open_first && s.first >= 0) SA_HIP(hipEventSynchronize(s.ev_out));
    *host_image = s.h_img + (size_t)(frame - p->open_first) * p->width * p->height;
    if (row_stride) *row_stride = p->width;
    return SLAM_AMD_OK;
}

// Queues the collected frames [open_first, submitted) as one batch.  Returns at the first failure, possibly with part
// of the batch queued.
static int queue_batch(slam_amd_pipeline* p) {
    const int nb = (int)(p->submitted - p->open_first);
    auto& s = p->slots[(size_t)(p->batches % p->depth)];
    const bool reused = s.first >= 0;
    const size_t img = (size_t)p->width * p->height, cap = (size_t)p->capacity;
    // upload: the slot's device images were last read by the extraction of the batch `depth` batches back
    if (reused) SA_HIP(hipStreamWaitEvent(p->s_copy, s.ev_done, 0));
    SA_HIP(hipMemcpyAsync(s.d_img, s.h_img, img * nb, hipMemcpyHostToDevice, p->s_copy));
    SA_HIP(hipEventRecord(s.ev_h2d, p->s_copy));
    // compute: after the upload, and after that batch's results have left the slot
    SA_HIP(hipStreamWaitEvent(p->s_comp, s.ev_h2d, 0));
    if (reused) SA_HIP(hipStreamWaitEvent(p->s_comp, s.ev_out, 0));
    int rc = slam_amd_extract_batch_async(p->ex, s.d_img, nb, img, p->width, s.d_kp, s.d_desc, s.d_cnt, p->capacity, p->s_comp);
    if (rc) return rc;
    // every frame against the one before it: the batch's first against the last frame of the batch before
    // (none for frame 0: its match count stays 0)
    if (p->batches > 0) {
        const auto& prev = p->slots[(size_t)((p->batches - 1) % p->depth)];
        const size_t last = (size_t)(prev.count - 1);
        rc = slam_amd_frame_match_chain_async(p->m, nb, prev.d_desc + last * cap * 32, prev.d_kp + last * cap, prev.d_cnt + last,
                                              s.d_desc, s.d_kp, p->capacity, s.d_cnt, p->max_dist, p->area, p->max_level,
                                              s.d_match, s.d_nmatch, p->s_comp);
    } else {
        SA_HIP(hipMemsetAsync(s.d_nmatch, 0, sizeof(int32_t), p->s_comp));
        if (nb > 1)
            rc = slam_amd_frame_match_chain_async(p->m, nb - 1, s.d_desc, s.d_kp, s.d_cnt, s.d_desc + cap * 32, s.d_kp + cap,
                                                  p->capacity, s.d_cnt + 1, p->max_dist, p->area, p->max_level,
                                                  s.d_match + cap, s.d_nmatch + 1, p->s_comp);
    }
    if (rc) return rc;
    SA_HIP(hipEventRecord(s.ev_done, p->s_comp));
    // download
    SA_HIP(hipStreamWaitEvent(p->s_back, s.ev_done, 0));
    SA_HIP(hipMemcpyAsync(s.h_cnt, s.d_cnt, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, p->s_back));
    SA_HIP(hipMemcpyAsync(s.h_nmatch, s.d_nmatch, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, p->s_back));
    SA_HIP(hipMemcpyAsync(s.h_kp, s.d_kp, cap * nb * sizeof(slam_amd_keypoint), hipMemcpyDeviceToHost, p->s_back));
    SA_HIP(hipMemcpyAsync(s.h_desc, s.d_desc, cap * nb * 32, hipMemcpyDeviceToHost, p->s_back));
    SA_HIP(hipMemcpyAsync(s.h_match, s.d_match, cap * nb * sizeof(slam_amd_dmatch), hipMemcpyDeviceToHost, p->s_back));
    SA_HIP(hipEventRecord(s.ev_out, p->s_back));
    s.first = p->open_first;
    s.count = nb;
    p->open_first = p->submitted;
    p->batches++;
    return SLAM_AMD_OK;
}

static int queue_batch_or_fail(slam_amd_pipeline* p) {
    const int rc = queue_batch(p);
    if (rc) {
        // part of the batch may be queued while the slot's events still belong to its previous occupants: let
        // the streams drain (nothing reads the slot's buffers after that) and refuse further use.  The results
        // of the batches queued before stay readable.
        const std::string why = slam_amd_last_error();
        for (hipStream_t st : {p->s_copy, p->s_comp, p->s_back}) (void)hipStreamSynchronize(st);
        p->failed = true;
        set_error(why);
    }
    return rc;
}

extern "C" int slam_amd_pipeline_submit(slam_amd_pipeline* p, long long frame) {
    if (!p) SA_FAIL(SLAM_AMD_ERR_ARG, "pipeline is null");
    SA_PIPE_USABLE(p);
    if (frame != p->submitted) SA_FAIL(SLAM_AMD_ERR_ARG, "frames are submitted in order");
    SA_HIP(hipSetDevice(p->device));
    p->submitted = frame + 1;
    if (p->submitted - p->open_first < p->batch) return SLAM_AMD_OK;
    return queue_batch_or_fail(p);
}

extern "C" int slam_amd_pipeline_flush(slam_amd_pipeline* p) {
    if (!p) SA_FAIL(SLAM_AMD_ERR_ARG, "pipeline is null");
    SA_PIPE_USABLE(p);
    if (p->submitted == p->open_first) return SLAM_AMD_OK;
    SA_HIP(hipSetDevice(p->device));
    return queue_batch_or_fail(p);
}

extern "C" int slam_amd_pipeline_result(slam_amd_pipeline* p, long long frame, const slam_amd_keypoint** kp,
                                        const uint8_t** desc, int* n, const slam_amd_dmatch** matches, int* n_matches) {
    if (!p) SA_FAIL(SLAM_AMD_ERR_ARG, "pipeline is null");
    if (frame < 0 || frame >= p->submitted) SA_FAIL(SLAM_AMD_ERR_ARG, "frame has not been submitted");
    if (frame >= p->open_first) SA_FAIL(SLAM_AMD_ERR_ARG, "the frame's batch is not complete yet: submit the rest of it, or flush");
    const slam_amd_pipeline::Slot* s = nullptr;
    for (const auto& c : p->slots)
        if (c.first >= 0 && frame >= c.first && frame < c.first + c.count) s = &c;
    if (!s) SA_FAIL(SLAM_AMD_ERR_ARG, "the slot of this frame has been reused (results live until depth - 1 more batches are queued)");
    SA_HIP(hipSetDevice(p->device));
    SA_HIP(hipEventSynchronize(s->ev_out));
    const size_t k = (size_t)(frame - s->first), cap = (size_t)p->capacity;
    if (kp) *kp = s->h_kp + k * cap;
    if (desc) *desc = s->h_desc + k * cap * 32;
    if (n) *n = std::min(s->h_cnt[k], p->capacity);
    if (matches) *matches = s->h_match + k * cap;
    if (n_matches) *n_matches = frame > 0 ? s->h_nmatch[k] : 0;
    return SLAM_AMD_OK;
}
