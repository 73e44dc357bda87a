// Host side of the streaming configuration: frames arrive in pinned host memory, and the copy
// to the device, extraction + matching against the previous frame, and the copy of the results
// back run on three HIP streams over a ring of slots, so that frame i + 1 is uploaded and frame
// i - 1 downloaded while frame i is computed.  Plain C++ over the HIP runtime; no kernel here.
#include <algorithm>
#include <new>
#include <vector>

#include "common.h"

using namespace sa;

struct slam_amd_pipeline {
    int device = 0, width = 0, height = 0, capacity = 0, depth = 0;
    float max_dist = 300.f, area = -1.f;
    int max_level = -1;
    slam_amd_extractor* ex = nullptr;
    slam_amd_matcher* m = nullptr;
    hipStream_t s_copy = nullptr, s_comp = nullptr, s_back = nullptr;
    struct Slot {
        uint8_t* h_img = nullptr;            // pinned
        uint8_t* d_img = nullptr;
        slam_amd_keypoint* d_kp = nullptr; uint8_t* d_desc = nullptr; int32_t* d_cnt = nullptr;
        slam_amd_dmatch* d_match = nullptr; int32_t* d_nmatch = nullptr;
        slam_amd_keypoint* h_kp = nullptr; uint8_t* h_desc = nullptr; int32_t* h_cnt = nullptr;   // pinned
        slam_amd_dmatch* h_match = nullptr; int32_t* h_nmatch = nullptr;                          // pinned
        hipEvent_t ev_h2d = nullptr, ev_done = nullptr, ev_out = nullptr;
        long long frame = -1;                // frame whose results the slot holds (or will hold)
    };
    std::vector<Slot> slots;
    long long submitted = 0;                 // frames submitted so far (frame indices are 0, 1, 2, ...)
};

static void free_slot(slam_amd_pipeline::Slot& s) {
    if (s.h_img) (void)hipHostFree(s.h_img);
    if (s.d_img) (void)hipFree(s.d_img);
    if (s.d_kp) (void)hipFree(s.d_kp);
    if (s.d_desc) (void)hipFree(s.d_desc);
    if (s.d_cnt) (void)hipFree(s.d_cnt);
    if (s.d_match) (void)hipFree(s.d_match);
    if (s.d_nmatch) (void)hipFree(s.d_nmatch);
    if (s.h_kp) (void)hipHostFree(s.h_kp);
    if (s.h_desc) (void)hipHostFree(s.h_desc);
    if (s.h_cnt) (void)hipHostFree(s.h_cnt);
    if (s.h_match) (void)hipHostFree(s.h_match);
    if (s.h_nmatch) (void)hipHostFree(s.h_nmatch);
    if (s.ev_h2d) (void)hipEventDestroy(s.ev_h2d);
    if (s.ev_done) (void)hipEventDestroy(s.ev_done);
    if (s.ev_out) (void)hipEventDestroy(s.ev_out);
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

extern "C" int slam_amd_pipeline_create(const slam_amd_orb_params* params, int width, int height, int capacity,
                                        int depth, int device, float max_dist, float area, int max_level,
                                        slam_amd_pipeline** out) {
    if (!out) SA_FAIL(SLAM_AMD_ERR_ARG, "out is null");
    *out = nullptr;
    if (depth < 3 || depth > 64) SA_FAIL(SLAM_AMD_ERR_ARG, "depth must be 3..64 (the previous frame's slot must outlive the match)");
    if (capacity <= 0) SA_FAIL(SLAM_AMD_ERR_ARG, "capacity must be positive");
    slam_amd_pipeline* p = new (std::nothrow) slam_amd_pipeline();
    if (!p) SA_FAIL(SLAM_AMD_ERR_NOMEM, "out of host memory");
    p->device = device; p->width = width; p->height = height; p->capacity = capacity; p->depth = depth;
    p->max_dist = max_dist; p->area = area; p->max_level = max_level;
    auto fail = [&](int code) { slam_amd_pipeline_destroy(p); return code; };
    int rc;
    if ((rc = slam_amd_extractor_create(params, width, height, 1, capacity, device, &p->ex))) return fail(rc);
    if ((rc = slam_amd_matcher_create(device, &p->m))) return fail(rc);
    if ((rc = slam_amd_matcher_reserve(p->m, 1, capacity, capacity))) return fail(rc);
    auto hip = [&](hipError_t e, const char* what) { return e == hipSuccess ? 0 : hip_fail(e, what, __FILE__, __LINE__); };
    if ((rc = hip(hipSetDevice(device), "hipSetDevice"))) return fail(rc);
    for (hipStream_t* st : {&p->s_copy, &p->s_comp, &p->s_back})
        if ((rc = hip(hipStreamCreateWithFlags(st, hipStreamNonBlocking), "hipStreamCreateWithFlags"))) return fail(rc);
    p->slots.resize((size_t)depth);
    const size_t img = (size_t)width * height, cap = (size_t)capacity;
    for (auto& s : p->slots) {
        if ((rc = hip(hipHostMalloc((void**)&s.h_img, img, hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_img, img), "hipMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_kp, cap * sizeof(slam_amd_keypoint)), "hipMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_desc, cap * 32), "hipMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_cnt, 256), "hipMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_match, cap * sizeof(slam_amd_dmatch)), "hipMalloc")) ||
            (rc = hip(hipMalloc((void**)&s.d_nmatch, 256), "hipMalloc")) ||
            (rc = hip(hipHostMalloc((void**)&s.h_kp, cap * sizeof(slam_amd_keypoint), hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipHostMalloc((void**)&s.h_desc, cap * 32, hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipHostMalloc((void**)&s.h_cnt, 256, hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipHostMalloc((void**)&s.h_match, cap * sizeof(slam_amd_dmatch), hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipHostMalloc((void**)&s.h_nmatch, 256, hipHostMallocDefault), "hipHostMalloc")) ||
            (rc = hip(hipEventCreateWithFlags(&s.ev_h2d, hipEventDisableTiming), "hipEventCreate")) ||
            (rc = hip(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming), "hipEventCreate")) ||
            (rc = hip(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming), "hipEventCreate")) ||
            (rc = hip(hipMemset(s.d_cnt, 0, 256), "hipMemset")) || (rc = hip(hipMemset(s.d_nmatch, 0, 256), "hipMemset")))
            return fail(rc);
        *s.h_cnt = 0; *s.h_nmatch = 0;
    }
    *out = p;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pipeline_input(slam_amd_pipeline* p, long long frame, uint8_t** host_image, int* row_stride) {
    if (!p || !host_image) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    if (frame != p->submitted) SA_FAIL(SLAM_AMD_ERR_ARG, "frames are submitted in order: ask for the input buffer of the next frame");
    auto& s = p->slots[(size_t)(frame % p->depth)];
    SA_HIP(hipSetDevice(p->device));
    // the slot's previous occupant (frame - depth): its image has been uploaded and its results are out
    if (s.frame >= 0) SA_HIP(hipEventSynchronize(s.ev_out));
    *host_image = s.h_img;
    if (row_stride) *row_stride = p->width;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pipeline_submit(slam_amd_pipeline* p, long long frame) {
    if (!p) SA_FAIL(SLAM_AMD_ERR_ARG, "pipeline is null");
    if (frame != p->submitted) SA_FAIL(SLAM_AMD_ERR_ARG, "frames are submitted in order");
    SA_HIP(hipSetDevice(p->device));
    auto& s = p->slots[(size_t)(frame % p->depth)];
    const bool reused = s.frame >= 0;
    // upload: the slot's device image was last read by frame - depth's extraction
    if (reused) SA_HIP(hipStreamWaitEvent(p->s_copy, s.ev_done, 0));
    SA_HIP(hipMemcpyAsync(s.d_img, s.h_img, (size_t)p->width * p->height, hipMemcpyHostToDevice, p->s_copy));
    SA_HIP(hipEventRecord(s.ev_h2d, p->s_copy));
    // compute: after the upload, and after frame - depth's results have left the slot
    SA_HIP(hipStreamWaitEvent(p->s_comp, s.ev_h2d, 0));
    if (reused) SA_HIP(hipStreamWaitEvent(p->s_comp, s.ev_out, 0));
    int rc = slam_amd_extract_batch_async(p->ex, s.d_img, 1, 0, p->width, s.d_kp, s.d_desc, s.d_cnt, p->capacity, p->s_comp);
    if (rc) return rc;
    if (frame > 0) {
        auto& prev = p->slots[(size_t)((frame - 1) % p->depth)];
        rc = slam_amd_frame_match_async(p->m, prev.d_desc, prev.d_kp, p->capacity, prev.d_cnt, s.d_desc, s.d_kp, p->capacity,
                                        s.d_cnt, nullptr, 0, nullptr, p->max_dist, p->area, p->max_level, s.d_match,
                                        s.d_nmatch, p->s_comp);
        if (rc) return rc;
    }
    SA_HIP(hipEventRecord(s.ev_done, p->s_comp));
    // download
    SA_HIP(hipStreamWaitEvent(p->s_back, s.ev_done, 0));
    const size_t cap = (size_t)p->capacity;
    SA_HIP(hipMemcpyAsync(s.h_cnt, s.d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, p->s_back));
    SA_HIP(hipMemcpyAsync(s.h_kp, s.d_kp, cap * sizeof(slam_amd_keypoint), hipMemcpyDeviceToHost, p->s_back));
    SA_HIP(hipMemcpyAsync(s.h_desc, s.d_desc, cap * 32, hipMemcpyDeviceToHost, p->s_back));
    if (frame > 0) {
        SA_HIP(hipMemcpyAsync(s.h_nmatch, s.d_nmatch, sizeof(int32_t), hipMemcpyDeviceToHost, p->s_back));
        SA_HIP(hipMemcpyAsync(s.h_match, s.d_match, cap * sizeof(slam_amd_dmatch), hipMemcpyDeviceToHost, p->s_back));
    }
    SA_HIP(hipEventRecord(s.ev_out, p->s_back));
    s.frame = frame;
    p->submitted = frame + 1;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_pipeline_result(slam_amd_pipeline* p, long long frame, const slam_amd_keypoint** kp,
                                        const uint8_t** desc, int* n, const slam_amd_dmatch** matches, int* n_matches) {
    if (!p) SA_FAIL(SLAM_AMD_ERR_ARG, "pipeline is null");
    if (frame < 0 || frame >= p->submitted) SA_FAIL(SLAM_AMD_ERR_ARG, "frame has not been submitted");
    auto& s = p->slots[(size_t)(frame % p->depth)];
    if (s.frame != frame) SA_FAIL(SLAM_AMD_ERR_ARG, "the slot of this frame has been reused (results live for depth - 1 more frames)");
    SA_HIP(hipSetDevice(p->device));
    SA_HIP(hipEventSynchronize(s.ev_out));
    if (kp) *kp = s.h_kp;
    if (desc) *desc = s.h_desc;
    if (n) *n = std::min(*s.h_cnt, p->capacity);
    if (matches) *matches = s.h_match;
    if (n_matches) *n_matches = frame > 0 ? *s.h_nmatch : 0;
    return SLAM_AMD_OK;
}
