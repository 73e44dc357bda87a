// The 4K stream configuration driven from C++ alone (no Python in the process): 3840 x 2160 frames in the
// pipeline's pinned input ring, 10000 features, every frame matched against the one before, results read back
// on the host.  Prints the sustained rate.  This is the process profiled for the copy / compute overlap
// (rocprofv3 --kernel-trace --memory-copy-trace -- ./stream4k_host <batch> <frames>).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/slam_amd.hpp"

static unsigned rng_state = 777;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static void paint(uint8_t* img, int w, int h, int stride, int shift) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)y * stride + x] = (uint8_t)(60 + 40 * std::sin((x + shift) * 0.05) + 30 * std::cos(y * 0.07));
    rng_state = 777;
    for (int i = 0; i < 40000; i++) {
        const int cx = (int)(rnd() % w) + shift, cy = (int)(rnd() % h), r = 3 + (int)(rnd() % 12);
        const uint8_t v = (uint8_t)(rnd() % 256);
        for (int y = std::max(0, cy - r); y < std::min(h, cy + r); y++)
            for (int x = std::max(0, cx - r); x < std::min(w, cx + r); x++) img[(size_t)y * stride + x] = v;
    }
}

int main(int argc, char** argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 1, frames = argc > 2 ? atoi(argv[2]) : 200;
    const int W = 3840, H = 2160, NF = 10000, CAP = 12288, DEPTH = 4;
    try {
        slam_amd_orb_params p;
        slam_amd_default_orb_params(&p);
        p.nfeatures = NF;
        slam_amd::FramePipeline pipe(p, W, H, CAP, DEPTH, 300, -1, -1, 0, batch);
        const int ring = DEPTH * batch;
        long long i = 0;
        for (; i < 2 * ring; i++) {                 // fill the ring once (a camera's DMA would keep writing there), warm up
            int stride = 0;
            uint8_t* buf = pipe.input(&stride);
            if (i < ring) paint(buf, W, H, stride, 5 * (int)i);
            pipe.submit();
        }
        pipe.result(i - 1);
        const auto t0 = std::chrono::steady_clock::now();
        const long long end = i + (long long)(frames / batch) * batch;
        long long kp_total = 0, match_total = 0;
        for (; i < end; i++) {
            int stride = 0;
            pipe.input(&stride);
            pipe.submit();
            const auto r = pipe.result(i - 2 * batch);   // the consumer reads two batches behind the producer
            kp_total += r.n; match_total += r.nMatches;
        }
        const auto last = pipe.result(end - 1);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("batch %d: %lld frames in %.3f ms = %.1f frames/s (%.2f GB/s uploaded); last frame %d keypoints, %d matches\n", batch,
               end - 2 * ring, dt * 1e3, (double)(end - 2 * ring) / dt, (double)(end - 2 * ring) * W * H / dt / 1e9, last.n, last.nMatches);
        (void)kp_total; (void)match_total;
    } catch (const std::exception& e) {
        printf("exception: %s\n", e.what());
        return 2;
    }
    return 0;
}
