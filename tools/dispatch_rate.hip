// Micro-benchmark: how fast can the chip start workgroups of 256 threads with a given LDS allocation that do
// (almost) nothing, and how long does a grid of 32768 of them take?  The floor under any kernel of that grid.
// Build: hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate.hip -o tools/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LDS_BYTES, int SPIN>
__global__ __launch_bounds__(256) void k(unsigned* out, int n) {
    __shared__ unsigned s[LDS_BYTES / 4];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned v = s[(threadIdx.x + 1) & 255];
    for (int i = 0; i < SPIN; i++) v = v * 1664525u + 1013904223u;      // SPIN dependent multiply-adds per thread
    if (v == 0x12345678u && n == -1) out[blockIdx.x] = v;
}

template <int LDS_BYTES, int SPIN>
static void run(unsigned* d, int grid, int threads) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<LDS_BYTES, SPIN>), dim3(grid), dim3(threads), 0, 0, d, 0);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("grid %6d x %d threads, %5d B LDS, %4d multiply-adds per thread: %7.1f us = %.2f ns per workgroup\n", grid, threads, LDS_BYTES, SPIN,
           best * 1e3, best * 1e6 / grid);
}

int main() {
    unsigned* d; CHECK(hipMalloc(&d, 1 << 20));
    for (int grid : {32768, 16384, 8192}) {
        run<1024, 0>(d, grid, 256);
        run<20480, 0>(d, grid, 256);
        run<20480, 100>(d, grid, 256);
        run<20480, 400>(d, grid, 256);
        run<20480, 800>(d, grid, 256);
        run<40960, 800>(d, grid, 256);
        run<40960, 1600>(d, grid, 256);
    }
    run<40960, 1600>(d, 16384, 512);
    run<40960, 800>(d, 16384, 512);
    return 0;
}
