// Micro-benchmark: the FAST ring of a pixel gathered from a byte tile in LDS (pitch 72) by every lane at its own
// position -- as 17 byte loads, or as 7 loads of 8 bytes at unaligned addresses (one per row of the 7 x 7 window)
// with the ring bytes picked out of the registers.  Checks that both give the same sums and times them at 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_gather.hip -o tools/lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int P = 72, ROWS = 40;

__device__ __forceinline__ uint2 lds_read8(const unsigned char* p) {
    uint2 v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* seeds, unsigned* out, int iters) {
    __shared__ __align__(16) unsigned char tile[ROWS * P + 16];
    for (int i = threadIdx.x; i < ROWS * P + 16; i += 256) tile[i] = (unsigned char)(i * 37 + (i >> 5) * 11 + blockIdx.x);
    __syncthreads();
    unsigned s = seeds[threadIdx.x] + blockIdx.x, acc = 0;
    for (int it = 0; it < iters; it++) {
        s = s * 1664525u + 1013904223u;
        int row = 3 + (int)((s >> 8) % 34u), col = 3 + (int)((s >> 16) % 66u);
        if (MODE == 3) {     // the lanes of a wave at positions in raster order, about one pixel in seven (byte loads)
            const unsigned lin = ((__builtin_amdgcn_readfirstlane(s) >> 8) % 1700u) + (threadIdx.x & 63u) * 7u + ((s >> 20) % 5u);
            row = 3 + (int)((lin / 66u) % 34u); col = 3 + (int)(lin % 66u);
        }
        const unsigned char* p = tile + row * P + col;
        unsigned sum = 0;
        if (MODE == 0 || MODE == 3) {
            constexpr int off[16] = {3 * P, 3 * P + 1, 2 * P + 2, P + 3, 3, -P + 3, -2 * P + 2, -3 * P + 1,
                                     -3 * P, -3 * P - 1, -2 * P - 2, -P - 3, -3, P - 3, 2 * P - 2, 3 * P - 1};
#pragma unroll
            for (int j = 0; j < 16; j++) sum += (unsigned)p[off[j]] * (unsigned)(j + 1);
            sum += p[0];
        } else {
            uint2 w[7];
#pragma unroll
            for (int dy = -3; dy <= 3; dy++) {
                const unsigned char* q = p + dy * P - 3;
                if (MODE == 1) {
                    const unsigned a = (unsigned)(size_t)q;
                    asm volatile("ds_read_b64 %0, %1" : "=v"(w[dy + 3]) : "v"(a) : "memory");
                } else {
                    w[dy + 3] = *reinterpret_cast<const uint2*>(q - ((size_t)q & 3));   // aligned stand-in (wrong bytes): cost of aligned 8-byte gathers
                }
            }
            if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            auto b = [&](int dy, int dx) { const uint2 v = w[dy + 3]; const int kx = dx + 3; return kx < 4 ? (v.x >> (8 * kx)) & 255u : (v.y >> (8 * (kx - 4))) & 255u; };
            constexpr int dys[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
            constexpr int dxs[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
#pragma unroll
            for (int j = 0; j < 16; j++) sum += b(dys[j], dxs[j]) * (unsigned)(j + 1);
            sum += b(0, 0);
        }
        acc += sum;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const int grid = 256 * 8, iters = 2000;
    unsigned *d_seeds, *d_out[3];
    CHECK(hipMalloc(&d_seeds, 1024));
    unsigned hs[256];
    for (int i = 0; i < 256; i++) hs[i] = 2654435761u * (i + 1);
    CHECK(hipMemcpy(d_seeds, hs, 1024, hipMemcpyHostToDevice));
    for (auto& p : d_out) CHECK(hipMalloc(&p, grid * 256 * 4));
    const char* names[4] = {"17 byte loads", "7 unaligned 8-byte loads", "7 aligned 8-byte loads (wrong bytes)", "17 byte loads, lanes in raster order"};
    for (int mode = 0; mode < 4; mode++) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d_seeds, d_out[0], iters);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d_seeds, d_out[1], iters);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d_seeds, d_out[2], iters);
            else hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d_seeds, d_out[2], iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // wave-level gathers per CU: grid * 4 waves * iters / 256 CUs
        printf("%-40s %8.1f us: %.1f ns per ring and wave on a CU\n", names[mode], best * 1e3, best * 1e6 / ((double)grid * 4 * iters / 256));
    }
    unsigned* h0 = (unsigned*)malloc(grid * 256 * 4); unsigned* h1 = (unsigned*)malloc(grid * 256 * 4);
    CHECK(hipMemcpy(h0, d_out[0], grid * 256 * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h1, d_out[1], grid * 256 * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < grid * 256; i++) bad += h0[i] != h1[i];
    printf("unaligned 8-byte loads: %d of %d results differ from the byte loads\n", bad, grid * 256);
    return 0;
}
