// Micro-benchmark: what clock does gfx950 hold while every SIMD issues v_mfma_i32_32x32x32_i8 back to back,
// and how many cycles does a product take then?  Launches of growing length (the power controller needs time
// to react), 1 and 2 waves per SIMD, with and without an LDS operand read per product.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o tools/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <bool LDS>
__global__ __launch_bounds__(256) void k(const unsigned* in, unsigned* out, int iters, unsigned long long* clk) {
    __shared__ v4i tile[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tile[i] = v4i{(int)in[i & 1023], 1, 2, 3};
    __syncthreads();
    v4i fa = {(int)in[threadIdx.x], 1, 2, 3}, fb = {(int)in[threadIdx.x + 256], 4, 5, 6};
    v16i acc0 = {}, acc1 = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (LDS) fa = tile[(threadIdx.x + 64 * m + it) & 1023];
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, fa, acc1, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    unsigned r = 0;
    for (int i = 0; i < 16; i++) r ^= (unsigned)acc0[i] ^ (unsigned)acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <bool LDS>
static void run(const unsigned* d_in, unsigned* d_out, unsigned long long* d_clk) {
    for (int wps : {1, 2})
        for (int iters : {200, 2000, 20000, 200000}) {
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL((k<LDS>), dim3(256 * wps), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ev_ms = 0; CHECK(hipEventElapsedTime(&ev_ms, e0, e1));
            unsigned long long clk[2];
            CHECK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
            const double us = (double)clk[1] / 100.0;
            const double ghz = (double)clk[0] / us / 1e3;
            printf("%s %d waves/SIMD %7d iterations: block 0 ran %9.1f us, kernel (events) %9.1f us, %.3f GHz, %.1f cycles per product and SIMD by the kernel time, %.0f Tmac/s\n",
                   LDS ? "LDS operand " : "register ops", wps, iters, us, ev_ms * 1e3, ghz, ev_ms * 1e3 * ghz * 1e3 / ((double)iters * 16 * wps),
                   1024.0 * wps * iters * 16 * 32768.0 / (ev_ms * 1e-3) / 1e12);
        }
}

int main() {
    unsigned* d_in; unsigned* d_out; unsigned long long* d_clk;
    CHECK(hipMalloc(&d_in, 8192)); CHECK(hipMalloc(&d_out, 1024 * 256 * 4)); CHECK(hipMalloc(&d_clk, 16));
    unsigned h[2048];
    for (int i = 0; i < 2048; i++) h[i] = 2654435761u * (i + 1);
    CHECK(hipMemcpy(d_in, h, 8192, hipMemcpyHostToDevice));
    run<false>(d_in, d_out, d_clk);
    run<true>(d_in, d_out, d_clk);
    return 0;
}
