// Micro-benchmark: do vector ALU instructions issue while the matrix pipe works on gfx950?
// Every wave loops over {M products v_mfma_i32_32x32x32_i8 on two accumulators, V independent
// v_add_u32}; the products and the adds either sit in two separate runs (as a compiler leaves a
// loop of "products, then epilogue") or interleaved, 2 products then V / (M / 2) adds.
// Prints cycles per loop iteration for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int M, int V, int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* in, unsigned* out, int iters, unsigned long long* clk) {
    unsigned a[16];
    for (int i = 0; i < 16; i++) a[i] = in[(threadIdx.x + i * 7) & 1023];
    unsigned b = in[blockIdx.x & 1023];
    v4i fa = {(int)a[0], (int)a[1], (int)a[2], (int)a[3]}, fb = {(int)a[4], (int)a[5], (int)a[6], (int)a[7]};
    v16i acc0 = {}, acc1 = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < M; m += 2) {
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc1) : "v"(fa), "v"(fb));
            }
#pragma unroll
            for (int v = 0; v < V; v++) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[v & 15]) : "v"(b));
        } else {
#pragma unroll
            for (int m = 0; m < M; m += 2) {
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc1) : "v"(fa), "v"(fb));
#pragma unroll
                for (int v = 0; v < V / (M / 2); v++) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[v & 15]) : "v"(b));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    unsigned r = 0;
    for (int i = 0; i < 16; i++) r ^= a[i] ^ (unsigned)acc0[i] ^ (unsigned)acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int M, int V, int MODE>
static void run(const char* what, const unsigned* d_in, unsigned* d_out, unsigned long long* d_clk) {
    for (int wps : {1, 2, 4}) {
        const int iters = 2000, blocks = 256 * wps;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k<M, V, MODE>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 100, d_clk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<M, V, MODE>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long clk[2];
        CHECK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)clk[0] / ((double)clk[1] / 100e6) / 1e9;
        const double cyc = ms * 1e-3 * ghz * 1e9 / (double)(wps * iters);
        printf("%-34s %d waves/SIMD: %.0f cycles per wave-iteration (products alone would be %d, adds alone ~%d)\n", what, wps, cyc,
               M * 32, V * 2);
    }
}

int main() {
    unsigned* d_in; unsigned* d_out; unsigned long long* d_clk;
    CHECK(hipMalloc(&d_in, 4096)); CHECK(hipMalloc(&d_out, 1024 * 256 * 4)); CHECK(hipMalloc(&d_clk, 16));
    unsigned h[1024];
    for (int i = 0; i < 1024; i++) h[i] = 2654435761u * (i + 1);
    CHECK(hipMemcpy(d_in, h, 4096, hipMemcpyHostToDevice));
    run<36, 0, 0>("36 products", d_in, d_out, d_clk);
    run<0, 288, 0>("288 adds", d_in, d_out, d_clk);
    run<36, 288, 0>("36 products, then 288 adds", d_in, d_out, d_clk);
    run<36, 288, 1>("2 products + 16 adds, 18 times", d_in, d_out, d_clk);
    run<36, 144, 0>("36 products, then 144 adds", d_in, d_out, d_clk);
    run<36, 144, 1>("2 products + 8 adds, 18 times", d_in, d_out, d_clk);
    return 0;
}
