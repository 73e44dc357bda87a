// Micro-benchmark: do scalar (SALU) instructions issue beside vector ones on gfx950, or in their place?
// Loop body: 16 independent full-rate v_add_u32 plus K scalar instructions (independent s_add_u32 /
// s_and_b64 on four registers), 8 waves per SIMD.  Prints cycles per loop iteration per SIMD, i.e.
// the issue time of (16 VALU + K SALU) x 8 waves / 8; if scalar instructions were free the number
// would not move with K until the scalar unit itself saturates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int K, int KIND>
__global__ __launch_bounds__(256) void mix_kernel(const unsigned* in, unsigned* out, int iters, unsigned long long* clk) {
    unsigned a[16];
    for (int i = 0; i < 16; i++) a[i] = in[(threadIdx.x + i * 7) & 1023];
    unsigned b = in[blockIdx.x & 1023];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            // K scalar instructions spread over the 16 vector ones
#pragma unroll
            for (int k = 0; k < (K * (i + 1)) / 16 - (K * i) / 16; k++) {
                if (KIND == 0) {
                    if ((i & 3) == 0) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
                    if ((i & 3) == 1) asm volatile("s_add_u32 s21, s21, 1" ::: "s21", "scc");
                    if ((i & 3) == 2) asm volatile("s_add_u32 s22, s22, 1" ::: "s22", "scc");
                    if ((i & 3) == 3) asm volatile("s_add_u32 s23, s23, 1" ::: "s23", "scc");
                } else if (KIND == 1) {
                    if ((i & 1) == 0) asm volatile("s_and_b64 s[20:21], s[20:21], exec" ::: "s20", "s21", "scc");
                    if ((i & 1) == 1) asm volatile("s_or_b64 s[22:23], s[22:23], exec" ::: "s22", "s23", "scc");
                } else if (KIND == 2) {
                    asm volatile("s_nop 0");
                } else if (KIND == 3) {
                    asm volatile("s_waitcnt lgkmcnt(0)");
                } else if (KIND == 4) {
                    asm volatile("s_mov_b32 s20, 0x12345" ::: "s20");
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    unsigned r = 0;
    for (int i = 0; i < 16; i++) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int K, int KIND>
static void run(const char* kind, const unsigned* d_in, unsigned* d_out, unsigned long long* d_clk) {
    const int iters = 4000, blocks = 256 * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix_kernel<K, KIND>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 200, d_clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix_kernel<K, KIND>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk[2];
    CHECK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)clk[0] / ((double)clk[1] / 100e6) / 1e9;
    // 8 waves per SIMD, each runs `iters` iterations
    const double cyc = ms * 1e-3 * ghz * 1e9 / (8.0 * iters);
    printf("%-12s 16 VALU + %2d scalar: %.3f ms, %.1f cycles per wave-iteration (%.2f per instruction)\n", kind, K, ms, cyc,
           cyc / (16 + K));
}

int main() {
    unsigned* d_in; unsigned* d_out; unsigned long long* d_clk;
    CHECK(hipMalloc(&d_in, 4096)); CHECK(hipMalloc(&d_out, 4096 * 256 * 4)); CHECK(hipMalloc(&d_clk, 16));
    unsigned h[1024];
    for (int i = 0; i < 1024; i++) h[i] = 2654435761u * (i + 1);
    CHECK(hipMemcpy(d_in, h, 4096, hipMemcpyHostToDevice));
    run<0, 0>("s_add_u32", d_in, d_out, d_clk);
    run<4, 0>("s_add_u32", d_in, d_out, d_clk);
    run<8, 0>("s_add_u32", d_in, d_out, d_clk);
    run<16, 0>("s_add_u32", d_in, d_out, d_clk);
    run<32, 0>("s_add_u32", d_in, d_out, d_clk);
    run<8, 1>("s_and/or_b64", d_in, d_out, d_clk);
    run<16, 1>("s_and/or_b64", d_in, d_out, d_clk);
    run<8, 2>("s_nop 0", d_in, d_out, d_clk);
    run<16, 2>("s_nop 0", d_in, d_out, d_clk);
    run<8, 3>("s_waitcnt", d_in, d_out, d_clk);
    run<16, 3>("s_waitcnt", d_in, d_out, d_clk);
    run<8, 4>("s_mov_b32", d_in, d_out, d_clk);
    run<16, 4>("s_mov_b32", d_in, d_out, d_clk);
    return 0;
}
