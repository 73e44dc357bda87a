// Micro-benchmark: issue rate of single VALU opcodes on gfx950 (independent chains,
// operands in registers) to tell which instruction bounds the Hamming inner loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void op_kernel(const unsigned* in, unsigned* out, int iters, unsigned long long* clk) {
    unsigned a[16];
    for (int i = 0; i < 16; i++) a[i] = in[(threadIdx.x + i * 7) & 1023];
    unsigned b = in[blockIdx.x & 1023];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 3) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 6) asm volatile("v_bcnt_u32_b32 %0, %0, 0" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    unsigned r = 0;
    for (int i = 0; i < 16; i++) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
static void run(const char* name, const unsigned* d_in, unsigned* d_out, unsigned long long* d_clk, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((op_kernel<OP>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 200, d_clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((op_kernel<OP>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk[2];
    CHECK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
    double ops = (double)blocks * 256 * 16 * iters;
    double ghz = (double)clk[0] / ((double)clk[1] / 100e6) / 1e9;
    printf("%-16s blocks=%5d %.3f ms  %.2f Tlane-op/s  clock %.2f GHz  lane-ops/clk/CU %.1f\n", name, blocks, ms,
           ops / ms / 1e9, ghz, ops / (ms * 1e-3) / (ghz * 1e9) / 256.0);
}

int main() {
    unsigned* d_in; unsigned* d_out; unsigned long long* d_clk;
    CHECK(hipMalloc(&d_in, 4096)); CHECK(hipMalloc(&d_out, 4096 * 256 * 4)); CHECK(hipMalloc(&d_clk, 16));
    unsigned h[1024];
    for (int i = 0; i < 1024; i++) h[i] = 2654435761u * (i + 1);
    CHECK(hipMemcpy(d_in, h, 4096, hipMemcpyHostToDevice));
    for (int bpc : {2, 8}) {
        int blocks = 256 * bpc;
        run<0>("v_xor_b32", d_in, d_out, d_clk, blocks);
        run<1>("v_bcnt acc", d_in, d_out, d_clk, blocks);
        run<6>("v_bcnt 0", d_in, d_out, d_clk, blocks);
        run<2>("v_add_u32", d_in, d_out, d_clk, blocks);
        run<3>("v_min_u32", d_in, d_out, d_clk, blocks);
        run<4>("v_lshl_or_b32", d_in, d_out, d_clk, blocks);
        run<5>("v_min3_u32", d_in, d_out, d_clk, blocks);
        run<7>("v_and_b32", d_in, d_out, d_clk, blocks);
    }
    return 0;
}
