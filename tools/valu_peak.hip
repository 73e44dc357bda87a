// Micro-benchmark: peak rate of 256-bit Hamming compares on gfx950 when nothing
// but the arithmetic runs (8 x v_xor_b32 + 8 x v_bcnt_u32_b32 per compare, operands
// in registers, no memory traffic).  Prints compares/s for a given number of waves
// per SIMD; the best figure is the integer ceiling matching kernels are priced against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) {
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

template <int CHAINS, bool WITH_KEY>
__global__ __launch_bounds__(256) void peak_kernel(const unsigned* in, unsigned* out, int iters) {
    unsigned q[CHAINS][8];
    for (int c = 0; c < CHAINS; c++)
        for (int w = 0; w < 8; w++) q[c][w] = in[(threadIdx.x * 8 + w + c * 13) & 1023];
    unsigned t[8];
    for (int w = 0; w < 8; w++) t[w] = in[(blockIdx.x + w) & 1023];
    unsigned best[CHAINS];
    for (int c = 0; c < CHAINS; c++) best[c] = 0xffffffffu;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            unsigned d = 0;
#pragma unroll
            for (int w = 0; w < 8; w++) d = bcnt_acc(q[c][w] ^ t[w], d);
            if (WITH_KEY) {
                unsigned key = (d << 20) | (unsigned)i;
                best[c] = min(best[c], key);
            } else {
                best[c] += d;
            }
        }
        // change the "train" operand cheaply so the compiler cannot hoist the loop body
#pragma unroll
        for (int w = 0; w < 8; w++) asm volatile("" : "+v"(t[w]));
    }
    unsigned r = 0;
    for (int c = 0; c < CHAINS; c++) r ^= best[c];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int CHAINS, bool WITH_KEY>
static void run(const char* name, const unsigned* d_in, unsigned* d_out, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((peak_kernel<CHAINS, WITH_KEY>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 200);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((peak_kernel<CHAINS, WITH_KEY>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double cmps = (double)blocks * 256 * CHAINS * iters;
    printf("%-28s blocks=%5d  %.3f ms  %.1f Gcmp/s\n", name, blocks, best, cmps / best / 1e6);
}

int main() {
    unsigned* d_in; unsigned* d_out;
    CHECK(hipMalloc(&d_in, 4096));
    CHECK(hipMalloc(&d_out, 4096 * 256 * 4));
    unsigned h[1024];
    for (int i = 0; i < 1024; i++) h[i] = 2654435761u * (i + 1);
    CHECK(hipMemcpy(d_in, h, 4096, hipMemcpyHostToDevice));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int bpc : {1, 2, 4, 8}) {
        int blocks = p.multiProcessorCount * bpc;
        run<2, false>("2 chains, sum", d_in, d_out, blocks);
        run<4, false>("4 chains, sum", d_in, d_out, blocks);
        run<4, true>("4 chains, key+min", d_in, d_out, blocks);
        run<8, true>("8 chains, key+min", d_in, d_out, blocks);
    }
    return 0;
}
