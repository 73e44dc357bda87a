// Probe: 256-bit Hamming distances of a 32 x 32 block of (train, query) pairs from four
// v_mfma_scale_f32_32x32x64_f8f6f4 products on FP4 (E2M1) operands, checked against popcounts, and the
// instruction's cycles per product.  Train bits -> nibbles {0, 0.5}, query bits -> {+1, -1}, query-side block
// scale 2^6: an accumulator ends as 32 (H - popcount(query)) + its initial value, exactly.
// Build: hipcc --offload-arch=gfx950 -O3 tools/f4_probe.hip -o tools/f4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// dword -> 4 dwords of 8 nibbles: nibble j of output k is bit k + 4 j
__device__ __forceinline__ void expand_train(unsigned w, unsigned (&o)[4]) {
    for (int k = 0; k < 4; k++) o[k] = (w >> k) & 0x11111111u;                       // 0b0001 = 0.5
}
__device__ __forceinline__ void expand_query(unsigned w, unsigned (&o)[4]) {
    for (int k = 0; k < 4; k++) o[k] = 0x22222222u | (((w >> k) & 0x11111111u) << 3);   // 0b0010 = +1, 0b1010 = -1
}

// one wave: train rows 0..31 and queries 0..31, descriptors of 8 dwords.  out[q * 32 + t] = accumulator
__global__ void hamming_block(const unsigned* train, const unsigned* query, float* out, int scale_exp) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v16f acc;
    for (int i = 0; i < 16; i++) acc[i] = (float)(8192 + 8 * (i >> 2) + 4 * h + (i & 3));
    const int sa = 127, sb = scale_exp;       // E8M0: 2^(e - 127), byte 0 of the scale registers
    // product s takes descriptor dwords 2s and 2s + 1: lane half h supplies dword 2s + h (32 bits = 32 FP4 values)
    for (int s = 0; s < 4; s++) {
        unsigned ta[4], qb[4];
        expand_train(train[r * 8 + 2 * s + h], ta);
        expand_query(query[r * 8 + 2 * s + h], qb);
        v8i A = {(int)ta[0], (int)ta[1], (int)ta[2], (int)ta[3], 0, 0, 0, 0};
        v8i B = {(int)qb[0], (int)qb[1], (int)qb[2], (int)qb[3], 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, sa, 0, sb);
    }
    for (int i = 0; i < 16; i++) {
        const int t = 8 * (i >> 2) + 4 * h + (i & 3);
        out[r * 32 + t] = acc[i];
    }
}

template <int KIND>   // 0: FP4 32x32x64, 1: i8 32x32x32, 2: FP4 16x16x128
__global__ __launch_bounds__(256) void rate(const unsigned* in, float* out, int iters, unsigned long long* clk) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v16i __attribute__((ext_vector_type(16)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    v8i A = {(int)in[threadIdx.x], 1, 2, 3, 0, 0, 0, 0}, B = {(int)in[threadIdx.x + 256], 4, 5, 6, 0, 0, 0, 0};
    v4i a4 = {A[0], A[1], A[2], A[3]}, b4 = {B[0], B[1], B[2], B[3]};
    v16f acc0 = {}, acc1 = {};
    v16i i0 = {}, i1 = {};
    v4f s0 = {}, s1 = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (KIND == 0) {
                acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc0, 4, 4, 0, 127, 0, 127);
                acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B, A, acc1, 4, 4, 0, 127, 0, 127);
            } else if (KIND == 1) {
                i0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, i0, 0, 0, 0);
                i1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b4, a4, i1, 0, 0, 0);
            } else {
                s0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, s0, 4, 4, 0, 127, 0, 127);
                s1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B, A, s1, 4, 4, 0, 127, 0, 127);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
    for (int i = 0; i < 16; i++) r += acc0[i] + acc1[i] + (float)i0[i] + (float)i1[i];
    for (int i = 0; i < 4; i++) r += s0[i] + s1[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    // ---- exactness ----
    std::vector<unsigned> train(32 * 8), query(32 * 8);
    unsigned x = 12345u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x ^ (x >> 13); };
    for (auto& v : train) v = rnd();
    for (auto& v : query) v = rnd();
    for (int d = 0; d < 8; d++) { train[3 * 8 + d] = 0u; train[4 * 8 + d] = 0xffffffffu; query[5 * 8 + d] = 0xffffffffu; query[6 * 8 + d] = 0u; query[7 * 8 + d] = train[9 * 8 + d]; }
    unsigned *d_t, *d_q; float* d_o;
    CHECK(hipMalloc(&d_t, 1024)); CHECK(hipMalloc(&d_q, 1024)); CHECK(hipMalloc(&d_o, 4096));
    CHECK(hipMemcpy(d_t, train.data(), 1024, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_q, query.data(), 1024, hipMemcpyHostToDevice));
    for (int se : {133, 132}) {
        hipLaunchKernelGGL(hamming_block, dim3(1), dim3(64), 0, 0, d_t, d_q, d_o, se);
        CHECK(hipDeviceSynchronize());
        std::vector<float> o(1024);
        CHECK(hipMemcpy(o.data(), d_o, 4096, hipMemcpyDeviceToHost));
        int bad = 0;
        const int unit = se == 133 ? 32 : 16;
        for (int q = 0; q < 32; q++)
            for (int t = 0; t < 32; t++) {
                int H = 0, pq = 0;
                for (int d = 0; d < 8; d++) { H += __builtin_popcount(train[t * 8 + d] ^ query[q * 8 + d]); pq += __builtin_popcount(query[q * 8 + d]); }
                const float want = (float)(8192 + t + unit * (H - pq));
                if (o[q * 32 + t] != want) { if (bad < 5) printf("  q %d t %d: got %.3f want %.1f (H %d pq %d)\n", q, t, o[q * 32 + t], want, H, pq); bad++; }
            }
        printf("scale exponent %d: %d of 1024 accumulators differ from %d (H - popcount(q)) + 8192 + row\n", se, bad, unit);
    }
    // ---- rate ----
    unsigned* d_in; float* d_out; unsigned long long* d_clk;
    CHECK(hipMalloc(&d_in, 8192)); CHECK(hipMalloc(&d_out, 1024 * 256 * 4)); CHECK(hipMalloc(&d_clk, 16));
    std::vector<unsigned> hin(2048);
    for (auto& v : hin) v = rnd() & 0x77777777u;
    CHECK(hipMemcpy(d_in, hin.data(), 8192, hipMemcpyHostToDevice));
    for (int kind = 0; kind < 3; kind++)
        for (int wps : {1, 2}) {
            const int iters = 4000;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256 * wps), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
                else if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(256 * wps), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
                else hipLaunchKernelGGL(rate<2>, dim3(256 * wps), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
            }
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long clk[2];
            CHECK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)clk[0] / ((double)clk[1] / 100.0) / 1e3;
            const double macs = kind == 0 ? 65536.0 : 32768.0;
            printf("%-22s %d waves/SIMD: %.1f cycles per product and SIMD, %.2f GHz, %.0f Tmac/s\n",
                   kind == 0 ? "FP4 32x32x64" : kind == 1 ? "i8 32x32x32" : "FP4 16x16x128", wps,
                   ms * 1e-3 * ghz * 1e9 / ((double)iters * 16 * wps), ghz, 1024.0 * wps * iters * 16 * macs / (ms * 1e-3) / 1e12);
        }
    return 0;
}
