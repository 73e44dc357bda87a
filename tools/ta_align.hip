// What a 16-byte-per-lane global load costs the vector memory path on gfx950, by the alignment of the lanes'
// addresses: 64 lanes read 16 bytes each from 32 image rows (two lanes per row, as the describe kernel's
// moment pass does), the row start shifted by `offset` bytes from a 64-byte boundary.  Working set 4 MB
// per XCD-sized slice (L2 resident after the first pass); many waves per CU, loads only.
//   hipcc --offload-arch=gfx950 -O3 tools/ta_align.hip -o tools/ta_align && tools/ta_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_a1;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int LANES_PER_ROW>
__global__ __launch_bounds__(256) void loads(const unsigned char* img, int pitch, int rows, int offset, int iters, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const int r = lane / LANES_PER_ROW, c = lane % LANES_PER_ROW;
    u32x4 acc = {0, 0, 0, 0};
    unsigned row0 = (unsigned)(wave * 37) % (unsigned)(rows - 64);
    unsigned col0 = (unsigned)(wave * 64) % (unsigned)(pitch - 512);
    for (int i = 0; i < iters; i++) {
        const unsigned char* p = img + (size_t)(row0 + r) * pitch + (col0 & ~63u) + offset + 16 * c;
        acc ^= *reinterpret_cast<const u32x4_a1*>(p);
        row0 += 31; if (row0 >= (unsigned)(rows - 64)) row0 -= (unsigned)(rows - 64);
        col0 += 192; if (col0 >= (unsigned)(pitch - 512)) col0 -= (unsigned)(pitch - 512);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

int main() {
    const int pitch = 2048, rows = 2048;
    unsigned char* img; unsigned* out;
    CHECK(hipMalloc(&img, (size_t)pitch * rows + 256));
    CHECK(hipMemset(img, 7, (size_t)pitch * rows + 256));
    CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * 8, iters = 256;
    const int offsets[] = {0, 16, 64, 4, 1};
    for (int lpr : {2, 4, 8, 16}) {
        for (int off : offsets) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                CHECK(hipEventRecord(e0));
                if (lpr == 2) hipLaunchKernelGGL(loads<2>, dim3(blocks), dim3(256), 0, 0, img, pitch, rows, off, iters, out);
                else if (lpr == 4) hipLaunchKernelGGL(loads<4>, dim3(blocks), dim3(256), 0, 0, img, pitch, rows, off, iters, out);
                else if (lpr == 8) hipLaunchKernelGGL(loads<8>, dim3(blocks), dim3(256), 0, 0, img, pitch, rows, off, iters, out);
                else hipLaunchKernelGGL(loads<16>, dim3(blocks), dim3(256), 0, 0, img, pitch, rows, off, iters, out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double wave_loads = (double)blocks * 4 * iters;
            // cycles of one CU's address unit per wave-load, if that unit is what bounds the kernel (2.4 GHz, 256 CUs)
            printf("lanes/row %d offset %2d: %.3f ms  %.1f cycles per wave-load per CU\n", lpr, off, best, best * 1e-3 * 2.4e9 * 256 / wave_loads);
        }
    }
    return 0;
}
