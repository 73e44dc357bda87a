// Input front end: BGR -> grey and bilinear resize to the working width in one pass
// (the reference does cv::cvtColor + cv::resize on the CPU, main.cpp:124-129).
// A destination pixel needs the grey value of four source pixels; they are computed
// from the BGR bytes on the fly with the rounding cvtColor applies to its 8-bit
// output, so the result equals the two-step form.  With the 3x reduction of a
// 1080p stream two thirds of the source pixels are never touched.
#include "common.h"
#include "kernels.h"

namespace sa {

__device__ __forceinline__ int grey_of(const uint8_t* p) {
    return (3735 * (int)p[0] + 19235 * (int)p[1] + 9798 * (int)p[2] + (1 << 14)) >> 15;
}

__global__ __launch_bounds__(256) void frontend_kernel(FrontendArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (x >= a.dw || y >= a.dh) return;
    // tables hold offsets already clamped to the image: every read below is in range
    const int x0 = a.xofs[x], x1 = min(x0 + 1, a.sw - 1);
    const int ax0 = a.xtap[2 * x], ax1 = a.xtap[2 * x + 1];
    const int yo = a.yofs[y];
    const int y0 = min(max(yo, 0), a.sh - 1), y1 = min(max(yo + 1, 0), a.sh - 1);
    const int by0 = a.ytap[2 * y], by1 = a.ytap[2 * y + 1];
    const uint8_t* src = a.src + (size_t)b * a.src_frame_pitch;
    const uint8_t* r0 = src + (size_t)y0 * a.src_pitch;
    const uint8_t* r1 = src + (size_t)y1 * a.src_pitch;
    int g00, g01, g10, g11;
    if (a.channels == 3) {
        g00 = grey_of(r0 + 3 * x0); g01 = grey_of(r0 + 3 * x1);
        g10 = grey_of(r1 + 3 * x0); g11 = grey_of(r1 + 3 * x1);
    } else {
        g00 = r0[x0]; g01 = r0[x1]; g10 = r1[x0]; g11 = r1[x1];
    }
    const int S0 = g00 * ax0 + g01 * ax1;
    const int S1 = g10 * ax0 + g11 * ax1;
    const int v = (((by0 * (S0 >> 4)) >> 16) + ((by1 * (S1 >> 4)) >> 16) + 2) >> 2;
    a.dst[(size_t)b * a.dst_frame_pitch + (size_t)y * a.dst_pitch + x] = (uint8_t)min(max(v, 0), 255);
}

int launch_frontend(const FrontendArgs& a, int batch, hipStream_t st) {
    if (a.dw <= 0 || a.dh <= 0 || batch <= 0) return SLAM_AMD_OK;
    dim3 grid(div_up(a.dw, 64), div_up(a.dh, 4), batch);
    hipLaunchKernelGGL(frontend_kernel, grid, dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
    return SLAM_AMD_OK;
}

}  // namespace sa
