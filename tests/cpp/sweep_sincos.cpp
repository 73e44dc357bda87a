// Exhaustive check of the library's cos / sin (slam_amd_sincos_f32: double polynomial, rounded to
// float -- the routine the descriptor kernel runs on the device and the host) against libm's
// double cos / sin rounded to float, which is what the CPU oracle uses
// (oracle/orb_oracle.c, orc_brief_descriptor): EVERY float from 0 to just above 2 pi, about
// 1.09e9 values.  rBRIEF rounds the rotated pattern coordinates to pixels, so one differing
// last bit could flip a descriptor bit; this sweep shows there is none.  CPU only.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/slam_amd.h"

int main() {
    const float top = 6.2831860f;                 // a few ulps above float(2 pi) = 6.2831855
    uint32_t last;
    std::memcpy(&last, &top, 4);
    long long bad = 0;
    uint32_t first_bad = 0;
#pragma omp parallel for schedule(static, 1 << 20) reduction(+ : bad)
    for (long long bits = 0; bits <= (long long)last; bits++) {
        const uint32_t u = (uint32_t)bits;
        float a;
        std::memcpy(&a, &u, 4);
        float c, s;
        slam_amd_sincos_f32(a, &c, &s);
        const float rc = (float)std::cos((double)a), rs = (float)std::sin((double)a);
        if (c != rc || s != rs) {
            bad++;
#pragma omp critical
            if (!first_bad) first_bad = u;
        }
    }
    printf("%lld of %lld float angles in [0, %.7f] differ from libm", bad, (long long)last + 1, (double)top);
    if (bad) printf(" (first: bits 0x%08x)", first_bad);
    printf("\n");
    return bad ? 1 : 0;
}
