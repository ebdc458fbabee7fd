// check_describe_math.cc -- the branch-free fastAtan2 / sincosf of k_describe (device code compiled for the host through the SIMT shim)
// against the host libm's sinf / cosf (glibc 2.35, the FMA ifunc variant the oracle was verified against exhaustively) and the oracle's
// cv::fastAtan2 restatement: every float angle in [0, 360) at 1e-4 steps, a sweep of float bit patterns up to 6.5, 10^7 moment pairs.
#include "hip/hip_runtime.h"
#include "extractor_kernels.hip.h"
#include <cmath>
extern "C" float orbo_fast_atan2(float y, float x);
extern "C" float orbo_fast_atan2_fma(float y, float x);   // the ORBX_FLAG_ATAN_FMA form
static void libm_sincosf(float y, float *s, float *c) { *s = sinf(y); *c = cosf(y); }
#include <random>
int main() {
    long bad = 0, n = 0;
    for (int i = 0; i < 3600000; i++) {
        const float ang = i * 0.0001f;
        const float y = ang * (float)(3.14159265358979323846 / 180.f);
        float s, c, s2, c2; libm_sincosf(y, &s, &c); orbx::glibc_sincosf(y, &s2, &c2); n++;
        if (memcmp(&s, &s2, 4) || memcmp(&c, &c2, 4)) { if (bad < 5) printf("sincos %g: %a %a vs %a %a\n", ang, s, c, s2, c2); bad++; }
    }
    // tiny and small arguments, bit patterns swept
    for (uint32_t u = 0; u < 0x40d00000u; u += 977) { float y; memcpy(&y, &u, 4); float s, c, s2, c2; libm_sincosf(y, &s, &c); orbx::glibc_sincosf(y, &s2, &c2); n++;
        if (memcmp(&s, &s2, 4) || memcmp(&c, &c2, 4)) { if (bad < 10) printf("sincos bits %08x: %a %a vs %a %a\n", u, s, c, s2, c2); bad++; } }
    std::mt19937 rng(1);
    for (long i = 0; i < 10000000; i++) {
        const int m01 = (int)(rng() % 5800001) - 2900000, m10 = (int)(rng() % 5800001) - 2900000;
        const float a = orbo_fast_atan2((float)m01, (float)m10), b = orbx::fast_atan2_deg((float)m01, (float)m10, false); n++;
        if (memcmp(&a, &b, 4)) { if (bad < 15) printf("atan2 %d %d: %a vs %a\n", m01, m10, a, b); bad++; }
        const float af = orbo_fast_atan2_fma((float)m01, (float)m10), bf = orbx::fast_atan2_deg((float)m01, (float)m10, true); n++;
        if (memcmp(&af, &bf, 4)) { if (bad < 15) printf("atan2 (fma form) %d %d: %a vs %a\n", m01, m10, af, bf); bad++; }
    }
    for (int m01 = -40; m01 <= 40; m01++) for (int m10 = -40; m10 <= 40; m10++) {
        const float a = orbo_fast_atan2((float)m01, (float)m10), b = orbx::fast_atan2_deg((float)m01, (float)m10, false); n++;
        if (memcmp(&a, &b, 4)) { if (bad < 20) printf("atan2 %d %d: %a vs %a\n", m01, m10, a, b); bad++; }
        const float af = orbo_fast_atan2_fma((float)m01, (float)m10), bf = orbx::fast_atan2_deg((float)m01, (float)m10, true); n++;
        if (memcmp(&af, &bf, 4)) { if (bad < 20) printf("atan2 (fma form) %d %d: %a vs %a\n", m01, m10, af, bf); bad++; }
    }
    printf("checked %ld bad %ld\n", n, bad);
    return bad != 0;
}
