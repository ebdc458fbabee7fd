// check_geometry_math.cc -- the device's glibc_atanf / glibc_atan2f (geometry_kernels.hip.h, compiled for the host through the SIMT shim) against the host
// libm's atanf / atan2f (glibc 2.35: the libm the reference links), and kb8_project against the oracle's restatement of KannalaBrandt8::project.
// argv[1] = stride of the atanf sweep over all float bit patterns (1 = exhaustive, 25 s; the test uses 7).
#include "hip/hip_runtime.h"
#include "geometry_kernels.hip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
extern "C" void orbo_kb8_project(const float *p, float X, float Y, float Z, float *u, float *v);
extern "C" void orbo_kb8_unproject(const float *p, float px, float py, float *ray3);
extern "C" float orbo_kb8_triangulate_matches(const float *cam1, const float *cam2, float x1, float y1, float x2, float y2, const float *R12, const float *t12,
                                              float sigmaLevel, float unc);
static uint32_t fw(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static float wf(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
int main(int argc, char **argv) {
    const uint64_t stride = argc > 1 ? (uint64_t)atoll(argv[1]) : 7;
    long bad = 0, n = 0;
    for (uint64_t u = 0; u < 0x100000000ull; u += stride) {
        const float x = wf((uint32_t)u);
        if (x != x) continue;
        const float a = atanf(x), b = orbx::glibc_atanf(x); n++;
        if (fw(a) != fw(b)) { if (bad < 10) printf("atanf %a: %a vs %a\n", x, a, b); bad++; }
    }
    std::mt19937 rng(7);
    for (long i = 0; i < 40000000; i++) {
        float y, x;
        const uint32_t a = rng(), b = rng();
        if (i & 1) { y = wf(a); x = wf(b); }
        else { y = ((int)(a % 2000001) - 1000000) * 1e-3f; x = ((int)(b % 2000001) - 1000000) * 1e-3f * ((i & 2) ? 1.f : 1e-3f); }
        if (y != y || x != x) continue;
        const float r = atan2f(y, x), s = orbx::glibc_atan2f(y, x); n++;
        if (fw(r) != fw(s)) { if (bad < 20) printf("atan2f %a %a: %a vs %a\n", y, x, r, s); bad++; }
    }
    const float sp[12][2] = {{0.f, 1.f}, {-0.f, 1.f}, {0.f, -1.f}, {-0.f, -1.f}, {1.f, 0.f}, {-1.f, 0.f}, {1.f, -0.f}, {INFINITY, 1.f}, {1.f, INFINITY}, {1.f, -INFINITY},
                             {INFINITY, -INFINITY}, {-INFINITY, INFINITY}};
    for (auto &q : sp) { const float r = atan2f(q[0], q[1]), s = orbx::glibc_atan2f(q[0], q[1]); n++; if (fw(r) != fw(s)) { printf("atan2f special %a %a: %a vs %a\n", q[0], q[1], r, s); bad++; } }
    // the projection: host cos / sin here on both sides, so bit-identical
    const float prm[8] = {190.978477f, 190.973307f, 254.931706f, 256.897442f, 0.0034823894f, 0.0007150348f, -0.0020532361f, 0.0002029367f};
    for (long i = 0; i < 4000000; i++) {
        const float X = ((int)(rng() % 20001) - 10000) * 1e-3f, Y = ((int)(rng() % 20001) - 10000) * 1e-3f, Z = ((int)(rng() % 14001) - 2000) * 1e-3f;
        float u0, v0, u1, v1;
        orbo_kb8_project(prm, X, Y, Z, &u0, &v0);
        orbx::kb8_project(prm, X, Y, Z, &u1, &v1); n++;
        if (fw(u0) != fw(u1) || fw(v0) != fw(v1)) { if (bad < 30) printf("kb8 %g %g %g: %a %a vs %a %a\n", X, Y, Z, u0, v0, u1, v1); bad++; }
    }
    // tanf on |x| < 3 pi / 4 (the exhaustive run, stride 1: 2 150 471 624 arguments, 0 differences)
    for (uint64_t u = 0; u < 0x4016cbe4ull; u += stride) {
        for (int sg = 0; sg < 2; sg++) {
            const float x = wf((uint32_t)u | (sg ? 0x80000000u : 0u));
            const float a = tanf(x), b = orbx::glibc_tanf(x); n++;
            if (fw(a) != fw(b)) { if (bad < 40) printf("tanf %a: %a vs %a\n", x, a, b); bad++; }
        }
    }
    // unproject and the whole epipolarConstrain chain (unproject x 2, JacobiSVD, project x 2, thresholds) against the oracle's restatement, on pairs made by
    // projecting a 3-D point into both cameras (+ pixel noise, so that every rejection branch is taken)
    const float prm2[8] = {190.442369f, 190.434438f, 252.598164f, 254.917230f, 0.0034003171f, 0.0017669271f, -0.0026631290f, 0.0003299517f};
    long n_ok = 0, n_rej[6] = {0, 0, 0, 0, 0, 0};
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    for (long i = 0; i < 300000; i++) {
        const float a = 0.05f * U(rng), b = 0.05f * U(rng), c = 0.05f * U(rng);
        const float R12[9] = {1.f, -c, b, c, 1.f, -a, -b, a, 1.f};   // (not exactly orthonormal: the reference's code does not require it)
        const float t12[3] = {0.1f + 0.05f * U(rng), 0.02f * U(rng), 0.02f * U(rng)};
        const float X1[3] = {2.f * U(rng), 2.f * U(rng), 0.8f + 3.f * (U(rng) + 1.f)};
        // X1 = R12 * X2 + t12  ->  X2 = R12^T (X1 - t12)
        const float d[3] = {X1[0] - t12[0], X1[1] - t12[1], X1[2] - t12[2]};
        const float X2[3] = {R12[0] * d[0] + R12[3] * d[1] + R12[6] * d[2], R12[1] * d[0] + R12[4] * d[1] + R12[7] * d[2], R12[2] * d[0] + R12[5] * d[1] + R12[8] * d[2]};
        float u1, v1, u2, v2;
        orbo_kb8_project(prm, X1[0], X1[1], X1[2], &u1, &v1);
        orbo_kb8_project(prm2, X2[0], X2[1], X2[2], &u2, &v2);
        const float noise = (i % 4 == 0) ? 6.f : (i % 4 == 1) ? 1.5f : 0.3f;
        u1 += noise * U(rng); v1 += noise * U(rng); u2 += noise * U(rng); v2 += noise * U(rng);
        if (i % 7 == 0) { u2 = u1; v2 = v1; }   // nearly parallel rays
        float ray_o[3], rx, ry;
        orbo_kb8_unproject(prm, u1, v1, ray_o);
        orbx::kb8_unproject(prm, u1, v1, 1e-6f, &rx, &ry); n++;
        if (fw(ray_o[0]) != fw(rx) || fw(ray_o[1]) != fw(ry)) { if (bad < 60) printf("unproject %g %g: %a %a vs %a %a\n", u1, v1, ray_o[0], ray_o[1], rx, ry); bad++; }
        const float s1 = 1.f + (i % 5) * 0.44f, s2 = 1.f + (i % 3) * 0.73f;
        const float z = orbo_kb8_triangulate_matches(prm, prm2, u1, v1, u2, v2, R12, t12, s1, s2);
        const bool want = z > 0.0001f, got = orbx::kb8_epipolar_constrain(prm, prm2, u1, v1, u2, v2, R12, t12, s1, s2); n++;
        if (want) n_ok++; else n_rej[z >= 0 ? 0 : (int)-z]++;
        if (want != got) { if (bad < 80) printf("epipolarConstrain %ld: oracle %g device %d\n", i, z, (int)got); bad++; }
    }
    printf("epipolarConstrain: accepted %ld, rejected by rule -1..-5: %ld %ld %ld %ld %ld\n", n_ok, n_rej[1], n_rej[2], n_rej[3], n_rej[4], n_rej[5]);
    if (n_ok < 20000 || n_rej[1] < 1000 || n_rej[4] < 1000 || n_rej[5] < 1000) { printf("the pairs do not exercise the gate\n"); bad++; }
    printf("checked %ld bad %ld\n", n, bad);
    return bad != 0;
}
