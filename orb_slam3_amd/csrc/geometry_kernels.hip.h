// geometry_kernels.hip.h -- candidate-generation pre-passes of the projection matchers, gfx950.
//   k_in_frustum   Frame::isInFrustum (Frame.cc:512-575, Nleft == -1) + MapPoint::PredictScale (MapPoint.cc:531-546) + Pinhole::project
//                  (CameraModels/Pinhole.cpp:43-49) for every (frame, map point): fills the arrays SearchByProjection reads
//   k_in_frustum_checks  Frame::isInFrustumChecks (Frame.cc:1168-1240: the fisheye rig's form, left and right camera) + KannalaBrandt8::project
//                  (CameraModels/KannalaBrandt8.cpp:67-85), with glibc's atan2f restated for the device
//   k_undistort    cv::undistortPoints as Frame::UndistortKeyPoints (Frame.cc:747-780) calls it [OCV-recalled]: 5 fixed-point
//                  iterations of the radial-tangential model in double, P = K
// Built with -ffp-contract=off: every operation rounds as the reference text writes it (a reference built against the real Eigen may
// evaluate mRcw * P with packet FMAs and differ in the last ulp; tolerance stated in DESIGN.md section 5).
#pragma once

#include <cmath>
#include <cstring>

#include "orbx_internal.h"

namespace orbx {

struct FrustumFrame {   // what Frame::isInFrustum reads of the frame
    float Rcw[9], tcw[3], Ow[3];
    float fx, fy, cx, cy, mbf;
    float minx, maxx, miny, maxy;
    float log_scale_factor;
    int nlevels;
    float cos_limit;
};

// grid (ceil(n_mp / 256), n_frames), block 256.  Map-point data are shared by all frames; outputs are [n_frames][n_mp].
static __global__ __launch_bounds__(256) void k_in_frustum(const FrustumFrame *__restrict__ frames, int n_mp, const float *__restrict__ pos,
                                                    const float *__restrict__ normal, const float *__restrict__ min_dist,
                                                    const float *__restrict__ max_dist, uint8_t *__restrict__ in_view,
                                                    float *__restrict__ proj_x, float *__restrict__ proj_y, float *__restrict__ proj_xr,
                                                    float *__restrict__ depth, int32_t *__restrict__ level, float *__restrict__ view_cos) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_mp) return;
    const FrustumFrame F = frames[blockIdx.y];
    const size_t o = (size_t)blockIdx.y * n_mp + i;
    uint8_t iv = 0;
    float px = -1.f, py = -1.f, pxr = 0.f, dep = 0.f, vc = 0.f;   // :515-516: mTrackProjX/Y = -1 until the bounds test passes
    int lvl = 0;
    const float P0 = pos[3 * i], P1 = pos[3 * i + 1], P2 = pos[3 * i + 2];
    // every input of the map point requested at once (the tests below read them conditionally: each would be a dependent round trip -- over the host
    // link when the call runs in the matcher context's direct mode)
    const float mn_in = min_dist[i], mx = max_dist[i], N0 = normal[3 * i], N1 = normal[3 * i + 1], N2 = normal[3 * i + 2];
    float Pc[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
        Pc[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(F.Rcw[3 * r], P0), __fmul_rn(F.Rcw[3 * r + 1], P1)), __fmul_rn(F.Rcw[3 * r + 2], P2)), F.tcw[r]);
    const float Pc_dist = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(Pc[0], Pc[0])), __fmul_rn(Pc[1], Pc[1])), __fmul_rn(Pc[2], Pc[2])));
    const float invz = __fdiv_rn(1.0f, Pc[2]);
    if (!(Pc[2] < 0.0f)) {
        const float u = __fadd_rn(__fdiv_rn(__fmul_rn(F.fx, Pc[0]), Pc[2]), F.cx), v = __fadd_rn(__fdiv_rn(__fmul_rn(F.fy, Pc[1]), Pc[2]), F.cy);
        if (!(u < F.minx || u > F.maxx) && !(v < F.miny || v > F.maxy)) {
            px = u; py = v;
            const float PO0 = __fsub_rn(P0, F.Ow[0]), PO1 = __fsub_rn(P1, F.Ow[1]), PO2 = __fsub_rn(P2, F.Ow[2]);
            const float dist = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(PO0, PO0)), __fmul_rn(PO1, PO1)), __fmul_rn(PO2, PO2)));
            const float maxDistance = __fmul_rn(1.2f, mx), minDistance = __fmul_rn(0.8f, mn_in);
            if (!(dist < minDistance || dist > maxDistance)) {
                const float viewCos = __fdiv_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(PO0, N0)), __fmul_rn(PO1, N1)), __fmul_rn(PO2, N2)), dist);
                if (!(viewCos < F.cos_limit)) {
                    const float ratio = __fdiv_rn(mx, dist);
                    // logf through the double logarithm: correctly rounded to float but for double-rounding cases (libm's logf is
                    // within 1 ulp as well); the level changes only if log(ratio)/logScale sits within an ulp of an integer
                    int nScale = (int)ceilf(__fdiv_rn((float)log((double)ratio), F.log_scale_factor));
                    if (nScale < 0) nScale = 0;
                    else if (nScale >= F.nlevels) nScale = F.nlevels - 1;
                    iv = 1;
                    pxr = __fsub_rn(u, __fmul_rn(F.mbf, invz));
                    dep = Pc_dist; lvl = nScale; vc = viewCos;
                }
            }
        }
    }
    in_view[o] = iv; proj_x[o] = px; proj_y[o] = py; proj_xr[o] = pxr; depth[o] = dep; level[o] = lvl; view_cos[o] = vc;
}

// ---------------------------------------------------------------------------------------------------------
// glibc 2.35 atanf / atan2f (sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c: the fdlibm forms, plain float operations -- no FMA variant exists for them),
// restated operation for operation.  tests/simt/check_geometry_math.cc compares them on the CPU with the host libm: atanf on EVERY float, atan2f on
// 4 * 10^8 pairs -- bit-identical.
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ inline float glibc_atanf(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    uint32_t ux;
    memcpy(&ux, &x, 4);
    const int32_t hx = (int32_t)ux, ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {   // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {    // |x| < 0.4375
        if (ix < 0x31000000) return x;   // |x| < 2^-29 (raises inexact in the original)
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {           // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float hi = id == 0 ? atanhi[0] : id == 1 ? atanhi[1] : id == 2 ? atanhi[2] : atanhi[3];   // (selects: an indexed local array would live in scratch memory)
    const float lo = id == 0 ? atanlo[0] : id == 1 ? atanlo[1] : id == 2 ? atanlo[2] : atanlo[3];
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}
__host__ __device__ inline float glibc_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    uint32_t ux, uy;
    memcpy(&ux, &x, 4);
    memcpy(&uy, &y, 4);
    const int32_t hx = (int32_t)ux, ix = hx & 0x7fffffff, hy = (int32_t)uy, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return glibc_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : m == 2 ? pi + tiny : -pi - tiny;
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = glibc_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return -z;
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// One camera of a fisheye rig as Frame::isInFrustumChecks sees it (Frame.cc:1172-1186): the caller evaluates the reference's Eigen expressions
// (bRight: mR = Rrl * mRcw, mt = Rrl * mtcw + trl, twc = mRwc * mTlr.translation() + mOw; left: mRcw, mtcw, mOw)
struct FisheyeView {
    float R[9], t[3], twc[3];
    float p[8];   // KannalaBrandt8::mvParameters: fx, fy, cx, cy, k0 .. k3
};
struct FrustumChecks {
    FisheyeView view[2];
    float minx, maxx, miny, maxy;
    float log_scale_factor;
    int nlevels;
    float cos_limit;
};

// KannalaBrandt8::project(const Eigen::Vector3f &) (KannalaBrandt8.cpp:67-85).  The translation unit has no `using namespace std`, so its
// `cos(psi)` / `sin(psi)` are ::cos(double) / ::sin(double): the products fx * r * cos(psi) + cx are evaluated in double and rounded to float once.
// The device's double cos / sin are within 1-2 ulp (double) of glibc's: the float results agree except where the double value lies that close to a
// float rounding boundary (the same situation as logf in k_in_frustum; the tests allow one float ulp and count the differences: none seen).
__host__ __device__ inline void kb8_project(const float *p, float X, float Y, float Z, float *u, float *v) {
    const float x2_plus_y2 = X * X + Y * Y;
    const float theta = glibc_atan2f(sqrtf(x2_plus_y2), Z);
    const float psi = glibc_atan2f(Y, X);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + p[4] * theta3 + p[5] * theta5 + p[6] * theta7 + p[7] * theta9;
    *u = (float)((double)(p[0] * r) * cos((double)psi) + (double)p[2]);
    *v = (float)((double)(p[1] * r) * sin((double)psi) + (double)p[3]);
}

// glibc 2.35 tanf (sysdeps/ieee754/flt-32/s_tanf.c, k_tanf.c, e_rem_pio2f.c: fdlibm) for |x| < 3 pi / 4 -- the range KannalaBrandt8::unproject's
// theta lives in (theta_d is clamped to pi / 2); beyond it the double tangent.  Against the host libm on EVERY float of that range (2 150 471 624 arguments):
// bit-identical -- with the argument reduction in double (the fdlibm float reduction, 24 + 24 (+ 24) bits of pi / 2, differs from glibc on 1034 of them).
// tests/simt/check_geometry_math.cc repeats a strided sweep.
__host__ __device__ inline float glibc_kernel_tanf(float x, float y, int iy) {
    const float T[13] = {3.3333334327e-01f, 1.3333334029e-01f, 5.3968254477e-02f, 2.1869488060e-02f, 8.8632395491e-03f, 3.5920790397e-03f, 1.4562094584e-03f,
                         5.8804126456e-04f, 2.4646313977e-04f, 7.8179444245e-05f, 7.1407252108e-05f, -1.8558637748e-05f, 2.5907305826e-05f};
    const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    uint32_t ux;
    memcpy(&ux, &x, 4);
    const int32_t hx = (int32_t)ux, ix = hx & 0x7fffffff;
    float z, r, v, w, s;
    if (ix < 0x39000000 && (int)x == 0) {   // |x| < 2^-13
        if ((ix | (iy + 1)) == 0) return 1.0f / fabsf(x);
        return iy == 1 ? x : -1.0f / x;
    }
    if (ix >= 0x3f2ca140) {   // |x| >= 0.6744
        if (hx < 0) { x = -x; y = -y; }
        z = pio4 - x; w = pio4lo - y; x = z + w; y = 0.0f;
        if (fabsf(x) < 0x1p-13f) return (1 - ((hx >> 30) & 2)) * iy * (1.0f - 2 * iy * x);
    }
    z = x * x; w = z * z;
    r = T[1] + w * (T[3] + w * (T[5] + w * (T[7] + w * (T[9] + w * T[11]))));
    v = z * (T[2] + w * (T[4] + w * (T[6] + w * (T[8] + w * (T[10] + w * T[12])))));
    s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T[0] * s;
    w = x + r;
    if (ix >= 0x3f2ca140) { v = (float)iy; return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r))); }
    if (iy == 1) return w;
    uint32_t u;
    memcpy(&u, &w, 4); u &= 0xfffff000u; memcpy(&z, &u, 4);
    v = r - (z - x);
    float a, t;
    t = a = -1.0f / w;
    memcpy(&u, &t, 4); u &= 0xfffff000u; memcpy(&t, &u, 4);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}
__host__ __device__ inline float glibc_tanf(float x) {
    uint32_t ux;
    memcpy(&ux, &x, 4);
    const int32_t hx = (int32_t)ux, ix = hx & 0x7fffffff;
    if (ix <= 0x3f490fda) return glibc_kernel_tanf(x, 0.0f, 1);
    if (ix >= 0x4016cbe4) return (float)tan((double)x);   // outside the camera model's range
    // __ieee754_rem_pio2f for pi/4 < |x| < 3 pi/4 (n = +-1): the remainder in double, handed to the kernel as head + tail
    const double r = hx > 0 ? (double)x - 1.5707963267948966 : (double)x + 1.5707963267948966;
    const float y0 = (float)r, y1 = (float)(r - (double)y0);
    return glibc_kernel_tanf(y0, y1, -1);   // odd quadrant
}

// KannalaBrandt8::unproject (KannalaBrandt8.cpp:107-142): Newton iterations on the distortion polynomial; precision = 1e-6 (KannalaBrandt8.h:42-59)
__host__ __device__ inline void kb8_unproject(const float *p, float px, float py, float precision, float *rx, float *ry) {
    const float pwx = (px - p[2]) / p[0], pwy = (py - p[3]) / p[1];
    float scale = 1.f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    theta_d = fminf(fmaxf((float)(-3.1415926535897932384626433832795 / 2.f), theta_d), (float)(3.1415926535897932384626433832795 / 2.f));
    if ((double)theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const float k0_theta2 = p[4] * theta2, k1_theta4 = p[5] * theta4, k2_theta6 = p[6] * theta6, k3_theta8 = p[7] * theta8;
            const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                    (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < precision) break;
        }
        scale = glibc_tanf(theta) / theta_d;
    }
    *rx = pwx * scale; *ry = pwy * scale;   // ray (rx, ry, 1)
}

// Eigen::JacobiSVD<Matrix4f>(A, ComputeFullV).matrixV().col(3) -- the two-sided Jacobi iteration of Eigen 3.3 / 3.4 (Eigen/src/SVD/JacobiSVD.h,
// Eigen/src/Jacobi/Jacobi.h) restated from the published source: scaling by the largest coefficient, sweeps over (p, q), real_2x2_jacobi_svd, rotations
// applied in Eigen's operation order, singular values sorted with column swaps.  PARITY UNPINNED: Eigen is not vendored by the reference and absent from
// this image; what the gate does with the result are threshold tests, which a last-bit difference moves only at a boundary.
__host__ __device__ inline void eigen_jacobi_svd4_v3(const float (*A)[4], float *out) {
    const float fmin_ = 1.17549435e-38f, eps2 = 2.f * 1.1920929e-07f;
    float W[4][4], V[4][4];
    float scale = 0.f;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) scale = fmaxf(scale, fabsf(A[i][j]));
    if (scale == 0.f) scale = 1.f;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { W[i][j] = A[i][j] / scale; V[i][j] = i == j ? 1.f : 0.f; }
    float maxDiag = 0.f;
    for (int i = 0; i < 4; i++) maxDiag = fmaxf(maxDiag, fabsf(W[i][i]));
    bool finished = false;
    for (int sweep = 0; !finished && sweep < 64; sweep++) {   // (Eigen loops until a sweep changes nothing; 64 bounds a NaN input)
        finished = true;
#pragma unroll
        for (int p = 1; p < 4; p++)
#pragma unroll
            for (int q = 0; q < p; q++) {
                const float threshold = fmaxf(fmin_, eps2 * maxDiag);
                if (!(fabsf(W[p][q]) > threshold || fabsf(W[q][p]) > threshold)) continue;
                finished = false;
                // real_2x2_jacobi_svd on (W(p,p) W(p,q); W(q,p) W(q,q))
                const float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
                const float t = m00 + m11, d = m10 - m01;
                float c1, s1;
                if (fabsf(d) < fmin_) { s1 = 0.f; c1 = 1.f; }
                else { const float u = t / d, tmp = sqrtf(1.f + u * u); s1 = 1.f / tmp; c1 = u / tmp; }
                const float n00 = c1 * m00 + s1 * m10, n01 = c1 * m01 + s1 * m11, n11 = -s1 * m01 + c1 * m11;
                float cr, sr;   // j_right.makeJacobi(n00, n01, n11)
                const float deno = 2.f * fabsf(n01);
                if (deno < fmin_) { cr = 1.f; sr = 0.f; }
                else {
                    const float tau = (n00 - n11) / deno, w = sqrtf(tau * tau + 1.f);
                    const float tt = tau > 0.f ? 1.f / (tau + w) : 1.f / (tau - w);
                    const float sign_t = tt > 0.f ? 1.f : -1.f, n = 1.f / sqrtf(tt * tt + 1.f);
                    sr = -sign_t * (n01 / fabsf(n01)) * fabsf(tt) * n;
                    cr = n;
                }
                const float cl = c1 * cr - s1 * (-sr), sl = c1 * (-sr) + s1 * cr;   // j_left = rot1 * j_right.transpose()
                if (!(cl == 1.f && sl == 0.f))
#pragma unroll
                    for (int i = 0; i < 4; i++) { const float xp = W[p][i], xq = W[q][i]; W[p][i] = cl * xp + sl * xq; W[q][i] = -sl * xp + cl * xq; }
                if (!(cr == 1.f && sr == 0.f))   // applyOnTheRight(p, q, j) rotates the columns with j.transpose() = (cr, -sr)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float xp = W[i][p], xq = W[i][q]; W[i][p] = cr * xp + (-sr) * xq; W[i][q] = sr * xp + cr * xq;
                        const float vp = V[i][p], vq = V[i][q]; V[i][p] = cr * vp + (-sr) * vq; V[i][q] = sr * vp + cr * vq;
                    }
                maxDiag = fmaxf(maxDiag, fmaxf(fabsf(W[p][p]), fabsf(W[q][q])));
            }
    }
    float sv[4];
    int col[4] = {0, 1, 2, 3};
#pragma unroll
    for (int i = 0; i < 4; i++) sv[i] = fabsf(W[i][i]);
    bool stop = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {   // descending, first maximum, as maxCoeff(&pos); static indices only (an indexed local array would live in scratch memory)
        int pos = i;
        float best = sv[i];
#pragma unroll
        for (int j = i + 1; j < 4; j++) if (sv[j] > best) { best = sv[j]; pos = j; }
        stop = stop || best == 0.f;
#pragma unroll
        for (int j = i + 1; j < 4; j++)
            if (!stop && pos == j) { const float ts = sv[i]; sv[i] = sv[j]; sv[j] = ts; const int tc = col[i]; col[i] = col[j]; col[j] = tc; }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float v = V[i][0];
        v = col[3] == 1 ? V[i][1] : v; v = col[3] == 2 ? V[i][2] : v; v = col[3] == 3 ? V[i][3] : v;
        out[i] = v;
    }
}

// KannalaBrandt8::epipolarConstrain = TriangulateMatches(...) > 0.0001f (KannalaBrandt8.cpp:216-221, 305-368, Triangulate :387-400) for keypoint kp1 of camera
// cam1 and kp2 of cam2, R12 / t12 the relative pose (row-major), sigmaLevel / unc the two level variances
__host__ __device__ inline bool kb8_epipolar_constrain(const float *cam1, const float *cam2, float x1, float y1, float x2, float y2, const float *R12,
                                                       const float *t12, float sigmaLevel, float unc) {
    float r1[3], r2[3];
    kb8_unproject(cam1, x1, y1, 1e-6f, &r1[0], &r1[1]); r1[2] = 1.f;
    kb8_unproject(cam2, x2, y2, 1e-6f, &r2[0], &r2[1]); r2[2] = 1.f;
    float r21[3];
    for (int i = 0; i < 3; i++) r21[i] = ((0.f + R12[3 * i] * r2[0]) + R12[3 * i + 1] * r2[1]) + R12[3 * i + 2] * r2[2];
    const float dot = ((0.f + r1[0] * r21[0]) + r1[1] * r21[1]) + r1[2] * r21[2];
    const float n1 = sqrtf(((0.f + r1[0] * r1[0]) + r1[1] * r1[1]) + r1[2] * r1[2]), n21 = sqrtf(((0.f + r21[0] * r21[0]) + r21[1] * r21[1]) + r21[2] * r21[2]);
    const float cosParallaxRays = dot / (n1 * n21);
    if ((double)cosParallaxRays > 0.9998) return false;
    float R21[3][3], T2[3][4];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R21[i][j] = R12[3 * j + i];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T2[i][j] = R21[i][j];
        T2[i][3] = ((0.f + (-R21[i][0]) * t12[0]) + (-R21[i][1]) * t12[1]) + (-R21[i][2]) * t12[2];   // -R21 * t12
    }
    const float T1[3][4] = {{1.f, 0.f, 0.f, 0.f}, {0.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f}};
    float A[4][4];
    for (int j = 0; j < 4; j++) {
        A[0][j] = r1[0] * T1[2][j] - T1[0][j];
        A[1][j] = r1[1] * T1[2][j] - T1[1][j];
        A[2][j] = r2[0] * T2[2][j] - T2[0][j];
        A[3][j] = r2[1] * T2[2][j] - T2[1][j];
    }
    float xh[4];
    eigen_jacobi_svd4_v3(A, xh);
    const float X[3] = {xh[0] / xh[3], xh[1] / xh[3], xh[2] / xh[3]};
    const float z1 = X[2];
    if (z1 <= 0) return false;
    const float z2 = (((0.f + R21[2][0] * X[0]) + R21[2][1] * X[1]) + R21[2][2] * X[2]) + T2[2][3];
    if (z2 <= 0) return false;
    float u, v;
    kb8_project(cam1, X[0], X[1], X[2], &u, &v);
    const float errX1 = u - x1, errY1 = v - y1;
    if ((double)(errX1 * errX1 + errY1 * errY1) > 5.991 * (double)sigmaLevel) return false;
    float X2[3];
    for (int i = 0; i < 3; i++) X2[i] = (((0.f + R21[i][0] * X[0]) + R21[i][1] * X[1]) + R21[i][2] * X[2]) + T2[i][3];
    kb8_project(cam2, X2[0], X2[1], X2[2], &u, &v);
    const float errX2 = u - x2, errY2 = v - y2;
    if ((double)(errX2 * errX2 + errY2 * errY2) > 5.991 * (double)unc) return false;
    return z1 > 0.0001f;
}

// grid (ceil(n_mp / 256), n_views), block 256.  Outputs [n_views][n_mp]: the MapPoint fields the function writes when every test passes
// (mTrackProjX/Y[R], mnTrackScaleLevel[R], mTrackViewCos[R], mTrackDepth[R]); otherwise in_view = 0, level = -1 (Frame.cc:579-580) and zeros.
static __global__ __launch_bounds__(256) void k_in_frustum_checks(const FrustumChecks *__restrict__ fc, int n_mp, const float *__restrict__ pos,
                                                           const float *__restrict__ normal, const float *__restrict__ min_dist,
                                                           const float *__restrict__ max_dist, uint8_t *__restrict__ in_view,
                                                           float *__restrict__ proj_x, float *__restrict__ proj_y, float *__restrict__ depth,
                                                           int32_t *__restrict__ level, float *__restrict__ view_cos) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_mp) return;
    const FrustumChecks &F = *fc;
    const FisheyeView &V = fc->view[blockIdx.y];
    const size_t o = (size_t)blockIdx.y * n_mp + i;
    const float P0 = pos[3 * i], P1 = pos[3 * i + 1], P2 = pos[3 * i + 2];
    const float mn_in = min_dist[i], mx = max_dist[i], N0 = normal[3 * i], N1 = normal[3 * i + 1], N2 = normal[3 * i + 2];
    uint8_t iv = 0;
    float px = 0.f, py = 0.f, dep = 0.f, vc = 0.f;
    int lvl = -1;
    float Pc[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
        Pc[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(V.R[3 * r], P0), __fmul_rn(V.R[3 * r + 1], P1)), __fmul_rn(V.R[3 * r + 2], P2)), V.t[r]);
    const float Pc_dist = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(Pc[0], Pc[0])), __fmul_rn(Pc[1], Pc[1])), __fmul_rn(Pc[2], Pc[2])));
    if (!(Pc[2] < 0.0f)) {
        float u, v;
        kb8_project(V.p, Pc[0], Pc[1], Pc[2], &u, &v);
        if (!(u < F.minx || u > F.maxx) && !(v < F.miny || v > F.maxy)) {
            const float PO0 = __fsub_rn(P0, V.twc[0]), PO1 = __fsub_rn(P1, V.twc[1]), PO2 = __fsub_rn(P2, V.twc[2]);
            const float dist = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(PO0, PO0)), __fmul_rn(PO1, PO1)), __fmul_rn(PO2, PO2)));
            const float maxDistance = __fmul_rn(1.2f, mx), minDistance = __fmul_rn(0.8f, mn_in);
            if (!(dist < minDistance || dist > maxDistance)) {
                const float viewCos = __fdiv_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(PO0, N0)), __fmul_rn(PO1, N1)), __fmul_rn(PO2, N2)), dist);
                if (!(viewCos < F.cos_limit)) {
                    const float ratio = __fdiv_rn(mx, dist);
                    int nScale = (int)ceilf(__fdiv_rn((float)log((double)ratio), F.log_scale_factor));   // as k_in_frustum
                    if (nScale < 0) nScale = 0;
                    else if (nScale >= F.nlevels) nScale = F.nlevels - 1;
                    iv = 1; px = u; py = v; dep = Pc_dist; lvl = nScale; vc = viewCos;
                }
            }
        }
    }
    in_view[o] = iv; proj_x[o] = px; proj_y[o] = py; depth[o] = dep; level[o] = lvl; view_cos[o] = vc;
}

struct CameraModel {   // Pinhole intrinsics (Frame::mK) + radial-tangential distortion (Frame::mDistCoef)
    float fx, fy, cx, cy, k1, k2, p1, p2, k3;
};

// plain operators: the translation unit is built with -ffp-contract=off, so host and device evaluate exactly the operations written
__host__ __device__ inline void undistort_point(const CameraModel &c, float xin, float yin, float *xo, float *yo) {
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy;
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k[5] = {c.k1, c.k2, c.p1, c.p2, c.k3};
    const double u = xin, v = yin;
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);   // k4..k6 = 0
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);   // P = K, R = I
    *xo = (float)(xx * ww);
    *yo = (float)(yy * ww);
}

// Frame::ComputeImageBounds (Frame.cc:782-810)
inline void image_bounds(const CameraModel &c, int width, int height, float *b) {
    if (c.k1 == 0.0f) { b[0] = 0.0f; b[1] = (float)width; b[2] = 0.0f; b[3] = (float)height; return; }
    const float cx[4] = {0.f, (float)width, 0.f, (float)width}, cy[4] = {0.f, 0.f, (float)height, (float)height};
    float ox[4], oy[4];
    for (int i = 0; i < 4; i++) undistort_point(c, cx[i], cy[i], &ox[i], &oy[i]);
    b[0] = std::fmin(ox[0], ox[2]); b[1] = std::fmax(ox[1], ox[3]);
    b[2] = std::fmin(oy[0], oy[1]); b[3] = std::fmax(oy[2], oy[3]);
}

// Frame::UndistortKeyPoints for B frames of keypoints [B][cap] (count[f] valid each; count == NULL: all `cap` entries of one frame):
// mvKeysUn[i] = mvKeys[i] with the undistorted point.  k1 == 0 copies (:749-753).  grid (ceil(cap / 256), B)
static __global__ __launch_bounds__(256) void k_undistort(CameraModel c, const orbx_keypoint *__restrict__ kps, const int32_t *__restrict__ count, int cap,
                                                   orbx_keypoint *__restrict__ kps_un) {
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    const int n = count ? count[f] : cap;
    if (i >= n) return;
    orbx_keypoint kp = kps[(size_t)f * cap + i];
    if (c.k1 != 0.0f) undistort_point(c, kp.x, kp.y, &kp.x, &kp.y);
    kps_un[(size_t)f * cap + i] = kp;
}

}  // namespace orbx
