/* orb_oracle_geom.cc -- CPU oracle (TEST INFRASTRUCTURE ONLY) for the candidate-generation pre-passes of the projection matchers
 * (SURVEY.md 8f-3): Frame::isInFrustum and Frame::UndistortKeyPoints / ComputeImageBounds.
 *
 * Pinning: orbo_is_in_frustum follows the reference's own text line by line and is compared bit for bit with that text compiled
 * where it lies over the stand-in float Eigen types (oracle/_ref/libframe_ref.so, tests/test_oracle_frame_vs_reference.py) -- that
 * pins every decision and the operation order AS WRITTEN; a build of the reference against the real Eigen may evaluate the 3x3
 * product with packet FMAs and differ in the last ulp.  orbo_undistort_points restates cv::undistortPoints
 * from the published algorithm [OCV-recalled]: PARITY UNPINNED (OpenCV is absent from the reference tree and from this image). */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>

#include "orb_oracle.h"

extern "C" {

/* Frame::isInFrustum, Nleft == -1 branch (Frame.cc:512-575) with MapPoint::PredictScale(dist, Frame*) (MapPoint.cc:531-546) and
 * Pinhole::project (CameraModels/Pinhole.cpp:43-49).  Per map point: in_view (mbTrackInView), proj_x / proj_y (mTrackProjX/Y: -1
 * unless the point passed the image-bounds test, Frame.cc:515-516, :541-542), and -- only meaningful where in_view -- proj_xr, depth
 * (mTrackDepth), level (mnTrackScaleLevel), view_cos (mTrackViewCos).  min_dist / max_dist = the map point's mfMinDistance / mfMaxDistance. */
void orbo_is_in_frustum(const float *Rcw, const float *tcw, const float *Ow, float fx, float fy, float cx, float cy, float mbf,
                        const float *bounds, float log_scale_factor, int nlevels, float viewing_cos_limit, int n, const float *pos,
                        const float *normal, const float *min_dist, const float *max_dist, uint8_t *in_view, float *proj_x, float *proj_y,
                        float *proj_xr, float *depth, int32_t *level, float *view_cos) {
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    for (int i = 0; i < n; i++) {
        in_view[i] = 0; proj_x[i] = -1; proj_y[i] = -1;
        proj_xr[i] = 0; depth[i] = 0; level[i] = 0; view_cos[i] = 0;
        const float P[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
        float Pc[3];
        for (int r = 0; r < 3; r++) Pc[r] = (Rcw[3 * r] * P[0] + Rcw[3 * r + 1] * P[1] + Rcw[3 * r + 2] * P[2]) + tcw[r];   /* mRcw * P + mtcw */
        const float Pc_dist = std::sqrt((0.f + Pc[0] * Pc[0]) + Pc[1] * Pc[1] + Pc[2] * Pc[2]);
        const float PcZ = Pc[2];
        const float invz = 1.0f / PcZ;
        if (PcZ < 0.0f) continue;
        const float u = fx * Pc[0] / Pc[2] + cx, v = fy * Pc[1] / Pc[2] + cy;   /* Pinhole::project */
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        proj_x[i] = u; proj_y[i] = v;
        const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
        const float dist = std::sqrt((0.f + PO[0] * PO[0]) + PO[1] * PO[1] + PO[2] * PO[2]);
        const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];   /* Get{Max,Min}DistanceInvariance, MapPoint.cc:502-512 */
        if (dist < minDistance || dist > maxDistance) continue;
        const float viewCos = ((0.f + PO[0] * normal[3 * i]) + PO[1] * normal[3 * i + 1] + PO[2] * normal[3 * i + 2]) / dist;
        if (viewCos < viewing_cos_limit) continue;
        /* MapPoint::PredictScale: ratio = mfMaxDistance / currentDist; nScale = ceil(log(ratio) / mfLogScaleFactor), clamped */
        const float ratio = max_dist[i] / dist;
        int nScale = (int)std::ceil(std::log(ratio) / log_scale_factor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= nlevels) nScale = nlevels - 1;
        in_view[i] = 1;
        proj_xr[i] = u - mbf * invz;
        depth[i] = Pc_dist;
        level[i] = nScale;
        view_cos[i] = viewCos;
    }
}

/* KannalaBrandt8::project(const Eigen::Vector3f &) (CameraModels/KannalaBrandt8.cpp:67-85).  The reference's translation unit has no `using namespace std`:
 * its cos(psi) / sin(psi) are the C library's double functions, the sums are double and round to float once (pinned by compiling that text in a
 * translation unit of the same kind: oracle/ref_kb8_shim.cc). */
void orbo_kb8_project(const float *p, float X, float Y, float Z, float *u, float *v) {
    const float x2_plus_y2 = X * X + Y * Y;
    const float theta = atan2f(sqrtf(x2_plus_y2), Z);
    const float psi = atan2f(Y, X);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + p[4] * theta3 + p[5] * theta5 + p[6] * theta7 + p[7] * theta9;
    *u = (float)(p[0] * r * ::cos((double)psi) + p[2]);
    *v = (float)(p[1] * r * ::sin((double)psi) + p[3]);
}

/* KannalaBrandt8::unproject (KannalaBrandt8.cpp:107-142); precision = 1e-6 (KannalaBrandt8.h).  std::tan(float) = libm tanf. */
void orbo_kb8_unproject(const float *p, float px, float py, float *ray3) {
    const float precision = 1e-6f;
    const float pwx = (px - p[2]) / p[0], pwy = (py - p[3]) / p[1];
    float scale = 1.f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    theta_d = fminf(fmaxf(-3.1415926535897932384626433832795 / 2.f, theta_d), 3.1415926535897932384626433832795 / 2.f);   /* CV_PI is a double constant */
    if (theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            float k0_theta2 = p[4] * theta2, k1_theta4 = p[5] * theta4;
            float k2_theta6 = p[6] * theta6, k3_theta8 = p[7] * theta8;
            float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                              (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < precision) break;
        }
        scale = std::tan(theta) / theta_d;
    }
    ray3[0] = pwx * scale; ray3[1] = pwy * scale; ray3[2] = 1.f;
}

/* Eigen::JacobiSVD<Matrix4f>(A, ComputeFullV).matrixV() [EIGEN-recalled: Eigen is not vendored by the reference and absent from this image -- PARITY UNPINNED;
 * restated from Eigen 3.3/3.4's JacobiSVD.h (two-sided Jacobi, real_2x2_jacobi_svd, JacobiRotation::makeJacobi, sorted singular values)].  A, V row-major. */
namespace {
struct Rot { float c, s; };
inline void rot_rows(float (*M)[4], int p, int q, Rot j) {            /* MatrixBase::applyOnTheLeft(p, q, j) */
    if (j.c == 1.f && j.s == 0.f) return;
    for (int i = 0; i < 4; i++) { const float x = M[p][i], y = M[q][i]; M[p][i] = j.c * x + j.s * y; M[q][i] = -j.s * x + j.c * y; }
}
inline void rot_cols(float (*M)[4], int p, int q, Rot j) {            /* MatrixBase::applyOnTheRight(p, q, j): the columns rotate with j.transpose() */
    const Rot t = {j.c, -j.s};
    if (t.c == 1.f && t.s == 0.f) return;
    for (int i = 0; i < 4; i++) { const float x = M[i][p], y = M[i][q]; M[i][p] = t.c * x + t.s * y; M[i][q] = -t.s * x + t.c * y; }
}
}  // namespace
void orbo_eigen_jacobi_svd4_V(const float *A16, float *V16, float *sv4) {
    const float tiny = std::numeric_limits<float>::min(), precision = 2.f * std::numeric_limits<float>::epsilon();
    float W[4][4], V[4][4];
    float scale = 0.f;
    for (int i = 0; i < 16; i++) scale = std::max(scale, std::fabs(A16[i]));
    if (scale == 0.f) scale = 1.f;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { W[i][j] = A16[4 * i + j] / scale; V[i][j] = i == j ? 1.f : 0.f; }
    float maxDiagEntry = 0.f;
    for (int i = 0; i < 4; i++) maxDiagEntry = std::max(maxDiagEntry, std::fabs(W[i][i]));
    bool finished = false;
    for (int guard = 0; !finished && guard < 64; guard++) {
        finished = true;
        for (int p = 1; p < 4; ++p)
            for (int q = 0; q < p; ++q) {
                const float threshold = std::max(tiny, precision * maxDiagEntry);
                if (std::fabs(W[p][q]) > threshold || std::fabs(W[q][p]) > threshold) {
                    finished = false;
                    /* real_2x2_jacobi_svd(W, p, q, &j_left, &j_right) */
                    float m[2][2] = {{W[p][p], W[p][q]}, {W[q][p], W[q][q]}};
                    Rot rot1;
                    const float t = m[0][0] + m[1][1], d = m[1][0] - m[0][1];
                    if (std::fabs(d) < tiny) { rot1.s = 0.f; rot1.c = 1.f; }
                    else { const float u = t / d, tmp = std::sqrt(1.f + u * u); rot1.s = 1.f / tmp; rot1.c = u / tmp; }
                    {   /* m.applyOnTheLeft(0, 1, rot1) */
                        for (int i = 0; i < 2; i++) { const float x = m[0][i], y = m[1][i]; m[0][i] = rot1.c * x + rot1.s * y; m[1][i] = -rot1.s * x + rot1.c * y; }
                    }
                    Rot jr;   /* j_right.makeJacobi(m, 0, 1): x = m(0,0), y = m(0,1), z = m(1,1) */
                    {
                        const float x = m[0][0], y = m[0][1], z = m[1][1], deno = 2.f * std::fabs(y);
                        if (deno < tiny) { jr.c = 1.f; jr.s = 0.f; }
                        else {
                            const float tau = (x - z) / deno, w = std::sqrt(tau * tau + 1.f);
                            const float tt = tau > 0.f ? 1.f / (tau + w) : 1.f / (tau - w);
                            const float sign_t = tt > 0.f ? 1.f : -1.f, n = 1.f / std::sqrt(tt * tt + 1.f);
                            jr.s = -sign_t * (y / std::fabs(y)) * std::fabs(tt) * n;
                            jr.c = n;
                        }
                    }
                    const Rot jrt = {jr.c, -jr.s};
                    const Rot jl = {rot1.c * jrt.c - rot1.s * jrt.s, rot1.c * jrt.s + rot1.s * jrt.c};   /* rot1 * j_right.transpose() */
                    rot_rows(W, p, q, jl);
                    rot_cols(W, p, q, jr);
                    rot_cols(V, p, q, jr);
                    maxDiagEntry = std::max(maxDiagEntry, std::max(std::fabs(W[p][p]), std::fabs(W[q][q])));
                }
            }
    }
    float sv[4];
    for (int i = 0; i < 4; i++) sv[i] = std::fabs(W[i][i]) * scale;
    for (int i = 0; i < 4; i++) {
        int pos = 0;
        float best = sv[i];
        for (int j = 1; j < 4 - i; j++) if (sv[i + j] > best) { best = sv[i + j]; pos = j; }
        if (best == 0.f) break;
        if (pos) { pos += i; std::swap(sv[i], sv[pos]); for (int r = 0; r < 4; r++) std::swap(V[r][i], V[r][pos]); }
    }
    for (int i = 0; i < 4; i++) { for (int j = 0; j < 4; j++) V16[4 * i + j] = V[i][j]; if (sv4) sv4[i] = sv[i]; }
}

/* KannalaBrandt8::TriangulateMatches (KannalaBrandt8.cpp:305-368) with Triangulate (:387-400): the value the reference returns (z1, or -1 .. -5) */
float orbo_kb8_triangulate_matches(const float *cam1, const float *cam2, float x1, float y1, float x2, float y2, const float *R12, const float *t12,
                                   float sigmaLevel, float unc) {
    float r1[3], r2[3], r21[3];
    orbo_kb8_unproject(cam1, x1, y1, r1);
    orbo_kb8_unproject(cam2, x2, y2, r2);
    for (int i = 0; i < 3; i++) r21[i] = (0.f + R12[3 * i] * r2[0]) + R12[3 * i + 1] * r2[1] + R12[3 * i + 2] * r2[2];
    const float dot = (0.f + r1[0] * r21[0]) + r1[1] * r21[1] + r1[2] * r21[2];
    const float nr1 = std::sqrt((0.f + r1[0] * r1[0]) + r1[1] * r1[1] + r1[2] * r1[2]), nr21 = std::sqrt((0.f + r21[0] * r21[0]) + r21[1] * r21[1] + r21[2] * r21[2]);
    const float cosParallaxRays = dot / (nr1 * nr21);
    if (cosParallaxRays > 0.9998) return -1;
    float Tcw1[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}}, Tcw2[3][4], R21[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R21[i][j] = R12[3 * j + i];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Tcw2[i][j] = R21[i][j];
        Tcw2[i][3] = (0.f + -R21[i][0] * t12[0]) + -R21[i][1] * t12[1] + -R21[i][2] * t12[2];   /* -R21 * t12 */
    }
    float A[16];
    for (int j = 0; j < 4; j++) {
        A[j] = r1[0] * Tcw1[2][j] - Tcw1[0][j];
        A[4 + j] = r1[1] * Tcw1[2][j] - Tcw1[1][j];
        A[8 + j] = r2[0] * Tcw2[2][j] - Tcw2[0][j];
        A[12 + j] = r2[1] * Tcw2[2][j] - Tcw2[1][j];
    }
    float V[16];
    orbo_eigen_jacobi_svd4_V(A, V, nullptr);
    const float x3D[3] = {V[3] / V[15], V[7] / V[15], V[11] / V[15]};   /* matrixV().col(3).head(3) / x3Dh(3) */
    const float z1 = x3D[2];
    if (z1 <= 0) return -2;
    const float z2 = ((0.f + R21[2][0] * x3D[0]) + R21[2][1] * x3D[1] + R21[2][2] * x3D[2]) + Tcw2[2][3];
    if (z2 <= 0) return -3;
    float u, v;
    orbo_kb8_project(cam1, x3D[0], x3D[1], x3D[2], &u, &v);
    const float errX1 = u - x1, errY1 = v - y1;
    if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaLevel) return -4;
    float x3D2[3];
    for (int i = 0; i < 3; i++) x3D2[i] = ((0.f + R21[i][0] * x3D[0]) + R21[i][1] * x3D[1] + R21[i][2] * x3D[2]) + Tcw2[i][3];
    orbo_kb8_project(cam2, x3D2[0], x3D2[1], x3D2[2], &u, &v);
    const float errX2 = u - x2, errY2 = v - y2;
    if ((errX2 * errX2 + errY2 * errY2) > 5.991 * unc) return -5;
    return z1;
}
/* KannalaBrandt8::epipolarConstrain (:216-221) for n keypoint pairs */
void orbo_kb8_epipolar_constrain(const float *cam1, const float *cam2, int n, const float *xy1, const float *xy2, const float *R12, const float *t12,
                                 const float *sigma1, const float *sigma2, uint8_t *ok, float *tm_value) {
    for (int i = 0; i < n; i++) {
        const float z = orbo_kb8_triangulate_matches(cam1, cam2, xy1[2 * i], xy1[2 * i + 1], xy2[2 * i], xy2[2 * i + 1], R12, t12, sigma1[i], sigma2[i]);
        ok[i] = z > 0.0001f ? 1 : 0;
        if (tm_value) tm_value[i] = z;
    }
}

/* Frame::isInFrustumChecks(pMP, viewingCosLimit, bRight) (Frame.cc:1168-1240) for n map points and one camera of the rig: R, t, twc are what lines
 * 1172-1186 compute (the caller does; the tests take them from the reference text itself).  Outputs as orbx_is_in_frustum_checks documents. */
void orbo_is_in_frustum_checks(const float *R, const float *t, const float *twc, const float *params8, const float *bounds, float log_scale_factor,
                               int nlevels, float viewing_cos_limit, int n, const float *pos, const float *normal, const float *min_dist,
                               const float *max_dist, uint8_t *in_view, float *proj_x, float *proj_y, float *depth, int32_t *level, float *view_cos) {
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    for (int i = 0; i < n; i++) {
        in_view[i] = 0; proj_x[i] = 0; proj_y[i] = 0; depth[i] = 0; level[i] = -1; view_cos[i] = 0;
        const float P[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
        float Pc[3];
        for (int r = 0; r < 3; r++) Pc[r] = (R[3 * r] * P[0] + R[3 * r + 1] * P[1] + R[3 * r + 2] * P[2]) + t[r];   /* mR * P + mt */
        const float Pc_dist = std::sqrt((0.f + Pc[0] * Pc[0]) + Pc[1] * Pc[1] + Pc[2] * Pc[2]);
        if (Pc[2] < 0.0f) continue;
        float u, v;
        orbo_kb8_project(params8, Pc[0], Pc[1], Pc[2], &u, &v);
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
        const float PO[3] = {P[0] - twc[0], P[1] - twc[1], P[2] - twc[2]};
        const float dist = std::sqrt((0.f + PO[0] * PO[0]) + PO[1] * PO[1] + PO[2] * PO[2]);
        if (dist < minDistance || dist > maxDistance) continue;
        const float viewCos = ((0.f + PO[0] * normal[3 * i]) + PO[1] * normal[3 * i + 1] + PO[2] * normal[3 * i + 2]) / dist;
        if (viewCos < viewing_cos_limit) continue;
        const float ratio = max_dist[i] / dist;   /* MapPoint::PredictScale */
        int nScale = (int)std::ceil(std::log(ratio) / log_scale_factor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= nlevels) nScale = nlevels - 1;
        in_view[i] = 1; proj_x[i] = u; proj_y[i] = v; depth[i] = Pc_dist; level[i] = nScale; view_cos[i] = viewCos;
    }
}

/* [OCV-recalled] cv::undistortPoints(src, dst, K, distCoeffs, R = I, P = K) as Frame::UndistortKeyPoints (Frame.cc:747-780) and
 * ComputeImageBounds (:782-810) call it: double arithmetic, default criteria (COUNT = 5 fixed-point iterations), radial-tangential
 * model k1, k2, p1, p2, k3 (mDistCoef with 4 or 5 entries: k3 = 0 for 4).  xy: n interleaved (x, y) float pairs, in and out. */
void orbo_undistort_points(int n, const float *xy_in, float fx, float fy, float cx, float cy, float k1, float k2, float p1, float p2,
                           float k3, float *xy_out) {
    const double dfx = fx, dfy = fy, dcx = cx, dcy = cy;
    const double ifx = 1. / dfx, ify = 1. / dfy;
    const double k[5] = {k1, k2, p1, p2, k3};
    for (int i = 0; i < n; i++) {
        double x = xy_in[2 * i], y = xy_in[2 * i + 1];
        const double u = x, v = y;
        x = (x - dcx) * ifx;
        y = (y - dcy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);   /* k4..k6 = 0 */
            if (icdist < 0) { x = (u - dcx) * ifx; y = (v - dcy) * ify; break; }
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        /* P = K, R = I: xx = fx*x + cx, yy = fy*y + cy, ww = 1 */
        const double xx = dfx * x + 0. * y + dcx, yy = 0. * x + dfy * y + dcy, ww = 1. / (0. * x + 0. * y + 1.);
        xy_out[2 * i] = (float)(xx * ww);
        xy_out[2 * i + 1] = (float)(yy * ww);
    }
}

/* Frame::ComputeImageBounds (Frame.cc:782-810): bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY} */
void orbo_image_bounds(int width, int height, float fx, float fy, float cx, float cy, float k1, float k2, float p1, float p2, float k3,
                       float *bounds) {
    if (k1 == 0.0f) { bounds[0] = 0.0f; bounds[1] = (float)width; bounds[2] = 0.0f; bounds[3] = (float)height; return; }
    const float c[8] = {0.f, 0.f, (float)width, 0.f, 0.f, (float)height, (float)width, (float)height};
    float o[8];
    orbo_undistort_points(4, c, fx, fy, cx, cy, k1, k2, p1, p2, k3, o);
    bounds[0] = std::fmin(o[0], o[4]);
    bounds[1] = std::fmax(o[2], o[6]);
    bounds[2] = std::fmin(o[1], o[3]);
    bounds[3] = std::fmax(o[5], o[7]);
}

}  // extern "C"
