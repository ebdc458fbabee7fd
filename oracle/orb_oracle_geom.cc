/* orb_oracle_geom.cc -- CPU oracle (TEST INFRASTRUCTURE ONLY) for the candidate-generation pre-passes of the projection matchers
 * (SURVEY.md 8f-3): Frame::isInFrustum and Frame::UndistortKeyPoints / ComputeImageBounds.
 *
 * Pinning: orbo_is_in_frustum follows the reference's own text line by line and is compared bit for bit with that text compiled
 * where it lies over the stand-in float Eigen types (oracle/_ref/libframe_ref.so, tests/test_oracle_frame_vs_reference.py) -- that
 * pins every decision and the operation order AS WRITTEN; a build of the reference against the real Eigen may evaluate the 3x3
 * product with packet FMAs and differ in the last ulp.  orbo_undistort_points restates cv::undistortPoints
 * from the published algorithm [OCV-recalled]: PARITY UNPINNED (OpenCV is absent from the reference tree and from this image). */
#include <cmath>
#include <cstdint>

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
