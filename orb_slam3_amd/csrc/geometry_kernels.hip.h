// geometry_kernels.hip.h -- candidate-generation pre-passes of the projection matchers, gfx950.
//   k_in_frustum   Frame::isInFrustum (Frame.cc:512-575, Nleft == -1) + MapPoint::PredictScale (MapPoint.cc:531-546) + Pinhole::project
//                  (CameraModels/Pinhole.cpp:43-49) for every (frame, map point): fills the arrays SearchByProjection reads
//   k_undistort    cv::undistortPoints as Frame::UndistortKeyPoints (Frame.cc:747-780) calls it [OCV-recalled]: 5 fixed-point
//                  iterations of the radial-tangential model in double, P = K
// Built with -ffp-contract=off: every operation rounds as the reference text writes it (a reference built against the real Eigen may
// evaluate mRcw * P with packet FMAs and differ in the last ulp; tolerance stated in DESIGN.md section 5).
#pragma once

#include <cmath>

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
