/* TEST INFRASTRUCTURE ONLY.  oracle/_ref/libfrustum_ref.so: Frame::isInFrustum, MapPoint::PredictScale(dist, Frame*) and
 * Pinhole::project(Vector3f) compiled from the reference's own text (excerpted by oracle/Makefile into a temporary file, never in the
 * repo) inside the class shells of mock_frame/frustum_mock.h, with -ffp-contract=off so that every float operation rounds as the text
 * writes it.  Pins orbo_is_in_frustum (orb_oracle_geom.cc): decisions, thresholds, operation order. */
#include <cstdint>
#include "frustum_mock.h"

using namespace std;

namespace ORB_SLAM3 {
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;
#include "ref_frustum_excerpt.inc"
}  // namespace ORB_SLAM3

using namespace ORB_SLAM3;

extern "C" void frustumref_is_in_frustum(const float *Rcw, const float *tcw, const float *Ow, float fx, float fy, float cx, float cy, float mbf,
                                         const float *bounds, float log_scale_factor, int nlevels, float viewing_cos_limit, int n,
                                         const float *pos, const float *normal, const float *min_dist, const float *max_dist,
                                         uint8_t *in_view, float *proj_x, float *proj_y, float *proj_xr, float *depth, int32_t *level,
                                         float *view_cos, uint8_t *ret) {
    Pinhole cam;
    cam.mvParameters = {fx, fy, cx, cy};
    Frame F;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) F.mRcw(r, c) = Rcw[3 * r + c];
        F.mtcw(r) = tcw[r];
        F.mOw(r) = Ow[r];
    }
    F.mpCamera = &cam;
    F.mbf = mbf;
    F.mnScaleLevels = nlevels;
    F.mfLogScaleFactor = log_scale_factor;
    Frame::mnMinX = bounds[0]; Frame::mnMaxX = bounds[1]; Frame::mnMinY = bounds[2]; Frame::mnMaxY = bounds[3];
    for (int i = 0; i < n; i++) {
        MapPoint p;
        p.mWorldPos = Eigen::Vector3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        p.mNormalVector = Eigen::Vector3f(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
        p.mfMinDistance = min_dist[i]; p.mfMaxDistance = max_dist[i];
        ret[i] = F.isInFrustum(&p, viewing_cos_limit) ? 1 : 0;
        in_view[i] = p.mbTrackInView ? 1 : 0;
        proj_x[i] = p.mTrackProjX; proj_y[i] = p.mTrackProjY; proj_xr[i] = p.mTrackProjXR;
        depth[i] = p.mTrackDepth; level[i] = p.mnTrackScaleLevel; view_cos[i] = p.mTrackViewCos;
    }
}

/* Frame::isInFrustumChecks (Frame.cc:1168-1240) for one camera of a fisheye rig (b_right) over KannalaBrandt8::project (ref_kb8_shim.cc): Rcw / tcw / Ow
 * the frame's pose, Rrl / trl = mTrl, tlr = mTlr.translation(), Rwc = mRwc.  Also returns what lines 1172-1186 computed (mR, mt, twc) re-derived here with the
 * same expressions, so that the callers of the oracle / the C ABI can pass exactly those. */
extern "C" void frustumref_is_in_frustum_checks(const float *Rcw, const float *tcw, const float *Ow, const float *Rwc, const float *Rrl, const float *trl,
                                                const float *tlr, const float *params_l, const float *params_r, int b_right, const float *bounds,
                                                float log_scale_factor, int nlevels, float viewing_cos_limit, int n, const float *pos, const float *normal,
                                                const float *min_dist, const float *max_dist, uint8_t *in_view, float *proj_x, float *proj_y, float *depth,
                                                int32_t *level, float *view_cos, float *view_out /* R[9], t[3], twc[3] */) {
    KannalaBrandt8 camL, camR;
    camL.mvParameters.assign(params_l, params_l + 8);
    camR.mvParameters.assign(params_r, params_r + 8);
    Frame F;
    Eigen::Matrix3f mRrl;
    Eigen::Vector3f vtrl, vtlr;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) { F.mRcw(r, c) = Rcw[3 * r + c]; F.mRwc(r, c) = Rwc[3 * r + c]; mRrl(r, c) = Rrl[3 * r + c]; }
        F.mtcw(r) = tcw[r]; F.mOw(r) = Ow[r]; vtrl(r) = trl[r]; vtlr(r) = tlr[r];
    }
    F.mTrl = Sophus::SE3f(mRrl, vtrl);
    F.mTlr = Sophus::SE3f(mRrl.transpose(), vtlr);
    F.mpCamera = &camL; F.mpCamera2 = &camR;
    F.Nleft = 1;
    F.mnScaleLevels = nlevels;
    F.mfLogScaleFactor = log_scale_factor;
    Frame::mnMinX = bounds[0]; Frame::mnMaxX = bounds[1]; Frame::mnMinY = bounds[2]; Frame::mnMaxY = bounds[3];
    {   /* the text of lines 1172-1186, for the caller */
        Eigen::Matrix3f mR; Eigen::Vector3f mt, twc;
        if (b_right) { Eigen::Matrix3f Rrl_ = F.mTrl.rotationMatrix(); Eigen::Vector3f trl_ = F.mTrl.translation(); mR = Rrl_ * F.mRcw; mt = Rrl_ * F.mtcw + trl_; twc = F.mRwc * F.mTlr.translation() + F.mOw; }
        else { mR = F.mRcw; mt = F.mtcw; twc = F.mOw; }
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) view_out[3 * r + c] = mR(r, c); view_out[9 + r] = mt(r); view_out[12 + r] = twc(r); }
    }
    for (int i = 0; i < n; i++) {
        MapPoint p;
        p.mWorldPos = Eigen::Vector3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        p.mNormalVector = Eigen::Vector3f(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
        p.mfMinDistance = min_dist[i]; p.mfMaxDistance = max_dist[i];
        p.mnTrackScaleLevel = -1; p.mnTrackScaleLevelR = -1;   /* Frame.cc:579-580 */
        const bool ok = F.isInFrustumChecks(&p, viewing_cos_limit, b_right != 0);
        in_view[i] = ok ? 1 : 0;
        if (b_right) { proj_x[i] = p.mTrackProjXR; proj_y[i] = p.mTrackProjYR; depth[i] = p.mTrackDepthR; level[i] = p.mnTrackScaleLevelR; view_cos[i] = p.mTrackViewCosR; }
        else { proj_x[i] = p.mTrackProjX; proj_y[i] = p.mTrackProjY; depth[i] = p.mTrackDepth; level[i] = p.mnTrackScaleLevel; view_cos[i] = p.mTrackViewCos; }
    }
}
