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
