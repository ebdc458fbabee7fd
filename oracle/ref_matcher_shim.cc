/* TEST INFRASTRUCTURE ONLY.  C entry points over the reference's own ORBmatcher (src/ORBmatcher.cc compiled where it lies
 * against oracle/mock_slam + oracle/ocv_shim; see oracle/Makefile, _ref/libmatcher_ref.so).  Each function takes the same
 * flattened arrays the oracle's orbo_* matcher takes, rebuilds the stand-in Frame / KeyFrame / MapPoint objects from them,
 * runs the reference's member function and flattens its result back, so tests/test_oracle_vs_reference.py can compare the
 * two on identical inputs.  Geometry is driven exactly: poses are identity (plus a z translation where the function reads
 * the camera motion), the camera's project() is (x, y), so a map point at world (u, v, z) lands on pixel (u, v). */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
#include "ORBmatcher.h"

using namespace ORB_SLAM3;

/* class Pinhole of mock_slam/slam_mock.h: member bodies from the reference's CameraModels/Pinhole.cpp, handed over by oracle/Makefile through a
 * temporary include (ref_excerpt.awk; no reference text in the repo).  The adapter build gets project() only. */
namespace ORB_SLAM3 {
#include "ref_matcher_excerpt.inc"
#ifdef MATREF_ADAPTER_BUILD
bool KannalaBrandt8::epipolarConstrain(GeometricCamera *, const cv::KeyPoint &, const cv::KeyPoint &, const Eigen::Matrix3f &, const Eigen::Vector3f &, const float,
                                       const float) {
    fprintf(stderr, "libmatcher_adapter: KannalaBrandt8::epipolarConstrain called on the host -- the adapter must route KannalaBrandt8 rigs to the device gate\n");
    abort();
}
#else
bool KannalaBrandt8::epipolarConstrain(GeometricCamera *pCamera2, const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const Eigen::Matrix3f &R12,
                                       const Eigen::Vector3f &t12, const float sigmaLevel, const float unc) {
    KannalaBrandt8 *o = dynamic_cast<KannalaBrandt8 *>(pCamera2);
    if (!o) { fprintf(stderr, "KannalaBrandt8 shell: the other camera is not a KannalaBrandt8\n"); abort(); }
    float R[9], t[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[3 * i + j] = R12(i, j); t[i] = t12(i); }
    return orbo_kb8_triangulate_matches(mvParameters.data(), o->mvParameters.data(), kp1.pt.x, kp1.pt.y, kp2.pt.x, kp2.pt.y, R, t, sigmaLevel, unc) > 0.0001f;
}
#endif
#ifdef MATREF_ADAPTER_BUILD
bool Pinhole::epipolarConstrain(GeometricCamera *, const cv::KeyPoint &, const cv::KeyPoint &, const Eigen::Matrix3f &, const Eigen::Vector3f &, const float,
                                const float) {
    fprintf(stderr, "libmatcher_adapter: Pinhole::epipolarConstrain called on the host -- the adapter must route pinhole key frames to the device gates\n");
    abort();
}
#endif
}  // namespace ORB_SLAM3

namespace {

struct Bounds { float minx, maxx, miny, maxy; };

void fill(FeatureHolder &h, const orbo_keypoint *kps, int n, const uint8_t *desc, const Bounds &b, const float *scale,
          const float *sigma2, const float *inv_sigma2, int nlevels, const float *u_right, GeometricCamera *cam) {
    h.N = n;
    h.kps_un.assign(kps, kps + n);
    h.mvKeysUn.resize(n);
    for (int i = 0; i < n; i++) {
        cv::KeyPoint k(kps[i].x, kps[i].y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave, i);
        h.mvKeysUn[i] = k;
    }
    h.mvKeys = h.mvKeysUn;
    h.mDescriptors = cv::Mat(n > 0 ? n : 1, 32, CV_8UC1);
    if (n) std::memcpy(h.mDescriptors.data, desc, (size_t)n * 32);
    h.mvuRight.assign(n, -1.f);
    if (u_right) h.mvuRight.assign(u_right, u_right + n);
    h.mvDepth.assign(n, -1.f);
    h.mvScaleFactors.assign(scale, scale + nlevels);
    h.mvLevelSigma2.assign(nlevels, 1.f);
    h.mvInvLevelSigma2.assign(nlevels, 1.f);
    if (sigma2) h.mvLevelSigma2.assign(sigma2, sigma2 + nlevels);
    if (inv_sigma2) h.mvInvLevelSigma2.assign(inv_sigma2, inv_sigma2 + nlevels);
    h.mnMinX = b.minx; h.mnMaxX = b.maxx; h.mnMinY = b.miny; h.mnMaxY = b.maxy;
    h.mpCamera = cam;
    h.grid = orbo_grid_create(h.kps_un.data(), n, b.minx, b.maxx, b.miny, b.maxy);
}

cv::Mat desc_row(const uint8_t *d) {
    cv::Mat m(1, 32, CV_8UC1);
    std::memcpy(m.data, d, 32);
    return m;
}

void featvec(DBoW2::FeatureVector &fv, const orbo_featvec *f) {
    for (int k = 0; k < f->n_nodes; k++)
        for (int j = f->node_ptr[k]; j < f->node_ptr[k + 1]; j++) fv.addFeature(f->node_id[k], (unsigned)f->index[j]);
}

/* a map point that only marks "this feature slot is taken" */
MapPoint *marker(std::vector<std::unique_ptr<MapPoint>> &pool, int nobs = 1, bool bad = false) {
    pool.emplace_back(new MapPoint());
    pool.back()->nobs = nobs;
    pool.back()->bad = bad;
    pool.back()->id = -2;
    return pool.back().get();
}

const float kOnes[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

}  // namespace

extern "C" {

int matref_descriptor_distance(const uint8_t *a, const uint8_t *b) {
    return ORBmatcher::DescriptorDistance(desc_row(a), desc_row(b));
}

void matref_three_maxima(const int *sizes, int L, int *i1, int *i2, int *i3) {
    std::vector<std::vector<int>> h(L);
    for (int i = 0; i < L; i++) h[i].assign(sizes[i], 0);
    ORBmatcher m(0.6f, true);
    int a = -1, b = -1, c = -1;
    m.ComputeThreeMaxima(h.data(), L, a, b, c);
    *i1 = a; *i2 = b; *i3 = c;
}

/* M1  ORBmatcher.cc:43-213, mono form */
int matref_search_by_projection_mappoints(const orbo_keypoint *kps, const uint8_t *desc, int n, const float *bounds,
                                          const float *scale, int nlevels, const float *u_right,
                                          const uint8_t *occupied, int n_mp, const float *proj_x, const float *proj_y,
                                          const float *proj_xr, const int32_t *level, const float *view_cos,
                                          const uint8_t *mp_desc, const uint8_t *in_view, const uint8_t *has_obs, float th,
                                          float nnratio, int32_t *frame_match) {
    GeometricCamera cam;
    Frame F;
    fill(F, kps, n, desc, Bounds{bounds[0], bounds[1], bounds[2], bounds[3]}, scale, nullptr, nullptr, nlevels, u_right, &cam);
    std::vector<std::unique_ptr<MapPoint>> pool;
    F.mvpMapPoints.assign(n, nullptr);
    for (int i = 0; i < n; i++)
        if (occupied && occupied[i]) F.mvpMapPoints[i] = marker(pool);
    std::vector<MapPoint> mps(n_mp);
    std::vector<MapPoint *> vp(n_mp);
    for (int j = 0; j < n_mp; j++) {
        MapPoint &p = mps[j];
        p.id = j;
        p.mbTrackInView = in_view[j] != 0;
        p.mTrackProjX = proj_x[j]; p.mTrackProjY = proj_y[j]; p.mTrackProjXR = proj_xr[j];
        p.mnTrackScaleLevel = level[j];
        p.mTrackViewCos = view_cos[j];
        p.nobs = has_obs[j] ? 1 : 0;
        p.desc = desc_row(mp_desc + (size_t)j * 32);
        vp[j] = &p;
    }
    ORBmatcher m(nnratio, true);
    int r = m.SearchByProjection(F, vp, th, false, 50.0f);
    for (int i = 0; i < n; i++) frame_match[i] = (F.mvpMapPoints[i] && F.mvpMapPoints[i]->id >= 0) ? F.mvpMapPoints[i]->id : -1;
    return r;
}

/* M1 in its fisheye-stereo form (F.Nleft != -1): ORBmatcher.cc:43-213 whole, left search + right-camera twin.  desc / occupied /
 * frame_match cover the n_left + n_right features in the reference's order (left first). */
int matref_search_by_projection_mappoints_fisheye(const orbo_keypoint *kps_left, int n_left, const orbo_keypoint *kps_right, int n_right,
                                                  const uint8_t *desc, const float *bounds, const float *scale, int nlevels,
                                                  const int32_t *l2r, const int32_t *r2l, const uint8_t *occupied, int n_mp,
                                                  const uint8_t *in_view, const float *proj_x, const float *proj_y, const int32_t *level,
                                                  const float *view_cos, const uint8_t *in_view_r, const float *proj_xr,
                                                  const float *proj_yr, const int32_t *level_r, const float *view_cos_r,
                                                  const uint8_t *mp_desc, const uint8_t *has_obs, float th, float nnratio,
                                                  int32_t *frame_match) {
    GeometricCamera cam;
    Frame F;
    const int N = n_left + n_right;
    fill(F, kps_left, n_left, desc, Bounds{bounds[0], bounds[1], bounds[2], bounds[3]}, scale, nullptr, nullptr, nlevels, nullptr, &cam);
    F.N = N; F.Nleft = n_left; F.NLeft = n_left;
    F.mDescriptors = cv::Mat(N > 0 ? N : 1, 32, CV_8UC1);
    if (N) std::memcpy(F.mDescriptors.data, desc, (size_t)N * 32);
    F.kps_right.assign(kps_right, kps_right + n_right);
    F.mvKeysRight.resize(n_right);
    for (int i = 0; i < n_right; i++)
        F.mvKeysRight[i] = cv::KeyPoint(kps_right[i].x, kps_right[i].y, kps_right[i].size, kps_right[i].angle, kps_right[i].response, kps_right[i].octave, i);
    F.grid_right = orbo_grid_create(F.kps_right.data(), n_right, bounds[0], bounds[1], bounds[2], bounds[3]);
    F.mvLeftToRightMatch.assign(l2r, l2r + n_left);
    F.mvRightToLeftMatch.assign(r2l, r2l + n_right);
    F.mvuRight.assign(N, -1.f);
    std::vector<std::unique_ptr<MapPoint>> pool;
    F.mvpMapPoints.assign(N, nullptr);
    for (int i = 0; i < N; i++)
        if (occupied && occupied[i]) F.mvpMapPoints[i] = marker(pool);
    std::vector<MapPoint> mps(n_mp);
    std::vector<MapPoint *> vp(n_mp);
    for (int j = 0; j < n_mp; j++) {
        MapPoint &p = mps[j];
        p.id = j;
        p.mbTrackInView = in_view[j] != 0; p.mbTrackInViewR = in_view_r[j] != 0;
        p.mTrackProjX = proj_x[j]; p.mTrackProjY = proj_y[j]; p.mTrackProjXR = proj_xr[j]; p.mTrackProjYR = proj_yr[j];
        p.mnTrackScaleLevel = level[j]; p.mnTrackScaleLevelR = level_r[j];
        p.mTrackViewCos = view_cos[j]; p.mTrackViewCosR = view_cos_r[j];
        p.nobs = has_obs[j] ? 1 : 0;
        p.desc = desc_row(mp_desc + (size_t)j * 32);
        vp[j] = &p;
    }
    ORBmatcher m(nnratio, true);
    int r = m.SearchByProjection(F, vp, th, false, 50.0f);
    for (int i = 0; i < N; i++) frame_match[i] = (F.mvpMapPoints[i] && F.mvpMapPoints[i]->id >= 0) ? F.mvpMapPoints[i]->id : -1;
    return r;
}

/* M2  ORBmatcher.cc:1676-1885.  q_z = camera-frame depth of the projected point (the reference derives the stereo
 * coordinate as u - mbf / z with mbf = 1 here).  mode 0 mono, 1 forward, 2 backward. */
int matref_search_by_projection_frame(const orbo_keypoint *kps, const uint8_t *desc, int n, const float *bounds,
                                      const float *scale, int nlevels, const float *u_right, const uint8_t *occupied,
                                      int n_q, const float *q_u, const float *q_v, const float *q_z,
                                      const int32_t *q_octave, const float *q_angle, const uint8_t *q_desc,
                                      const uint8_t *q_has_obs, float th, int mode, int check_orientation,
                                      int32_t *cur_match) {
    GeometricCamera cam;
    Frame Cur, Last;
    fill(Cur, kps, n, desc, Bounds{bounds[0], bounds[1], bounds[2], bounds[3]}, scale, nullptr, nullptr, nlevels, u_right, &cam);
    Cur.mbf = 1.f;
    Cur.mb = 0.5f;
    const float tz = mode == 1 ? -1.f : mode == 2 ? 1.f : 0.f; /* camera centre at z = -tz in the last frame */
    Cur.Tcw.t = Eigen::Vector3f(0.f, 0.f, tz);
    std::vector<std::unique_ptr<MapPoint>> pool;
    Cur.mvpMapPoints.assign(n, nullptr);
    for (int i = 0; i < n; i++)
        if (occupied && occupied[i]) Cur.mvpMapPoints[i] = marker(pool);
    Last.N = n_q;
    Last.mvKeys.resize(n_q);
    Last.mvKeysUn.resize(n_q);
    Last.mvbOutlier.assign(n_q, false);
    std::vector<MapPoint> mps(n_q);
    Last.mvpMapPoints.resize(n_q);
    for (int j = 0; j < n_q; j++) {
        MapPoint &p = mps[j];
        p.id = j;
        p.pos = Eigen::Vector3f(q_u[j], q_v[j], q_z[j] - tz);
        p.nobs = q_has_obs[j] ? 1 : 0;
        p.desc = desc_row(q_desc + (size_t)j * 32);
        Last.mvpMapPoints[j] = &p;
        Last.mvKeys[j].octave = q_octave[j];
        Last.mvKeys[j].angle = q_angle[j];
        Last.mvKeysUn[j] = Last.mvKeys[j];
    }
    ORBmatcher m(0.9f, check_orientation != 0);
    int r = m.SearchByProjection(Cur, Last, th, mode == 0);
    for (int i = 0; i < n; i++) cur_match[i] = (Cur.mvpMapPoints[i] && Cur.mvpMapPoints[i]->id >= 0) ? Cur.mvpMapPoints[i]->id : -1;
    return r;
}

/* M2 in its fisheye-stereo form: Trl = [[1,0,1],[0,1,0],[0,0,1] | 0] on the stand-in matrix type, so a point at camera depth z
 * projects to (u + z, v) in the right camera -- the test picks z per query.  Features of the current frame: left then right. */
int matref_search_by_projection_frame_fisheye(const orbo_keypoint *kps_left, int n_left, const orbo_keypoint *kps_right, int n_right,
                                              const uint8_t *desc, const float *bounds, const float *scale, int nlevels,
                                              const uint8_t *occupied, int n_q, const float *q_u, const float *q_v, const float *q_z,
                                              const int32_t *q_octave, const float *q_angle, const uint8_t *q_desc,
                                              const uint8_t *q_has_obs, float th, int mode, int check_orientation, int32_t *cur_match) {
    GeometricCamera cam;
    Frame Cur, Last;
    const int N = n_left + n_right;
    fill(Cur, kps_left, n_left, desc, Bounds{bounds[0], bounds[1], bounds[2], bounds[3]}, scale, nullptr, nullptr, nlevels, nullptr, &cam);
    Cur.N = N; Cur.Nleft = n_left; Cur.NLeft = n_left;
    Cur.mDescriptors = cv::Mat(N > 0 ? N : 1, 32, CV_8UC1);
    if (N) std::memcpy(Cur.mDescriptors.data, desc, (size_t)N * 32);
    Cur.kps_right.assign(kps_right, kps_right + n_right);
    Cur.mvKeysRight.resize(n_right);
    for (int i = 0; i < n_right; i++)
        Cur.mvKeysRight[i] = cv::KeyPoint(kps_right[i].x, kps_right[i].y, kps_right[i].size, kps_right[i].angle, kps_right[i].response, kps_right[i].octave, i);
    Cur.grid_right = orbo_grid_create(Cur.kps_right.data(), n_right, bounds[0], bounds[1], bounds[2], bounds[3]);
    Cur.mvuRight.assign(N, -1.f);
    Cur.mbf = 1.f;
    Cur.mb = 0.5f;
    const float tz = mode == 1 ? -1.f : mode == 2 ? 1.f : 0.f;
    Cur.Tcw.t = Eigen::Vector3f(0.f, 0.f, tz);
    Cur.Trl.R(0, 2) = 1.f;
    std::vector<std::unique_ptr<MapPoint>> pool;
    Cur.mvpMapPoints.assign(N, nullptr);
    for (int i = 0; i < N; i++)
        if (occupied && occupied[i]) Cur.mvpMapPoints[i] = marker(pool);
    Last.N = n_q;
    Last.mvKeys.resize(n_q);
    Last.mvKeysUn.resize(n_q);
    Last.mvbOutlier.assign(n_q, false);
    std::vector<MapPoint> mps(n_q);
    Last.mvpMapPoints.resize(n_q);
    for (int j = 0; j < n_q; j++) {
        MapPoint &p = mps[j];
        p.id = j;
        p.pos = Eigen::Vector3f(q_u[j], q_v[j], q_z[j] - tz);
        p.nobs = q_has_obs[j] ? 1 : 0;
        p.desc = desc_row(q_desc + (size_t)j * 32);
        Last.mvpMapPoints[j] = &p;
        Last.mvKeys[j].octave = q_octave[j];
        Last.mvKeys[j].angle = q_angle[j];
        Last.mvKeysUn[j] = Last.mvKeys[j];
    }
    ORBmatcher m(0.9f, check_orientation != 0);
    int r = m.SearchByProjection(Cur, Last, th, mode == 0);
    for (int i = 0; i < N; i++) cur_match[i] = (Cur.mvpMapPoints[i] && Cur.mvpMapPoints[i]->id >= 0) ? Cur.mvpMapPoints[i]->id : -1;
    return r;
}

/* M3  ORBmatcher.cc:1887-2010 (relocalisation): query i = key frame feature i with its map point */
int matref_search_by_projection_keyframe(const orbo_keypoint *kps, const uint8_t *desc, int n, const float *bounds,
                                         const float *scale, int nlevels, const uint8_t *occupied, int n_q,
                                         const float *q_x, const float *q_y, const int32_t *q_level,
                                         const float *q_angle, const uint8_t *q_desc, const uint8_t *q_skip, float th,
                                         int orb_dist, int check_orientation, int32_t *match) {
    GeometricCamera cam;
    Frame Cur;
    KeyFrame KF;
    fill(Cur, kps, n, desc, Bounds{bounds[0], bounds[1], bounds[2], bounds[3]}, scale, nullptr, nullptr, nlevels, nullptr, &cam);
    std::vector<std::unique_ptr<MapPoint>> pool;
    Cur.mvpMapPoints.assign(n, nullptr);
    for (int i = 0; i < n; i++)
        if (occupied && occupied[i]) Cur.mvpMapPoints[i] = marker(pool, 0); /* any non-null pointer blocks the slot */
    KF.N = n_q;
    KF.mvKeysUn.resize(n_q);
    KF.mvpMapPoints.resize(n_q);
    std::vector<MapPoint> mps(n_q);
    std::set<MapPoint *> found;
    for (int j = 0; j < n_q; j++) {
        MapPoint &p = mps[j];
        p.id = j;
        p.pos = Eigen::Vector3f(q_x[j], q_y[j], 1.f);
        p.pred_scale = q_level[j];
        p.desc = desc_row(q_desc + (size_t)j * 32);
        KF.mvKeysUn[j].angle = q_angle[j];
        KF.mvpMapPoints[j] = &p;
        if (q_skip && q_skip[j] == 1) KF.mvpMapPoints[j] = nullptr;
        if (q_skip && q_skip[j] == 2) p.bad = true;
        if (q_skip && q_skip[j] == 3) found.insert(&p);
    }
    ORBmatcher m(0.9f, check_orientation != 0);
    int r = m.SearchByProjection(Cur, &KF, found, th, orb_dist);
    for (int i = 0; i < n; i++) match[i] = (Cur.mvpMapPoints[i] && Cur.mvpMapPoints[i]->id >= 0) ? Cur.mvpMapPoints[i]->id : -1;
    return r;
}

/* M3 on a fisheye-stereo frame (CurrentFrame.Nleft != -1): the reference has no special case -- GetFeaturesInArea's default bRight = false
 * searches the LEFT camera's grid (mvKeys), mvKeysUn == mvKeys (Frame.cc:751), so the right camera's features [n_left, N) are neither candidates
 * nor written; desc / occupied cover all N features */
int matref_search_by_projection_keyframe_fisheye(const orbo_keypoint *kps_left, int n_left, const orbo_keypoint *kps_right, int n_right,
                                                 const uint8_t *desc, const float *bounds, const float *scale, int nlevels,
                                                 const uint8_t *occupied, int n_q, const float *q_x, const float *q_y, const int32_t *q_level,
                                                 const float *q_angle, const uint8_t *q_desc, const uint8_t *q_skip, float th, int orb_dist,
                                                 int check_orientation, int32_t *match) {
    GeometricCamera cam;
    Frame Cur;
    KeyFrame KF;
    const int N = n_left + n_right;
    fill(Cur, kps_left, n_left, desc, Bounds{bounds[0], bounds[1], bounds[2], bounds[3]}, scale, nullptr, nullptr, nlevels, nullptr, &cam);
    Cur.N = N; Cur.Nleft = n_left; Cur.NLeft = n_left;
    Cur.mDescriptors = cv::Mat(N > 0 ? N : 1, 32, CV_8UC1);
    if (N) std::memcpy(Cur.mDescriptors.data, desc, (size_t)N * 32);
    Cur.kps_right.assign(kps_right, kps_right + n_right);
    Cur.mvKeysRight.resize(n_right);
    for (int i = 0; i < n_right; i++)
        Cur.mvKeysRight[i] = cv::KeyPoint(kps_right[i].x, kps_right[i].y, kps_right[i].size, kps_right[i].angle, kps_right[i].response, kps_right[i].octave, i);
    Cur.grid_right = orbo_grid_create(Cur.kps_right.data(), n_right, bounds[0], bounds[1], bounds[2], bounds[3]);
    Cur.mvuRight.assign(N, -1.f);
    std::vector<std::unique_ptr<MapPoint>> pool;
    Cur.mvpMapPoints.assign(N, nullptr);
    for (int i = 0; i < N; i++)
        if (occupied && occupied[i]) Cur.mvpMapPoints[i] = marker(pool, 0);
    KF.N = n_q;
    KF.mvKeysUn.resize(n_q);
    KF.mvpMapPoints.resize(n_q);
    std::vector<MapPoint> mps(n_q);
    std::set<MapPoint *> found;
    for (int j = 0; j < n_q; j++) {
        MapPoint &p = mps[j];
        p.id = j;
        p.pos = Eigen::Vector3f(q_x[j], q_y[j], 1.f);
        p.pred_scale = q_level[j];
        p.desc = desc_row(q_desc + (size_t)j * 32);
        KF.mvKeysUn[j].angle = q_angle[j];
        KF.mvpMapPoints[j] = &p;
        if (q_skip && q_skip[j] == 1) KF.mvpMapPoints[j] = nullptr;
        if (q_skip && q_skip[j] == 2) p.bad = true;
        if (q_skip && q_skip[j] == 3) found.insert(&p);
    }
    ORBmatcher m(0.9f, check_orientation != 0);
    int r = m.SearchByProjection(Cur, &KF, found, th, orb_dist);
    for (int i = 0; i < N; i++) match[i] = (Cur.mvpMapPoints[i] && Cur.mvpMapPoints[i]->id >= 0) ? Cur.mvpMapPoints[i]->id : -1;
    return r;
}

/* M4  ORBmatcher.cc:427-535 (variant 0) and :537-646 (variant 1, with the parallel key frame vectors) */
int matref_search_by_projection_sim3(const orbo_keypoint *kps, const uint8_t *desc, int n, const float *bounds,
                                     const float *scale, int nlevels, const uint8_t *occupied, int n_q, const float *q_x,
                                     const float *q_y, const int32_t *q_level, const uint8_t *q_desc, int th,
                                     float ratio_hamming, int variant, int32_t *match) {
    GeometricCamera cam;
    KeyFrame KF, other;
    fill(KF, kps, n, desc, Bounds{bounds[0], bounds[1], bounds[2], bounds[3]}, scale, nullptr, nullptr, nlevels, nullptr, &cam);
    std::vector<std::unique_ptr<MapPoint>> pool;
    std::vector<MapPoint *> matched(n, nullptr);
    std::vector<KeyFrame *> matchedKF(n, nullptr);
    for (int i = 0; i < n; i++)
        if (occupied && occupied[i]) matched[i] = marker(pool);
    std::vector<MapPoint> mps(n_q);
    std::vector<MapPoint *> vp(n_q);
    std::vector<KeyFrame *> vkf(n_q, &other);
    for (int j = 0; j < n_q; j++) {
        MapPoint &p = mps[j];
        p.id = j;
        p.pos = Eigen::Vector3f(q_x[j], q_y[j], 1.f);
        p.normal = Eigen::Vector3f(0.f, 0.f, 1.0e30f); /* PO . n >> |PO| / 2 : the viewing-angle gate always passes */
        p.pred_scale = q_level[j];
        p.desc = desc_row(q_desc + (size_t)j * 32);
        vp[j] = &p;
    }
    Sophus::Sim3f Scw;
    ORBmatcher m(0.9f, true);
    int r = variant == 0 ? m.SearchByProjection(&KF, Scw, vp, matched, th, ratio_hamming)
                         : m.SearchByProjection(&KF, Scw, vp, vkf, matched, matchedKF, th, ratio_hamming);
    for (int i = 0; i < n; i++) match[i] = (matched[i] && matched[i]->id >= 0) ? matched[i]->id : -1;
    return r;
}

/* M5a  ORBmatcher.cc:223-425 (mono).  kf_valid: 0 = no / bad map point */
int matref_search_by_bow_frame(const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid, int n_kf,
                               const orbo_featvec *kf_fv, const uint8_t *f_desc, const float *f_angle, int n_f,
                               const orbo_featvec *f_fv, float nnratio, int check_orientation, int32_t *f_match) {
    KeyFrame KF;
    Frame F;
    KF.N = n_kf; F.N = n_f;
    KF.mvKeysUn.resize(n_kf); F.mvKeys.resize(n_f);
    KF.mDescriptors = cv::Mat(n_kf > 0 ? n_kf : 1, 32, CV_8UC1);
    F.mDescriptors = cv::Mat(n_f > 0 ? n_f : 1, 32, CV_8UC1);
    std::memcpy(KF.mDescriptors.data, kf_desc, (size_t)n_kf * 32);
    std::memcpy(F.mDescriptors.data, f_desc, (size_t)n_f * 32);
    std::vector<MapPoint> mps(n_kf);
    KF.mvpMapPoints.assign(n_kf, nullptr);
    for (int i = 0; i < n_kf; i++) {
        KF.mvKeysUn[i].angle = kf_angle[i];
        mps[i].id = i;
        if (kf_valid[i]) KF.mvpMapPoints[i] = &mps[i];
        else if (i & 1) { mps[i].bad = true; KF.mvpMapPoints[i] = &mps[i]; } /* both "absent" paths of :255-259 */
    }
    for (int i = 0; i < n_f; i++) F.mvKeys[i].angle = f_angle[i];
    F.mvKeysUn = F.mvKeys;
    featvec(KF.mFeatVec, kf_fv);
    featvec(F.mFeatVec, f_fv);
    std::vector<MapPoint *> out;
    ORBmatcher m(nnratio, check_orientation != 0);
    int r = m.SearchByBoW(&KF, F, out);
    for (int i = 0; i < n_f; i++) f_match[i] = out[i] ? out[i]->id : -1;
    return r;
}

/* M5a with F.Nleft != -1 (:283-392): both objects carry a second camera, the frame's features are left [0, n_f_left) then right */
int matref_search_by_bow_frame_fisheye(const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid, int n_kf,
                                       const orbo_featvec *kf_fv, const uint8_t *f_desc, const float *f_angle, int n_f, int n_f_left,
                                       const orbo_featvec *f_fv, float nnratio, int check_orientation, int32_t *f_match) {
    GeometricCamera cam;
    KeyFrame KF;
    Frame F;
    KF.N = n_kf; F.N = n_f;
    KF.mpCamera = KF.mpCamera2 = &cam; F.mpCamera = F.mpCamera2 = &cam;
    KF.NLeft = KF.Nleft = n_kf;          /* every key frame feature is a left one: kp = pKF->mvKeys[realIdxKF] (:325-327) */
    KF.mvKeys.resize(n_kf); KF.mvKeysUn.resize(n_kf);
    F.Nleft = F.NLeft = n_f_left;
    F.mvKeys.resize(n_f_left); F.mvKeysRight.resize(n_f - n_f_left);
    KF.mDescriptors = cv::Mat(n_kf > 0 ? n_kf : 1, 32, CV_8UC1);
    F.mDescriptors = cv::Mat(n_f > 0 ? n_f : 1, 32, CV_8UC1);
    std::memcpy(KF.mDescriptors.data, kf_desc, (size_t)n_kf * 32);
    std::memcpy(F.mDescriptors.data, f_desc, (size_t)n_f * 32);
    std::vector<MapPoint> mps(n_kf);
    KF.mvpMapPoints.assign(n_kf, nullptr);
    for (int i = 0; i < n_kf; i++) {
        KF.mvKeys[i].angle = kf_angle[i];
        KF.mvKeysUn[i].angle = -1000.f;    /* must not be read in this configuration */
        mps[i].id = i;
        if (kf_valid[i]) KF.mvpMapPoints[i] = &mps[i];
        else if (i & 1) { mps[i].bad = true; KF.mvpMapPoints[i] = &mps[i]; }
    }
    for (int i = 0; i < n_f_left; i++) F.mvKeys[i].angle = f_angle[i];
    for (int i = n_f_left; i < n_f; i++) F.mvKeysRight[i - n_f_left].angle = f_angle[i];
    featvec(KF.mFeatVec, kf_fv);
    featvec(F.mFeatVec, f_fv);
    std::vector<MapPoint *> out;
    ORBmatcher m(nnratio, check_orientation != 0);
    int r = m.SearchByBoW(&KF, F, out);
    for (int i = 0; i < n_f; i++) f_match[i] = out[i] ? out[i]->id : -1;
    return r;
}

/* M5b  ORBmatcher.cc:765-905; nleft >= 0: a fisheye-stereo key frame (NLeft = nleft, mvKeysUn holds the nleft left-camera keypoints only,
 * the features from nleft on are the right camera's: :800-802 / :820-822 skip them) */
static int bow_keyframes(const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1, int nleft1,
                         const orbo_featvec *fv1, const uint8_t *desc2, const float *angle2,
                         const uint8_t *valid2, int n2, int nleft2, const orbo_featvec *fv2, float nnratio,
                         int check_orientation, int32_t *match12) {
    KeyFrame K1, K2;
    const int NL[2] = {nleft1, nleft2};
    KeyFrame *K[2] = {&K1, &K2};
    const uint8_t *D[2] = {desc1, desc2};
    const float *A[2] = {angle1, angle2};
    const uint8_t *V[2] = {valid1, valid2};
    const int N[2] = {n1, n2};
    std::vector<MapPoint> mps[2];
    for (int s = 0; s < 2; s++) {
        KeyFrame &k = *K[s];
        k.N = N[s];
        const int nun = NL[s] >= 0 ? NL[s] : N[s];
        if (NL[s] >= 0) { k.NLeft = NL[s]; k.Nleft = NL[s]; }
        k.mvKeysUn.resize(nun);
        k.mDescriptors = cv::Mat(N[s] > 0 ? N[s] : 1, 32, CV_8UC1);
        std::memcpy(k.mDescriptors.data, D[s], (size_t)N[s] * 32);
        mps[s].resize(N[s]);
        k.mvpMapPoints.assign(N[s], nullptr);
        for (int i = 0; i < N[s]; i++) {
            if (i < nun) k.mvKeysUn[i].angle = A[s][i];
            mps[s][i].id = i;
            if (V[s][i]) k.mvpMapPoints[i] = &mps[s][i];
            else if (i & 1) { mps[s][i].bad = true; k.mvpMapPoints[i] = &mps[s][i]; }
        }
    }
    featvec(K1.mFeatVec, fv1);
    featvec(K2.mFeatVec, fv2);
    std::vector<MapPoint *> out;
    ORBmatcher m(nnratio, check_orientation != 0);
    int r = m.SearchByBoW(&K1, &K2, out);
    for (int i = 0; i < n1; i++) match12[i] = out[i] ? out[i]->id : -1;
    return r;
}
int matref_search_by_bow_keyframes(const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1,
                                   const orbo_featvec *fv1, const uint8_t *desc2, const float *angle2,
                                   const uint8_t *valid2, int n2, const orbo_featvec *fv2, float nnratio,
                                   int check_orientation, int32_t *match12) {
    return bow_keyframes(desc1, angle1, valid1, n1, -1, fv1, desc2, angle2, valid2, n2, -1, fv2, nnratio, check_orientation, match12);
}
int matref_search_by_bow_keyframes_fisheye(const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1, int nleft1,
                                           const orbo_featvec *fv1, const uint8_t *desc2, const float *angle2,
                                           const uint8_t *valid2, int n2, int nleft2, const orbo_featvec *fv2, float nnratio,
                                           int check_orientation, int32_t *match12) {
    return bow_keyframes(desc1, angle1, valid1, n1, nleft1, fv1, desc2, angle2, valid2, n2, nleft2, fv2, nnratio, check_orientation, match12);
}

/* M6  ORBmatcher.cc:648-763 */
int matref_search_for_initialization(const orbo_keypoint *kps1, const uint8_t *desc1, int n1, const orbo_keypoint *kps2,
                                     const uint8_t *desc2, int n2, const float *bounds, float *prev_matched,
                                     int window_size, float nnratio, int check_orientation, int32_t *matches12) {
    GeometricCamera cam;
    Frame F1, F2;
    const Bounds b{bounds[0], bounds[1], bounds[2], bounds[3]};
    fill(F1, kps1, n1, desc1, b, kOnes, nullptr, nullptr, 16, nullptr, &cam);
    fill(F2, kps2, n2, desc2, b, kOnes, nullptr, nullptr, 16, nullptr, &cam);
    std::vector<cv::Point2f> prev(n1);
    for (int i = 0; i < n1; i++) prev[i] = cv::Point2f(prev_matched[2 * i], prev_matched[2 * i + 1]);
    std::vector<int> m12;
    ORBmatcher m(nnratio, check_orientation != 0);
    int r = m.SearchForInitialization(F1, F2, prev, m12, window_size);
    for (int i = 0; i < n1; i++) {
        matches12[i] = m12[i];
        prev_matched[2 * i] = prev[i].x;
        prev_matched[2 * i + 1] = prev[i].y;
    }
    return r;
}

/* M7  ORBmatcher.cc:907-1146, pinhole key frames without a second camera.  pair_ok: n1 x n2 verdicts of
 * GeometricCamera::epipolarConstrain (NULL = all pass), looked up by keypoint identity.  kps1 / kps2 (optional) give the keypoints
 * their position and octave, u_right* the stereo flags, scale2 = pKF2->mvScaleFactors and (ep_x, ep_y) the epipole, so the
 * reference's own epipole-distance gate (:1026-1034) runs on real data; without kps the epipole is parked far outside the image
 * and the gate never fires.  coarse = bCoarse. */
int matref_search_for_triangulation_geo(const orbo_keypoint *kps1, const uint8_t *desc1, const float *angle1, const uint8_t *skip1,
                                        const float *u_right1, int n1, const orbo_featvec *fv1, const orbo_keypoint *kps2,
                                        const uint8_t *desc2, const float *angle2, const uint8_t *skip2, const float *u_right2, int n2,
                                        const orbo_featvec *fv2, const float *scale2, int nlevels, float ep_x, float ep_y,
                                        int check_orientation, const uint8_t *pair_ok, int coarse, int32_t *matches12) {
    GeometricCamera cam;
    cam.epi_ok = pair_ok;
    cam.epi_n2 = n2;
    KeyFrame K1, K2;
    KeyFrame *K[2] = {&K1, &K2};
    const orbo_keypoint *P[2] = {kps1, kps2};
    const uint8_t *D[2] = {desc1, desc2};
    const float *A[2] = {angle1, angle2};
    const uint8_t *S[2] = {skip1, skip2};
    const float *U[2] = {u_right1, u_right2};
    const int N[2] = {n1, n2};
    std::vector<std::unique_ptr<MapPoint>> pool;
    for (int s = 0; s < 2; s++) {
        KeyFrame &k = *K[s];
        k.N = N[s];
        k.mvKeysUn.resize(N[s]);
        k.mDescriptors = cv::Mat(N[s] > 0 ? N[s] : 1, 32, CV_8UC1);
        std::memcpy(k.mDescriptors.data, D[s], (size_t)N[s] * 32);
        k.mvpMapPoints.assign(N[s], nullptr);
        k.mvuRight.assign(N[s], -1.f);
        if (U[s]) k.mvuRight.assign(U[s], U[s] + N[s]);
        k.mvScaleFactors.assign(16, 1.f);
        if (s == 1 && scale2) k.mvScaleFactors.assign(scale2, scale2 + nlevels);
        k.mvLevelSigma2.assign(16, 1.f);
        k.mpCamera = &cam;
        for (int i = 0; i < N[s]; i++) {
            if (P[s]) k.mvKeysUn[i] = cv::KeyPoint(P[s][i].x, P[s][i].y, P[s][i].size, P[s][i].angle, P[s][i].response, P[s][i].octave, i);
            else k.mvKeysUn[i].octave = 0;
            k.mvKeysUn[i].angle = A[s][i];
            k.mvKeysUn[i].class_id = i;
            if (S[s][i]) k.mvpMapPoints[i] = marker(pool);
        }
    }
    /* camera centre of KF1 = -R^T t = (ep_x, ep_y, 0); KF2 at identity and project() = (x, y): the epipole is (ep_x, ep_y) */
    K1.Tcw.t = kps1 ? Eigen::Vector3f(-ep_x, -ep_y, 0.f) : Eigen::Vector3f(-1.0e6f, -1.0e6f, 0.f);
    featvec(K1.mFeatVec, fv1);
    featvec(K2.mFeatVec, fv2);
    std::vector<std::pair<size_t, size_t>> pairs;
    ORBmatcher m(0.6f, check_orientation != 0);
    int r = m.SearchForTriangulation(&K1, &K2, pairs, false, coarse != 0);
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    for (auto &p : pairs) matches12[p.first] = (int)p.second;
    return r;
}

/* M7 between two key frames with REAL pinhole cameras and real poses (LocalMapping::CreateNewMapPoints' shape, LocalMapping.cc:466): K1 / K2 =
 * fx, fy, cx, cy; pose1 / pose2 = Tcw as 9 floats of R (row-major) + 3 of t.  In libmatcher_ref.so every gate is the reference's own text:
 * the epipole via Pinhole::project, the epipole-distance test of ORBmatcher.cc:1026-1034, and Pinhole::epipolarConstrain (which derives
 * F12 from K1, K2, R12, t12 for every pair).  In libmatcher_adapter.so the same call must take the adapter's CAM_PINHOLE route (F12 built
 * once by the adapter with the same expression, both gates inside k_replay_bow).  F12_out (optional, 9 floats): the last 3x3 product the
 * run computed -- in the reference build the F12 of the last pair evaluated, in the adapter build the F12 the adapter built. */
int matref_search_for_triangulation_pinhole_cams(const orbo_keypoint *kps1, const uint8_t *desc1, const uint8_t *skip1, const float *u_right1, int n1,
                                                 const orbo_featvec *fv1, const orbo_keypoint *kps2, const uint8_t *desc2, const uint8_t *skip2,
                                                 const float *u_right2, int n2, const orbo_featvec *fv2, const float *scale2, const float *sigma2_2,
                                                 int nlevels, const float *K1, const float *K2, const float *pose1, const float *pose2,
                                                 int check_orientation, int only_stereo, int coarse, int32_t *matches12, float *F12_out) {
    Pinhole cam1(K1[0], K1[1], K1[2], K1[3]), cam2(K2[0], K2[1], K2[2], K2[3]);
    KeyFrame KF1, KF2;
    KeyFrame *K[2] = {&KF1, &KF2};
    const orbo_keypoint *P[2] = {kps1, kps2};
    const uint8_t *D[2] = {desc1, desc2};
    const uint8_t *S[2] = {skip1, skip2};
    const float *U[2] = {u_right1, u_right2};
    const float *T[2] = {pose1, pose2};
    const int N[2] = {n1, n2};
    GeometricCamera *cams[2] = {&cam1, &cam2};
    std::vector<std::unique_ptr<MapPoint>> pool;
    for (int s = 0; s < 2; s++) {
        KeyFrame &k = *K[s];
        k.N = N[s];
        k.mvKeysUn.resize(N[s]);
        k.mDescriptors = cv::Mat(N[s] > 0 ? N[s] : 1, 32, CV_8UC1);
        std::memcpy(k.mDescriptors.data, D[s], (size_t)N[s] * 32);
        k.mvpMapPoints.assign(N[s], nullptr);
        k.mvuRight.assign(N[s], -1.f);
        if (U[s]) k.mvuRight.assign(U[s], U[s] + N[s]);
        k.mvScaleFactors.assign(scale2, scale2 + nlevels);
        k.mvLevelSigma2.assign(sigma2_2, sigma2_2 + nlevels);
        k.mpCamera = cams[s];
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) k.Tcw.R(i, j) = T[s][3 * i + j]; k.Tcw.t(i) = T[s][9 + i]; }
        for (int i = 0; i < N[s]; i++) {
            k.mvKeysUn[i] = cv::KeyPoint(P[s][i].x, P[s][i].y, P[s][i].size, P[s][i].angle, P[s][i].response, P[s][i].octave, i);
            if (S[s][i]) k.mvpMapPoints[i] = marker(pool);
        }
    }
    featvec(KF1.mFeatVec, fv1);
    featvec(KF2.mFeatVec, fv2);
    std::vector<std::pair<size_t, size_t>> pairs;
    ORBmatcher m(0.6f, check_orientation != 0);
    int r = m.SearchForTriangulation(&KF1, &KF2, pairs, only_stereo != 0, coarse != 0);
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    for (auto &p : pairs) matches12[p.first] = (int)p.second;
    if (F12_out) for (int i = 0; i < 9; i++) F12_out[i] = Eigen::last_product()(i / 3, i % 3);
    return r;
}

int matref_search_for_triangulation(const uint8_t *desc1, const float *angle1, const uint8_t *skip1, int n1,
                                    const orbo_featvec *fv1, const uint8_t *desc2, const float *angle2,
                                    const uint8_t *skip2, int n2, const orbo_featvec *fv2, int check_orientation,
                                    const uint8_t *pair_ok, int coarse, int32_t *matches12) {
    return matref_search_for_triangulation_geo(nullptr, desc1, angle1, skip1, nullptr, n1, fv1, nullptr, desc2, angle2, skip2, nullptr,
                                               n2, fv2, nullptr, 0, 0.f, 0.f, check_orientation, pair_ok, coarse, matches12);
}

/* Fuse  ORBmatcher.cc:1148-1338 (variant 0; q_z = camera depth, ur = u - 1/z) and :1340-1455 (variant 1, Sim3 form, no
 * chi2 gate).  Every key frame slot is empty, so each accepted query records AddObservation(pKF, bestIdx): best_idx[i]. */
int matref_fuse(const orbo_keypoint *kps, const uint8_t *desc, int n, const float *bounds, const float *scale,
                const float *inv_sigma2, int nlevels, const float *u_right, int n_q, const float *q_u, const float *q_v,
                const float *q_z, const int32_t *q_level, const uint8_t *q_desc, float th, int variant, int32_t *best_idx) {
    GeometricCamera cam;
    KeyFrame KF;
    fill(KF, kps, n, desc, Bounds{bounds[0], bounds[1], bounds[2], bounds[3]}, scale, nullptr, inv_sigma2, nlevels, u_right, &cam);
    KF.mbf = 1.f;
    KF.mvpMapPoints.assign(n, nullptr);
    KF.probe = true;
    std::vector<MapPoint> mps(n_q);
    std::vector<MapPoint *> vp(n_q), repl(n_q, nullptr);
    for (int j = 0; j < n_q; j++) {
        MapPoint &p = mps[j];
        p.id = j;
        p.pos = Eigen::Vector3f(q_u[j], q_v[j], q_z[j]);
        p.normal = Eigen::Vector3f(0.f, 0.f, 1.0e30f);
        p.pred_scale = q_level[j];
        p.desc = desc_row(q_desc + (size_t)j * 32);
        vp[j] = &p;
    }
    ORBmatcher m(0.6f, true);
    Sophus::Sim3f Scw;
    int r = variant == 0 ? m.Fuse(&KF, vp, th, false) : m.Fuse(&KF, Scw, vp, th, repl);
    for (int j = 0; j < n_q; j++) best_idx[j] = mps[j].added_obs.empty() ? -1 : mps[j].added_obs[0].second;
    return r;
}

/* Fuse(pKF, vpMapPoints, th, bRight = true)  ORBmatcher.cc:1148-1337 on a fisheye-stereo key frame: the map points are projected with
 * the right camera's pose (Trl = identity here) and searched in the right camera's grid; a fused feature is reported with its
 * global index idx + NLeft (:1296).  mvuRight holds -1 everywhere (Frame.cc:1137), so the monocular chi2 gate applies. */
int matref_fuse_right(const orbo_keypoint *kps_left, int n_left, const orbo_keypoint *kps_right, int n_right, const uint8_t *desc,
                      const float *bounds, const float *scale, const float *inv_sigma2, int nlevels, int n_q, const float *q_u,
                      const float *q_v, const float *q_z, const int32_t *q_level, const uint8_t *q_desc, float th, int32_t *best_idx) {
    GeometricCamera cam, cam2;
    KeyFrame KF;
    const int N = n_left + n_right;
    fill(KF, kps_left, n_left, desc, Bounds{bounds[0], bounds[1], bounds[2], bounds[3]}, scale, nullptr, inv_sigma2, nlevels, nullptr, &cam);
    KF.N = N; KF.Nleft = n_left; KF.NLeft = n_left;
    KF.mpCamera2 = &cam2;
    KF.mDescriptors = cv::Mat(N > 0 ? N : 1, 32, CV_8UC1);
    if (N) std::memcpy(KF.mDescriptors.data, desc, (size_t)N * 32);
    KF.kps_right.assign(kps_right, kps_right + n_right);
    KF.mvKeysRight.resize(n_right);
    for (int i = 0; i < n_right; i++)
        KF.mvKeysRight[i] = cv::KeyPoint(kps_right[i].x, kps_right[i].y, kps_right[i].size, kps_right[i].angle, kps_right[i].response, kps_right[i].octave, i);
    KF.grid_right = orbo_grid_create(KF.kps_right.data(), n_right, bounds[0], bounds[1], bounds[2], bounds[3]);
    KF.mvuRight.assign(N, -1.f);
    KF.mbf = 1.f;
    KF.mvpMapPoints.assign(N, nullptr);
    KF.probe = true;
    std::vector<MapPoint> mps(n_q);
    std::vector<MapPoint *> vp(n_q);
    for (int j = 0; j < n_q; j++) {
        MapPoint &p = mps[j];
        p.id = j;
        p.pos = Eigen::Vector3f(q_u[j], q_v[j], q_z[j]);
        p.normal = Eigen::Vector3f(0.f, 0.f, 1.0e30f);
        p.pred_scale = q_level[j];
        p.desc = desc_row(q_desc + (size_t)j * 32);
        vp[j] = &p;
    }
    ORBmatcher m(0.6f, true);
    int r = m.Fuse(&KF, vp, th, true);
    for (int j = 0; j < n_q; j++) best_idx[j] = mps[j].added_obs.empty() ? -1 : mps[j].added_obs[0].second;
    return r;
}

/* SearchForTriangulation between two fisheye-stereo key frames (both have mpCamera2), ORBmatcher.cc:907-1146 with the four camera
 * pairings of :1036-1069.  Features [0, n_left) are mvKeys, the rest mvKeysRight; angle* / skip* / the feature vectors cover all of
 * them; pair_ok is indexed by the combined feature indices.  The four relative poses get distinct x translations (left KF1 at x = 0,
 * its right camera at +1; left KF2 at 0, its right camera at +4: t12.x = 0 ll, +4 lr, -1 rl, +3 rr) and the stand-in cameras refuse a
 * pair whose camera objects or t12 do not belong to the pairing of the two keypoints. */
int matref_search_for_triangulation_fisheye(const uint8_t *desc1, const float *angle1, const uint8_t *skip1, int n1, int n_left1,
                                            const orbo_featvec *fv1, const uint8_t *desc2, const float *angle2, const uint8_t *skip2, int n2,
                                            int n_left2, const orbo_featvec *fv2, int check_orientation, const uint8_t *pair_ok, int coarse,
                                            int32_t *matches12) {
    static const float t12x[4] = {0.f, 4.f, -1.f, 3.f};   /* [2 * right1 + right2] */
    GeometricCamera cam[2][2];
    KeyFrame K1, K2;
    KeyFrame *K[2] = {&K1, &K2};
    const uint8_t *D[2] = {desc1, desc2};
    const float *A[2] = {angle1, angle2};
    const uint8_t *S[2] = {skip1, skip2};
    const int N[2] = {n1, n2}, NL[2] = {n_left1, n_left2};
    std::vector<std::unique_ptr<MapPoint>> pool;
    for (int s = 0; s < 2; s++) {
        KeyFrame &k = *K[s];
        for (int c = 0; c < 2; c++) {
            cam[s][c].cam_id = c; cam[s][c].epi_ok = pair_ok; cam[s][c].epi_n2 = n2;
            cam[s][c].pair_left1 = n_left1; cam[s][c].pair_left2 = n_left2; cam[s][c].t12x_expect = t12x;
        }
        k.N = N[s]; k.Nleft = NL[s]; k.NLeft = NL[s];
        k.mpCamera = &cam[s][0]; k.mpCamera2 = &cam[s][1];
        k.mvKeys.resize(NL[s]); k.mvKeysUn.resize(NL[s]); k.mvKeysRight.resize(N[s] - NL[s]);
        k.mDescriptors = cv::Mat(N[s] > 0 ? N[s] : 1, 32, CV_8UC1);
        std::memcpy(k.mDescriptors.data, D[s], (size_t)N[s] * 32);
        k.mvpMapPoints.assign(N[s], nullptr);
        k.mvuRight.assign(N[s], -1.f);
        k.mvScaleFactors.assign(16, 1.f);
        k.mvLevelSigma2.assign(16, 1.f);
        for (int i = 0; i < N[s]; i++) {
            cv::KeyPoint &kp = i < NL[s] ? k.mvKeys[i] : k.mvKeysRight[i - NL[s]];
            kp.octave = 0; kp.angle = A[s][i]; kp.class_id = i;   /* class_id = combined index: the verdict table's key */
            if (S[s][i]) k.mvpMapPoints[i] = marker(pool);
        }
        k.mvKeysUn = k.mvKeys;
    }
    K1.Trl.t = Eigen::Vector3f(-1.f, 0.f, 0.f);   /* right camera of KF1 sits at x = +1 */
    K2.Trl.t = Eigen::Vector3f(-4.f, 0.f, 0.f);   /* right camera of KF2 sits at x = +4 */
    featvec(K1.mFeatVec, fv1);
    featvec(K2.mFeatVec, fv2);
    std::vector<std::pair<size_t, size_t>> pairs;
    ORBmatcher m(0.6f, check_orientation != 0);
    int r = m.SearchForTriangulation(&K1, &K2, pairs, false, coarse != 0);
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    for (auto &p : pairs) matches12[p.first] = (int)p.second;
    return r;
}

/* M7 between two key frames of a KannalaBrandt8 stereo rig with real poses (LocalMapping::CreateNewMapPoints' shape for TUM-VI): features [0, n_left) = mvKeys,
 * the rest mvKeysRight; cam_l / cam_r = the rig's two parameter sets (both key frames), pose1 / pose2 = Tcw of the LEFT cameras, trl = Trl (9 floats of R
 * row-major + 3 of t each).  In libmatcher_ref.so the reference's own SearchForTriangulation forms Tll .. Trr (:934-944), picks cameras and pose per pair
 * (:1036-1069) and calls the shell's epipolarConstrain (= the oracle's, pinned to the reference's KannalaBrandt8 text elsewhere); in libmatcher_adapter.so the
 * same call must take the adapter's KannalaBrandt8 route (the shell aborts there).  R12_out [36] / t12_out [12]: the four relative poses ll, lr, rl, rr as the
 * stand-in Sophus of this build computes them (inputs of the oracle's form of the search). */
int matref_search_for_triangulation_kb8_cams(const orbo_keypoint *kps1, int n_left1, const uint8_t *desc1, const uint8_t *skip1, int n1, const orbo_featvec *fv1,
                                             const orbo_keypoint *kps2, int n_left2, const uint8_t *desc2, const uint8_t *skip2, int n2, const orbo_featvec *fv2,
                                             const float *sigma2, int nlevels, const float *cam_l, const float *cam_r, const float *pose1, const float *pose2,
                                             const float *trl, int check_orientation, int coarse, int32_t *matches12, float *R12_out, float *t12_out) {
    KannalaBrandt8 cl1(cam_l), cr1(cam_r), cl2(cam_l), cr2(cam_r);
    KeyFrame KF1, KF2;
    KeyFrame *K[2] = {&KF1, &KF2};
    const orbo_keypoint *P[2] = {kps1, kps2};
    const uint8_t *D[2] = {desc1, desc2};
    const uint8_t *S[2] = {skip1, skip2};
    const float *T[2] = {pose1, pose2};
    const int N[2] = {n1, n2}, NL[2] = {n_left1, n_left2};
    GeometricCamera *cams[2][2] = {{&cl1, &cr1}, {&cl2, &cr2}};
    std::vector<std::unique_ptr<MapPoint>> pool;
    for (int s = 0; s < 2; s++) {
        KeyFrame &k = *K[s];
        k.N = N[s]; k.Nleft = NL[s]; k.NLeft = NL[s];
        k.mpCamera = cams[s][0]; k.mpCamera2 = cams[s][1];
        k.mvKeys.resize(NL[s]); k.mvKeysRight.resize(N[s] - NL[s]);
        k.mDescriptors = cv::Mat(N[s] > 0 ? N[s] : 1, 32, CV_8UC1);
        std::memcpy(k.mDescriptors.data, D[s], (size_t)N[s] * 32);
        k.mvpMapPoints.assign(N[s], nullptr);
        k.mvuRight.assign(N[s], -1.f);
        k.mvScaleFactors.assign(nlevels, 1.f);
        k.mvLevelSigma2.assign(sigma2, sigma2 + nlevels);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) { k.Tcw.R(i, j) = T[s][3 * i + j]; k.Trl.R(i, j) = trl[3 * i + j]; }
            k.Tcw.t(i) = T[s][9 + i]; k.Trl.t(i) = trl[9 + i];
        }
        for (int i = 0; i < N[s]; i++) {
            cv::KeyPoint &kp = i < NL[s] ? k.mvKeys[i] : k.mvKeysRight[i - NL[s]];
            kp = cv::KeyPoint(P[s][i].x, P[s][i].y, P[s][i].size, P[s][i].angle, P[s][i].response, P[s][i].octave, i);
            if (S[s][i]) k.mvpMapPoints[i] = marker(pool);
        }
        k.mvKeysUn = k.mvKeys;
    }
    featvec(KF1.mFeatVec, fv1);
    featvec(KF2.mFeatVec, fv2);
    if (R12_out && t12_out) {   /* ORBmatcher.cc:925-944 on this build's stand-in Sophus */
        const Sophus::SE3f T1w = KF1.GetPose(), Tw2 = KF2.GetPoseInverse(), Tr1w = KF1.GetRightPose(), Twr2 = KF2.GetRightPoseInverse();
        const Sophus::SE3f Tp[4] = {T1w * Tw2, T1w * Twr2, Tr1w * Tw2, Tr1w * Twr2};
        for (int q = 0; q < 4; q++) {
            const Eigen::Matrix3f R = Tp[q].rotationMatrix();
            const Eigen::Vector3f t = Tp[q].translation();
            for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R12_out[9 * q + 3 * i + j] = R(i, j); t12_out[3 * q + i] = t(i); }
        }
    }
    std::vector<std::pair<size_t, size_t>> pairs;
    ORBmatcher m(0.6f, check_orientation != 0);
    int r = m.SearchForTriangulation(&KF1, &KF2, pairs, false, coarse != 0);
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    for (auto &p : pairs) matches12[p.first] = (int)p.second;
    return r;
}

/* SearchBySim3  ORBmatcher.cc:1457-1674: both key frames at identity, S12 = identity, fx = fy = 1, cx = cy = 0, points at
 * z = 1: map point i of KF1 projects to (x1[i], y1[i]) in KF2 and vice versa.  valid: 0 none, 1 map point, 2 bad.
 * match12[i1] = KF2 feature index or -1. */
int matref_search_by_sim3(const orbo_keypoint *kps1, const uint8_t *desc1, int n1, const orbo_keypoint *kps2,
                          const uint8_t *desc2, int n2, const float *bounds, const float *scale, int nlevels,
                          const uint8_t *valid1, const float *x1, const float *y1, const int32_t *lvl1,
                          const uint8_t *mpdesc1, const uint8_t *valid2, const float *x2, const float *y2,
                          const int32_t *lvl2, const uint8_t *mpdesc2, float th, int32_t *match12) {
    GeometricCamera cam;
    KeyFrame K1, K2;
    const Bounds b{bounds[0], bounds[1], bounds[2], bounds[3]};
    fill(K1, kps1, n1, desc1, b, scale, nullptr, nullptr, nlevels, nullptr, &cam);
    fill(K2, kps2, n2, desc2, b, scale, nullptr, nullptr, nlevels, nullptr, &cam);
    std::vector<MapPoint> m1(n1), m2(n2);
    K1.mvpMapPoints.assign(n1, nullptr);
    K2.mvpMapPoints.assign(n2, nullptr);
    for (int i = 0; i < n1; i++) {
        m1[i].id = i; m1[i].pos = Eigen::Vector3f(x1[i], y1[i], 1.f); m1[i].pred_scale = lvl1[i];
        m1[i].desc = desc_row(mpdesc1 + (size_t)i * 32); m1[i].bad = valid1[i] == 2;
        if (valid1[i]) K1.mvpMapPoints[i] = &m1[i];
    }
    for (int i = 0; i < n2; i++) {
        m2[i].id = i; m2[i].pos = Eigen::Vector3f(x2[i], y2[i], 1.f); m2[i].pred_scale = lvl2[i];
        m2[i].desc = desc_row(mpdesc2 + (size_t)i * 32); m2[i].bad = valid2[i] == 2;
        if (valid2[i]) K2.mvpMapPoints[i] = &m2[i];
    }
    std::vector<MapPoint *> out(n1, nullptr);
    Sophus::Sim3f S12;
    ORBmatcher m(0.75f, true);
    int r = m.SearchBySim3(&K1, &K2, out, S12, th);
    for (int i = 0; i < n1; i++) match12[i] = out[i] ? out[i]->id : -1;
    return r;
}

#ifdef ORBX_WITH_SLAM_TYPES
/* adapter build only (libmatcher_adapter.so): the product's ORBmatcher::ComputeStereoFishEyeMatches (kNN-2 on the GPU) on the arrays
 * frameref_stereo_fisheye_matches (ref_frame_shim.cc: the reference's own Frame::ComputeStereoFishEyeMatches) takes */
int matref_adapter_stereo_fisheye_matches(const orbo_keypoint *kl, const uint8_t *dl, int nl, int mono_left, const orbo_keypoint *kr,
                                          const uint8_t *dr, int nr, int mono_right, const float *level_sigma2, int nlevels,
                                          orbo_triangulate_fn tri, void *ctx, int32_t *l2r, int32_t *r2l, float *depth, float *u_right, float *p3d) {
    ORBmatcher m(0.75f, true);
    std::vector<int> a, b;
    std::vector<float> d, u;
    std::vector<std::array<float, 3>> p;
    (void)nlevels;
    const int n = m.ComputeStereoFishEyeMatches((const orbx_keypoint *)kl, dl, nl, mono_left, (const orbx_keypoint *)kr, dr, nr, mono_right, level_sigma2,
                                                [&](int il, int ir, float s1, float s2, float *p3) { return tri(ctx, il, ir, s1, s2, p3); }, a, b, d, u, p);
    for (int i = 0; i < nl; i++) { l2r[i] = a[i]; depth[i] = d[i]; u_right[i] = u[i]; for (int c = 0; c < 3; c++) p3d[3 * i + c] = p[i][c]; }
    for (int i = 0; i < nr; i++) r2l[i] = b[i];
    return n;
}
#endif

}  // extern "C"
