/* TEST INFRASTRUCTURE ONLY.  oracle/_ref/libframe_ref.so: member functions of the reference's Frame.cc, KeyFrame.cc and MapPoint.cc
 * that the hot path's restatements follow -- Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea / ComputeStereoMatches,
 * KeyFrame::GetFeaturesInArea, MapPoint::ComputeDistinctiveDescriptors, Pinhole::epipolarConstrain -- compiled from the reference's own text.  Those files cannot
 * be compiled whole in this image (g2o, Eigen, OpenCV calib3d ...), so oracle/Makefile writes the seven function definitions, verbatim
 * and untouched, into a temporary ref_excerpt.inc (ref_excerpt.awk; deleted after the build, never in the repo) which is included
 * below inside class shells that declare only the members they touch (mock_frame/frame_mock.h).  ORBextractor (for mvImagePyramid)
 * is the reference's own class; ORBmatcher::DescriptorDistance / TH_* come from the reference's ORBmatcher.cc in libmatcher_ref.so. */
#include <cstring>
#include <memory>
#include "ref_excerpt_defs.inc" /* the two FRAME_GRID_* #define lines of the reference's include/Frame.h */
#include "frame_mock.h"
#include "../orb_oracle.h"

using namespace std;

namespace ORB_SLAM3 {
cv::BFMatcher Frame::BFmatcher = cv::BFMatcher(cv::NORM_HAMMING);   /* as Frame.cc:43 defines it */
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv;
#include "ref_excerpt.inc"
}  // namespace ORB_SLAM3

using namespace ORB_SLAM3;

namespace {
void keys(std::vector<cv::KeyPoint> &dst, const orbo_keypoint *k, int n) {
    dst.resize(n);
    for (int i = 0; i < n; i++) dst[i] = cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id);
}
cv::Mat rows32(const uint8_t *d, int n) {
    cv::Mat m(n > 0 ? n : 1, 32, CV_8UC1);
    if (n) std::memcpy(m.data, d, (size_t)n * 32);
    return m;
}
void set_bounds(float minx, float maxx, float miny, float maxy) { /* Frame.cc:340-343 (first-frame initialisation) */
    Frame::mnMinX = minx; Frame::mnMaxX = maxx; Frame::mnMinY = miny; Frame::mnMaxY = maxy;
    Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(maxx - minx);
    Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(maxy - miny);
}
struct GridHandle {
    Frame F;
    KeyFrame KF;
};
}  // namespace

extern "C" {

int frameref_grid_dims(int *cols, int *rows) { *cols = FRAME_GRID_COLS; *rows = FRAME_GRID_ROWS; return 0; }

/* Frame with mono keypoints, grid assigned by the reference's AssignFeaturesToGrid; the KeyFrame copy holds the same grid the
 * way KeyFrame's constructor copies it (KeyFrame.cc:60-78: mGrid[i][j] = F.mGrid[i][j]) */
void *frameref_grid_create(const orbo_keypoint *kps_un, int n, float minx, float maxx, float miny, float maxy) {
    GridHandle *h = new GridHandle();
    set_bounds(minx, maxx, miny, maxy);
    h->F.N = n;
    keys(h->F.mvKeysUn, kps_un, n);
    h->F.mvKeys = h->F.mvKeysUn;
    h->F.AssignFeaturesToGrid();
    KeyFrame &K = h->KF;
    K.N = n; K.mvKeysUn = h->F.mvKeysUn; K.mvKeys = K.mvKeysUn;
    K.mnGridCols = FRAME_GRID_COLS; K.mnGridRows = FRAME_GRID_ROWS;
    K.mnMinX = minx; K.mnMaxX = maxx; K.mnMinY = miny; K.mnMaxY = maxy;
    K.mfGridElementWidthInv = Frame::mfGridElementWidthInv; K.mfGridElementHeightInv = Frame::mfGridElementHeightInv;
    K.mGrid.resize(K.mnGridCols);
    for (int i = 0; i < K.mnGridCols; i++) {
        K.mGrid[i].resize(K.mnGridRows);
        for (int j = 0; j < K.mnGridRows; j++) K.mGrid[i][j] = h->F.mGrid[i][j];
    }
    return h;
}
void frameref_grid_destroy(void *h) { delete (GridHandle *)h; }
/* the fisheye-stereo layout (Nleft != -1): features [0, n_left) are mvKeys, the rest mvKeysRight; AssignFeaturesToGrid fills mGrid
 * and mGridRight (Frame.cc:395-413) */
void *frameref_grid_create_stereo(const orbo_keypoint *kps_left, int n_left, const orbo_keypoint *kps_right, int n_right, float minx,
                                  float maxx, float miny, float maxy) {
    GridHandle *h = new GridHandle();
    set_bounds(minx, maxx, miny, maxy);
    h->F.N = n_left + n_right;
    h->F.Nleft = n_left;
    keys(h->F.mvKeys, kps_left, n_left);
    keys(h->F.mvKeysRight, kps_right, n_right);
    h->F.AssignFeaturesToGrid();
    return h;
}
/* Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel, bRight) of such a frame */
int frameref_grid_query_stereo(void *hv, int right, float minx, float maxx, float miny, float maxy, float x, float y, float r,
                               int min_level, int max_level, int32_t *out, int cap) {
    GridHandle *h = (GridHandle *)hv;
    set_bounds(minx, maxx, miny, maxy);
    std::vector<size_t> v = h->F.GetFeaturesInArea(x, y, r, min_level, max_level, right != 0);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = (int32_t)v[i];
    return (int)v.size();
}


/* which = 0: Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel); 1: KeyFrame::GetFeaturesInArea(x, y, r) */
int frameref_grid_query(void *hv, int which, float minx, float maxx, float miny, float maxy, float x, float y, float r,
                        int min_level, int max_level, int32_t *out, int cap) {
    GridHandle *h = (GridHandle *)hv;
    set_bounds(minx, maxx, miny, maxy); /* the Frame statics are process-wide: re-assert this handle's */
    std::vector<size_t> v = which == 0 ? h->F.GetFeaturesInArea(x, y, r, min_level, max_level) : h->KF.GetFeaturesInArea(x, y, r);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = (int32_t)v[i];
    return (int)v.size();
}

/* Frame::ComputeStereoMatches on flattened inputs; pyramids: per level the ROI origin, size and stride */
int frameref_compute_stereo_matches(const orbo_keypoint *kl, const uint8_t *dl, int nl, const orbo_keypoint *kr, const uint8_t *dr,
                                    int nr, const float *scale, const float *inv_scale, int nlevels,
                                    const uint8_t *const *pyr_left, const uint8_t *const *pyr_right, const int *pyr_w,
                                    const int *pyr_h, const size_t *pyr_stride, float bf, float b, float *u_right, float *depth) {
    ORBextractor exL(100, 1.2f, nlevels, 20, 7), exR(100, 1.2f, nlevels, 20, 7);
    for (int l = 0; l < nlevels; l++) {
        const uint8_t *src[2] = {pyr_left[l], pyr_right[l]};
        ORBextractor *ex[2] = {&exL, &exR};
        for (int s = 0; s < 2; s++) {
            cv::Mat m(pyr_h[l], pyr_w[l], CV_8UC1);
            for (int y = 0; y < pyr_h[l]; y++) std::memcpy(m.ptr(y), src[s] + (size_t)y * pyr_stride[l], pyr_w[l]);
            ex[s]->mvImagePyramid[l] = m;
        }
    }
    Frame F;
    F.N = nl;
    keys(F.mvKeys, kl, nl);
    keys(F.mvKeysRight, kr, nr);
    F.mDescriptors = rows32(dl, nl);
    F.mDescriptorsRight = rows32(dr, nr);
    F.mvScaleFactors.assign(scale, scale + nlevels);
    F.mvInvScaleFactors.assign(inv_scale, inv_scale + nlevels);
    F.mbf = bf; F.mb = b;
    F.mpORBextractorLeft = &exL; F.mpORBextractorRight = &exR;
    F.ComputeStereoMatches();
    int n = 0;
    for (int i = 0; i < nl; i++) {
        u_right[i] = F.mvuRight[i];
        depth[i] = F.mvDepth[i];
        n += F.mvDepth[i] > 0;
    }
    return n;
}

/* MapPoint::ComputeDistinctiveDescriptors for one observation set (descriptors in observation order); writes the chosen
 * descriptor, returns 0, or -1 when the reference returns early (empty set) */
int frameref_distinctive_descriptor(const uint8_t *desc, int n, uint8_t *out32) {
    std::vector<KeyFrame> kfs(n); /* contiguous: std::map<KeyFrame*> iterates in index order */
    MapPoint mp;
    for (int i = 0; i < n; i++) {
        kfs[i].mDescriptors = rows32(desc + (size_t)i * 32, 1);
        mp.mObservations[&kfs[i]] = std::tuple<int, int>(0, -1);
    }
    mp.ComputeDistinctiveDescriptors();
    if (mp.mDescriptor.empty()) return -1;
    std::memcpy(out32, mp.mDescriptor.data, 32);
    return 0;
}

/* Pinhole::epipolarConstrain (CameraModels/Pinhole.cpp:107-129) for n keypoint pairs.  K1, K2: fx, fy, cx, cy.  R12 row-major.
 * F12_out receives the fundamental matrix the reference code used (the stand-in Matrix3f records its last product), so a test can
 * hand exactly that matrix to the oracle / the device. */
void frameref_epipolar_pinhole(const float *K1, const float *K2, const float *R12, const float *t12, int n, const float *x1,
                               const float *y1, const float *x2, const float *y2, const float *unc, uint8_t *ok, float *F12_out) {
    Pinhole c1, c2;
    c1.mvParameters.assign(K1, K1 + 4);
    c2.mvParameters.assign(K2, K2 + 4);
    Eigen::Matrix3f R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = R12[3 * i + j];
    const Eigen::Vector3f t(t12[0], t12[1], t12[2]);
    for (int i = 0; i < n; i++) {
        cv::KeyPoint a(x1[i], y1[i], 31.f), b(x2[i], y2[i], 31.f);
        ok[i] = c1.epipolarConstrain(&c2, a, b, R, t, 1.0f, unc[i]) ? 1 : 0;
    }
    const Eigen::Matrix3f &F = Eigen::last_product();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) F12_out[3 * i + j] = F(i, j);
}

/* Frame::ComputeStereoFishEyeMatches (Frame.cc:1126-1166, the reference's text) on flattened inputs; the camera's TriangulateMatches is
 * the stand-in of mock_frame/frame_mock.h.  p3d is reported for accepted matches only (the reference leaves the others unset). */
int frameref_stereo_fisheye_matches(const orbo_keypoint *kl, const uint8_t *dl, int nl, int mono_left, const orbo_keypoint *kr,
                                    const uint8_t *dr, int nr, int mono_right, const float *level_sigma2, int nlevels,
                                    orbo_triangulate_fn tri, void *ctx, int32_t *l2r, int32_t *r2l, float *depth, float *u_right, float *p3d) {
    Frame F;
    KannalaBrandt8 cam, cam2;
    keys(F.mvKeys, kl, nl);
    keys(F.mvKeysRight, kr, nr);
    F.mDescriptors = cv::Mat(nl, 32, CV_8UC1);
    F.mDescriptorsRight = cv::Mat(nr, 32, CV_8UC1);
    if (nl) std::memcpy(F.mDescriptors.data, dl, (size_t)nl * 32);
    if (nr) std::memcpy(F.mDescriptorsRight.data, dr, (size_t)nr * 32);
    F.Nleft = nl; F.Nright = nr; F.N = nl + nr; F.monoLeft = mono_left; F.monoRight = mono_right;
    F.mvLevelSigma2.assign(level_sigma2, level_sigma2 + nlevels);
    cam.frame = &F; cam.fn = tri; cam.ctx = ctx;
    F.mpCamera = &cam; F.mpCamera2 = &cam2;
    F.ComputeStereoFishEyeMatches();
    int n = 0;
    for (int i = 0; i < nl; i++) {
        l2r[i] = F.mvLeftToRightMatch[i]; depth[i] = F.mvDepth[i]; u_right[i] = F.mvuRight[i];
        const bool ok = l2r[i] >= 0;
        for (int c = 0; c < 3; c++) p3d[3 * i + c] = ok ? F.mvStereo3Dpoints[i](c) : 0.f;
        n += ok;
    }
    for (int i = 0; i < nr; i++) r2l[i] = F.mvRightToLeftMatch[i];
    return n;
}

}  // extern "C"
