/* TEST INFRASTRUCTURE ONLY.  Stand-in SLAM types (Frame / KeyFrame / MapPoint / GeometricCamera) plus a minimal Eigen /
 * Sophus vocabulary, just large enough that the reference's own src/ORBmatcher.cc compiles WHERE IT LIES (see
 * oracle/Makefile, target _ref/libmatcher_ref.so).  Nothing here restates matcher logic: the matcher loops that run are
 * the reference's; these types only carry the flattened test data into them.  Frame/KeyFrame::GetFeaturesInArea forward to
 * the oracle's grid (orbo_grid_query), so the grid lookup itself stays a restatement (Frame.cc / KeyFrame.cc cannot be
 * built here: they pull in g2o, Eigen, OpenCV calib3d ...).  The math types do plain float arithmetic; the parity tests
 * drive the geometry with identity poses and a camera whose project() returns (x, y) unchanged, so every projected pixel
 * is exactly the number the test chose. */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstddef>
#include <map>
#include <set>
#include <vector>
#include <list>
#include <utility>
#include <tuple>
#include <opencv2/core/core.hpp>
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "../orb_oracle.h"

#include "mini_eigen.h"

namespace ORB_SLAM3 {

class KeyFrame;
class Frame;

/* project() hands back (x, y) of the camera-frame point: the test stores the wanted pixel in the map point itself */
class GeometricCamera {
public:
    virtual ~GeometricCamera() {}
    virtual Eigen::Vector2f project(const Eigen::Vector3f &p) { return Eigen::Vector2f(p(0), p(1)); }
    /* include/CameraModels/GeometricCamera.h:77,93-96,105.  The table-driven stand-in below is "some camera model whose epipolarConstrain is
     * opaque host code" (as KannalaBrandt8's is to the adapter): it reports CAM_FISHEYE; class Pinhole further down reports CAM_PINHOLE */
    const static unsigned int CAM_PINHOLE = 0;
    const static unsigned int CAM_FISHEYE = 1;
    unsigned int mnType = CAM_FISHEYE;
    unsigned int GetType() { return mnType; }
    virtual Eigen::Matrix3f toK_() { return Eigen::Matrix3f(); }
    /* GeometricCamera.h:85-88: parameter access.  The table-driven stand-in has none (size() == 0): the adapter then keeps the host callback, which is what
     * the shim's table-driven fisheye cases test; class KannalaBrandt8 below has 8 and goes to the device gate (matref_search_for_triangulation_kb8_cams) */
    virtual size_t size() { return 0; }
    virtual float getParameter(const int) { return 0.f; }
    /* the verdicts of the epipolar test are test data: ok[idx1 * n2 + idx2], keyed by keypoint identity (class_id) */
    const uint8_t *epi_ok = nullptr;
    int epi_n2 = 0;
    /* fisheye-stereo pairings (ORBmatcher.cc:1036-1069): when pair_left1 >= 0 the verdict also demands that the caller picked the
     * camera objects and the relative translation of the pairing the two keypoints belong to (class_id >= n_left: right camera);
     * t12x_expect[2 * right1 + right2] is the x component the test gave that pairing's t12 */
    int cam_id = 0, pair_left1 = -1, pair_left2 = -1;
    const float *t12x_expect = nullptr;
    virtual bool epipolarConstrain(GeometricCamera *other, const cv::KeyPoint &kp1, const cv::KeyPoint &kp2,
                                   const Eigen::Matrix3f &, const Eigen::Vector3f &t12, const float, const float) {
        if (pair_left1 >= 0) {
            const int r1 = kp1.class_id >= pair_left1, r2 = kp2.class_id >= pair_left2;
            if (cam_id != r1 || !other || other->cam_id != r2) return false;
            if (t12x_expect && t12(0) != t12x_expect[2 * r1 + r2]) return false;
        }
        return epi_ok ? epi_ok[(size_t)kp1.class_id * epi_n2 + kp2.class_id] != 0 : true;
    }
};

/* shell for CameraModels/Pinhole: project(Vector3f) and epipolarConstrain are the reference's own text in libmatcher_ref.so (excerpted by
 * oracle/Makefile into a temporary include of ref_matcher_shim.cc); toK_ is Pinhole.cpp:100-104 restated.  In libmatcher_adapter.so
 * epipolarConstrain aborts: the adapter must take pinhole key frames through the on-device gates and never call it. */
class Pinhole : public GeometricCamera {
public:
    std::vector<float> mvParameters;
    Pinhole(float fx, float fy, float cx, float cy) : mvParameters{fx, fy, cx, cy} { mnType = CAM_PINHOLE; }
    Eigen::Vector2f project(const Eigen::Vector3f &v3D) override;
    Eigen::Matrix3f toK_() override {
        Eigen::Matrix3f K;
        K(0, 0) = mvParameters[0]; K(0, 1) = 0.f; K(0, 2) = mvParameters[2];
        K(1, 0) = 0.f; K(1, 1) = mvParameters[1]; K(1, 2) = mvParameters[3];
        K(2, 0) = 0.f; K(2, 1) = 0.f; K(2, 2) = 1.f;
        return K;
    }
    bool epipolarConstrain(GeometricCamera *pCamera2, const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const Eigen::Matrix3f &R12,
                           const Eigen::Vector3f &t12, const float sigmaLevel, const float unc) override;
};

/* shell for CameraModels/KannalaBrandt8 (the reference's CAM_FISHEYE model: 8 parameters, include/CameraModels/KannalaBrandt8.h).  In libmatcher_ref.so
 * epipolarConstrain is the oracle's KannalaBrandt8::epipolarConstrain (orbo_kb8_triangulate_matches > 0.0001, pinned against the reference's own text in
 * tests/test_oracle_geometry.py): the reference's SearchForTriangulation picks the camera objects and the relative pose per pair (ORBmatcher.cc:1036-1069) and calls
 * it.  In libmatcher_adapter.so it aborts: the adapter must take such key frames to the device gate (orbx_search_for_triangulation_kb8) and never call it. */
class KannalaBrandt8 : public GeometricCamera {
public:
    std::vector<float> mvParameters;
    explicit KannalaBrandt8(const float *p8) : mvParameters(p8, p8 + 8) { mnType = CAM_FISHEYE; }
    size_t size() override { return mvParameters.size(); }
    float getParameter(const int i) override { return mvParameters[i]; }
    bool epipolarConstrain(GeometricCamera *pCamera2, const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const Eigen::Matrix3f &R12,
                           const Eigen::Vector3f &t12, const float sigmaLevel, const float unc) override;
};

class MapPoint {
public:
    /* tracking fields the matchers read (MapPoint.h) */
    float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackDepthR = 0, mTrackProjXR = 0, mTrackProjYR = 0;
    bool mbTrackInView = false, mbTrackInViewR = false;
    int mnTrackScaleLevel = 0, mnTrackScaleLevelR = -1;
    float mTrackViewCos = 1.f, mTrackViewCosR = 1.f;
    /* test data */
    int id = -1;
    bool bad = false;
    int nobs = 1;
    int pred_scale = 0;
    Eigen::Vector3f pos, normal;
    float min_dist = 0.f, max_dist = 3.0e38f;
    cv::Mat desc; /* 1 x 32 */
    std::map<KeyFrame *, int> in_kf;
    std::vector<std::pair<KeyFrame *, int>> added_obs;
    MapPoint *replaced_by = nullptr;

    bool isBad() { return bad; }
    int Observations() { return nobs; }
    cv::Mat GetDescriptor() { return desc; }
    Eigen::Vector3f GetWorldPos() { return pos; }
    Eigen::Vector3f GetNormal() { return normal; }
    float GetMinDistanceInvariance() { return min_dist; }
    float GetMaxDistanceInvariance() { return max_dist; }
    int PredictScale(const float &, KeyFrame *) { return pred_scale; }
    int PredictScale(const float &, Frame *) { return pred_scale; }
    bool IsInKeyFrame(KeyFrame *kf) { return in_kf.count(kf) != 0; }
    std::tuple<int, int> GetIndexInKeyFrame(KeyFrame *kf) {
        auto it = in_kf.find(kf);
        return it == in_kf.end() ? std::tuple<int, int>(-1, -1) : std::tuple<int, int>(it->second, -1);
    }
    void AddObservation(KeyFrame *kf, int idx) { added_obs.push_back({kf, idx}); in_kf[kf] = idx; }
    void Replace(MapPoint *p) { replaced_by = p; }
};

struct FeatureHolder {
    int N = 0;
    int Nleft = -1, NLeft = -1;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
    GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
    float fx = 1, fy = 1, cx = 0, cy = 0, mbf = 0, mb = 0;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
    Sophus::SE3f Tcw, Trl;
    /* grid over mvKeysUn held by the oracle; kps_un is the flattened copy it indexes */
    std::vector<orbo_keypoint> kps_un, kps_right;
    orbo_grid *grid = nullptr, *grid_right = nullptr; /* fisheye stereo: mGridRight over mvKeysRight (Frame.cc:410-413) */
    ~FeatureHolder() { if (grid) orbo_grid_destroy(grid); if (grid_right) orbo_grid_destroy(grid_right); }
    std::vector<size_t> area(float x, float y, float r, int minLevel, int maxLevel, bool bRight = false) const {
        std::vector<int32_t> tmp(N > 0 ? N : 1);
        int n = orbo_grid_query(bRight ? grid_right : grid, x, y, r, minLevel, maxLevel, tmp.data(), (int)tmp.size());
        return std::vector<size_t>(tmp.begin(), tmp.begin() + n);
    }
};

class Frame : public FeatureHolder {
public:
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel = -1,
                                          const int maxLevel = -1, const bool bRight = false) const {
        return area(x, y, r, minLevel, maxLevel, bRight);
    }
    Sophus::SE3f GetPose() const { return Tcw; }
    Sophus::SE3f GetRelativePoseTrl() const { return Trl; }
};

class KeyFrame : public FeatureHolder {
public:
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<std::pair<MapPoint *, int>> added;
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    std::set<MapPoint *> GetMapPoints() {
        std::set<MapPoint *> s;
        for (MapPoint *p : mvpMapPoints) if (p && !p->isBad()) s.insert(p);
        return s;
    }
    /* probe: slots read as empty and stay empty, so Fuse reports every accepted query through AddObservation */
    bool probe = false;
    MapPoint *GetMapPoint(const size_t &idx) { return probe ? nullptr : mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint *p, const size_t &idx) { if (!probe) mvpMapPoints[idx] = p; added.push_back({p, (int)idx}); }
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const bool bRight = false) const {
        return area(x, y, r, -1, -1, bRight);
    }
    bool IsInImage(const float &x, const float &y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }
    Sophus::SE3f GetPose() { return Tcw; }
    Sophus::SE3f GetPoseInverse() { return Tcw.inverse(); }
    Sophus::SE3f GetRightPose() { return Trl * Tcw; }
    Sophus::SE3f GetRightPoseInverse() { return (Trl * Tcw).inverse(); }
    Sophus::SE3f GetRelativePoseTrl() { return Trl; }
    Eigen::Vector3f GetCameraCenter() { return Tcw.inverse().translation(); }
    Eigen::Vector3f GetRightCameraCenter() { return (Trl * Tcw).inverse().translation(); }
};

}  // namespace ORB_SLAM3
