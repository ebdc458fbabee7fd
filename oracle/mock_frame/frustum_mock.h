/* TEST INFRASTRUCTURE ONLY.  Class shells for Frame::isInFrustum (Frame.cc:512-586), MapPoint::PredictScale(dist, Frame*)
 * (MapPoint.cc:531-546) and Pinhole::project(Vector3f) (CameraModels/Pinhole.cpp:43-49): the data members those three functions touch,
 * with the reference's names.  Their BODIES are the reference's own text, excerpted at build time (oracle/Makefile,
 * _ref/libfrustum_ref.so).  Eigen is the stand-in of mock_slam/mini_eigen.h: plain float arithmetic in the order written there. */
#pragma once
#include <cmath>
#include <mutex>
#include <vector>
#include "../mock_slam/mini_eigen.h"

namespace Eigen {
template <class T, int R, int C> struct MatrixSel;
template <> struct MatrixSel<float, 3, 1> { typedef Vector3f type; };
template <> struct MatrixSel<float, 3, 3> { typedef Matrix3f type; };
template <class T, int R, int C> using Matrix = typename MatrixSel<T, R, C>::type;
}  // namespace Eigen

namespace ORB_SLAM3 {

class Frame;
class KeyFrame;

class GeometricCamera {
public:
    std::vector<float> mvParameters; /* fx, fy, cx, cy */
    virtual ~GeometricCamera() {}
    virtual Eigen::Vector2f project(const Eigen::Vector3f &v3D) = 0;
};
class Pinhole : public GeometricCamera {
public:
    Eigen::Vector2f project(const Eigen::Vector3f &v3D) override;
};
class KannalaBrandt8 : public GeometricCamera {   /* mvParameters: fx, fy, cx, cy, k0 .. k3; body: ref_kb8_shim.cc (a translation unit WITHOUT `using namespace std`, as the reference's) */
public:
    Eigen::Vector2f project(const Eigen::Vector3f &v3D) override;
};

class MapPoint {
public:
    /* MapPoint.h:171-179 */
    float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackDepthR = 0, mTrackProjXR = 0, mTrackProjYR = 0;
    bool mbTrackInView = false, mbTrackInViewR = false;
    int mnTrackScaleLevel = 0, mnTrackScaleLevelR = 0;
    float mTrackViewCos = 0, mTrackViewCosR = 0;
    /* position, normal, scale-invariance distances */
    Eigen::Vector3f mWorldPos, mNormalVector;
    float mfMinDistance = 0, mfMaxDistance = 0;
    std::mutex mMutexPos;
    Eigen::Vector3f GetWorldPos() { return mWorldPos; }
    Eigen::Vector3f GetNormal() { return mNormalVector; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }   /* MapPoint.cc:502-512 */
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    int PredictScale(const float &currentDist, Frame *pF);
};

class Frame {
public:
    int Nleft = -1;
    Eigen::Matrix<float, 3, 1> mOw;
    Eigen::Matrix<float, 3, 3> mRcw;
    Eigen::Matrix<float, 3, 1> mtcw;
    GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
    Eigen::Matrix<float, 3, 3> mRwc;
    Sophus::SE3f mTrl, mTlr;   /* Frame.h: Sophus::SE3<float> mTlr, mTrl */
    float mbf = 0;
    int mnScaleLevels = 0;
    float mfLogScaleFactor = 0;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    bool isInFrustum(MapPoint *pMP, float viewingCosLimit);
    bool isInFrustumChecks(MapPoint *pMP, float viewingCosLimit, bool bRight = false);
};

}  // namespace ORB_SLAM3
