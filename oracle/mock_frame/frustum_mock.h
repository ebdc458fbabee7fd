/* TEST INFRASTRUCTURE ONLY.  Class shells for Frame::isInFrustum (Frame.cc:512-586), MapPoint::PredictScale(dist, Frame*)
 * (MapPoint.cc:531-546) and Pinhole::project(Vector3f) (CameraModels/Pinhole.cpp:43-49): the data members those three functions touch,
 * with the reference's names.  Their BODIES are the reference's own text, excerpted at build time (oracle/Makefile,
 * _ref/libfrustum_ref.so).  Eigen is the stand-in of mock_slam/mini_eigen.h: plain float arithmetic in the order written there. */
#pragma once
#include <cmath>
#include <mutex>
#include <vector>
#include "../mock_slam/mini_eigen.h"

extern "C" void orbo_eigen_jacobi_svd4_V(const float *A16, float *V16, float *sv4);   /* liborb_oracle.so: the restated Eigen::JacobiSVD (PARITY UNPINNED) */

namespace Eigen {
/* what KannalaBrandt8::TriangulateMatches / Triangulate (KannalaBrandt8.cpp:305-400) spell: 3x4 and 4x4 float matrices with comma initialisation from blocks,
 * row / column access, and JacobiSVD<Matrix4f>(A, ComputeFullV).matrixV() -- the decomposition itself is the oracle's restatement of Eigen's algorithm
 * (Eigen is absent here: that part of the pin is the restatement checked against itself; everything around it is the reference's text) */
typedef Vec<4> Vector4f;
template <int R, int C>
struct Mat {
    float m[R][C];
    Mat() { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) m[i][j] = 0.f; }
    float &operator()(int i, int j) { return m[i][j]; }
    float operator()(int i, int j) const { return m[i][j]; }
    struct RowRef {
        Mat &M; int i;
        RowRef &operator=(const Vec<C> &v) { for (int j = 0; j < C; j++) M.m[i][j] = v.v[j]; return *this; }
        operator Vec<C>() const { Vec<C> r; for (int j = 0; j < C; j++) r.v[j] = M.m[i][j]; return r; }
    };
    RowRef row(int i) { return RowRef{*this, i}; }
    Vec<C> row(int i) const { Vec<C> r; for (int j = 0; j < C; j++) r.v[j] = m[i][j]; return r; }
    Vec<R> col(int j) const { Vec<R> r; for (int i = 0; i < R; i++) r.v[i] = m[i][j]; return r; }
    struct Comma {   /* M << block, block: blocks are laid left to right (a 3x3 block, then a column) */
        Mat &M; int c;
        Comma &operator,(const Matrix3f &b) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M.m[i][c + j] = b.m[i][j]; c += 3; return *this; }
        Comma &operator,(const Vec<R> &v) { for (int i = 0; i < R; i++) M.m[i][c] = v.v[i]; c += 1; return *this; }
    };
    Comma operator<<(const Matrix3f &b) { Comma k{*this, 0}; k, b; return k; }
};
enum { ComputeFullV = 8 };
typedef Mat<4, 4> Matrix4f;
template <class M> struct JacobiSVD;
template <> struct JacobiSVD<Matrix4f> {
    Matrix4f V;
    JacobiSVD(const Matrix4f &A, int) { orbo_eigen_jacobi_svd4_V(&A.m[0][0], &V.m[0][0], nullptr); }
    const Matrix4f &matrixV() const { return V; }
};
template <class T, int R, int C> struct MatrixSel;
template <> struct MatrixSel<float, 3, 1> { typedef Vector3f type; };
template <> struct MatrixSel<float, 3, 3> { typedef Matrix3f type; };
template <> struct MatrixSel<float, 3, 4> { typedef Mat<3, 4> type; };
template <> struct MatrixSel<float, 4, 4> { typedef Mat<4, 4> type; };
template <class T, int R, int C> using Matrix = typename MatrixSel<T, R, C>::type;
}  // namespace Eigen

namespace cv {   /* the three OpenCV value types the camera model spells */
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float a, float b) : x(a), y(b) {} };
struct Point3f { float x, y, z; Point3f() : x(0), y(0), z(0) {} Point3f(float a, float b, float c) : x(a), y(b), z(c) {} };
struct KeyPoint { Point2f pt; int octave = 0; };
}  // namespace cv
#define CV_PI 3.1415926535897932384626433832795

namespace ORB_SLAM3 {

class Frame;
class KeyFrame;

class GeometricCamera {
public:
    std::vector<float> mvParameters; /* fx, fy, cx, cy */
    virtual ~GeometricCamera() {}
    virtual Eigen::Vector2f project(const Eigen::Vector3f &v3D) = 0;
    virtual Eigen::Vector3f unprojectEig(const cv::Point2f &p2D) { return Eigen::Vector3f(); }
};
class Pinhole : public GeometricCamera {
public:
    Eigen::Vector2f project(const Eigen::Vector3f &v3D) override;
};
class KannalaBrandt8 : public GeometricCamera {   /* mvParameters: fx, fy, cx, cy, k0 .. k3; bodies: ref_kb8_shim.cc (a translation unit WITHOUT `using namespace std`, as the reference's) */
public:
    const float precision = 1e-6f;   /* KannalaBrandt8.h:42-59 */
    Eigen::Vector2f project(const Eigen::Vector3f &v3D) override;
    cv::Point3f unproject(const cv::Point2f &p2D);
    Eigen::Vector3f unprojectEig(const cv::Point2f &p2D) override;
    bool epipolarConstrain(GeometricCamera *pCamera2, const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const Eigen::Matrix3f &R12, const Eigen::Vector3f &t12,
                           const float sigmaLevel, const float unc);
    float TriangulateMatches(GeometricCamera *pCamera2, const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const Eigen::Matrix3f &R12, const Eigen::Vector3f &t12,
                             const float sigmaLevel, const float unc, Eigen::Vector3f &p3D);
    void Triangulate(const cv::Point2f &p1, const cv::Point2f &p2, const Eigen::Matrix<float, 3, 4> &Tcw1, const Eigen::Matrix<float, 3, 4> &Tcw2, Eigen::Vector3f &x3D);
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
