/* TEST INFRASTRUCTURE ONLY.  Class shells for the member functions of the reference's Frame.cc / KeyFrame.cc / MapPoint.cc that
 * oracle/Makefile excerpts at build time (_ref/libframe_ref.so): just the data members those functions touch, with the
 * reference's names.  The function BODIES compiled into the library are the reference's own text (see ref_frame_shim.cc);
 * FRAME_GRID_ROWS / FRAME_GRID_COLS are taken from the reference's include/Frame.h by the same recipe. */
#pragma once
#include <climits>
#include <cmath>
#include <cstddef>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include <algorithm>
#include <opencv2/opencv.hpp>
#include "ORBextractor.h" /* the reference's own header (include/ORBextractor.h): mvImagePyramid */
#include "../mock_slam/mini_eigen.h"

namespace ORB_SLAM3 {

/* defined by the reference's ORBmatcher.cc inside _ref/libmatcher_ref.so */
class ORBmatcher {
public:
    static const int TH_LOW, TH_HIGH, HISTO_LENGTH;
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);
};

class GeometricCamera;
class Frame {
public:
    int N = 0, Nleft = -1;
    /* fisheye stereo (Frame::ComputeStereoFishEyeMatches, Frame.cc:1126-1166) */
    int Nright = -1, monoLeft = -1, monoRight = -1, mnCloseMPs = 0;
    std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
    std::vector<Eigen::Vector3f> mvStereo3Dpoints;
    std::vector<float> mvLevelSigma2;
    GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
    Eigen::Matrix3f mRlr;
    Eigen::Vector3f mtlr;
    static cv::BFMatcher BFmatcher;   /* Frame.cc:43: cv::BFMatcher(cv::NORM_HAMMING) */
    void ComputeStereoFishEyeMatches();
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<float> mvScaleFactors, mvInvScaleFactors;
    float mb = 0, mbf = 0;
    ORBextractor *mpORBextractorLeft = nullptr, *mpORBextractorRight = nullptr;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    std::vector<std::size_t> mGridRight[FRAME_GRID_COLS][FRAME_GRID_ROWS];

    void AssignFeaturesToGrid();
    bool PosInGrid(const cv::KeyPoint &kp, int &posX, int &posY);
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel = -1,
                                          const int maxLevel = -1, const bool bRight = false) const;
    void ComputeStereoMatches();
};

class KeyFrame {
public:
    int N = 0, NLeft = -1;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    cv::Mat mDescriptors;
    int mnGridCols = 0, mnGridRows = 0;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    std::vector<std::vector<std::vector<size_t>>> mGrid, mGridRight;
    bool bad = false;
    bool isBad() { return bad; }
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const bool bRight = false) const;
};

class MapPoint {
public:
    std::mutex mMutexFeatures;
    bool mbBad = false;
    std::map<KeyFrame *, std::tuple<int, int>> mObservations;
    cv::Mat mDescriptor;
    void ComputeDistinctiveDescriptors();
};

/* shells for CameraModels/Pinhole.cpp's epipolarConstrain (the body is the reference's; toK_ is Pinhole.cpp:100-104 restated) */
class GeometricCamera {
public:
    std::vector<float> mvParameters; /* fx, fy, cx, cy */
    virtual ~GeometricCamera() {}
    virtual Eigen::Matrix3f toK_() = 0;
};
class Pinhole : public GeometricCamera {
public:
    Eigen::Matrix3f toK_() override {
        Eigen::Matrix3f K;
        K(0, 0) = mvParameters[0]; K(0, 1) = 0.f; K(0, 2) = mvParameters[2];
        K(1, 0) = 0.f; K(1, 1) = mvParameters[1]; K(1, 2) = mvParameters[3];
        K(2, 0) = 0.f; K(2, 1) = 0.f; K(2, 2) = 1.f;
        return K;
    }
    bool epipolarConstrain(GeometricCamera *pCamera2, const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const Eigen::Matrix3f &R12,
                           const Eigen::Vector3f &t12, const float sigmaLevel, const float unc);
};

/* stand-in for CameraModels/KannalaBrandt8 (host geometry of the caller's camera objects, outside the path): TriangulateMatches
 * (KannalaBrandt8.cpp:306) hands the two keypoints' ABSOLUTE indices and the level sigmas to a callback the test supplies */
typedef float (*frame_mock_triangulate_fn)(void *ctx, int i_left, int i_right, float sigma1, float sigma2, float *p3d);
class KannalaBrandt8 : public GeometricCamera {
public:
    Frame *frame = nullptr;
    frame_mock_triangulate_fn fn = nullptr;
    void *ctx = nullptr;
    Eigen::Matrix3f toK_() override { return Eigen::Matrix3f(); }
    float TriangulateMatches(GeometricCamera *pCamera2, const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const Eigen::Matrix3f &R12,
                             const Eigen::Vector3f &t12, const float sigmaLevel, const float unc, Eigen::Vector3f &p3D) {
        float p[3] = {0.f, 0.f, 0.f};
        const float d = fn(ctx, (int)(&kp1 - frame->mvKeys.data()), (int)(&kp2 - frame->mvKeysRight.data()), sigmaLevel, unc, p);
        p3D(0) = p[0]; p3D(1) = p[1]; p3D(2) = p[2];
        return d;
    }
};

}  // namespace ORB_SLAM3
