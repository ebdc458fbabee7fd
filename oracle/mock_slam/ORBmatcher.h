/* TEST INFRASTRUCTURE ONLY.  Found ahead of the reference's include/ORBmatcher.h on the include path when the reference's
 * src/ORBmatcher.cc is compiled for oracle/_ref/libmatcher_ref.so.  It declares the class that file defines (the member
 * definitions there must find matching declarations here) over the stand-in Frame / KeyFrame / MapPoint types of
 * slam_mock.h; everything is public because the test shim also calls the two helper members directly. */
#pragma once
#include "slam_mock.h"
using std::pair;
using std::vector;
namespace ORB_SLAM3 {
struct ORBmatcher {
    typedef std::vector<MapPoint *> MPs;
    typedef std::vector<KeyFrame *> KFs;
    typedef Sophus::Sim3<float> S3;

    float mfNNratio;
    bool mbCheckOrientation;
    static const int TH_LOW, TH_HIGH, HISTO_LENGTH;

    ORBmatcher(float nnratio = 0.6, bool checkOri = true);
    static int DescriptorDistance(const cv::Mat &, const cv::Mat &);
    float RadiusByViewingCos(const float &);
    void ComputeThreeMaxima(std::vector<int> *, const int, int &, int &, int &);

    /* five projection searches: tracked map points, last frame, relocalisation key frame, two Sim3 forms */
    int SearchByProjection(Frame &, const MPs &, const float = 3, const bool = false, const float = 50.0f);
    int SearchByProjection(Frame &, const Frame &, const float, const bool);
    int SearchByProjection(Frame &, KeyFrame *, const std::set<MapPoint *> &, const float, const int);
    int SearchByProjection(KeyFrame *, S3 &, const MPs &, MPs &, int, float = 1.0);
    int SearchByProjection(KeyFrame *, S3 &, const MPs &, const KFs &, MPs &, KFs &, int, float = 1.0);
    /* vocabulary-guided searches */
    int SearchByBoW(KeyFrame *, Frame &, MPs &);
    int SearchByBoW(KeyFrame *, KeyFrame *, MPs &);
    int SearchForTriangulation(KeyFrame *, KeyFrame *, std::vector<pair<size_t, size_t>> &, const bool, const bool = false);
    /* initialisation, Sim3 agreement, fusion */
    int SearchForInitialization(Frame &, Frame &, std::vector<cv::Point2f> &, std::vector<int> &, int = 10);
    int SearchBySim3(KeyFrame *, KeyFrame *, MPs &, const S3 &, const float);
    int Fuse(KeyFrame *, const MPs &, const float = 3.0, const bool = false);
    int Fuse(KeyFrame *, S3 &, const MPs &, float, MPs &);
};
}  // namespace ORB_SLAM3
