// ORBmatcher.h -- adapter with the surface of /root/reference/include/ORBmatcher.h:36-103 over the C ABI.
//
// The reference matchers take Frame / KeyFrame / MapPoint pointer graphs guarded by mutexes.  The adapter's job
// (see INTEGRATION.md for the Frame-typed overloads a maintainer adds inside the ORB-SLAM3 tree) is to flatten what
// the loops read under the reference's locks, call the kernels, and write the assignments back in reference order.
// This header holds the dependency-free core: same constants, same constructor, DescriptorDistance, and the
// flattened forms of the two SearchByProjection variants and of the stereo matchers.
#ifndef ORBX_ADAPTER_ORBMATCHER_H
#define ORBX_ADAPTER_ORBMATCHER_H

#include <cstdint>
#include <cstring>
#include <set>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <array>
#include <vector>

#include "../../include/orbx.h"

namespace ORB_SLAM3 {

// What the projection matchers read of a Frame (Frame.h: mvKeysUn, mDescriptors, mnMinX.., mvScaleFactors, mvuRight)
struct FrameView {
    const orbx_keypoint *mvKeysUn = nullptr;
    const uint8_t *mDescriptors = nullptr;  // N x 32
    int N = 0;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
    const float *mvScaleFactors = nullptr;
    int nlevels = 0;
    const float *mvuRight = nullptr;  // NULL for monocular
    orbx_frame_desc c() const { return orbx_frame_desc{mvKeysUn, mDescriptors, N, mnMinX, mnMaxX, mnMinY, mnMaxY, mvScaleFactors, nlevels, mvuRight}; }
};

// The MapPoint scratch fields SearchByProjection reads (MapPoint.h:171-179), one entry per map point
struct MapPointBatch {
    std::vector<float> mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos;
    std::vector<int32_t> mnTrackScaleLevel;
    std::vector<uint8_t> descriptors;  // n x 32 (MapPoint::GetDescriptor)
    std::vector<uint8_t> inView;       // mbTrackInView && !isBad() && !(bFarPoints && mTrackDepth > thFarPoints)
    std::vector<uint8_t> hasObservations;  // Observations() > 0
    int size() const { return (int)mTrackProjX.size(); }
};

class ORBmatcher {
public:
    static const int TH_LOW = ORBX_TH_LOW;
    static const int TH_HIGH = ORBX_TH_HIGH;
    static const int HISTO_LENGTH = ORBX_HISTO_LENGTH;

    ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
        const int st = orbx_matcher_create(device, &m_);
        if (st != ORBX_OK) throw std::runtime_error(std::string("orbx_matcher_create: ") + orbx_status_string(st) + " " + orbx_last_error());
    }
    ~ORBmatcher() { orbx_matcher_destroy(m_); }
    ORBmatcher(const ORBmatcher &) = delete;
    ORBmatcher &operator=(const ORBmatcher &) = delete;

    // ORBmatcher::DescriptorDistance (ORBmatcher.cc:2058-2074); scalar host helper for the adapter's own glue
    static int DescriptorDistance(const uint8_t *a, const uint8_t *b) {
        int dist = 0;
        for (int i = 0; i < 4; i++) {
            uint64_t x, y;
            __builtin_memcpy(&x, a + 8 * i, 8);
            __builtin_memcpy(&y, b + 8 * i, 8);
            dist += __builtin_popcountll(x ^ y);
        }
        return dist;
    }

    // SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, bFarPoints, thFarPoints)
    // (ORBmatcher.cc:43-213, Nleft == -1).  vpMatch[i] = index into `mps` assigned to feature i, or -1.
    int SearchByProjection(const FrameView &F, const std::vector<uint8_t> &occupied, const MapPointBatch &mps, float th,
                           std::vector<int32_t> &vpMatch) {
        vpMatch.assign(F.N, -1);
        orbx_frame_desc fd = F.c();
        const int r = orbx_search_by_projection_mappoints(
            m_, &fd, occupied.empty() ? nullptr : occupied.data(), mps.size(), mps.mTrackProjX.data(), mps.mTrackProjY.data(),
            mps.mTrackProjXR.empty() ? nullptr : mps.mTrackProjXR.data(), mps.mnTrackScaleLevel.data(), mps.mTrackViewCos.data(),
            mps.descriptors.data(), mps.inView.empty() ? nullptr : mps.inView.data(),
            mps.hasObservations.empty() ? nullptr : mps.hasObservations.data(), th, mfNNratio, vpMatch.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_search_by_projection_mappoints: ") + orbx_status_string(r));
        return r;
    }

    // SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (ORBmatcher.cc:1676-1887) after the
    // adapter projected LastFrame's map points with CurrentFrame's pose (host float math, unchanged).
    struct ProjectedQueries {
        std::vector<float> u, v, ur, angle;
        std::vector<int32_t> octave;
        std::vector<uint8_t> descriptors, hasObservations;
    };
    int SearchByProjection(const FrameView &Cur, const std::vector<uint8_t> &occupied, const ProjectedQueries &q, float th,
                           bool bForward, bool bBackward, std::vector<int32_t> &vpMatch) {
        vpMatch.assign(Cur.N, -1);
        orbx_frame_desc fd = Cur.c();
        const int mode = bForward ? 1 : (bBackward ? 2 : 0);
        const int r = orbx_search_by_projection_frame(
            m_, &fd, occupied.empty() ? nullptr : occupied.data(), (int)q.u.size(), q.u.data(), q.v.data(),
            q.ur.empty() ? nullptr : q.ur.data(), q.octave.data(), q.angle.data(), q.descriptors.data(),
            q.hasObservations.empty() ? nullptr : q.hasObservations.data(), th, mode, mbCheckOrientation ? 1 : 0, vpMatch.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_search_by_projection_frame: ") + orbx_status_string(r));
        return r;   // vpMatch[i]: query index, -1 = untouched, -2 = assigned then cleared by the rotation check (slot becomes NULL)
    }

    // SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)
    // (ORBmatcher.cc:1889-2010) and SearchByProjection(KeyFrame*, Sim3f&, vpPoints[, vpPointsKFs], vpMatched[, vpMatchedKF],
    // th, ratioHamming) (ORBmatcher.cc:427-646) share this window form; the adapter projects and fills `q`.
    struct WindowQueries {
        std::vector<float> x, y, r, angle;
        std::vector<int32_t> minLevel, maxLevel;
        std::vector<uint8_t> descriptors;
    };
    int SearchByProjectionWindow(const FrameView &F, const std::vector<uint8_t> &occupied, const WindowQueries &q, float maxDist,
                                 bool checkOrientation, std::vector<int32_t> &vpMatch) {
        vpMatch.assign(F.N, -1);
        orbx_frame_desc fd = F.c();
        const int r = orbx_search_by_projection_window(m_, &fd, occupied.empty() ? nullptr : occupied.data(), (int)q.x.size(), q.x.data(),
                                                       q.y.data(), q.r.data(), q.minLevel.data(), q.maxLevel.data(),
                                                       q.angle.empty() ? nullptr : q.angle.data(), q.descriptors.data(), nullptr, maxDist,
                                                       checkOrientation ? 1 : 0, vpMatch.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_search_by_projection_window: ") + orbx_status_string(r));
        return r;
    }

    // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:648-763)
    int SearchForInitialization(const orbx_keypoint *mvKeysUn1, const uint8_t *mDescriptors1, int N1, const FrameView &F2,
                                std::vector<float> &vbPrevMatchedXY, std::vector<int> &vnMatches12, int windowSize = 10) {
        vnMatches12.assign(N1, -1);
        orbx_frame_desc fd = F2.c();
        const int r = orbx_search_for_initialization(m_, mvKeysUn1, mDescriptors1, N1, &fd, vbPrevMatchedXY.data(), windowSize, mfNNratio,
                                                     mbCheckOrientation ? 1 : 0, vnMatches12.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_search_for_initialization: ") + orbx_status_string(r));
        return r;
    }

    // DBoW2::FeatureVector flattened by the adapter (iterate the std::map in order)
    struct FeatVec {
        std::vector<uint32_t> node_id;
        std::vector<int32_t> node_ptr{0}, index;
        template <class Map> static FeatVec from(const Map &fv) {  // Map = DBoW2::FeatureVector
            FeatVec f;
            for (const auto &kv : fv) {
                f.node_id.push_back((uint32_t)kv.first);
                for (unsigned int i : kv.second) f.index.push_back((int32_t)i);
                f.node_ptr.push_back((int32_t)f.index.size());
            }
            return f;
        }
        orbx_featvec c() const { return orbx_featvec{node_id.data(), node_ptr.data(), index.data(), (int32_t)node_id.size()}; }
    };

    // SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (ORBmatcher.cc:223-425): vpMatch[iF] = KF feature index or -1
    int SearchByBoW(const uint8_t *descKF, const float *angleKF, const uint8_t *validKF, int nKF, const FeatVec &fvKF,
                    const uint8_t *descF, const float *angleF, int nF, const FeatVec &fvF, std::vector<int32_t> &vpMatch) {
        vpMatch.assign(nF, -1);
        orbx_featvec a = fvKF.c(), b = fvF.c();
        const int r = orbx_search_by_bow_frame(m_, descKF, angleKF, validKF, nKF, &a, descF, angleF, nF, &b, mfNNratio,
                                               mbCheckOrientation ? 1 : 0, vpMatch.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_search_by_bow_frame: ") + orbx_status_string(r));
        return r;
    }
    // SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (ORBmatcher.cc:765-905): vpMatches12[i1] = i2 or -1
    int SearchByBoW(const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1, const FeatVec &fv1, const uint8_t *desc2,
                    const float *angle2, const uint8_t *valid2, int n2, const FeatVec &fv2, std::vector<int32_t> &vpMatches12) {
        vpMatches12.assign(n1, -1);
        orbx_featvec a = fv1.c(), b = fv2.c();
        const int r = orbx_search_by_bow_keyframes(m_, desc1, angle1, valid1, n1, &a, desc2, angle2, valid2, n2, &b, mfNNratio,
                                                   mbCheckOrientation ? 1 : 0, vpMatches12.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_search_by_bow_keyframes: ") + orbx_status_string(r));
        return r;
    }
    // SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (ORBmatcher.cc:907-1146).  `gate` is a callable
    // bool(size_t idx1, size_t idx2) holding the reference's epipole-distance test and pCamera1->epipolarConstrain(...) (or
    // `return true` for bCoarse) -- unchanged host float math.
    template <class Gate>
    int SearchForTriangulation(const uint8_t *desc1, const float *angle1, const uint8_t *skip1, int n1, const FeatVec &fv1,
                               const uint8_t *desc2, const float *angle2, const uint8_t *skip2, int n2, const FeatVec &fv2, Gate gate,
                               std::vector<std::pair<size_t, size_t>> &vMatchedPairs) {
        std::vector<int32_t> m12(n1, -1);
        orbx_featvec a = fv1.c(), b = fv2.c();
        auto thunk = [](void *user, int i1, int i2) -> int { return (*static_cast<Gate *>(user))((size_t)i1, (size_t)i2) ? 1 : 0; };
        const int r = orbx_search_for_triangulation(m_, desc1, angle1, skip1, n1, &a, desc2, angle2, skip2, n2, &b,
                                                    mbCheckOrientation ? 1 : 0, thunk, &gate, m12.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_search_for_triangulation: ") + orbx_status_string(r));
        vMatchedPairs.clear();
        for (int i = 0; i < n1; i++) if (m12[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m12[i]);  // :1138-1143
        return r;
    }

    // The same for pinhole key frames with both gates on the device (no callback): `gate` carries mvKeysUn / mvuRight of both key frames,
    // pKF2's mvScaleFactors / mvLevelSigma2, the epipole (:919-921) and F12 = K1^-T [t12]x R12 K2^-1 computed by the caller's Eigen as
    // Pinhole::epipolarConstrain (CameraModels/Pinhole.cpp:107-112) does; bCoarse = gate.coarse.
    int SearchForTriangulation(const uint8_t *desc1, const uint8_t *skip1, int n1, const FeatVec &fv1, const uint8_t *desc2,
                               const uint8_t *skip2, int n2, const FeatVec &fv2, const orbx_pinhole_gate &gate,
                               std::vector<std::pair<size_t, size_t>> &vMatchedPairs) {
        std::vector<int32_t> m12(n1, -1);
        orbx_featvec a = fv1.c(), b = fv2.c();
        const int r = orbx_search_for_triangulation_pinhole(m_, desc1, skip1, n1, &a, desc2, skip2, n2, &b, mbCheckOrientation ? 1 : 0, &gate,
                                                            m12.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_search_for_triangulation_pinhole: ") + orbx_status_string(r));
        vMatchedPairs.clear();
        for (int i = 0; i < n1; i++) if (m12[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m12[i]);
        return r;
    }

    // Matching core of Fuse(KeyFrame*, vpMapPoints, th, bRight) (ORBmatcher.cc:1148-1337) and Fuse(KeyFrame*, Sim3f&, vpPoints, th,
    // vpReplacePoint) (:1339-1455).  The caller projects the map points exactly as :1186-1244 / :1376-1403 do (host float math), passes
    // the survivors, and afterwards runs the reference's own tail on (bestIdx[i], bestDist[i] <= TH_LOW): Replace / AddObservation /
    // AddMapPoint (:1309-1330) or vpReplacePoint[iMP] = pMPinKF (:1436-1449).  mvInvLevelSigma2 = nullptr selects the Sim3 overload
    // (no chi2 gate).
    struct FuseQueries {
        std::vector<float> u, v, ur, radius;   // radius = th * pKF->mvScaleFactors[nPredictedLevel]
        std::vector<int32_t> nPredictedLevel;
        std::vector<uint8_t> descriptors;      // 32 B each (pMP->GetDescriptor())
    };
    void FuseSearch(const FrameView &KF, const float *mvInvLevelSigma2, const FuseQueries &q, std::vector<int32_t> &bestIdx,
                    std::vector<int32_t> &bestDist, bool strictFloat = false) {
        const int nq = (int)q.u.size();
        bestIdx.assign(nq, -1); bestDist.assign(nq, 256);
        orbx_frame_desc fd = KF.c();
        const int r = orbx_fuse_search(m_, &fd, mvInvLevelSigma2, nq, q.u.data(), q.v.data(), q.ur.empty() ? nullptr : q.ur.data(),
                                       q.radius.data(), q.nPredictedLevel.data(), q.descriptors.data(), strictFloat ? 1 : 0, bestIdx.data(),
                                       bestDist.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_fuse_search: ") + orbx_status_string(r));
    }

    // SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (ORBmatcher.cc:1457-1674): the two projection searches are gate-less fuse searches
    // (KeyFrame::GetFeaturesInArea, octave gate [l-1,l], first minimum wins, accept bestDist <= TH_HIGH :1569/:1656), followed by the
    // mutual-agreement pass (:1662-1675).  q1 = KF1's map points transformed by S21 and projected into KF2 (one entry per KF1 feature,
    // radius = th * pKF2->mvScaleFactors[level]); q2 = the converse.  use1[i] / use2[i] == 0 drops a slot (no / bad map point,
    // vbAlreadyMatched, failed depth / image / distance gate :1495-1527).  vnMatch12[i1] = KF2 feature index or -1; returns nFound.
    int SearchBySim3(const FrameView &KF1, const FrameView &KF2, const FuseQueries &q1, const std::vector<uint8_t> &use1,
                     const FuseQueries &q2, const std::vector<uint8_t> &use2, std::vector<int32_t> &vnMatch12) {
        std::vector<int32_t> bi1, bd1, bi2, bd2;
        FuseSearch(KF2, nullptr, q1, bi1, bd1);
        FuseSearch(KF1, nullptr, q2, bi2, bd2);
        const int n1 = (int)bi1.size(), n2 = (int)bi2.size();
        vnMatch12.assign(n1, -1);
        int nFound = 0;
        for (int i1 = 0; i1 < n1; i1++) {
            if (!use1[i1] || bd1[i1] > TH_HIGH || bi1[i1] < 0) continue;
            const int idx2 = bi1[i1];
            if (idx2 >= n2 || !use2[idx2] || bd2[idx2] > TH_HIGH) continue;
            if (bi2[idx2] == i1) { vnMatch12[i1] = idx2; nFound++; }
        }
        return nFound;
    }

    // MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:329-403) for a batch of map points: the observations' descriptors of map
    // point p are rows [setPtr[p], setPtr[p+1]) of `descriptors`; bestIdx[p] is the row offset (within the set) the reference keeps.
    void ComputeDistinctiveDescriptors(const std::vector<uint8_t> &descriptors, const std::vector<int32_t> &setPtr, std::vector<int32_t> &bestIdx) {
        const int n_sets = (int)setPtr.size() - 1;
        bestIdx.assign(n_sets > 0 ? n_sets : 0, -1);
        if (n_sets <= 0) return;
        const int r = orbx_distinctive_descriptors(m_, descriptors.data(), setPtr.data(), n_sets, bestIdx.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_distinctive_descriptors: ") + orbx_status_string(r));
    }

    // Frame::ComputeStereoFishEyeMatches (Frame.cc:1126-1166) on host vectors: the brute-force kNN-2 of the two lapping-area descriptor
    // tails (BFmatcher.knnMatch, :1144) on the device, Lowe's ratio (:1151) and the bookkeeping (:1157-1162) here.  `triangulate(iLeft,
    // iRight, sigma1, sigma2, p3D) -> depth` is the caller's own KannalaBrandt8::TriangulateMatches (host geometry of the camera objects,
    // not part of the path) on mvKeys[iLeft] / mvKeysRight[iRight]; p3D is a float[3].  Returns nMatches; mvStereo3Dpoints[i] is written
    // for accepted matches only, as in the reference.  The body that replaces the reference's member is in INTEGRATION.md.
    template <class Triangulate>
    int ComputeStereoFishEyeMatches(const orbx_keypoint *mvKeys, const uint8_t *mDescriptors, int Nleft, int monoLeft,
                                    const orbx_keypoint *mvKeysRight, const uint8_t *mDescriptorsRight, int Nright, int monoRight,
                                    const float *mvLevelSigma2, Triangulate &&triangulate, std::vector<int> &mvLeftToRightMatch,
                                    std::vector<int> &mvRightToLeftMatch, std::vector<float> &mvDepth, std::vector<float> &mvuRight,
                                    std::vector<std::array<float, 3>> &mvStereo3Dpoints, int *descMatches = nullptr) {
        mvLeftToRightMatch.assign(Nleft > 0 ? Nleft : 0, -1);
        mvRightToLeftMatch.assign(Nright > 0 ? Nright : 0, -1);
        mvDepth.assign(Nleft > 0 ? Nleft : 0, -1.0f);
        mvuRight.assign(Nleft > 0 ? Nleft : 0, -1.0f);
        mvStereo3Dpoints.assign(Nleft > 0 ? Nleft : 0, std::array<float, 3>{0.f, 0.f, 0.f});
        const int nq = Nleft - monoLeft, nt = Nright - monoRight;
        int nMatches = 0, nDesc = 0;
        if (nq > 0) {
            std::vector<int32_t> idx(2 * (size_t)nq), dist(2 * (size_t)nq);
            const int r = orbx_knn2(m_, mDescriptors + (size_t)monoLeft * 32, nq, mDescriptorsRight + (size_t)monoRight * 32, nt > 0 ? nt : 0,
                                    idx.data(), dist.data());
            if (r < 0) throw std::runtime_error(std::string("orbx_knn2: ") + orbx_status_string(r));
            for (int q = 0; q < nq; q++) {
                if (idx[2 * q + 1] < 0) continue;                                     // fewer than two neighbours
                if (!((float)dist[2 * q] < (float)dist[2 * q + 1] * 0.7)) continue;     // :1151 (float against double)
                nDesc++;
                const int iL = q + monoLeft, iR = idx[2 * q] + monoRight;
                float p3D[3] = {0.f, 0.f, 0.f};
                const float depth = triangulate(iL, iR, mvLevelSigma2[mvKeys[iL].octave], mvLevelSigma2[mvKeysRight[iR].octave], p3D);
                if (depth > 0.0001f) {
                    mvLeftToRightMatch[iL] = iR;
                    mvRightToLeftMatch[iR] = iL;
                    mvStereo3Dpoints[iL] = {p3D[0], p3D[1], p3D[2]};
                    mvDepth[iL] = depth;
                    nMatches++;
                }
            }
        }
        if (descMatches) *descMatches = nDesc;
        return nMatches;
    }

    // Frame::ComputeStereoMatches (Frame.cc:811-981) on host vectors: mvuRight / mvDepth out.  pyrLeft/pyrRight[l] are the level ROI
    // origins of the two extractors' mvImagePyramid (host copies, e.g. ORBextractor::GetPyramidLevel).  For the device-resident
    // batched form (no pyramid transfer) use orbx_stereo_batch_device / orbx_stereo_batch_download on the two extractors.
    int ComputeStereoMatches(const orbx_keypoint *mvKeys, const uint8_t *mDescriptors, int N, const orbx_keypoint *mvKeysRight,
                             const uint8_t *mDescriptorsRight, int Nr, const float *mvScaleFactors, const float *mvInvScaleFactors, int nlevels,
                             const uint8_t *const *pyrLeft, const uint8_t *const *pyrRight, const int32_t *pyrW, const int32_t *pyrH,
                             const size_t *pyrStride, float mbf, float mb, std::vector<float> &mvuRight, std::vector<float> &mvDepth) {
        mvuRight.assign(N, -1.0f); mvDepth.assign(N, -1.0f);
        const int r = orbx_compute_stereo_matches(m_, mvKeys, mDescriptors, N, mvKeysRight, mDescriptorsRight, Nr, mvScaleFactors, mvInvScaleFactors,
                                                  nlevels, pyrLeft, pyrRight, pyrW, pyrH, pyrStride, mbf, mb, mvuRight.data(), mvDepth.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_compute_stereo_matches: ") + orbx_status_string(r));
        return r;
    }

#ifdef ORBX_WITH_SLAM_TYPES
    // the reference's own signatures (Frame / KeyFrame / MapPoint graphs): see ORBmatcher_slam.inl
#include "ORBmatcher_slam.inl"
#endif

    orbx_matcher *handle() { return m_; }

protected:
    float mfNNratio;
    bool mbCheckOrientation;
    orbx_matcher *m_ = nullptr;
};

// ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) flattened for the device: the k-ary tree as CSR children, one
// 32-byte descriptor and one word id (-1 for inner nodes) per node.  transform() is the tree descent of
// TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1151-1193)
// as called from Frame::ComputeBoW (Frame.cc:462-470) / KeyFrame::ComputeBoW: per feature the word id and the node id `levelsup`
// levels above the leaf; the caller folds them into BowVector (addWeight) and FeatureVector (addFeature) in feature order.
class ORBVocabularyDevice {
public:
    ORBVocabularyDevice(int L, const std::vector<int32_t> &childPtr, const std::vector<int32_t> &childIdx, const std::vector<uint8_t> &nodeDesc,
                        const std::vector<int32_t> &wordId, int device = 0) {
        const int st = orbx_vocabulary_create(device, L, (int)wordId.size(), childPtr.data(), childIdx.data(), nodeDesc.data(), wordId.data(), &v_);
        if (st != ORBX_OK) throw std::runtime_error(std::string("orbx_vocabulary_create: ") + orbx_status_string(st));
    }
    ~ORBVocabularyDevice() { orbx_vocabulary_destroy(v_); }
    ORBVocabularyDevice(const ORBVocabularyDevice &) = delete;
    ORBVocabularyDevice &operator=(const ORBVocabularyDevice &) = delete;
    void transform(ORBmatcher &m, const uint8_t *descriptors, int n, int levelsup, std::vector<int32_t> &wordId, std::vector<int32_t> &nodeId) const {
        wordId.assign(n, -1); nodeId.assign(n, -1);
        if (n == 0) return;
        const int r = orbx_bow_transform(m.handle(), v_, descriptors, n, levelsup, wordId.data(), nodeId.data());
        if (r < 0) throw std::runtime_error(std::string("orbx_bow_transform: ") + orbx_status_string(r));
    }

private:
    orbx_vocabulary *v_ = nullptr;
};

}  // namespace ORB_SLAM3

#endif
