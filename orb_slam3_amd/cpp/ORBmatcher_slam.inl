// ORBmatcher_slam.inl -- the REFERENCE signatures of ORBmatcher (/root/reference/include/ORBmatcher.h:47-87) over the C ABI.
//
// Included inside class ORB_SLAM3::ORBmatcher (ORBmatcher.h) when ORBX_WITH_SLAM_TYPES is defined, i.e. inside a tree that has
// Frame / KeyFrame / MapPoint (ORB-SLAM3 itself, or the stand-in types of oracle/mock_slam in this repo's tests).  Each overload
// does what the reference's member does around its candidate loop -- the same accessor calls, pose transforms, projection gates
// and write-backs, in the same order -- and hands the candidate search / Hamming / best-second / taken-mask / rotation-histogram
// part to the device.  Fisheye-stereo frames (Nleft != -1) take the twin entry points of the C ABI where the reference has a twin.
//
// Requires before inclusion: Frame.h, KeyFrame.h, MapPoint.h (and with them cv::Mat, cv::KeyPoint, Sophus::SE3f, Eigen).

private:
    template <class Holder> static FrameView view_of(const Holder &h, const std::vector<cv::KeyPoint> &keysUn, bool withRight) {
        static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint must be the 28-byte OpenCV layout");
        FrameView v;
        v.mvKeysUn = reinterpret_cast<const orbx_keypoint *>(keysUn.data());
        v.mDescriptors = h.mDescriptors.data;
        v.N = (int)keysUn.size();
        v.mnMinX = h.mnMinX; v.mnMaxX = h.mnMaxX; v.mnMinY = h.mnMinY; v.mnMaxY = h.mnMaxY;
        v.mvScaleFactors = h.mvScaleFactors.data();
        v.nlevels = (int)h.mvScaleFactors.size();
        v.mvuRight = (withRight && !h.mvuRight.empty()) ? h.mvuRight.data() : nullptr;
        return v;
    }
    static void push_desc(std::vector<uint8_t> &dst, const cv::Mat &d) { dst.insert(dst.end(), d.data, d.data + 32); }

public:
    // ORBmatcher.cc:43-213 (Tracking::SearchLocalPoints, Tracking.cc:3390-3413)
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3, const bool bFarPoints = false,
                           const float thFarPoints = 50.0f) {
        if (F.Nleft != -1) return SearchByProjectionFisheye(F, vpMapPoints, th, bFarPoints, thFarPoints);
        MapPointBatch b;
        const size_t n = vpMapPoints.size();
        b.mTrackProjX.reserve(n); b.descriptors.reserve(32 * n);
        for (MapPoint *pMP : vpMapPoints) {
            // :52-59 -- the three skip rules, evaluated per point as the loop does
            const bool in = pMP->mbTrackInView && !(bFarPoints && pMP->mTrackDepth > thFarPoints) && !pMP->isBad();
            b.inView.push_back(in ? 1 : 0);
            b.mTrackProjX.push_back(pMP->mTrackProjX); b.mTrackProjY.push_back(pMP->mTrackProjY);
            b.mTrackProjXR.push_back(pMP->mTrackProjXR); b.mTrackViewCos.push_back(pMP->mTrackViewCos);
            b.mnTrackScaleLevel.push_back(pMP->mnTrackScaleLevel);
            b.hasObservations.push_back(pMP->Observations() > 0 ? 1 : 0);
            if (in) push_desc(b.descriptors, pMP->GetDescriptor());   // copy under the MapPoint's own mutex
            else b.descriptors.insert(b.descriptors.end(), 32, 0);
        }
        std::vector<uint8_t> occupied(F.N);
        for (int i = 0; i < F.N; i++) occupied[i] = (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;   // :88-90
        std::vector<int32_t> match;
        const int nmatches = SearchByProjection(view_of(F, F.mvKeysUn, true), occupied, b, th, match);
        for (int i = 0; i < F.N; i++)
            if (match[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[match[i]];   // :129
        return nmatches;
    }

    // the same member for a fisheye-stereo frame (F.Nleft != -1): left search + right-camera twin per map point, ORBmatcher.cc:43-213 whole
    int SearchByProjectionFisheye(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th, const bool bFarPoints, const float thFarPoints) {
        const size_t n = vpMapPoints.size();
        std::vector<uint8_t> inL(n), inR(n), hasObs(n), desc(32 * n, 0);
        std::vector<float> px(n), py(n), vc(n), pxr(n), pyr(n), vcr(n);
        std::vector<int32_t> lv(n), lvr(n);
        for (size_t k = 0; k < n; k++) {
            MapPoint *pMP = vpMapPoints[k];
            const bool ok = (pMP->mbTrackInView || pMP->mbTrackInViewR) && !(bFarPoints && pMP->mTrackDepth > thFarPoints) && !pMP->isBad();   // :52-59
            inL[k] = (ok && pMP->mbTrackInView) ? 1 : 0;
            inR[k] = (ok && pMP->mbTrackInViewR) ? 1 : 0;
            px[k] = pMP->mTrackProjX; py[k] = pMP->mTrackProjY; lv[k] = pMP->mnTrackScaleLevel; vc[k] = pMP->mTrackViewCos;
            pxr[k] = pMP->mTrackProjXR; pyr[k] = pMP->mTrackProjYR; lvr[k] = pMP->mnTrackScaleLevelR; vcr[k] = pMP->mTrackViewCosR;
            hasObs[k] = pMP->Observations() > 0 ? 1 : 0;
            if (ok) { const cv::Mat d = pMP->GetDescriptor(); std::memcpy(&desc[32 * k], d.data, 32); }
        }
        std::vector<uint8_t> occupied(F.N);
        for (int i = 0; i < F.N; i++) occupied[i] = (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;
        std::vector<int32_t> match(F.N, -1);
        FrameView fv = view_of(F, F.mvKeys, false);
        fv.N = F.Nleft;
        orbx_frame_desc fd = fv.c();
        const int nmatches = orbx_search_by_projection_mappoints_fisheye(
            m_, &fd, reinterpret_cast<const orbx_keypoint *>(F.mvKeysRight.data()), F.N - F.Nleft, F.mvLeftToRightMatch.data(), F.mvRightToLeftMatch.data(),
            occupied.data(), (int)n, inL.data(), px.data(), py.data(), lv.data(), vc.data(), inR.data(), pxr.data(), pyr.data(), lvr.data(), vcr.data(),
            desc.data(), hasObs.data(), th, mfNNratio, match.data());
        if (nmatches < 0) throw std::runtime_error(std::string("orbx_search_by_projection_mappoints_fisheye: ") + orbx_status_string(nmatches));
        for (int i = 0; i < F.N; i++)
            if (match[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[match[i]];
        return nmatches;
    }

    // ORBmatcher.cc:1676-1887 (Tracking::TrackWithMotionModel)
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) {
        const bool fisheye = CurrentFrame.Nleft != -1;
        const Sophus::SE3f Tcw = CurrentFrame.GetPose();
        const Eigen::Vector3f twc = Tcw.inverse().translation();
        const Sophus::SE3f Tlw = LastFrame.GetPose();
        const Eigen::Vector3f tlc = Tlw * twc;
        const bool bForward = tlc(2) > CurrentFrame.mb && !bMono;    // :1692-1693
        const bool bBackward = -tlc(2) > CurrentFrame.mb && !bMono;
        ProjectedQueries q;
        std::vector<float> vr;   // fisheye-stereo: v of the projection into the right camera (q.ur then holds its u)
        std::vector<MapPoint *> live;
        for (int i = 0; i < LastFrame.N; i++) {
            MapPoint *pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;
            Eigen::Vector3f x3Dw = pMP->GetWorldPos();
            Eigen::Vector3f x3Dc = Tcw * x3Dw;
            const float invzc = 1.0 / x3Dc(2);
            if (invzc < 0) continue;
            Eigen::Vector2f uv = CurrentFrame.mpCamera->project(x3Dc);
            if (uv(0) < CurrentFrame.mnMinX || uv(0) > CurrentFrame.mnMaxX) continue;
            if (uv(1) < CurrentFrame.mnMinY || uv(1) > CurrentFrame.mnMaxY) continue;
            q.u.push_back(uv(0)); q.v.push_back(uv(1));
            const bool lastLeft = LastFrame.Nleft == -1 || i < LastFrame.Nleft;
            q.octave.push_back(lastLeft ? LastFrame.mvKeys[i].octave : LastFrame.mvKeysRight[i - LastFrame.Nleft].octave);   // :1719-1720
            q.angle.push_back(LastFrame.Nleft == -1 ? LastFrame.mvKeysUn[i].angle                                           // :1776-1778
                                                    : (lastLeft ? LastFrame.mvKeys[i].angle : LastFrame.mvKeysRight[i - LastFrame.Nleft].angle));
            if (!fisheye) q.ur.push_back(uv(0) - CurrentFrame.mbf * invzc);   // :1751 (stereo-coordinate gate of the rectified form)
            else {                                                            // :1795-1796: the point seen from the right camera
                Eigen::Vector3f x3Dr = CurrentFrame.GetRelativePoseTrl() * x3Dc;
                Eigen::Vector2f uvr = CurrentFrame.mpCamera->project(x3Dr);
                q.ur.push_back(uvr(0)); vr.push_back(uvr(1));
            }
            q.hasObservations.push_back(pMP->Observations() > 0 ? 1 : 0);
            push_desc(q.descriptors, pMP->GetDescriptor());
            live.push_back(pMP);
        }
        std::vector<uint8_t> occupied(CurrentFrame.N);
        for (int i = 0; i < CurrentFrame.N; i++)
            occupied[i] = (CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;   // :1744-1746
        std::vector<int32_t> match;
        int nmatches;
        if (!fisheye) nmatches = SearchByProjection(view_of(CurrentFrame, CurrentFrame.mvKeysUn, true), occupied, q, th, bForward, bBackward, match);
        else {
            match.assign(CurrentFrame.N, -1);
            FrameView fv = view_of(CurrentFrame, CurrentFrame.mvKeys, false);   // left camera: mvKeys; descriptors: all N rows
            fv.N = CurrentFrame.Nleft;
            orbx_frame_desc fd = fv.c();
            nmatches = orbx_search_by_projection_frame_fisheye(
                m_, &fd, reinterpret_cast<const orbx_keypoint *>(CurrentFrame.mvKeysRight.data()), CurrentFrame.N - CurrentFrame.Nleft, occupied.data(),
                (int)q.u.size(), q.u.data(), q.v.data(), q.ur.data(), vr.data(), q.octave.data(), q.angle.data(), q.descriptors.data(),
                q.hasObservations.data(), th, bForward ? 1 : (bBackward ? 2 : 0), mbCheckOrientation ? 1 : 0, match.data());
            if (nmatches < 0) throw std::runtime_error(std::string("orbx_search_by_projection_frame_fisheye: ") + orbx_status_string(nmatches));
        }
        for (int i = 0; i < CurrentFrame.N; i++) {
            if (match[i] >= 0) CurrentFrame.mvpMapPoints[i] = live[match[i]];                // :1771
            else if (match[i] == -2) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint *>(NULL);   // :1876
        }
        return nmatches;
    }

    // ORBmatcher.cc:1889-2010 (Tracking::Relocalization)
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist) {
        // a fisheye-stereo frame (Nleft != -1) takes the same path over its LEFT camera: the reference's GetFeaturesInArea call has the default
        // bRight = false, mvKeysUn == mvKeys there (Frame.cc:751); the right camera's features are neither candidates nor written
        const bool fisheye = CurrentFrame.Nleft != -1;
        const int NL = fisheye ? CurrentFrame.Nleft : CurrentFrame.N;
        const Sophus::SE3f Tcw = CurrentFrame.GetPose();
        Eigen::Vector3f Ow = Tcw.inverse().translation();
        const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
        WindowQueries q;
        std::vector<MapPoint *> live;
        for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
            MapPoint *pMP = vpMPs[i];
            if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
            Eigen::Vector3f x3Dw = pMP->GetWorldPos();
            Eigen::Vector3f x3Dc = Tcw * x3Dw;
            const Eigen::Vector2f uv = CurrentFrame.mpCamera->project(x3Dc);
            if (uv(0) < CurrentFrame.mnMinX || uv(0) > CurrentFrame.mnMaxX) continue;
            if (uv(1) < CurrentFrame.mnMinY || uv(1) > CurrentFrame.mnMaxY) continue;
            Eigen::Vector3f PO = x3Dw - Ow;
            float dist3D = PO.norm();
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
            q.x.push_back(uv(0)); q.y.push_back(uv(1));
            q.r.push_back(th * CurrentFrame.mvScaleFactors[nPredictedLevel]);        // :1940
            q.minLevel.push_back(nPredictedLevel - 1); q.maxLevel.push_back(nPredictedLevel + 1);
            q.angle.push_back(i < pKF->mvKeysUn.size() ? pKF->mvKeysUn[i].angle : 0.f);   // :1973 (the reference reads past mvKeysUn for a right-camera feature of a fisheye key frame)
            push_desc(q.descriptors, pMP->GetDescriptor());
            live.push_back(pMP);
        }
        std::vector<uint8_t> occupied(NL);
        for (int i = 0; i < NL; i++) occupied[i] = CurrentFrame.mvpMapPoints[i] ? 1 : 0;   // :1955: any map point blocks the slot
        std::vector<int32_t> match;
        FrameView fv = view_of(CurrentFrame, fisheye ? CurrentFrame.mvKeys : CurrentFrame.mvKeysUn, false);
        fv.N = NL;
        const int nmatches = SearchByProjectionWindow(fv, occupied, q, (float)ORBdist, mbCheckOrientation, match);
        for (int i = 0; i < NL; i++) {
            if (match[i] >= 0) CurrentFrame.mvpMapPoints[i] = live[match[i]];
            else if (match[i] == -2) CurrentFrame.mvpMapPoints[i] = NULL;           // :2001
        }
        return nmatches;
    }

    // ORBmatcher.cc:648-763 (Tracking::MonocularInitialization)
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10) {
        static_assert(sizeof(cv::Point2f) == 8, "cv::Point2f layout");
        const int N1 = (int)F1.mvKeysUn.size();
        std::vector<float> prev(2 * (size_t)N1);
        for (int i = 0; i < N1; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
        const int nmatches = SearchForInitialization(reinterpret_cast<const orbx_keypoint *>(F1.mvKeysUn.data()), F1.mDescriptors.data, N1,
                                                     view_of(F2, F2.mvKeysUn, false), prev, vnMatches12, windowSize);
        for (int i = 0; i < N1; i++) { vbPrevMatched[i].x = prev[2 * i]; vbPrevMatched[i].y = prev[2 * i + 1]; }   // :757-760
        return nmatches;
    }

    // ORBmatcher.cc:223-425 (Tracking::TrackReferenceKeyFrame, Relocalization)
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches) {
        const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));
        const int nKF = (int)vpMapPointsKF.size();
        std::vector<uint8_t> valid(nKF);
        std::vector<float> angKF(nKF), angF(F.N);
        for (int i = 0; i < nKF; i++) {
            MapPoint *pMP = vpMapPointsKF[i];
            valid[i] = (pMP && !pMP->isBad()) ? 1 : 0;                              // :252-256
            angKF[i] = pKF->mvKeysUn[i].angle;                                      // :335
        }
        std::vector<int32_t> match;
        int nmatches;
        if (F.Nleft == -1) {
            for (int i = 0; i < F.N; i++) angF[i] = F.mvKeysUn[i].angle;
            nmatches = SearchByBoW(pKF->mDescriptors.data, angKF.data(), valid.data(), nKF, FeatVec::from(pKF->mFeatVec), F.mDescriptors.data,
                                   angF.data(), F.N, FeatVec::from(F.mFeatVec), match);
        } else {   // :283-392: the frame's features split into the two cameras; key-frame / frame angles from mvKeys / mvKeysRight (:331-343)
            for (int i = 0; i < nKF; i++)
                angKF[i] = (!pKF->mpCamera2) ? pKF->mvKeysUn[i].angle : (i >= pKF->NLeft ? pKF->mvKeysRight[i - pKF->NLeft].angle : pKF->mvKeys[i].angle);
            for (int i = 0; i < F.N; i++)
                angF[i] = (i >= F.Nleft) ? F.mvKeysRight[i - F.Nleft].angle : F.mvKeys[i].angle;   // :337-340, :367-370 wherever they are well defined
            match.assign(F.N, -1);
            FeatVec fa = FeatVec::from(pKF->mFeatVec), fb = FeatVec::from(F.mFeatVec);
            orbx_featvec a = fa.c(), b = fb.c();
            nmatches = orbx_search_by_bow_frame_fisheye(m_, pKF->mDescriptors.data, angKF.data(), valid.data(), nKF, &a, F.mDescriptors.data, angF.data(),
                                                        F.N, F.Nleft, &b, mfNNratio, mbCheckOrientation ? 1 : 0, match.data());
            if (nmatches < 0) throw std::runtime_error(std::string("orbx_search_by_bow_frame_fisheye: ") + orbx_status_string(nmatches));
        }
        for (int i = 0; i < F.N; i++)
            if (match[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[match[i]];     // :329
        return nmatches;
    }

    // ORBmatcher.cc:765-905 (LoopClosing / place recognition)
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12) {
        const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
        const std::vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
        const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
        vpMatches12 = std::vector<MapPoint *>(n1, static_cast<MapPoint *>(NULL));
        std::vector<uint8_t> v1(n1), v2(n2);
        std::vector<float> a1(n1, 0.f), a2(n2, 0.f);
        // fisheye-stereo key frames (NLeft != -1): the features of the right camera (index >= mvKeysUn.size()) take no part, neither as
        // queries (:800-802) nor as candidates (:820-822) -- the same as a feature without a map point
        const int l1 = pKF1->NLeft != -1 ? std::min(n1, (int)pKF1->mvKeysUn.size()) : n1, l2 = pKF2->NLeft != -1 ? std::min(n2, (int)pKF2->mvKeysUn.size()) : n2;
        for (int i = 0; i < n1; i++) { MapPoint *p = vpMapPoints1[i]; v1[i] = (i < l1 && p && !p->isBad()) ? 1 : 0; if (i < l1) a1[i] = pKF1->mvKeysUn[i].angle; }
        for (int i = 0; i < n2; i++) { MapPoint *p = vpMapPoints2[i]; v2[i] = (i < l2 && p && !p->isBad()) ? 1 : 0; if (i < l2) a2[i] = pKF2->mvKeysUn[i].angle; }
        std::vector<int32_t> m12;
        const int nmatches = SearchByBoW(pKF1->mDescriptors.data, a1.data(), v1.data(), n1, FeatVec::from(pKF1->mFeatVec), pKF2->mDescriptors.data,
                                         a2.data(), v2.data(), n2, FeatVec::from(pKF2->mFeatVec), m12);
        for (int i = 0; i < n1; i++)
            if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];                 // :861
        return nmatches;
    }

    // ORBmatcher.cc:2058-2074 on the reference's argument type
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return DescriptorDistance((const uint8_t *)a.data, (const uint8_t *)b.data); }

    // ORBmatcher.cc:2012-2053 (kept for callers that build their own histograms)
    void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) {
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < L; i++) {
            const int s = (int)histo[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    }

private:
    // the projection gates shared by the two Sim3 SearchByProjection overloads (:452-487 / :569-604); proj(p3Dc) -> (u, v)
    template <class Proj>
    int sim3_projection(KeyFrame *pKF, Sophus::Sim3f &Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th,
                        float ratioHamming, Proj proj, std::vector<int32_t> &match, std::vector<int> &live) {
        Sophus::SE3f Tcw = Sophus::SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale());
        Eigen::Vector3f Ow = Tcw.inverse().translation();
        std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
        spAlreadyFound.erase(static_cast<MapPoint *>(NULL));
        WindowQueries q;
        live.clear();
        for (int iMP = 0, iendMP = (int)vpPoints.size(); iMP < iendMP; iMP++) {
            MapPoint *pMP = vpPoints[iMP];
            if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
            Eigen::Vector3f p3Dw = pMP->GetWorldPos();
            Eigen::Vector3f p3Dc = Tcw * p3Dw;
            if (p3Dc(2) < 0.0) continue;
            const Eigen::Vector2f uv = proj(p3Dc);
            if (!pKF->IsInImage(uv(0), uv(1))) continue;
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            Eigen::Vector3f PO = p3Dw - Ow;
            const float dist = PO.norm();
            if (dist < minDistance || dist > maxDistance) continue;
            Eigen::Vector3f Pn = pMP->GetNormal();
            if (PO.dot(Pn) < 0.5 * dist) continue;
            int nPredictedLevel = pMP->PredictScale(dist, pKF);
            q.x.push_back(uv(0)); q.y.push_back(uv(1));
            q.r.push_back(th * pKF->mvScaleFactors[nPredictedLevel]);
            q.minLevel.push_back(nPredictedLevel - 1); q.maxLevel.push_back(nPredictedLevel);   // :504-507
            push_desc(q.descriptors, pMP->GetDescriptor());
            live.push_back(iMP);
        }
        const int N = (int)pKF->mvKeysUn.size();
        std::vector<uint8_t> occupied(N);
        for (int i = 0; i < N; i++) occupied[i] = vpMatched[i] ? 1 : 0;   // :499
        return SearchByProjectionWindow(view_of(*pKF, pKF->mvKeysUn, false), occupied, q, TH_LOW * ratioHamming, false, match);
    }

public:
    // ORBmatcher.cc:427-538 (LoopClosing: projection of the loop key frame's covisible map points)
    int SearchByProjection(KeyFrame *pKF, Sophus::Sim3f &Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th,
                           float ratioHamming = 1.0) {
        std::vector<int32_t> match;
        std::vector<int> live;
        const int nmatches = sim3_projection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming,
                                             [&](const Eigen::Vector3f &p) { return pKF->mpCamera->project(p); }, match, live);
        for (size_t i = 0; i < match.size(); i++)
            if (match[i] >= 0) vpMatched[i] = vpPoints[live[match[i]]];   // :527
        return nmatches;
    }

    // ORBmatcher.cc:540-646 (the variant that also records the key frame each point came from; pinhole projection written out, :573-578)
    int SearchByProjection(KeyFrame *pKF, Sophus::Sim3<float> &Scw, const std::vector<MapPoint *> &vpPoints, const std::vector<KeyFrame *> &vpPointsKFs,
                           std::vector<MapPoint *> &vpMatched, std::vector<KeyFrame *> &vpMatchedKF, int th, float ratioHamming = 1.0) {
        const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
        std::vector<int32_t> match;
        std::vector<int> live;
        const int nmatches = sim3_projection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming, [&](const Eigen::Vector3f &p3Dc) {
            const float invz = 1 / p3Dc(2);
            const float x = p3Dc(0) * invz, y = p3Dc(1) * invz;
            return Eigen::Vector2f(fx * x + cx, fy * y + cy);
        }, match, live);
        for (size_t i = 0; i < match.size(); i++)
            if (match[i] >= 0) { vpMatched[i] = vpPoints[live[match[i]]]; vpMatchedKF[i] = vpPointsKFs[live[match[i]]]; }   // :638-639
        return nmatches;
    }

    // ORBmatcher.cc:907-1146 (LocalMapping::CreateNewMapPoints, LocalMapping.cc:466: 10-30 calls per key frame).
    // PINHOLE key frames (both cameras CAM_PINHOLE -- every monocular / stereo / RGB-D configuration of the reference): both geometric gates run
    // on the device (orbx_search_for_triangulation_pinhole: one upload, one launch chain, one download of the matches; no candidate distance
    // comes back to the host).  F12 is built HERE, once, with the very Eigen expression Pinhole::epipolarConstrain evaluates for every pair
    // (CameraModels/Pinhole.cpp:109-112) -- same types, same operand order, hence the same floats as inside the reference.
    // Any other GeometricCamera (KannalaBrandt8::epipolarConstrain triangulates): the epipole-distance test and the camera model's
    // epipolarConstrain are evaluated by a callback exactly where the reference evaluates them.
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<std::pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo,
                               const bool bCoarse = false) {
        if (pKF1->mpCamera2 || pKF2->mpCamera2) return SearchForTriangulationFisheye(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse);
        Sophus::SE3f T1w = pKF1->GetPose();
        Sophus::SE3f T2w = pKF2->GetPose();
        Sophus::SE3f Tw2 = pKF2->GetPoseInverse();
        Eigen::Vector3f Cw = pKF1->GetCameraCenter();
        Eigen::Vector3f C2 = T2w * Cw;
        Eigen::Vector2f ep = pKF2->mpCamera->project(C2);
        Sophus::SE3f T12 = T1w * Tw2;
        Eigen::Matrix3f R12 = T12.rotationMatrix();
        Eigen::Vector3f t12 = T12.translation();
        GeometricCamera *pCamera1 = pKF1->mpCamera, *pCamera2 = pKF2->mpCamera;
        const int n1 = pKF1->N, n2 = pKF2->N;
        std::vector<uint8_t> skip1(n1), skip2(n2);
        std::vector<float> a1(n1), a2(n2);
        for (int i = 0; i < n1; i++) {
            const bool bStereo1 = pKF1->mvuRight[i] >= 0;
            skip1[i] = (pKF1->GetMapPoint(i) || (bOnlyStereo && !bStereo1)) ? 1 : 0;   // :971-983
            a1[i] = pKF1->mvKeysUn[i].angle;
        }
        for (int i = 0; i < n2; i++) {
            const bool bStereo2 = pKF2->mvuRight[i] >= 0;
            skip2[i] = (pKF2->GetMapPoint(i) || (bOnlyStereo && !bStereo2)) ? 1 : 0;   // :1002-1012
            a2[i] = pKF2->mvKeysUn[i].angle;
        }
        if (pCamera1->GetType() == GeometricCamera::CAM_PINHOLE && pCamera2->GetType() == GeometricCamera::CAM_PINHOLE) {
            static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint must be the 28-byte OpenCV layout");
            Eigen::Matrix3f t12x = Sophus::SO3f::hat(t12);                                   // Pinhole.cpp:109
            Eigen::Matrix3f K1 = pCamera1->toK_();                                           // :110
            Eigen::Matrix3f K2 = pCamera2->toK_();                                           // :111
            Eigen::Matrix3f F12 = K1.transpose().inverse() * t12x * R12 * K2.inverse();      // :112
            orbx_pinhole_gate g;
            std::memset(&g, 0, sizeof(g));
            g.kps1_un = reinterpret_cast<const orbx_keypoint *>(pKF1->mvKeysUn.data());
            g.kps2_un = reinterpret_cast<const orbx_keypoint *>(pKF2->mvKeysUn.data());
            g.u_right1 = pKF1->mvuRight.data(); g.u_right2 = pKF2->mvuRight.data();          // bStereo1 / bStereo2 of :1024-1025
            g.scale_factors2 = pKF2->mvScaleFactors.data();                                  // :1030
            g.level_sigma2_2 = pKF2->mvLevelSigma2.data();                                   // `unc` of :1072
            g.nlevels = (int)std::min(pKF2->mvScaleFactors.size(), pKF2->mvLevelSigma2.size());
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) g.F12[3 * r + c] = F12(r, c);
            g.ep_x = ep(0); g.ep_y = ep(1);
            g.coarse = bCoarse ? 1 : 0;
            g.strict_fp = 0;
            return SearchForTriangulation(pKF1->mDescriptors.data, skip1.data(), n1, FeatVec::from(pKF1->mFeatVec), pKF2->mDescriptors.data, skip2.data(), n2,
                                          FeatVec::from(pKF2->mFeatVec), g, vMatchedPairs);
        }
        auto gate = [&](size_t idx1, size_t idx2) -> bool {
            const cv::KeyPoint &kp1 = pKF1->mvKeysUn[idx1];
            const cv::KeyPoint &kp2 = pKF2->mvKeysUn[idx2];
            const bool bStereo1 = pKF1->mvuRight[idx1] >= 0, bStereo2 = pKF2->mvuRight[idx2] >= 0;
            if (!bStereo1 && !bStereo2) {   // :1026-1034
                const float distex = ep(0) - kp2.pt.x;
                const float distey = ep(1) - kp2.pt.y;
                if (distex * distex + distey * distey < 100 * pKF2->mvScaleFactors[kp2.octave]) return false;
            }
            return bCoarse || pCamera1->epipolarConstrain(pCamera2, kp1, kp2, R12, t12, pKF1->mvLevelSigma2[kp1.octave], pKF2->mvLevelSigma2[kp2.octave]);
        };
        return SearchForTriangulation(pKF1->mDescriptors.data, a1.data(), skip1.data(), n1, FeatVec::from(pKF1->mFeatVec), pKF2->mDescriptors.data,
                                      a2.data(), skip2.data(), n2, FeatVec::from(pKF2->mFeatVec), gate, vMatchedPairs);
    }

    // the same member between fisheye-stereo key frames (both carry mpCamera2): the four camera pairings of ORBmatcher.cc:1036-1069.
    // No feature is "stereo" (bStereo needs !mpCamera2, :980/:1008), so bOnlyStereo leaves no query, and the epipole test (:1026) is off.
    int SearchForTriangulationFisheye(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<std::pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo,
                                      const bool bCoarse) {
        if (!pKF1->mpCamera2 || !pKF2->mpCamera2) throw std::runtime_error("SearchForTriangulation: one key frame is fisheye-stereo and the other is not");
        Sophus::SE3f T1w = pKF1->GetPose();
        Sophus::SE3f Tw2 = pKF2->GetPoseInverse();
        Sophus::SE3f Tr1w = pKF1->GetRightPose();
        Sophus::SE3f Twr2 = pKF2->GetRightPoseInverse();
        const Sophus::SE3f T[2][2] = {{T1w * Tw2, T1w * Twr2}, {Tr1w * Tw2, Tr1w * Twr2}};   // [right1][right2]: Tll, Tlr, Trl, Trr (:936-939)
        Eigen::Matrix3f R[2][2];
        Eigen::Vector3f t[2][2];
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { R[a][b] = T[a][b].rotationMatrix(); t[a][b] = T[a][b].translation(); }
        GeometricCamera *cam1[2] = {pKF1->mpCamera, pKF1->mpCamera2}, *cam2[2] = {pKF2->mpCamera, pKF2->mpCamera2};
        const int n1 = pKF1->N, n2 = pKF2->N;
        auto key = [](KeyFrame *k, size_t idx) -> const cv::KeyPoint & {                       // :984-986, :1020-1022
            return (k->NLeft == -1) ? k->mvKeysUn[idx] : ((int)idx < k->NLeft ? k->mvKeys[idx] : k->mvKeysRight[idx - k->NLeft]);
        };
        std::vector<uint8_t> skip1(n1), skip2(n2);
        std::vector<float> a1(n1), a2(n2);
        for (int i = 0; i < n1; i++) { skip1[i] = (pKF1->GetMapPoint(i) || bOnlyStereo) ? 1 : 0; a1[i] = key(pKF1, i).angle; }
        for (int i = 0; i < n2; i++) { skip2[i] = (pKF2->GetMapPoint(i) || bOnlyStereo) ? 1 : 0; a2[i] = key(pKF2, i).angle; }
        // KannalaBrandt8 cameras (the reference's only CAM_FISHEYE model: 8 parameters) on both rigs: KannalaBrandt8::epipolarConstrain runs on the device
        // (orbx_search_for_triangulation_kb8: no candidate distance returns to the host); any other camera object keeps the callback below
        bool kb8 = pKF1->NLeft >= 0 && pKF2->NLeft >= 0;
        for (GeometricCamera *c : {cam1[0], cam1[1], cam2[0], cam2[1]}) kb8 = kb8 && c->GetType() == GeometricCamera::CAM_FISHEYE && c->size() == 8;
        if (kb8) {
            std::vector<orbx_keypoint> k1(n1), k2(n2);
            auto flat = [](const cv::KeyPoint &k) { orbx_keypoint o; o.x = k.pt.x; o.y = k.pt.y; o.size = k.size; o.angle = k.angle; o.response = k.response; o.octave = k.octave; o.class_id = k.class_id; return o; };
            for (int i = 0; i < n1; i++) k1[i] = flat(key(pKF1, i));
            for (int i = 0; i < n2; i++) k2[i] = flat(key(pKF2, i));
            orbx_kb8_gate g;
            memset(&g, 0, sizeof(g));
            g.kps1 = k1.data(); g.kps2 = k2.data();
            g.n_left1 = pKF1->NLeft; g.n_left2 = pKF2->NLeft;
            g.level_sigma2_1 = pKF1->mvLevelSigma2.data(); g.level_sigma2_2 = pKF2->mvLevelSigma2.data();
            g.nlevels = (int)std::min(pKF1->mvLevelSigma2.size(), pKF2->mvLevelSigma2.size());
            for (int c = 0; c < 2; c++)
                for (int k = 0; k < 8; k++) { g.cam1[c][k] = cam1[c]->getParameter(k); g.cam2[c][k] = cam2[c]->getParameter(k); }
            for (int a = 0; a < 2; a++)
                for (int b = 0; b < 2; b++)
                    for (int r = 0; r < 3; r++) {
                        for (int c = 0; c < 3; c++) g.R12[2 * a + b][3 * r + c] = R[a][b](r, c);
                        g.t12[2 * a + b][r] = t[a][b](r);
                    }
            g.coarse = bCoarse ? 1 : 0;
            std::vector<int32_t> m12(n1, -1);
            const FeatVec f1 = FeatVec::from(pKF1->mFeatVec), f2 = FeatVec::from(pKF2->mFeatVec);
            orbx_featvec fa = f1.c(), fb = f2.c();
            const int r = orbx_search_for_triangulation_kb8(m_, pKF1->mDescriptors.data, skip1.data(), n1, &fa, pKF2->mDescriptors.data, skip2.data(), n2, &fb,
                                                            mbCheckOrientation ? 1 : 0, &g, m12.data());
            if (r < 0) throw std::runtime_error(std::string("orbx_search_for_triangulation_kb8: ") + orbx_status_string(r));
            vMatchedPairs.clear();
            for (int i = 0; i < n1; i++) if (m12[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m12[i]);
            return r;
        }
        auto gate = [&](size_t idx1, size_t idx2) -> bool {
            if (bCoarse) return true;
            const cv::KeyPoint &kp1 = key(pKF1, idx1);
            const cv::KeyPoint &kp2 = key(pKF2, idx2);
            const int r1 = !(pKF1->NLeft == -1 || (int)idx1 < pKF1->NLeft), r2 = !(pKF2->NLeft == -1 || (int)idx2 < pKF2->NLeft);   // bRight1 / bRight2
            return cam1[r1]->epipolarConstrain(cam2[r2], kp1, kp2, R[r1][r2], t[r1][r2], pKF1->mvLevelSigma2[kp1.octave], pKF2->mvLevelSigma2[kp2.octave]);
        };
        return SearchForTriangulation(pKF1->mDescriptors.data, a1.data(), skip1.data(), n1, FeatVec::from(pKF1->mFeatVec), pKF2->mDescriptors.data,
                                      a2.data(), skip2.data(), n2, FeatVec::from(pKF2->mFeatVec), gate, vMatchedPairs);
    }

    // ORBmatcher.cc:1148-1337 (LocalMapping::SearchInNeighbors)
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0, const bool bRight = false) {
        // :1150-1163: the right camera of a fisheye-stereo key frame has its own pose, centre, camera model, grid and keypoints
        Sophus::SE3f Tcw = bRight ? pKF->GetRightPose() : pKF->GetPose();
        Eigen::Vector3f Ow = bRight ? pKF->GetRightCameraCenter() : pKF->GetCameraCenter();
        GeometricCamera *pCamera = bRight ? pKF->mpCamera2 : pKF->mpCamera;
        const int NLeft = pKF->NLeft;
        const int idxOffset = bRight ? NLeft : 0;                                                       // :1296
        const float &bf = pKF->mbf;
        FuseQueries q;
        std::vector<int> live;
        const int nMPs = (int)vpMapPoints.size();
        for (int i = 0; i < nMPs; i++) {
            MapPoint *pMP = vpMapPoints[i];
            if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
            Eigen::Vector3f p3Dw = pMP->GetWorldPos();
            Eigen::Vector3f p3Dc = Tcw * p3Dw;
            if (p3Dc(2) < 0.0f) continue;
            const float invz = 1 / p3Dc(2);
            const Eigen::Vector2f uv = pCamera->project(p3Dc);
            if (!pKF->IsInImage(uv(0), uv(1))) continue;
            const float ur = uv(0) - bf * invz;
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            Eigen::Vector3f PO = p3Dw - Ow;
            const float dist3D = PO.norm();
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            Eigen::Vector3f Pn = pMP->GetNormal();
            if (PO.dot(Pn) < 0.5 * dist3D) continue;
            int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
            q.u.push_back(uv(0)); q.v.push_back(uv(1)); q.ur.push_back(ur);
            q.radius.push_back(th * pKF->mvScaleFactors[nPredictedLevel]);
            q.nPredictedLevel.push_back(nPredictedLevel);
            push_desc(q.descriptors, pMP->GetDescriptor());
            live.push_back(i);
        }
        std::vector<int32_t> bestIdx, bestDist;
        // :1260-1262: the candidates' keypoints are mvKeysUn (Nleft == -1), mvKeys (left of a fisheye rig) or mvKeysRight (bRight)
        FrameView kfv = (NLeft == -1) ? view_of(*pKF, pKF->mvKeysUn, true) : view_of(*pKF, bRight ? pKF->mvKeysRight : pKF->mvKeys, false);
        if (bRight) kfv.mDescriptors += (size_t)NLeft * 32;                                              // :1296-1298: row idx + NLeft
        FuseSearch(kfv, pKF->mvInvLevelSigma2.data(), q, bestIdx, bestDist);
        int nFused = 0;
        for (size_t k = 0; k < live.size(); k++) {   // the reference's tail, in its order, on the live graph (:1309-1330)
            MapPoint *pMP = vpMapPoints[live[k]];
            if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;   // an earlier iteration's Replace / AddObservation may have retired this query
            if (bestIdx[k] < 0 || bestDist[k] > TH_LOW) continue;
            bestIdx[k] += idxOffset;
            MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx[k]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx[k]);
                pKF->AddMapPoint(pMP, bestIdx[k]);
            }
            nFused++;
        }
        return nFused;
    }

    // ORBmatcher.cc:1339-1455 (LoopClosing::SearchAndFuse)
    int Fuse(KeyFrame *pKF, Sophus::Sim3f &Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint) {
        Sophus::SE3f Tcw = Sophus::SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale());
        Eigen::Vector3f Ow = Tcw.inverse().translation();
        const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();
        FuseQueries q;
        std::vector<int> live;
        const int nPoints = (int)vpPoints.size();
        for (int iMP = 0; iMP < nPoints; iMP++) {
            MapPoint *pMP = vpPoints[iMP];
            if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
            Eigen::Vector3f p3Dw = pMP->GetWorldPos();
            Eigen::Vector3f p3Dc = Tcw * p3Dw;
            if (p3Dc(2) < 0.0f) continue;
            const Eigen::Vector2f uv = pKF->mpCamera->project(p3Dc);
            if (!pKF->IsInImage(uv(0), uv(1))) continue;
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            Eigen::Vector3f PO = p3Dw - Ow;
            const float dist3D = PO.norm();
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            Eigen::Vector3f Pn = pMP->GetNormal();
            if (PO.dot(Pn) < 0.5 * dist3D) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
            q.u.push_back(uv(0)); q.v.push_back(uv(1));
            q.radius.push_back(th * pKF->mvScaleFactors[nPredictedLevel]);
            q.nPredictedLevel.push_back(nPredictedLevel);
            push_desc(q.descriptors, pMP->GetDescriptor());
            live.push_back(iMP);
        }
        std::vector<int32_t> bestIdx, bestDist;
        FuseSearch(view_of(*pKF, pKF->mvKeysUn, false), nullptr, q, bestIdx, bestDist);
        int nFused = 0;
        for (size_t k = 0; k < live.size(); k++) {   // :1436-1449
            if (bestIdx[k] < 0 || bestDist[k] > TH_LOW) continue;
            MapPoint *pMP = vpPoints[live[k]];
            MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx[k]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[live[k]] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, bestIdx[k]);
                pKF->AddMapPoint(pMP, bestIdx[k]);
            }
            nFused++;
        }
        return nFused;
    }

    // ORBmatcher.cc:1457-1674 (LoopClosing: guided matching after the Sim3 estimate)
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const Sophus::Sim3f &S12, const float th) {
        const float &fx = pKF1->fx, &fy = pKF1->fy, &cx = pKF1->cx, &cy = pKF1->cy;
        Sophus::SE3f T1w = pKF1->GetPose();
        Sophus::SE3f T2w = pKF2->GetPose();
        Sophus::Sim3f S21 = S12.inverse();
        const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
        const int N1 = (int)vpMapPoints1.size();
        const std::vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N2 = (int)vpMapPoints2.size();
        std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
        for (int i = 0; i < N1; i++) {
            MapPoint *pMP = vpMatches12[i];
            if (pMP) {
                vbAlreadyMatched1[i] = true;
                int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
                if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
            }
        }
        // one query slot per key-frame feature (use == 0: no / bad / already matched map point or a failed gate)
        auto side = [&](const std::vector<MapPoint *> &mps, const std::vector<bool> &done, auto toOther, KeyFrame *pKFo, FuseQueries &q,
                        std::vector<uint8_t> &use) {
            const int N = (int)mps.size();
            use.assign(N, 0);
            q.u.assign(N, 0.f); q.v.assign(N, 0.f); q.radius.assign(N, 0.f); q.nPredictedLevel.assign(N, 0); q.descriptors.assign(32 * (size_t)N, 0);
            for (int i = 0; i < N; i++) {
                MapPoint *pMP = mps[i];
                if (!pMP || done[i] || pMP->isBad()) continue;
                Eigen::Vector3f p3Dw = pMP->GetWorldPos();
                Eigen::Vector3f p3Dc = toOther(p3Dw);
                if (p3Dc(2) < 0.0) continue;
                const float invz = 1.0 / p3Dc(2);
                const float x = p3Dc(0) * invz, y = p3Dc(1) * invz;
                const float u = fx * x + cx, v = fy * y + cy;
                if (!pKFo->IsInImage(u, v)) continue;
                const float maxDistance = pMP->GetMaxDistanceInvariance();
                const float minDistance = pMP->GetMinDistanceInvariance();
                const float dist3D = p3Dc.norm();
                if (dist3D < minDistance || dist3D > maxDistance) continue;
                const int nPredictedLevel = pMP->PredictScale(dist3D, pKFo);
                q.u[i] = u; q.v[i] = v; q.radius[i] = th * pKFo->mvScaleFactors[nPredictedLevel]; q.nPredictedLevel[i] = nPredictedLevel;
                const cv::Mat d = pMP->GetDescriptor();
                std::memcpy(&q.descriptors[32 * (size_t)i], d.data, 32);
                use[i] = 1;
            }
        };
        FuseQueries q1, q2;
        std::vector<uint8_t> use1, use2;
        side(vpMapPoints1, vbAlreadyMatched1, [&](const Eigen::Vector3f &p) { Eigen::Vector3f c1 = T1w * p; Eigen::Vector3f c2 = S21 * c1; return c2; }, pKF2, q1, use1);
        side(vpMapPoints2, vbAlreadyMatched2, [&](const Eigen::Vector3f &p) { Eigen::Vector3f c2 = T2w * p; Eigen::Vector3f c1 = S12 * c2; return c1; }, pKF1, q2, use2);
        std::vector<int32_t> m12;
        const int nFound = SearchBySim3(view_of(*pKF1, pKF1->mvKeysUn, false), view_of(*pKF2, pKF2->mvKeysUn, false), q1, use1, q2, use2, m12);
        for (int i1 = 0; i1 < N1; i1++)
            if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];   // :1668
        return nFound;
    }
