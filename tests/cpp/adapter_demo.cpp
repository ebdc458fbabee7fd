// Exercises the C++ adapters (orb_slam3_amd/cpp/ORBextractor.h, ORBmatcher.h) exactly as ORB-SLAM3's host code would:
// reads a binary PGM, extracts ORB features, matches the frame against itself shifted, prints counts and FNV hashes.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "../../orb_slam3_amd/cpp/ORBextractor.h"
#include "../../orb_slam3_amd/cpp/ORBmatcher.h"

static uint64_t fnv(const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::string magic; int w, h, maxv;
    f >> magic >> w >> h >> maxv; f.get();
    std::vector<uint8_t> img((size_t)w * h);
    f.read((char *)img.data(), img.size());
    ORB_SLAM3::ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<orbx_keypoint> kps;
    std::vector<uint8_t> desc;
    std::vector<int> lap = {0, 1000};
    const int mono = ex(img.data(), w, h, (size_t)w, kps, desc, lap);
    std::printf("mono %d n %zu kps %016llx desc %016llx levels %d scale %.6f\n", mono, kps.size(),
                (unsigned long long)fnv(kps.data(), kps.size() * sizeof(orbx_keypoint)), (unsigned long long)fnv(desc.data(), desc.size()),
                ex.GetLevels(), ex.GetScaleFactor());
    ORB_SLAM3::ORBextractor::Level l3 = ex.GetPyramidLevel(3);
    std::printf("level3 %dx%d %016llx\n", l3.w, l3.h, (unsigned long long)fnv(l3.padded.data(), l3.padded.size()));
    // frame-to-frame SearchByProjection of the frame against itself
    ORB_SLAM3::ORBmatcher matcher(0.9f, true);
    std::vector<float> sf = ex.GetScaleFactors();
    ORB_SLAM3::FrameView F;
    F.mvKeysUn = kps.data(); F.mDescriptors = desc.data(); F.N = (int)kps.size();
    F.mnMinX = 0; F.mnMaxX = (float)w; F.mnMinY = 0; F.mnMaxY = (float)h; F.mvScaleFactors = sf.data(); F.nlevels = (int)sf.size();
    ORB_SLAM3::ORBmatcher::ProjectedQueries q;
    for (size_t i = 0; i < kps.size(); i++) {
        q.u.push_back(kps[i].x); q.v.push_back(kps[i].y); q.octave.push_back(kps[i].octave); q.angle.push_back(kps[i].angle);
        q.descriptors.insert(q.descriptors.end(), desc.begin() + 32 * i, desc.begin() + 32 * (i + 1));
    }
    std::vector<int32_t> match;
    const int nm = matcher.SearchByProjection(F, {}, q, 7.0f, false, false, match);
    int self = 0;
    for (size_t i = 0; i < match.size(); i++) self += (match[i] == (int)i);
    std::printf("matches %d self %d\n", nm, self);
    // Fuse matching core: the frame's own features as projected map points, Sim3 form (no chi2 gate)
    ORB_SLAM3::ORBmatcher::FuseQueries fq;
    for (size_t i = 0; i < kps.size(); i++) {
        fq.u.push_back(kps[i].x); fq.v.push_back(kps[i].y); fq.radius.push_back(3.0f * sf[kps[i].octave]); fq.nPredictedLevel.push_back(kps[i].octave);
    }
    fq.descriptors = desc;
    std::vector<int32_t> bi, bd;
    matcher.FuseSearch(F, nullptr, fq, bi, bd);
    int fself = 0;
    for (size_t i = 0; i < bi.size(); i++) fself += (bd[i] == 0);
    std::printf("fuse zero-distance %d of %zu\n", fself, bi.size());
    // SearchBySim3 of the frame against itself: every feature's map point projects onto the feature, mutual agreement everywhere
    std::vector<uint8_t> use(kps.size(), 1);
    std::vector<int32_t> m12;
    const int nSim3 = matcher.SearchBySim3(F, F, fq, use, fq, use, m12);
    int sself = 0;
    for (size_t i = 0; i < m12.size(); i++) sself += (m12[i] >= 0 && std::equal(desc.begin() + 32 * i, desc.begin() + 32 * (i + 1), desc.begin() + 32 * m12[i]));
    std::printf("sim3 found %d identical-descriptor %d\n", nSim3, sself);
    // ComputeDistinctiveDescriptors on sets of 5 consecutive descriptors
    std::vector<int32_t> setPtr, best;
    for (size_t i = 0; i + 5 <= kps.size(); i += 5) setPtr.push_back((int32_t)i);
    setPtr.push_back((int32_t)(kps.size() / 5 * 5));
    matcher.ComputeDistinctiveDescriptors(desc, setPtr, best);
    std::printf("distinctive sets %zu hash %016llx\n", best.size(), (unsigned long long)fnv(best.data(), best.size() * 4));
    return 0;
}
