// ocv_shim -- TEST INFRASTRUCTURE ONLY.  A minimal stand-in for the part of the OpenCV 4 API that
// /root/reference/src/ORBextractor.cc uses, so that the reference's OWN source file can be compiled where it lies
// (oracle/Makefile, target _ref/liborb_ref.so) and run next to the oracle's restatement of it.
// The container/type classes (Mat, KeyPoint, Point_, Size, Rect) are plain re-implementations of the documented
// OpenCV semantics the reference relies on (shared-buffer ROI views, row/col ranges, in-place borders).  The five
// arithmetic primitives (resize INTER_LINEAR 8U, copyMakeBorder REFLECT_101, FAST 9/16 + NMS, fixed-point GaussianBlur,
// fastAtan2) and cvRound are NOT OpenCV: they forward to the oracle's restated primitives (orb_oracle.h, [OCV-recalled]).
// What this pins: every line of the reference's own control flow and arithmetic (tables, cell loop and threshold
// fallback, quad-tree, orientation, steered BRIEF, output assembly).  What it cannot pin: the OpenCV primitives themselves.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>   // the real opencv2/core pulls these in; DBoW2's TemplatedVocabulary.h relies on it
#include <string>
#include <vector>

#include "../../orb_oracle.h"

typedef unsigned char uchar;
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_PI 3.1415926535897932384626433832795
#ifndef ORB_REF_BLUR_OCV440
#define ORB_REF_BLUR_OCV440 0   /* Gaussian taps of OpenCV >= 4.5.1 (the oracle's default); 1 = OpenCV <= 4.5.0 */
#endif

inline int cvRound(double v) { return (int)lrint(v); }   // [OCV] cvtsd2si: round half to even
inline int cvRound(float v) { return orbo_cv_round_f(v); }
inline int cvRound(int v) { return v; }
inline int cvFloor(double v) { return (int)std::floor(v); }
inline int cvCeil(double v) { return (int)std::ceil(v); }

namespace cv {

template <class T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T _x, T _y) : x(_x), y(_y) {}
    template <class U> Point_(const Point_<U> &o) : x((T)o.x), y((T)o.y) {}
    Point_ &operator*=(float s) { x = (T)(x * s); y = (T)(y * s); return *this; }
};
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};
struct Rect {
    int x, y, width, height;
    Rect(int _x, int _y, int w, int h) : x(_x), y(_y), width(w), height(h) {}
};

struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1)
        : pt(x, y), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id) {}
};

enum { INTER_LINEAR = 1 };
enum { BORDER_REFLECT_101 = 4, BORDER_ISOLATED = 16 };

class Mat {
public:
    struct Step {
        size_t v = 0;
        operator size_t() const { return v; }
    };
    int rows = 0, cols = 0;
    uchar *data = nullptr;
    Step step;
    // the allocation this view lives in (for in-place border detection)
    std::shared_ptr<uchar> buf;
    uchar *base = nullptr;
    int base_rows = 0, base_cols = 0;

    Mat() {}
    Mat(int r, int c, int type) { alloc(r, c, type); }
    Mat(Size s, int type) { alloc(s.height, s.width, type); }
    static Mat zeros(int r, int c, int type) { Mat m(r, c, type); if (m.data) memset(m.data, 0, (size_t)r * m.step.v); return m; }
    bool isContinuous() const { return step.v == (size_t)cols * esz(); }

    int type() const { return type_; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    size_t step1() const { return step.v; }
    Mat getMat() const { return *this; }
    void release() { *this = Mat(); }
    void create(int r, int c, int type) { if (r != rows || c != cols || type != type_ || !data) alloc(r, c, type); }

    Mat view(int x, int y, int w, int h) const {
        Mat m = *this;
        m.data = data + (size_t)y * step.v + x;
        m.rows = h; m.cols = w;
        return m;
    }
    Mat operator()(const Rect &r) const { return view(r.x, r.y, r.width, r.height); }
    Mat rowRange(int a, int b) const { return view(0, a, cols, b - a); }
    Mat colRange(int a, int b) const { return view(a, 0, b - a, rows); }
    Mat row(int i) const { return view(0, i, cols, 1); }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; r++) memcpy(m.data + (size_t)r * m.step.v, data + (size_t)r * step.v, (size_t)cols * esz());
        return m;
    }
    void copyTo(const Mat &dst) const {  // into an existing view of the same size (descriptor rows)
        assert(dst.rows == rows && dst.cols == cols);
        for (int r = 0; r < rows; r++) memcpy(dst.data + (size_t)r * dst.step.v, data + (size_t)r * step.v, (size_t)cols * esz());
    }
    template <class T> T &at(int r, int c) { return *(T *)(data + (size_t)r * step.v + (size_t)c * sizeof(T)); }
    template <class T> const T &at(int r, int c) const { return *(const T *)(data + (size_t)r * step.v + (size_t)c * sizeof(T)); }
    uchar *ptr(int r = 0) { return data + (size_t)r * step.v; }
    const uchar *ptr(int r = 0) const { return data + (size_t)r * step.v; }
    template <class T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step.v); }
    template <class T> const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step.v); }

private:
    int type_ = CV_8UC1;
    size_t esz() const { return type_ == CV_32F ? 4 : 1; }
    void alloc(int r, int c, int type) {
        rows = r; cols = c; type_ = type;
        step.v = (size_t)c * esz();
        buf = std::shared_ptr<uchar>(new uchar[(size_t)std::max(r, 1) * std::max(c, 1) * esz() + 64], std::default_delete<uchar[]>());
        data = base = buf.get();
        base_rows = r; base_cols = c;
    }
};

// cv::FileStorage / cv::FileNode: only named by the virtual save() / load() of DBoW2's TemplatedVocabulary, which the
// reference build never calls (the vocabulary is loaded with loadFromTextFile)
struct FileNode {
    enum { SEQ = 5 };
    FileNode operator[](const char *) const { return FileNode(); }
    FileNode operator[](const std::string &) const { return FileNode(); }
    FileNode operator[](int) const { return FileNode(); }
    size_t size() const { assert(!"cv::FileNode is outside the shim"); return 0; }
    int type() const { return 0; }
    template <class T> operator T() const { assert(!"cv::FileNode is outside the shim"); return T(); }
};
struct FileStorage {
    enum { READ = 0, WRITE = 1 };
    FileStorage() {}
    template <class S> FileStorage(const S &, int) { assert(!"cv::FileStorage is outside the shim"); }
    bool isOpened() const { return false; }
    void release() {}
    FileNode operator[](const char *) const { return FileNode(); }
    FileNode operator[](const std::string &) const { return FileNode(); }
    template <class T> FileStorage &operator<<(const T &) { return *this; }
};

typedef const Mat &InputArray;
typedef Mat &OutputArray;

inline float fastAtan2(float y, float x) { return orbo_fast_atan2(y, x); }

inline void resize(InputArray src, OutputArray dst, Size dsize, double /*fx*/ = 0, double /*fy*/ = 0, int interpolation = INTER_LINEAR) {
    assert(interpolation == INTER_LINEAR);
    dst.create(dsize.height, dsize.width, CV_8UC1);
    orbo_resize_linear_u8(src.data, src.cols, src.rows, src.step, dst.data, dst.cols, dst.rows, dst.step);
}

inline void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType) {
    assert((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101 && top == bottom && top == left && top == right);
    const int h = src.rows + 2 * top, w = src.cols + 2 * top;
    const bool inplace = dst.data && dst.rows == h && dst.cols == w && src.data == dst.data + (size_t)top * dst.step.v + left;
    if (!inplace) {
        // src is a stand-alone image (level 0) or must be treated as one (BORDER_ISOLATED)
        if (dst.rows != h || dst.cols != w || !dst.data) dst = Mat(h, w, CV_8UC1);
        for (int r = 0; r < src.rows; r++) memcpy(dst.data + (size_t)(r + top) * dst.step.v + left, src.data + (size_t)r * src.step.v, src.cols);
    }
    orbo_border_reflect101(dst.data, src.cols, src.rows, dst.step, top);
}

inline void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY, int borderType) {
    assert(ksize.width == 7 && ksize.height == 7 && sigmaX == 2 && sigmaY == 2 && borderType == BORDER_REFLECT_101);
    Mat in = src.clone();  // the reference blurs in place
    dst.create(in.rows, in.cols, CV_8UC1);
    orbo_gauss7_u8(in.data, in.cols, in.rows, in.step, dst.data, dst.step, ORB_REF_BLUR_OCV440);
}

inline void FAST(InputArray image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression = true) {
    assert(nonmaxSuppression);
    std::vector<orbo_keypoint> tmp((size_t)image.rows * image.cols / 2 + 16);
    const int n = orbo_fast9_16(image.data, image.cols, image.rows, image.step, threshold, tmp.data(), (int)tmp.size());
    keypoints.clear();
    for (int i = 0; i < n; i++) keypoints.push_back(KeyPoint(tmp[i].x, tmp[i].y, tmp[i].size, tmp[i].angle, tmp[i].response, tmp[i].octave, tmp[i].class_id));
}

// only referenced by the dead ComputeKeyPointsOld (ORBextractor.cc:898-1075, commented out at :1101); never executed
struct KeyPointsFilter {
    static void retainBest(std::vector<KeyPoint> &, int) { assert(!"KeyPointsFilter::retainBest is outside the shim"); }
};

// cv::norm(a, b, NORM_L1) on 8-bit windows (Frame::ComputeStereoMatches' SAD): an exact integer sum, returned as double
enum { NORM_L1 = 2, NORM_HAMMING = 6 };
inline double norm(const Mat &a, const Mat &b, int normType) {
    assert(normType == NORM_L1 && a.rows == b.rows && a.cols == b.cols);
    long s = 0;
    for (int r = 0; r < a.rows; r++) {
        const uchar *pa = a.ptr(r), *pb = b.ptr(r);
        for (int c = 0; c < a.cols; c++) s += pa[c] > pb[c] ? pa[c] - pb[c] : pb[c] - pa[c];
    }
    return (double)s;
}

// cv::DMatch / cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, matches, 2) as Frame::ComputeStereoFishEyeMatches uses them (Frame.cc:43,
// 1144) [OCV-recalled]: brute force, per query the min(k, train rows) nearest train rows in ascending distance, the lower train index
// first among equals; the arithmetic is the oracle's orbo_knn2
struct DMatch {
    int queryIdx = -1, trainIdx = -1, imgIdx = -1;
    float distance = 3.4028235e38f;
};
class BFMatcher {
public:
    BFMatcher(int normType = 4, bool crossCheck = false) : norm_(normType) { (void)crossCheck; }
    void knnMatch(const Mat &q, const Mat &t, std::vector<std::vector<DMatch>> &matches, int k) const {
        assert(norm_ == NORM_HAMMING && k == 2 && (q.rows == 0 || q.cols == 32) && (t.rows == 0 || t.cols == 32));
        std::vector<uchar> qb((size_t)q.rows * 32 + 32), tb((size_t)t.rows * 32 + 32);
        for (int r = 0; r < q.rows; r++) memcpy(qb.data() + (size_t)r * 32, q.ptr(r), 32);
        for (int r = 0; r < t.rows; r++) memcpy(tb.data() + (size_t)r * 32, t.ptr(r), 32);
        std::vector<int32_t> idx((size_t)2 * q.rows + 2), dist(idx.size());
        orbo_knn2(qb.data(), q.rows, tb.data(), t.rows, idx.data(), dist.data());
        matches.assign((size_t)q.rows, std::vector<DMatch>());
        for (int i = 0; i < q.rows; i++)
            for (int j = 0; j < 2; j++)
                if (idx[2 * i + j] >= 0) {
                    DMatch m;
                    m.queryIdx = i; m.trainIdx = idx[2 * i + j]; m.imgIdx = 0; m.distance = (float)dist[2 * i + j];
                    matches[i].push_back(m);
                }
    }
private:
    int norm_;
};
}  // namespace cv
