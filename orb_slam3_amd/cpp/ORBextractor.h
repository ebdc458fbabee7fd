// ORBextractor.h -- drop-in replacement for /root/reference/include/ORBextractor.h:43-109.
//
// Same namespace, class name, constructor and method names as the reference; every method forwards to the C ABI
// of liborbx.so (include/orbx.h).  With ORBX_WITH_OPENCV defined (a tree that has OpenCV, i.e. ORB-SLAM3 itself)
// operator() has the reference's exact signature (cv::InputArray / std::vector<cv::KeyPoint> / cv::OutputArray)
// and mvImagePyramid is a std::vector<cv::Mat>.  Without OpenCV (this repo's build box) the same class works on
// plain buffers so that it can be compiled and tested here.
#ifndef ORBX_ADAPTER_ORBEXTRACTOR_H
#define ORBX_ADAPTER_ORBEXTRACTOR_H

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/orbx.h"

#ifdef ORBX_WITH_OPENCV
#include <opencv2/opencv.hpp>
static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint layout");
#endif

namespace ORB_SLAM3 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0,
                 unsigned flags = 0)
        : nlevels_(nlevels), scaleFactor_(scaleFactor) {
        orbx_params p = {nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, flags};
        const int st = orbx_create(&p, device, 0, 0, 0, &ex_);
        if (st != ORBX_OK) throw std::runtime_error(std::string("orbx_create: ") + orbx_status_string(st) + " " + orbx_last_error());
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        orbx_get_scale_tables(ex_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data());
    }
    ~ORBextractor() { orbx_destroy(ex_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Plain-buffer form: 8-bit single-channel image; returns monoIndex, or -1 if the image is empty
    // (ORBextractor.cc:1090).  keypoints / descriptors (N x 32, row-major) are overwritten.
    int operator()(const uint8_t *image, int width, int height, size_t stride, std::vector<orbx_keypoint> &keypoints,
                   std::vector<uint8_t> &descriptors, const std::vector<int> &vLappingArea) {
        if (!image || width <= 0 || height <= 0) return -1;
        const int cap = orbx_output_capacity(ex_, width, height);
        if (cap < 0) throw std::runtime_error(std::string("orbx: ") + orbx_status_string(cap));
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        int n = 0, mono = 0;
        const int st = orbx_extract(ex_, image, width, height, stride, vLappingArea[0], vLappingArea[1], keypoints.data(),
                                    descriptors.data(), cap, &n, &mono);
        if (st == ORBX_E_EMPTY) return -1;
        if (st != ORBX_OK) throw std::runtime_error(std::string("orbx_extract: ") + orbx_status_string(st) + " " + orbx_last_error());
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);
        width_ = width; height_ = height;
        return mono;
    }

#ifdef ORBX_WITH_OPENCV
    // The reference signature (ORBextractor.h:56-58).  The mask is ignored, as in the reference.
    int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints,
                   cv::OutputArray _descriptors, std::vector<int> &vLappingArea) {
        if (_image.empty()) return -1;
        cv::Mat image = _image.getMat();
        if (image.type() != CV_8UC1) throw std::runtime_error("ORBextractor: CV_8UC1 image expected");   // assert(image.type() == CV_8UC1) :1094
        std::vector<orbx_keypoint> kps;
        std::vector<uint8_t> desc;
        const int mono = (*this)(image.data, image.cols, image.rows, image.step, kps, desc, vLappingArea);
        _keypoints.resize(kps.size());
        if (!kps.empty()) std::memcpy((void *)_keypoints.data(), kps.data(), kps.size() * sizeof(orbx_keypoint));
        if (kps.empty()) _descriptors.release();
        else {
            _descriptors.create((int)kps.size(), 32, CV_8U);
            std::memcpy(_descriptors.getMat().data, desc.data(), desc.size());
        }
        // the public pyramid (Frame::ComputeStereoMatches reads it, Frame.cc:818,908,923) is fetched lazily, level by level, on first use
        mvImagePyramid.reset(this, image.cols, image.rows);
        return mono;
    }
    // std::vector<cv::Mat> mvImagePyramid of the reference (ORBextractor.h:83), filled on demand: operator[] downloads the level the
    // first time it is touched after an extraction (one D2H of the padded level, as the reference's parent buffer holds it)
    class LazyPyramid {
    public:
        size_t size() const { return levels_.size(); }
        int downloads() const { return downloads_; }   // levels fetched from the device so far (diagnostic)
        cv::Mat &operator[](size_t l) {
            if (!valid_.at(l)) {
                int w, h;
                orbx_level_size(ex_->ex_, width_, height_, (int)l, &w, &h);
                cv::Mat padded(h + 38, w + 38, CV_8UC1);
                const int st = orbx_get_level(ex_->ex_, 0, (int)l, padded.data, padded.step);
                if (st != ORBX_OK) throw std::runtime_error(std::string("orbx_get_level: ") + orbx_status_string(st));
                levels_[l] = padded(cv::Rect(19, 19, w, h));
                valid_[l] = true;
                downloads_++;
            }
            return levels_[l];
        }
        void reset(ORBextractor *ex, int width, int height) {
            ex_ = ex; width_ = width; height_ = height;
            levels_.assign(ex->nlevels_, cv::Mat());
            valid_.assign(ex->nlevels_, false);
        }
    private:
        ORBextractor *ex_ = nullptr;
        int width_ = 0, height_ = 0, downloads_ = 0;
        std::vector<cv::Mat> levels_;
        std::vector<bool> valid_;
    };
    LazyPyramid mvImagePyramid;
#else
    // mvImagePyramid[level] as a padded host copy: ROI origin = data() + 19*stride + 19
    struct Level { int w = 0, h = 0; size_t stride = 0; std::vector<uint8_t> padded; const uint8_t *roi() const { return padded.data() + 19 * stride + 19; } };
    Level GetPyramidLevel(int level) {
        Level L;
        if (orbx_level_size(ex_, width_, height_, level, &L.w, &L.h) != ORBX_OK) throw std::runtime_error("bad level");
        L.stride = (size_t)L.w + 38;
        L.padded.resize(L.stride * (L.h + 38));
        const int st = orbx_get_level(ex_, 0, level, L.padded.data(), L.stride);
        if (st != ORBX_OK) throw std::runtime_error(std::string("orbx_get_level: ") + orbx_status_string(st));
        return L;
    }
#endif

    int inline GetLevels() { return nlevels_; }
    float inline GetScaleFactor() { return (float)scaleFactor_; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    orbx_extractor *handle() { return ex_; }

protected:
    orbx_extractor *ex_ = nullptr;
    int nlevels_;
    double scaleFactor_;
    int width_ = 0, height_ = 0;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace ORB_SLAM3

#endif
