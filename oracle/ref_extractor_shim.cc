// ref_extractor_shim.cc -- TEST INFRASTRUCTURE ONLY.  C entry points over the REFERENCE's ORB_SLAM3::ORBextractor, whose
// source (/root/reference/src/ORBextractor.cc) is compiled where it lies against oracle/ocv_shim (see the header there for
// what that does and does not pin).  Built by oracle/Makefile into oracle/_ref/liborb_ref.so; used by
// tests/test_oracle_vs_reference.py and tools/gen_ref_golden.py.  Nothing of the product links or loads this.
#include <cstring>
#include <vector>

#include "ORBextractor.h"   // the reference's header, -I/root/reference/include

extern "C" {

void *orbref_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
    return new ORB_SLAM3::ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th);
}
void orbref_destroy(void *h) { delete static_cast<ORB_SLAM3::ORBextractor *>(h); }

// ORBextractor::operator(); returns monoIndex (or -1), *n_out keypoints written (28 bytes each = cv::KeyPoint layout of the shim)
int orbref_extract(void *h, const unsigned char *img, int w, int hgt, size_t stride, int lap0, int lap1, float *kps7, unsigned char *desc,
                   int cap, int *n_out) {
    ORB_SLAM3::ORBextractor *ex = static_cast<ORB_SLAM3::ORBextractor *>(h);
    cv::Mat image(hgt, w, CV_8UC1);
    for (int r = 0; r < hgt; r++) memcpy(image.ptr(r), img + (size_t)r * stride, w);
    std::vector<cv::KeyPoint> keys;
    cv::Mat descriptors;
    std::vector<int> lap = {lap0, lap1};
    const int mono = (*ex)(image, cv::Mat(), keys, descriptors, lap);
    const int n = (int)keys.size();
    *n_out = n;
    for (int i = 0; i < n && i < cap; i++) {
        float *o = kps7 + 7 * (size_t)i;
        o[0] = keys[i].pt.x; o[1] = keys[i].pt.y; o[2] = keys[i].size; o[3] = keys[i].angle; o[4] = keys[i].response;
        memcpy(&o[5], &keys[i].octave, 4); memcpy(&o[6], &keys[i].class_id, 4);
        memcpy(desc + 32 * (size_t)i, descriptors.ptr(i), 32);
    }
    return mono;
}

// T1 tables of the reference object
void orbref_tables(void *h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2, int nlevels) {
    ORB_SLAM3::ORBextractor *ex = static_cast<ORB_SLAM3::ORBextractor *>(h);
    const std::vector<float> a = ex->GetScaleFactors(), b = ex->GetInverseScaleFactors(), c = ex->GetScaleSigmaSquares(),
                             d = ex->GetInverseScaleSigmaSquares();
    for (int i = 0; i < nlevels; i++) { scale[i] = a[i]; inv_scale[i] = b[i]; sigma2[i] = c[i]; inv_sigma2[i] = d[i]; }
}

// mvImagePyramid[level] (public member): ROI size and a copy of the padded image around it (19-px ring)
int orbref_level(void *h, int level, int *w, int *hgt, unsigned char *padded, size_t cap_bytes) {
    ORB_SLAM3::ORBextractor *ex = static_cast<ORB_SLAM3::ORBextractor *>(h);
    if (level < 0 || level >= (int)ex->mvImagePyramid.size()) return -1;
    const cv::Mat &m = ex->mvImagePyramid[level];
    *w = m.cols; *hgt = m.rows;
    const int pw = m.cols + 38, ph = m.rows + 38;
    if (padded && cap_bytes >= (size_t)pw * ph)
        for (int r = 0; r < ph; r++) memcpy(padded + (size_t)r * pw, m.data + ((ptrdiff_t)r - 19) * (ptrdiff_t)m.step.v - 19, pw);
    return 0;
}

}  // extern "C"
