// The extractor adapter exactly as ORB-SLAM3's Frame constructor uses it (Frame.cc:ExtractORB): the REFERENCE signature
//   int operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors,
//                  std::vector<int>& vLappingArea)
// on cv::Mat / cv::KeyPoint (here: the OpenCV stand-in of oracle/ocv_shim), then the public mvImagePyramid.
// usage: adapter_ocv_demo in.pgm out.bin     out.bin = int32 n, int32 monoIndex, n keypoints (28 B), n descriptors (32 B), level 3 ROI
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "../../orb_slam3_amd/cpp/ORBextractor.h"

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::string magic; int w, h, maxv;
    f >> magic >> w >> h >> maxv; f.get();
    cv::Mat im(h, w, CV_8UC1);
    f.read((char *)im.data, (size_t)w * h);
    ORB_SLAM3::ORBextractor extractor(1000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> mvKeys;
    cv::Mat mDescriptors;
    std::vector<int> vLapping = {0, 1000};
    const int monoLeft = extractor(im, cv::Mat(), mvKeys, mDescriptors, vLapping);
    const int32_t head[2] = {(int32_t)mvKeys.size(), monoLeft};
    std::ofstream o(argv[2], std::ios::binary);
    o.write((const char *)head, 8);
    o.write((const char *)mvKeys.data(), mvKeys.size() * sizeof(cv::KeyPoint));
    for (int i = 0; i < mDescriptors.rows; i++) o.write((const char *)mDescriptors.ptr(i), 32);
    // Frame::ComputeStereoMatches style access (Frame.cc:818, 908): only the touched level is downloaded
    const cv::Mat &l3 = extractor.mvImagePyramid[3];
    for (int r = 0; r < l3.rows; r++) o.write((const char *)l3.ptr(r), l3.cols);
    const int nRows = extractor.mvImagePyramid[3].rows;   // second access: cached
    std::printf("n %zu mono %d level3 %dx%d levels touched %d of %zu\n", mvKeys.size(), monoLeft, l3.cols, nRows, extractor.mvImagePyramid.downloads(), extractor.mvImagePyramid.size());
    return 0;
}
