/* TEST INFRASTRUCTURE ONLY.  Second translation unit of oracle/_ref/libfrustum_ref.so: KannalaBrandt8::project(const Eigen::Vector3f &), unproject,
 * unprojectEig, epipolarConstrain, TriangulateMatches and Triangulate compiled from the reference's own text (excerpted by oracle/Makefile into a temporary file).  Like src/CameraModels/KannalaBrandt8.cpp this unit has
 * NO `using namespace std` and includes <cmath> only, so the text's unqualified cos(psi) / sin(psi) resolve to the C library's double functions --
 * which is what pins orbo_kb8_project's double evaluation (orb_oracle_geom.cc). */
#include <cmath>
#include "frustum_mock.h"

namespace ORB_SLAM3 {
#include "ref_kb8_excerpt.inc"
}  // namespace ORB_SLAM3

using namespace ORB_SLAM3;

/* KannalaBrandt8::TriangulateMatches (the value epipolarConstrain compares with 0.0001) for n keypoint pairs; R12 row-major */
extern "C" void kb8ref_triangulate_matches(const float *cam1, const float *cam2, int n, const float *xy1, const float *xy2, const float *R12, const float *t12,
                                           const float *sigma1, const float *sigma2, unsigned char *ok, float *tm_value, float *rays /* [n][6]: unprojectEig of both */) {
    KannalaBrandt8 c1, c2;
    c1.mvParameters.assign(cam1, cam1 + 8);
    c2.mvParameters.assign(cam2, cam2 + 8);
    Eigen::Matrix3f R;
    Eigen::Vector3f t;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R(r, c) = R12[3 * r + c]; t(r) = t12[r]; }
    for (int i = 0; i < n; i++) {
        cv::KeyPoint k1, k2;
        k1.pt = cv::Point2f(xy1[2 * i], xy1[2 * i + 1]);
        k2.pt = cv::Point2f(xy2[2 * i], xy2[2 * i + 1]);
        Eigen::Vector3f p3D;
        tm_value[i] = c1.TriangulateMatches(&c2, k1, k2, R, t, sigma1[i], sigma2[i], p3D);
        ok[i] = c1.epipolarConstrain(&c2, k1, k2, R, t, sigma1[i], sigma2[i]) ? 1 : 0;
        if (rays) {
            const Eigen::Vector3f a = c1.unprojectEig(k1.pt), b = c2.unprojectEig(k2.pt);
            for (int k = 0; k < 3; k++) { rays[6 * i + k] = a(k); rays[6 * i + 3 + k] = b(k); }
        }
    }
}
