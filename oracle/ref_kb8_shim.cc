/* TEST INFRASTRUCTURE ONLY.  Second translation unit of oracle/_ref/libfrustum_ref.so: KannalaBrandt8::project(const Eigen::Vector3f &)
 * compiled from the reference's own text (excerpted by oracle/Makefile into a temporary file).  Like src/CameraModels/KannalaBrandt8.cpp this unit has
 * NO `using namespace std` and includes <cmath> only, so the text's unqualified cos(psi) / sin(psi) resolve to the C library's double functions --
 * which is what pins orbo_kb8_project's double evaluation (orb_oracle_geom.cc). */
#include <cmath>
#include "frustum_mock.h"

namespace ORB_SLAM3 {
#include "ref_kb8_excerpt.inc"
}  // namespace ORB_SLAM3
