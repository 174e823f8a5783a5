// Replaces source/vision/fundamental-matrix.cpp of the reference (find_fundamental_matrix :204-267; decl
// vision/fundamental-matrix.hpp:16-19): normalise both sets, 8-point with the 9x9 + 3x3 SVD, de-normalise -- one lane
// of the GPU kernel that runs 50 000 of these per pair, exposed for API parity.
#include <vision/fundamental-matrix.hpp>

#include <cassert>

#include "mvslam-hip-glue.hpp"

namespace mvSLAM
{
bool find_fundamental_matrix(const std::vector<Vector3Type> &p1_sample, const std::vector<Vector3Type> &p2_sample,
                             Matrix3Type &F21)
{
    assert(p1_sample.size() == 8 && p2_sample.size() == 8);   // fundamental-matrix.cpp:210-211
    double a[16], b[16], F[9];
    for (int i = 0; i < 8; ++i) {       // homogeneous (x, y, 1): the third coordinate is implied
        a[2 * i] = p1_sample[i][0]; a[2 * i + 1] = p1_sample[i][1];
        b[2 * i] = p2_sample[i][0]; b[2 * i + 1] = p2_sample[i][1];
    }
    const mvs_status st = mvs_find_fundamental_matrix(hip::context(), a, b, F);
    if (st != MVS_OK)
        return false;                   // degenerate sample: the reference asserts scale > epsilon (:45)
    F21 = hip::from_row_major(F);
    return true;
}
}  // namespace mvSLAM
