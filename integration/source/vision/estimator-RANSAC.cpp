// Replaces source/vision/estimator-RANSAC.cpp of the reference (ctor :8-14, compute :16-90; decl
// vision/estimator-RANSAC.hpp:10-50).  max_iteration hypotheses are solved and scored on the GPU; with the reference's
// own max_iteration = 1 (sfm-solve.cpp:67) and no sampler this IS the reference as shipped (identity sample).
// The private helpers propose_model / count_inliers (:92-129) have no caller left and are not defined.
#include <vision/estimator-RANSAC.hpp>

#include <cassert>

#include "mvslam-hip-glue.hpp"

namespace mvSLAM
{
FundamentalMatrixEstimatorRANSAC::FundamentalMatrixEstimatorRANSAC(ScalarType max_error_sq_, size_t max_iteration_)
    : max_error_sq(max_error_sq_), max_iteration(max_iteration_)
{
    assert(max_error_sq > epsilon);   // :12-13
    assert(max_iteration > 0);
}

bool FundamentalMatrixEstimatorRANSAC::compute(const std::vector<Vector3Type> &p1, const std::vector<Vector3Type> &p2,
                                               Matrix3Type &F21, std::vector<uint8_t> &inlier_mask)
{
    assert(p1.size() == p2.size());
    const size_t n = p1.size();
    if (n < MIN_DATA_POINT_COUNT)     // :25-29
        return false;
    std::vector<double> a(2 * n), b(2 * n);
    for (size_t i = 0; i < n; ++i) {
        a[2 * i] = p1[i][0]; a[2 * i + 1] = p1[i][1];
        b[2 * i] = p2[i][0]; b[2 * i + 1] = p2[i][1];
    }
    const mvs_params cfg = hip::two_view_params();
    const bool shipped = max_iteration == 1;     // the reference's shuffle is commented out (:41-42)
    double F[9];
    std::vector<uint8_t> mask(n);
    int best_hyp = -1, best_count = 0;
    double best_residual = 0.0;
    const mvs_status st = mvs_ransac_fundamental(hip::context(), a.data(), b.data(), (int)n, max_error_sq, (int)max_iteration,
                                                 shipped ? MVS_SAMPLER_IDENTITY : MVS_SAMPLER_PHILOX, cfg.seed, F, mask.data(),
                                                 &best_hyp, &best_count, &best_residual, nullptr, nullptr);
    if (st != MVS_OK)
        return false;                 // best_count == 0 (:89)
    F21 = hip::from_row_major(F);
    inlier_mask.swap(mask);
    return true;
}
}  // namespace mvSLAM
