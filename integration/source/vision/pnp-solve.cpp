// Replaces source/vision/pnp-solve.cpp of the reference (pnp_solve :16-104; decl vision/pnp.hpp:22-26): the
// cv::solvePnPRansac(SOLVEPNP_P3P, 100 iterations, reprojectionError 0.05, confidence 0.95) call (:43-64) becomes the
// library's batched P3P-RANSAC with the same constants; pose = SE3(R, t).inverse() as :99-101.
#include <vision/pnp.hpp>

#include <cassert>

#include "mvslam-hip-glue.hpp"

namespace mvSLAM
{
static constexpr size_t PNP_MIN_POINT_COUNT = 7;   // pnp-solve.cpp:13

bool pnp_solve(const std::vector<Point3> &world_points, const std::vector<ImagePoint> &image_points,
               const CameraIntrinsics &K, Transformation &pose, std::vector<size_t> &inlier_point_indexes)
{
    assert(world_points.size() >= PNP_MIN_POINT_COUNT);
    assert(world_points.size() == image_points.size());
    const int n = (int)world_points.size();
    mvs_pnp_params prm;
    mvs_pnp_params_default(&prm);      // 100 hypotheses, reprojection error 0.05 px (:47-49)
    prm.refit = 1;                     // cv::solvePnPRansac ends with a refit over all inliers (:53-64)
    double Kr[9], R[9], t[3];
    hip::to_row_major(K, Kr);
    std::vector<double> X(3 * (size_t)n);   // std::vector<Eigen::Vector3d> is packed, copied to stay independent of it
    for (int i = 0; i < n; ++i) {
        X[3 * i] = world_points[i][0]; X[3 * i + 1] = world_points[i][1]; X[3 * i + 2] = world_points[i][2];
    }
    std::vector<int64_t> idx(n);
    int ni = 0;
    if (mvs_pnp_solve(hip::context(), X.data(), &image_points[0].x, n, Kr, &prm, R, t, idx.data(), &ni, nullptr) != MVS_OK)
        return false;
    inlier_point_indexes.reserve(inlier_point_indexes.size() + ni);   // the reference appends (:69-73)
    for (int i = 0; i < ni; ++i)
        inlier_point_indexes.push_back((size_t)idx[i]);
    pose = hip::se3_from_arrays(R, t);
    return true;
}
}  // namespace mvSLAM
