// Replaces source/vision/sfm-solve.cpp of the reference (sfm_solve :285-368, sfm_triangulate :370-394; decl
// vision/sfm.hpp:30-53).  Build WITHOUT -DUSE_OPENCV_ESSENTIAL_MATRIX (SConstruct:82): this is the reference's own
// FundamentalMatrixEstimatorRANSAC branch (:64-90), run on the GPU.
#include <vision/sfm.hpp>

#include <cassert>

#include "mvslam-hip-glue.hpp"

namespace mvSLAM
{

bool sfm_solve(const std::vector<ImagePoint> &p1, const std::vector<ImagePoint> &p2, const CameraIntrinsics &K,
               Transformation &pose2in1_scaled, std::vector<Point3> &pointsin1_scaled, std::vector<size_t> &point_indexes)
{
    assert(p1.size() == p2.size());   // sfm-solve.cpp:292
    const int m = (int)p1.size();
    if (m < 1)
        return false;
    const mvs_params prm = hip::two_view_params();
    double Kr[9], R[9], t[3];
    hip::to_row_major(K, Kr);
    std::vector<double> pts(3 * (size_t)m);
    std::vector<int64_t> idx(m);
    int n = 0;
    // cv::Point_<double> is two packed doubles: &p1[0].x is an m x 2 row-major array
    if (mvs_two_view(hip::context(), &p1[0].x, &p2[0].x, m, Kr, &prm, R, t, pts.data(), idx.data(), &n, nullptr, nullptr) != MVS_OK)
        return false;                 // < 8 points, no model, < 8 inliers, no candidate with points (:319-356)
    pose2in1_scaled = hip::se3_from_arrays(R, t);
    std::vector<Point3> P(n);
    std::vector<size_t> I(n);
    for (int i = 0; i < n; ++i) {
        P[i] = Point3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        I[i] = (size_t)idx[i];
    }
    pointsin1_scaled.swap(P);         // the reference swaps its outputs in (:365-366)
    point_indexes.swap(I);
    return true;
}

void sfm_triangulate(const std::vector<ImagePoint> &p1, const std::vector<ImagePoint> &p2, const CameraIntrinsics &K,
                     const Transformation &pose1, const Transformation &pose2, std::vector<Point3> &points,
                     std::vector<size_t> &point_indexes)
{
    assert(p1.size() == p2.size() && !p1.empty());
    const Transformation T_1_to_2 = pose2.inverse() * pose1;   // sfm-solve.cpp:381
    const int m = (int)p1.size();
    double Kr[9], R[9];
    hip::to_row_major(K, Kr);
    hip::to_row_major(T_1_to_2.rotation().get_matrix(), R);
    const Vector3Type t = T_1_to_2.translation();
    std::vector<double> pts(3 * (size_t)m);
    std::vector<int64_t> idx(m);
    int n = 0;
    if (mvs_triangulate(hip::context(), &p1[0].x, &p2[0].x, m, Kr, R, t.data(), pts.data(), idx.data(), &n) != MVS_OK)
        n = 0;
    std::vector<Point3> P(n);
    std::vector<size_t> I(n);
    for (int i = 0; i < n; ++i) {
        P[i] = Point3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        I[i] = (size_t)idx[i];
    }
    points.swap(P);
    point_indexes.swap(I);
}

}  // namespace mvSLAM
