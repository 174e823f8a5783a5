// Glue shared by the forwarding translation units of this directory: replaces NOTHING in the reference, it is the one
// new header they include.  Written against the reference's real types (OpenCV + Eigen), which are not installed in
// the image this library is built in: these files are compiled in the mvSLAM tree, not here.  What they rely on:
//   * cv::Point_<double> is two packed doubles, cv::DMatch is {int queryIdx, trainIdx, imgIdx; float distance} (16 B =
//     mvs_match, checked by tests/test_abi.py), cv::KeyPoint is {Point2f pt; float size, angle, response; int octave,
//     class_id} (28 B = mvs_keypoint), a continuous CV_8UC1 cv::Mat is row-major bytes;
//   * Eigen fixed-size matrices are COLUMN-major: every 3x3 crosses the C ABI (row-major) through the two helpers below;
//     Eigen vectors (Vector2Type / Vector3Type) are packed doubles.
#pragma once
#include <mvslam_hip.h>

#include <Eigen/Core>
#include <cstdlib>
#include <math/lie-group.hpp>
#include <math/matrix.hpp>

namespace mvSLAM
{
namespace hip
{
typedef Eigen::Matrix<ScalarType, 3, 3, Eigen::RowMajor> RowMajor3;

// One context (device 0, own stream) per host thread.  The reference path is single-threaded.
inline mvs_ctx *context()
{
    static thread_local mvs_ctx *c = nullptr;
    if (!c && mvs_ctx_create(0, &c) != MVS_OK)
        std::abort();   // no HIP device: there is no CPU fallback to fall back to
    return c;
}

// What the reference hard-codes and the library parameterises.  MVSLAM_HIP_HYPOTHESES unset = the reference as shipped:
// one iteration on the first eight matches (sfm-solve.cpp:67, estimator-RANSAC.cpp:41-42).
inline mvs_params two_view_params()
{
    mvs_params p;
    mvs_params_default(&p);
    if (const char *h = std::getenv("MVSLAM_HIP_HYPOTHESES")) {
        p.num_hypotheses = std::atoi(h);
        p.sampler = MVS_SAMPLER_PHILOX;
        if (const char *s = std::getenv("MVSLAM_HIP_SEED"))
            p.seed = std::strtoull(s, nullptr, 0);
    }
    return p;
}

inline void to_row_major(const Matrix3Type &M, double out[9])
{
    Eigen::Map<RowMajor3> view(out);   // (a temporary `Eigen::Map<RowMajor3>(out) = M;` parses as a DECLARATION of `out`:
    view = M;                          // found by the syntax check of tests/test_integration_syntax.py)
}
inline Matrix3Type from_row_major(const double in[9])
{
    return Eigen::Map<const RowMajor3>(in);
}
// The library returns SE3(SO3(R), t).inverse() already rectified exactly as the reference's SO3(Matrix3) ctor does
// (lie-group.hpp:31-36,84-96); going through SO3(Matrix3) again re-rectifies an orthonormal matrix: a <= 1-ulp effect.
inline SE3 se3_from_arrays(const double R[9], const double t[3])
{
    return SE3(SO3(from_row_major(R)), Vector3Type(t[0], t[1], t[2]));
}
}  // namespace hip
}  // namespace mvSLAM
