// mvslam_compat.hpp -- the reference's C++ call surface for the two-view path, on top of the C ABI.
//
// Header-only.  Same names, argument meaning and error behaviour as the reference
// (namespace mvSLAM): VisualFeature::match_visual_features / match_and_filter_visual_features,
// sfm_solve, sfm_triangulate, find_fundamental_matrix, FundamentalMatrixEstimatorRANSAC, ImagePair,
// SO3 / SE3 / PinholeCamera.  The arithmetic of the path runs in libmvslam_hip.so (HIP kernels);
// this file only marshals arguments, exactly where the reference crosses from its callers
// (front-end/image-pair.cpp:57,146; utility/reconstruct-scene.cpp:40,48) into source/vision/.
//
// OpenCV and Eigen are not available in the build image, so layout-compatible stand-ins are
// defined here (KeyPoint, DMatch, Mat8u, Vector3Type, Matrix3Type) and this header is what the C++
// tests compile.  For the real tree the same forwarding bodies are written against the real types
// (cv::Mat, cv::KeyPoint, Eigen matrices) as replacement translation units: integration/source/vision/.
#pragma once

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <limits>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/mvslam_hip.h"

namespace mvSLAM
{
// ---- system-config.hpp:6-14 -----------------------------------------------------------------
using ScalarType = double;
constexpr ScalarType epsilon = std::numeric_limits<ScalarType>::epsilon();
constexpr ScalarType tolerance = epsilon * 1000;
constexpr ScalarType taylor_threshold = static_cast<ScalarType>(1e-5);
constexpr ScalarType infinity = std::numeric_limits<ScalarType>::max() / 10;

// ---- math/matrix.hpp: minimal fixed-size stand-ins for the Eigen typedefs --------------------
struct Vector3Type
{
    ScalarType v[3];
    Vector3Type() : v{0, 0, 0} {}
    Vector3Type(ScalarType x, ScalarType y, ScalarType z) : v{x, y, z} {}
    ScalarType &operator[](size_t i) { return v[i]; }
    const ScalarType &operator[](size_t i) const { return v[i]; }
    ScalarType &operator()(size_t i) { return v[i]; }
    const ScalarType &operator()(size_t i) const { return v[i]; }
    ScalarType x() const { return v[0]; }
    ScalarType y() const { return v[1]; }
    ScalarType z() const { return v[2]; }
    ScalarType norm() const { return std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
    static Vector3Type Zero() { return Vector3Type(); }
};
inline Vector3Type operator-(const Vector3Type &a) { return Vector3Type(-a[0], -a[1], -a[2]); }
inline Vector3Type operator+(const Vector3Type &a, const Vector3Type &b) { return Vector3Type(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }

struct Matrix3Type
{
    ScalarType m[9];  // row-major
    Matrix3Type() { std::memset(m, 0, sizeof(m)); }
    ScalarType &operator()(size_t r, size_t c) { return m[r * 3 + c]; }
    const ScalarType &operator()(size_t r, size_t c) const { return m[r * 3 + c]; }
    const ScalarType *data() const { return m; }
    ScalarType *data() { return m; }
    static Matrix3Type Identity()
    {
        Matrix3Type I;
        I(0, 0) = I(1, 1) = I(2, 2) = 1;
        return I;
    }
    Matrix3Type transpose() const
    {
        Matrix3Type T;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                T(j, i) = (*this)(i, j);
        return T;
    }
    ScalarType trace() const { return (m[0] + m[4]) + m[8]; }
};
inline Matrix3Type operator*(const Matrix3Type &A, const Matrix3Type &B)
{
    Matrix3Type C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C(i, j) = (A(i, 0) * B(0, j) + A(i, 1) * B(1, j)) + A(i, 2) * B(2, j);
    return C;
}
inline Vector3Type operator*(const Matrix3Type &A, const Vector3Type &x)
{
    Vector3Type y;
    for (int i = 0; i < 3; ++i)
        y[i] = (A(i, 0) * x[0] + A(i, 1) * x[1]) + A(i, 2) * x[2];
    return y;
}
using Vector6Type = std::array<ScalarType, 6>;

template <typename T>
constexpr T sqr(T x) { return x * x; }  // math/utility.hpp:8-12

// ---- math/lie-group.{hpp,cpp} -----------------------------------------------------------------
inline Matrix3Type skew_symmetric_matrix(const Vector3Type &v)
{
    Matrix3Type K;
    K(0, 1) = -v[2]; K(0, 2) = v[1];
    K(1, 0) = v[2];  K(1, 2) = -v[0];
    K(2, 0) = -v[1]; K(2, 1) = v[0];
    return K;
}

inline Matrix3Type rodrigues(const Vector3Type &v)  // lie-group.cpp:15-32
{
    const ScalarType theta = v.norm();
    ScalarType A, B;
    if (theta < epsilon) {
        A = 1.0 - sqr(theta) / 6.0;
        B = 0.5 - sqr(theta) / 24.0;
    } else {
        A = std::sin(theta) / theta;
        B = (1.0 - std::cos(theta)) / sqr(theta);
    }
    const Matrix3Type K = skew_symmetric_matrix(v);
    Matrix3Type BK;
    for (int i = 0; i < 9; ++i)
        BK.m[i] = B * K.m[i];
    const Matrix3Type BKK = BK * K;
    Matrix3Type R = Matrix3Type::Identity();
    for (int i = 0; i < 9; ++i)
        R.m[i] = (R.m[i] + A * K.m[i]) + BKK.m[i];
    return R;
}

class SO3
{
public:
    struct already_rectified_t {};
    SO3() : _R(Matrix3Type::Identity()) {}
    explicit SO3(const Matrix3Type &m) : _R(m) { rectify(); }  // lie-group.hpp:31-36
    // adopt a matrix that already went through the SO3 constructor inside the library (no second Gram-Schmidt)
    SO3(const Matrix3Type &m, already_rectified_t) : _R(m) {}
    SO3(ScalarType roll, ScalarType pitch, ScalarType yaw)     // lie-group.hpp:41-56
    {
        Matrix3Type Rx = Matrix3Type::Identity(), Ry = Rx, Rz = Rx;
        Rx(1, 1) = std::cos(roll);  Rx(1, 2) = -std::sin(roll); Rx(2, 1) = std::sin(roll);  Rx(2, 2) = std::cos(roll);
        Ry(0, 0) = std::cos(pitch); Ry(0, 2) = std::sin(pitch); Ry(2, 0) = -std::sin(pitch); Ry(2, 2) = std::cos(pitch);
        Rz(0, 0) = std::cos(yaw);   Rz(0, 1) = -std::sin(yaw);  Rz(1, 0) = std::sin(yaw);   Rz(1, 1) = std::cos(yaw);
        _R = Rz * Ry * Rx;
    }
    SO3(const Vector3Type &so3) : _R(rodrigues(so3)) {}
    const Matrix3Type &get_matrix() const { return _R; }
    SO3 inverse() const { return SO3(_R.transpose()); }
    void rectify()  // lie-group.hpp:84-96: row 1 is not re-normalised
    {
        Vector3Type u0(_R(0, 0), _R(0, 1), _R(0, 2));
        const ScalarType n = u0.norm();
        u0 = Vector3Type(u0[0] / n, u0[1] / n, u0[2] / n);
        Vector3Type u1(_R(1, 0), _R(1, 1), _R(1, 2));
        const ScalarType d = (u1[0] * u0[0] + u1[1] * u0[1]) + u1[2] * u0[2];
        u1 = Vector3Type(u1[0] - d * u0[0], u1[1] - d * u0[1], u1[2] - d * u0[2]);
        const Vector3Type u2(u0[1] * u1[2] - u0[2] * u1[1], u0[2] * u1[0] - u0[0] * u1[2], u0[0] * u1[1] - u0[1] * u1[0]);
        for (int k = 0; k < 3; ++k) {
            _R(0, k) = u0[k];
            _R(1, k) = u1[k];
            _R(2, k) = u2[k];
        }
    }
    ScalarType get_roll() const { return std::atan2(_R(2, 1), _R(2, 2)); }
    ScalarType get_pitch() const { return std::asin(-_R(2, 0)); }
    ScalarType get_yaw() const { return std::atan2(_R(1, 0), _R(0, 0)); }
    Vector3Type operator*(const Vector3Type &v) const { return _R * v; }
    SO3 operator*(const SO3 &rhs) const { return SO3(_R * rhs._R); }
    Vector3Type ln() const  // lie-group.hpp:138-162
    {
        ScalarType c = 0.5 * (_R.trace() - 1.0);
        c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);
        const ScalarType theta = std::acos(c);
        const Vector3Type v(_R(2, 1) - _R(1, 2), _R(0, 2) - _R(2, 0), _R(1, 0) - _R(0, 1));
        const ScalarType A = theta < taylor_threshold ? (1.0 + sqr(theta) / 6.0) * 0.5 : 0.5 * theta / std::sin(theta);
        return Vector3Type(v[0] * A, v[1] * A, v[2] * A);
    }
    static SO3 exp(const Vector3Type &so3) { return SO3(so3); }

private:
    Matrix3Type _R;
};

class SE3
{
public:
    SE3(const SO3 &r, const Vector3Type &t) : _R(r), _t(t) {}
    SE3() : _R(), _t() {}
    const SO3 &rotation() const { return _R; }
    const Vector3Type &translation() const { return _t; }
    SE3 inverse() const  // lie-group.hpp:212-216
    {
        const SO3 RT = _R.inverse();
        return SE3(RT, -(RT * _t));
    }
    Vector3Type operator*(const Vector3Type &v) const { return _R * v + _t; }
    SE3 operator*(const SE3 &rhs) const { return SE3(_R * rhs._R, _R * rhs._t + _t); }
    Vector6Type ln() const  // lie-group.hpp:245-269
    {
        const Vector3Type w = _R.ln();
        const ScalarType theta = w.norm();
        ScalarType G;
        if (theta < taylor_threshold) {
            G = 1.0 / 12.0 + sqr(theta) / 720.0;
        } else {
            const ScalarType A = std::sin(theta) / theta, B = (1.0 - std::cos(theta)) / sqr(theta);
            G = (1.0 - 0.5 * A / B) / sqr(theta);
        }
        const Matrix3Type K = skew_symmetric_matrix(w);
        Matrix3Type GK;
        for (int i = 0; i < 9; ++i)
            GK.m[i] = G * K.m[i];
        const Matrix3Type GKK = GK * K;
        Matrix3Type Vinv = Matrix3Type::Identity();
        for (int i = 0; i < 9; ++i)
            Vinv.m[i] = (Vinv.m[i] - 0.5 * K.m[i]) + GKK.m[i];
        const Vector3Type u = Vinv * _t;
        return Vector6Type{u[0], u[1], u[2], w[0], w[1], w[2]};
    }
    static SE3 exp(const Vector6Type &se3)  // lie-group.hpp:275-299
    {
        const Vector3Type u(se3[0], se3[1], se3[2]), w(se3[3], se3[4], se3[5]);
        const ScalarType theta = w.norm();
        ScalarType A, B, C;
        if (theta < taylor_threshold) {
            A = 1.0 - sqr(theta) / 6.0;
            B = 0.5 - sqr(theta) / 24.0;
            C = 1.0 / 6.0 - sqr(theta) / 120.0;
        } else {
            A = std::sin(theta) / theta;
            B = (1.0 - std::cos(theta)) / sqr(theta);
            C = (1.0 - A) / sqr(theta);
        }
        (void)A;
        const Matrix3Type K = skew_symmetric_matrix(w);
        Matrix3Type CK;
        for (int i = 0; i < 9; ++i)
            CK.m[i] = C * K.m[i];
        const Matrix3Type CKK = CK * K;
        Matrix3Type V = Matrix3Type::Identity();
        for (int i = 0; i < 9; ++i)
            V.m[i] = (V.m[i] + B * K.m[i]) + CKK.m[i];
        return SE3(SO3::exp(w), V * u);
    }

private:
    SO3 _R;
    Vector3Type _t;
};

// ---- base/data-type.hpp:19-32, base/image.hpp:37-51 ---------------------------------------------
using Point3 = Vector3Type;
using Transformation = SE3;
using CameraIntrinsics = Matrix3Type;
using CameraExtrinsics = SE3;
using IdealCameraImagePoint = Vector3Type;
struct ImagePoint  // cv::Point_<double>
{
    ScalarType x, y;
    ImagePoint() : x(0), y(0) {}
    ImagePoint(ScalarType x_, ScalarType y_) : x(x_), y(y_) {}
};
// ---- math/state-estimate.hpp:6-49 and base/data-type.hpp:19-29 -------------------------------------
struct Vector2Type
{
    ScalarType v[2];
    Vector2Type() : v{0, 0} {}
    Vector2Type(ScalarType x, ScalarType y) : v{x, y} {}
    ScalarType &operator[](size_t i) { return v[i]; }
    const ScalarType &operator[](size_t i) const { return v[i]; }
};
template <int N>
struct SquareMatrix  // row-major N x N stand-in for Eigen::Matrix<ScalarType, N, N>
{
    ScalarType m[N * N];
    SquareMatrix() { std::memset(m, 0, sizeof(m)); }
    static SquareMatrix Identity()
    {
        SquareMatrix I;
        for (int i = 0; i < N; ++i)
            I.m[i * N + i] = 1;
        return I;
    }
    ScalarType &operator()(size_t r, size_t c) { return m[r * N + c]; }
    const ScalarType &operator()(size_t r, size_t c) const { return m[r * N + c]; }
    const ScalarType *data() const { return m; }
};
template <int N>
inline SquareMatrix<N> operator*(ScalarType s, const SquareMatrix<N> &A)
{
    SquareMatrix<N> B;
    for (int i = 0; i < N * N; ++i)
        B.m[i] = s * A.m[i];
    return B;
}
using Matrix2Type = SquareMatrix<2>;
using Matrix6Type = SquareMatrix<6>;
template <typename MeanType, typename CovarType>
class StateEstimate
{
public:
    StateEstimate() {}
    StateEstimate(const MeanType &mean, const CovarType &covar) : _mean(mean), _covar(covar) {}
    const MeanType &mean() const { return _mean; }
    MeanType &mean() { return _mean; }
    const CovarType &covar() const { return _covar; }
    CovarType &covar() { return _covar; }

private:
    MeanType _mean;
    CovarType _covar;
};
using TransformationUncertainty = Matrix6Type;
using TransformationEstimate = StateEstimate<Transformation, TransformationUncertainty>;
using Point3Uncertainty = Matrix3Type;
using Point3Estimate = StateEstimate<Point3, Point3Uncertainty>;
using Point2 = Vector2Type;
using Point2Uncertainty = Matrix2Type;
using Point2Estimate = StateEstimate<Point2, Point2Uncertainty>;

struct Point2f { float x, y; };
struct KeyPoint  // cv::KeyPoint layout
{
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
using DMatch = mvs_match;  // cv::DMatch layout: queryIdx, trainIdx, imgIdx, distance
struct Mat8u               // cv::Mat CV_8U, rows x cols, row-major, not owning
{
    int rows = 0, cols = 0;
    std::vector<uint8_t> data;
    const uint8_t *row(int r) const { return data.data() + (size_t)r * cols; }
    void push_back_row(const uint8_t *p)
    {
        data.insert(data.end(), p, p + cols);
        ++rows;
    }
};
struct VisualFeatureConfig
{
    using DetectorResultType = std::vector<KeyPoint>;
    using ExtractorResultType = Mat8u;
    using MatchResultType = std::vector<DMatch>;
};

// ---- the HIP backend handle ---------------------------------------------------------------------
namespace hip
{
struct RansacConfig  // what the reference hard-codes (sfm-solve.cpp:67, estimator-RANSAC.cpp:41-42)
{
    int num_hypotheses = 1;
    int sampler = MVS_SAMPLER_IDENTITY;
    uint64_t seed = 0;
    double max_error_sq = 0.0;  // <= 0: 5e-2 / K00 / K11 (sfm-solve.cpp:311)
};
inline RansacConfig &ransac_config()
{
    static thread_local RansacConfig cfg;
    return cfg;
}
// one mvs_ctx per host thread; throws if there is no HIP device (there is no CPU fallback)
inline mvs_ctx *context()
{
    struct Holder
    {
        mvs_ctx *ctx = nullptr;
        ~Holder() { if (ctx) mvs_ctx_destroy(ctx); }
    };
    static thread_local Holder h;
    if (!h.ctx) {
        const mvs_status st = mvs_ctx_create(0, &h.ctx);
        if (st != MVS_OK)
            throw std::runtime_error(std::string("mvSLAM HIP backend: ") + mvs_status_str(st));
    }
    return h.ctx;
}
inline void check(mvs_status st, const char *what)
{
    if (st < 0)  // negative = the reference's assert()s / runtime failure
        throw std::runtime_error(std::string(what) + ": " + mvs_status_str(st) + " [" + mvs_last_error(context()) + "]");
}
}  // namespace hip

// ---- vision/camera.{hpp,cpp} (normalise / project only) -------------------------------------------
class PinholeCamera
{
public:
    PinholeCamera(const CameraIntrinsics &K_, const CameraExtrinsics &P_) : K(K_), P(P_) {}
    explicit PinholeCamera(const std::string &filename)  // camera.cpp:8-12,105-124: "fx fy shear px py" then an se3 line
    {
        const bool ok = load_from_file(filename);
        assert(ok);
        (void)ok;
    }
    bool load_from_file(const std::string &filename)
    {
        std::ifstream in(filename);
        K = Matrix3Type();
        in >> K(0, 0) >> K(1, 1) >> K(0, 1) >> K(0, 2) >> K(1, 2);
        K(2, 2) = 1;
        Vector6Type se3{};
        for (int i = 0; i < 6; ++i)
            in >> se3[i];
        P = SE3::exp(se3);
        return bool(in);
    }
    ImagePoint project_point(const Point3 &p_world) const  // camera.cpp:24-37
    {
        const Vector3Type pc = P * p_world;
        assert(pc[2] > 0);
        const Vector3Type ph = K * pc;
        return ImagePoint(ph[0] / ph[2], ph[1] / ph[2]);
    }
    std::vector<ImagePoint> project_points(const std::vector<Point3> &pts) const
    {
        std::vector<ImagePoint> r;
        r.reserve(pts.size());
        for (const auto &p : pts)
            r.push_back(project_point(p));
        return r;
    }
    const CameraIntrinsics &get_intrinsics() const { return K; }
    const CameraExtrinsics &get_extrinsics() const { return P; }

private:
    CameraIntrinsics K;
    CameraExtrinsics P;
};

// ---- vision/visual-feature.{hpp,cpp} --------------------------------------------------------------
class VisualFeature
{
public:
    VisualFeature() : m_image_width(-1), m_image_height(-1) {}
    VisualFeature(std::vector<KeyPoint> kp, Mat8u desc, int w, int h)
        : m_keypoints(std::move(kp)), m_descriptors(std::move(desc)), m_image_width(w), m_image_height(h) {}

    // visual-feature.cpp:51-80.  a vector of {trainIdx -> vf1, queryIdx -> vf2, distance}
    static VisualFeatureConfig::MatchResultType match_visual_features(const VisualFeature &vf1, const VisualFeature &vf2,
                                                                     ScalarType max_dist = -1)
    {
        assert(vf1.valid() && vf2.valid());
        VisualFeatureConfig::MatchResultType out(vf2.size());
        int n = 0;
        hip::check(mvs_match_hamming(hip::context(), vf1.m_descriptors.data.data(), (int)vf1.size(),
                                     vf2.m_descriptors.data.data(), (int)vf2.size(), vf1.m_descriptors.cols, 0.7,
                                     max_dist, out.data(), &n),
                   "match_visual_features");
        out.resize(n);
        return out;
    }
    // visual-feature.cpp:93-119 (the reference pushes vf2's descriptor rows into filtered1, :115; fixed here)
    static std::pair<VisualFeature, VisualFeature> match_and_filter_visual_features(const VisualFeature &vf1,
                                                                                   const VisualFeature &vf2,
                                                                                   ScalarType max_dist = -1)
    {
        const auto matches = match_visual_features(vf1, vf2, max_dist);
        assert(vf1.m_image_height == vf2.m_image_height && vf1.m_image_width == vf2.m_image_width);
        VisualFeature f1, f2;
        f1.m_image_height = f2.m_image_height = vf1.m_image_height;
        f1.m_image_width = f2.m_image_width = vf1.m_image_width;
        f1.m_descriptors.cols = f2.m_descriptors.cols = vf1.m_descriptors.cols;
        for (const auto &m : matches) {
            f1.m_keypoints.push_back(vf1.m_keypoints[m.trainIdx]);
            f1.m_descriptors.push_back_row(vf1.m_descriptors.row(m.trainIdx));
            f2.m_keypoints.push_back(vf2.m_keypoints[m.queryIdx]);
            f2.m_descriptors.push_back_row(vf2.m_descriptors.row(m.queryIdx));
        }
        return std::make_pair(f1, f2);
    }
    // VisualFeature::extract (visual-feature.cpp:40-49): _detector->detect + _extractor->compute with
    // cv::ORB::create(MAX_FEATURE_COUNT = 500) -> mvs_extract (ORB's pipeline, the library's own pattern: DESIGN.md 4.8)
    static VisualFeature extract(const Mat8u &image)
    {
        static_assert(sizeof(KeyPoint) == sizeof(mvs_keypoint), "KeyPoint has cv::KeyPoint's layout");
        assert(image.rows > 0 && image.cols > 0);
        mvs_orb_params prm;
        mvs_orb_params_default(&prm);
        std::vector<KeyPoint> kp(prm.nfeatures);
        Mat8u desc;
        desc.cols = 32;
        desc.data.resize((size_t)prm.nfeatures * 32);
        int32_t n = 0;
        hip::check(mvs_extract(hip::context(), image.data.data(), 1, image.cols, image.rows, &prm,
                               reinterpret_cast<mvs_keypoint *>(kp.data()), desc.data.data(), &n),
                   "VisualFeature::extract");
        kp.resize(n);
        desc.data.resize((size_t)n * 32);
        desc.rows = n;
        return VisualFeature(kp, desc, image.cols, image.rows);
    }
    size_t size() const { return m_keypoints.size(); }
    bool valid() const { return size() > 0 && m_image_width > 0 && m_image_height > 0; }  // :209-213
    const std::vector<KeyPoint> &get_keypoints() const { return m_keypoints; }
    const Mat8u &get_descriptors() const { return m_descriptors; }
    std::vector<ImagePoint> get_image_points() const  // :179-190
    {
        std::vector<ImagePoint> r;
        r.reserve(m_keypoints.size());
        for (const auto &kp : m_keypoints)
            r.emplace_back(kp.pt.x, kp.pt.y);
        return r;
    }

    std::vector<Point2Estimate> get_point_estimates() const  // :193-207: sigma = 2^octave * 0.5 px, isotropic
    {
        std::vector<Point2Estimate> r;
        r.reserve(m_keypoints.size());
        for (const auto &kp : m_keypoints) {
            const ScalarType stddev = static_cast<ScalarType>(1 << kp.octave) * 0.5;
            r.emplace_back(Point2(kp.pt.x, kp.pt.y), sqr(stddev) * Point2Uncertainty::Identity());
        }
        return r;
    }

private:
    std::vector<KeyPoint> m_keypoints;
    Mat8u m_descriptors;
    int m_image_width, m_image_height;
};

// ---- vision/fundamental-matrix.hpp:16-19 ----------------------------------------------------------
inline bool find_fundamental_matrix(const std::vector<Vector3Type> &p1_sample, const std::vector<Vector3Type> &p2_sample,
                                    Matrix3Type &F21)
{
    assert(p1_sample.size() == 8 && p2_sample.size() == 8);
    double a[16], b[16];
    for (int i = 0; i < 8; ++i) {
        a[2 * i] = p1_sample[i][0]; a[2 * i + 1] = p1_sample[i][1];
        b[2 * i] = p2_sample[i][0]; b[2 * i + 1] = p2_sample[i][1];
    }
    const mvs_status st = mvs_find_fundamental_matrix(hip::context(), a, b, F21.data());
    hip::check(st, "find_fundamental_matrix");
    return st == MVS_OK;
}

// ---- vision/estimator-RANSAC.hpp:10-50 -----------------------------------------------------------
class FundamentalMatrixEstimatorRANSAC
{
public:
    FundamentalMatrixEstimatorRANSAC(ScalarType max_error_sq_, size_t max_iteration_)
        : max_error_sq(max_error_sq_), max_iteration(max_iteration_)
    {
        assert(max_error_sq > epsilon);
        assert(max_iteration > 0);
    }
    bool compute(const std::vector<Vector3Type> &p1, const std::vector<Vector3Type> &p2, Matrix3Type &F21,
                 std::vector<uint8_t> &inlier_mask)
    {
        assert(p1.size() == p2.size());
        const size_t n = p1.size();
        std::vector<double> a(2 * n), b(2 * n);
        for (size_t i = 0; i < n; ++i) {
            a[2 * i] = p1[i][0]; a[2 * i + 1] = p1[i][1];
            b[2 * i] = p2[i][0]; b[2 * i + 1] = p2[i][1];
        }
        std::vector<uint8_t> mask(n ? n : 1);
        const auto &cfg = hip::ransac_config();
        const mvs_status st = mvs_ransac_fundamental(hip::context(), a.data(), b.data(), (int)n, max_error_sq,
                                                     (int)max_iteration, cfg.sampler, cfg.seed, F21.data(), mask.data(),
                                                     nullptr, nullptr, nullptr, nullptr, nullptr);
        hip::check(st, "FundamentalMatrixEstimatorRANSAC::compute");
        mask.resize(n);
        if (n >= 8)
            inlier_mask.swap(mask);
        return st == MVS_OK;
    }

private:
    const ScalarType max_error_sq;
    const size_t max_iteration;
};

// ---- vision/sfm.hpp:30-53 -------------------------------------------------------------------------
inline mvs_params make_params_()
{
    mvs_params p;
    mvs_params_default(&p);
    const auto &cfg = hip::ransac_config();
    p.num_hypotheses = cfg.num_hypotheses;
    p.sampler = cfg.sampler;
    p.seed = cfg.seed;
    p.max_error_sq = cfg.max_error_sq;
    return p;
}
inline SE3 se3_from_arrays_(const double R[9], const double t[3])
{
    Matrix3Type Rm;
    std::memcpy(Rm.data(), R, sizeof(double) * 9);
    // the library already returns SE3(SO3(R1to2), t1to2).inverse() (sfm-solve.cpp:364), rectified as the reference does
    return SE3(SO3(Rm, SO3::already_rectified_t()), Vector3Type(t[0], t[1], t[2]));
}

inline bool sfm_solve(const std::vector<ImagePoint> &p1, const std::vector<ImagePoint> &p2, const CameraIntrinsics &K,
                      Transformation &pose2in1_scaled, std::vector<Point3> &pointsin1_scaled,
                      std::vector<size_t> &point_indexes)
{
    assert(p1.size() == p2.size());
    const int m = (int)p1.size();
    static_assert(sizeof(ImagePoint) == 2 * sizeof(double), "ImagePoint is two packed doubles");
    std::vector<double> pts(3 * (size_t)(m ? m : 1));
    std::vector<int64_t> idx(m ? m : 1);
    double R[9], t[3];
    int n = 0;
    const mvs_params prm = make_params_();
    const mvs_status st = mvs_two_view(hip::context(), m ? &p1[0].x : nullptr, m ? &p2[0].x : nullptr, m, K.data(), &prm,
                                       R, t, pts.data(), idx.data(), &n, nullptr, nullptr);
    hip::check(st, "sfm_solve");
    if (st != MVS_OK)
        return false;
    pose2in1_scaled = se3_from_arrays_(R, t);
    std::vector<Point3> P(n);
    std::vector<size_t> I(n);
    for (int i = 0; i < n; ++i) {
        P[i] = Point3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        I[i] = (size_t)idx[i];
    }
    pointsin1_scaled.swap(P);
    point_indexes.swap(I);
    return true;
}

inline void sfm_triangulate(const std::vector<ImagePoint> &p1, const std::vector<ImagePoint> &p2, const CameraIntrinsics &K,
                            const Transformation &pose1, const Transformation &pose2, std::vector<Point3> &points,
                            std::vector<size_t> &point_indexes)
{
    assert(p1.size() == p2.size() && !p1.empty());
    const Transformation T_1_to_2 = pose2.inverse() * pose1;  // sfm-solve.cpp:381
    const int m = (int)p1.size();
    std::vector<double> pts(3 * (size_t)m);
    std::vector<int64_t> idx(m);
    int n = 0;
    const Vector3Type &t = T_1_to_2.translation();
    hip::check(mvs_triangulate(hip::context(), &p1[0].x, &p2[0].x, m, K.data(), T_1_to_2.rotation().get_matrix().data(),
                               t.v, pts.data(), idx.data(), &n),
               "sfm_triangulate");
    std::vector<Point3> P(n);
    std::vector<size_t> I(n);
    for (int i = 0; i < n; ++i) {
        P[i] = Point3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        I[i] = (size_t)idx[i];
    }
    points.swap(P);
    point_indexes.swap(I);
}

// ---- vision/pnp.hpp:22-26 -------------------------------------------------------------------------
namespace hip
{
struct PnpConfig  // what the reference hard-codes in pnp-solve.cpp:47-49
{
    int num_hypotheses = 100;
    int sampler = MVS_SAMPLER_PHILOX;
    uint64_t seed = 0;
    double reproj_error = 0.05;
    int refit = 1;   // cv::solvePnPRansac refits the pose on all inliers (pnp-solve.cpp:53-64)
};
inline PnpConfig &pnp_config()
{
    static thread_local PnpConfig cfg;
    return cfg;
}
}  // namespace hip

inline bool pnp_solve(const std::vector<Point3> &world_points, const std::vector<ImagePoint> &image_points,
                      const CameraIntrinsics &K, Transformation &pose, std::vector<size_t> &inlier_point_indexes)
{
    assert(world_points.size() >= 7);  // PNP_MIN_POINT_COUNT, pnp-solve.cpp:13,22
    assert(world_points.size() == image_points.size());
    const int n = (int)world_points.size();
    static_assert(sizeof(Point3) == 3 * sizeof(double), "Point3 is three packed doubles");
    mvs_pnp_params prm;
    mvs_pnp_params_default(&prm);
    const auto &cfg = hip::pnp_config();
    prm.num_hypotheses = cfg.num_hypotheses;
    prm.sampler = cfg.sampler;
    prm.seed = cfg.seed;
    prm.reproj_error = cfg.reproj_error;
    prm.refit = cfg.refit;
    std::vector<int64_t> idx(n);
    double R[9], t[3];
    int ni = 0;
    const mvs_status st = mvs_pnp_solve(hip::context(), world_points[0].v, &image_points[0].x, n, K.data(), &prm, R, t,
                                        idx.data(), &ni, nullptr);
    hip::check(st, "pnp_solve");
    if (st != MVS_OK)
        return false;
    inlier_point_indexes.reserve(inlier_point_indexes.size() + ni);  // the reference appends (pnp-solve.cpp:69-73)
    for (int i = 0; i < ni; ++i)
        inlier_point_indexes.push_back((size_t)idx[i]);
    pose = se3_from_arrays_(R, t);  // already SE3(R, t).inverse() (pnp-solve.cpp:99-101)
    return true;
}

// ---- vision/sfm.hpp:56-76 and vision/pnp.hpp:28-46 (row f4: the two callers of ba_frame_pose_and_point) ---------
namespace hip
{
inline mvs_refine_params &refine_config()
{
    static thread_local mvs_refine_params cfg = [] {
        mvs_refine_params p;
        mvs_refine_params_default(&p);
        return p;
    }();
    return cfg;
}
inline TransformationEstimate estimate_from_result_(const mvs_refine_result &r)
{
    TransformationUncertainty C;
    std::memcpy(C.m, r.pose_cov, sizeof(C.m));
    Matrix3Type Rm;
    std::memcpy(Rm.m, r.R, sizeof(Rm.m));
    // the optimiser's rotation is orthonormal to rounding; the reference's Pose3_to_SE3 goes through SO3(Matrix3)
    return TransformationEstimate(SE3(SO3(Rm), Vector3Type(r.t[0], r.t[1], r.t[2])), C);
}
}  // namespace hip

// ---- vision/ba.hpp:25-36: the one function that talks to GTSAM (ba.cpp:26-156) ------------------------------------
class Id  // base/data-type.hpp:12-17
{
public:
    using Type = std::size_t;
    static constexpr Type INVALID = static_cast<Type>(-1);
};
using PointIdToPoint2Estimate = std::unordered_map<Id::Type, Point2Estimate>;

// Same signature as the reference.  Supported: the configurations the reference builds -- one or two frames
// (sfm_refine, pnp_refine, VisualOdometer::track_refine); diagonal frame priors.  With this one function forwarded,
// sfm-refine.cpp, pnp-refine.cpp and visual-odometer.cpp stay untouched and GTSAM leaves the link line.
inline void ba_frame_pose_and_point(const CameraIntrinsics &ci, const std::unordered_set<Id::Type> &frame_id,
                                    const std::unordered_set<Id::Type> &point_id,
                                    const std::unordered_map<Id::Type, Transformation> &frame_pose_guess,
                                    const std::unordered_map<Id::Type, TransformationUncertainty> &frame_pose_prior,
                                    const std::unordered_map<Id::Type, Point3> &point_guess,
                                    const std::unordered_map<Id::Type, Point3Uncertainty> &point_prior,
                                    const std::unordered_map<Id::Type, PointIdToPoint2Estimate> &frame_observation,
                                    std::unordered_map<Id::Type, TransformationEstimate> &frame_pose_estimate,
                                    std::unordered_map<Id::Type, Point3Estimate> &point_estimate, ScalarType &final_error)
{
    assert(frame_id.size() > 0 && frame_id.size() <= 2);   // ba.cpp:39; more than two frames: not built
    assert(point_id.size() > 0);
    assert(frame_pose_guess.size() == frame_id.size() && point_guess.size() == point_id.size());
    assert(frame_pose_prior.size() + point_prior.size() >= 2);   // ba.cpp:43
    std::vector<Id::Type> fids(frame_id.begin(), frame_id.end()), pids(point_id.begin(), point_id.end());
    std::sort(fids.begin(), fids.end());
    std::sort(pids.begin(), pids.end());
    const int F = (int)fids.size(), m = (int)pids.size();
    std::unordered_map<Id::Type, int> pidx;
    for (int i = 0; i < m; ++i)
        pidx[pids[i]] = i;
    std::vector<double> pose(12 * (size_t)F), var(6 * (size_t)F, 0.0), pts(3 * (size_t)m), pcov(9 * (size_t)m, 0.0);
    std::vector<double> obs[2], ocov[2];
    std::vector<uint8_t> valid[2];
    for (int f = 0; f < F; ++f) {
        const Transformation &T = frame_pose_guess.at(fids[f]);
        std::memcpy(&pose[12 * f], T.rotation().get_matrix().m, 9 * sizeof(double));
        std::memcpy(&pose[12 * f + 9], T.translation().v, 3 * sizeof(double));
        auto pr = frame_pose_prior.find(fids[f]);
        if (pr != frame_pose_prior.end())
            for (int k = 0; k < 6; ++k)
                var[6 * f + k] = pr->second(k, k);
        obs[f].assign(2 * (size_t)m, 0.0);
        ocov[f].assign(4 * (size_t)m, 0.0);
        valid[f].assign(m, 0);
        auto ob = frame_observation.find(fids[f]);
        if (ob != frame_observation.end())
            for (const auto &kv : ob->second) {
                const int i = pidx.at(kv.first);
                std::memcpy(&obs[f][2 * i], kv.second.mean().v, 2 * sizeof(double));
                std::memcpy(&ocov[f][4 * i], kv.second.covar().m, 4 * sizeof(double));
                valid[f][i] = 1;
            }
    }
    for (int i = 0; i < m; ++i) {
        std::memcpy(&pts[3 * i], point_guess.at(pids[i]).v, 3 * sizeof(double));
        auto pr = point_prior.find(pids[i]);
        if (pr != point_prior.end())
            std::memcpy(&pcov[9 * i], pr->second.m, 9 * sizeof(double));
    }
    mvs_ba_problem pb;
    std::memset(&pb, 0, sizeof(pb));
    pb.n_frames = F;
    pb.n_points = m;
    pb.K = ci.data();
    pb.frame_pose = pose.data();
    pb.frame_prior_var = var.data();
    pb.points = pts.data();
    pb.point_prior_cov = pcov.data();
    for (int f = 0; f < F; ++f) {
        pb.obs[f] = obs[f].data();
        pb.obs_cov[f] = ocov[f].data();
        pb.obs_valid[f] = valid[f].data();
    }
    std::vector<mvs_refine_result> res(F);
    std::vector<double> po(3 * (size_t)m), pc(9 * (size_t)m);
    const mvs_status st = mvs_ba_refine(hip::context(), &pb, &hip::refine_config(), res.data(), po.data(), pc.data());
    hip::check(st, "ba_frame_pose_and_point");
    if (st != MVS_OK)
        throw std::runtime_error("ba_frame_pose_and_point: indeterminate system");   // GTSAM throws here as well
    frame_pose_estimate.clear();
    for (int f = 0; f < F; ++f)
        frame_pose_estimate[fids[f]] = hip::estimate_from_result_(res[f]);
    point_estimate.clear();
    for (int i = 0; i < m; ++i) {
        Point3Uncertainty C;
        std::memcpy(C.m, &pc[9 * (size_t)i], sizeof(C.m));
        point_estimate[pids[i]] = Point3Estimate(Point3(po[3 * i], po[3 * i + 1], po[3 * i + 2]), C);
    }
    final_error = res[0].error;
}

inline bool sfm_refine(const std::vector<Point2Estimate> &p1_estimate, const std::vector<Point2Estimate> &p2_estimate,
                       const CameraIntrinsics &ci, const Transformation &pose2in1_guess,
                       const std::vector<Point3> pointsin1_guess, TransformationEstimate &pose2in1_estimate,
                       std::vector<Point3Estimate> &pointsin1_estimate, ScalarType &error)
{
    assert(p1_estimate.size() == p2_estimate.size());      // sfm-refine.cpp:29-30
    assert(p1_estimate.size() == pointsin1_guess.size());
    const int m = (int)p1_estimate.size();
    std::vector<double> p1(2 * (size_t)m), p2(2 * (size_t)m), c1(4 * (size_t)m), c2(4 * (size_t)m), pts(3 * (size_t)m),
        cov(9 * (size_t)m);
    for (int i = 0; i < m; ++i) {
        std::memcpy(&p1[2 * i], p1_estimate[i].mean().v, 2 * sizeof(double));
        std::memcpy(&p2[2 * i], p2_estimate[i].mean().v, 2 * sizeof(double));
        std::memcpy(&c1[4 * i], p1_estimate[i].covar().m, 4 * sizeof(double));
        std::memcpy(&c2[4 * i], p2_estimate[i].covar().m, 4 * sizeof(double));
    }
    const Matrix3Type Rg = pose2in1_guess.rotation().get_matrix();
    const Vector3Type tg = pose2in1_guess.translation();
    mvs_refine_result res;
    const mvs_status st = mvs_sfm_refine(hip::context(), p1.data(), c1.data(), p2.data(), c2.data(), m, ci.data(), Rg.m, tg.v,
                                         pointsin1_guess[0].v, &hip::refine_config(), &res, pts.data(), cov.data());
    hip::check(st, "sfm_refine");
    if (st != MVS_OK)
        return false;
    pose2in1_estimate = hip::estimate_from_result_(res);
    pointsin1_estimate.clear();
    pointsin1_estimate.reserve(m);
    for (int i = 0; i < m; ++i) {
        Point3Uncertainty C;
        std::memcpy(C.m, &cov[9 * (size_t)i], sizeof(C.m));
        pointsin1_estimate.emplace_back(Point3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), C);
    }
    error = res.error;
    return true;
}

inline bool pnp_refine(const std::vector<Point3Estimate> &world_point_estimates,
                       const std::vector<Point2Estimate> &image_point_estimates, const CameraIntrinsics &ci,
                       const Transformation &pose_guess, TransformationEstimate &pose_estimate, ScalarType &error)
{
    assert(world_point_estimates.size() == image_point_estimates.size());  // pnp-refine.cpp:21
    const int m = (int)world_point_estimates.size();
    std::vector<double> X(3 * (size_t)m), XC(9 * (size_t)m), uv(2 * (size_t)m), uc(4 * (size_t)m);
    for (int i = 0; i < m; ++i) {
        std::memcpy(&X[3 * i], world_point_estimates[i].mean().v, 3 * sizeof(double));
        std::memcpy(&XC[9 * i], world_point_estimates[i].covar().m, 9 * sizeof(double));
        std::memcpy(&uv[2 * i], image_point_estimates[i].mean().v, 2 * sizeof(double));
        std::memcpy(&uc[4 * i], image_point_estimates[i].covar().m, 4 * sizeof(double));
    }
    const Matrix3Type Rg = pose_guess.rotation().get_matrix();
    const Vector3Type tg = pose_guess.translation();
    mvs_refine_result res;
    const mvs_status st = mvs_pnp_refine(hip::context(), X.data(), XC.data(), uv.data(), uc.data(), m, ci.data(), Rg.m, tg.v,
                                         &hip::refine_config(), &res);
    hip::check(st, "pnp_refine");
    if (st != MVS_OK)
        return false;
    pose_estimate = hip::estimate_from_result_(res);  // the points are not updated (pnp-refine.cpp:103-104)
    error = res.error;
    return true;
}

// ---- front-end/image-pair.{hpp,cpp}: ctor + reconstruct + refine --------------------------------------------------
struct Frame
{
    uint32_t id;
    VisualFeature visual_feature;
};
class ImagePair
{
public:
    struct MatchedPoint
    {
        Point3 position;
        size_t vf_idx_in_base, vf_idx_in_pair;
    };
    struct Params
    {
        ScalarType max_match_inlier_distance;   // image-pair.cpp:22-23
        bool refine_structure_in_constructor;   // image-pair.cpp:25-26
    };
    enum class State { INVALID, RECONSTRUCTED, REFINED };  // image-pair.hpp
    static Params get_default_params() { return Params{10, false}; }  // image-pair.cpp:17-28
    ImagePair(const Frame &base_frame_, const Frame &pair_frame_, const CameraIntrinsics &K,
              const Params &params = get_default_params())
        : valid(false), match_inlier_count(0), match_inlier_ssd(0), error(infinity), m_state(State::INVALID),
          m_base(&base_frame_), m_pair(&pair_frame_), m_K(K), m_params(params)   // error(infinity): image-pair.cpp:41
    {
        assert(base_frame_.id != pair_frame_.id);
        // match(base = train, pair = query) -> gather -> sfm_solve (image-pair.cpp:57-65,116-174) as ONE device pass
        // (mvs_image_pair): same results as the two calls, one upload / synchronisation / download instead of two
        const VisualFeature &vb = base_frame_.visual_feature, &vp = pair_frame_.visual_feature;
        assert(vb.valid() && vp.valid());
        auto pack = [](const VisualFeature &vf) {
            std::vector<float> xy(2 * vf.size());
            for (size_t i = 0; i < vf.size(); ++i) {
                xy[2 * i] = vf.get_keypoints()[i].pt.x;
                xy[2 * i + 1] = vf.get_keypoints()[i].pt.y;
            }
            return xy;
        };
        const std::vector<float> kb = pack(vb), kq = pack(vp);
        const int nq = (int)vp.size();
        mvs_params prm = make_params_();
        prm.ratio = 0.7;                                   // visual-feature.cpp:23
        prm.max_dist = params.max_match_inlier_distance;
        std::vector<DMatch> matches(nq);
        std::vector<double> pts(3 * (size_t)nq);
        std::vector<int64_t> idx(nq);
        mvs_pair_result res;
        const mvs_status st = mvs_image_pair(hip::context(), vb.get_descriptors().data.data(), kb.data(), (int)vb.size(),
                                             vp.get_descriptors().data.data(), kq.data(), nq, vb.get_descriptors().cols,
                                             K.data(), &prm, &res, matches.data(), nullptr,
                                             pts.data(), idx.data());
        hip::check(st, "ImagePair");
        valid = st == MVS_OK;
        if (valid) {
            T_pair_to_base = se3_from_arrays_(res.R, res.t);
            match_inlier_count = (uint32_t)res.n_points;
            for (int k = 0; k < res.n_points; ++k) {    // point k belongs to match idx[k] (the reference indexes
                const auto &m = matches[idx[k]];          // points[idx], image-pair.cpp:164 -- SURVEY Q12, not copied)
                matched_points.push_back(MatchedPoint{Point3(pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]),
                                                      (size_t)m.trainIdx, (size_t)m.queryIdx});
                match_inlier_ssd += (uint32_t)sqr(m.distance);
            }
            m_state = State::RECONSTRUCTED;
        }
        if (valid && params.refine_structure_in_constructor)  // image-pair.cpp:67-70
            refine();
    }
    // image-pair.cpp:176-238: sfm_refine on the matched keypoints' estimates, pose and points replaced on success.
    // The frames passed to the constructor must still be alive (the reference keeps shared_ptrs).
    bool refine()
    {
        assert(State::RECONSTRUCTED == m_state);
        assert(valid);
        const auto be = m_base->visual_feature.get_point_estimates(), pe = m_pair->visual_feature.get_point_estimates();
        std::vector<Point2Estimate> base_pe, pair_pe;
        std::vector<Point3> points;
        for (const auto &mp : matched_points) {
            base_pe.push_back(be[mp.vf_idx_in_base]);
            pair_pe.push_back(pe[mp.vf_idx_in_pair]);
            points.push_back(mp.position);
        }
        std::vector<Point3Estimate> point_estimates;
        TransformationEstimate T_est;
        valid = sfm_refine(base_pe, pair_pe, m_K, T_pair_to_base, points, T_est, point_estimates, error);
        if (valid) {
            T_pair_to_base = T_est.mean();
            T_pair_to_base_covar = T_est.covar();
            matched_points_covar.clear();
            for (size_t i = 0; i < matched_points.size(); ++i) {
                matched_points[i].position = point_estimates[i].mean();
                matched_points_covar.push_back(point_estimates[i].covar());
            }
            m_state = State::REFINED;
        }
        return valid;
    }
    // image-pair.cpp:77-114: would (base_frame, new_frame) make a better pair?  A light-weight reconstruction of the
    // candidate; it must not have fewer inliers nor a smaller descriptor SSD than this pair (the reference's two tests as
    // written, :96-100), is then refined, and replaces *this only if its refined error is below this pair's `error`
    // (infinity until refine() has run, :41).  `new_frame` must outlive the pair, as the frames of the constructor do.
    bool update(const Frame &new_frame)
    {
        if (new_frame.id == m_base->id || new_frame.id == m_pair->id)
            return false;
        Params params = m_params;
        params.refine_structure_in_constructor = false;
        ImagePair new_image_pair(*m_base, new_frame, m_K, params);
        if (!new_image_pair.valid)
            return false;
        if (new_image_pair.match_inlier_count < match_inlier_count || new_image_pair.match_inlier_ssd < match_inlier_ssd)
            return false;
        new_image_pair.refine();          // (result ignored as in :103: a failed refinement leaves error = infinity)
        if (new_image_pair.error < error) {
            std::swap(*this, new_image_pair);
            return true;
        }
        return false;
    }
    const Frame &base_frame() const { return *m_base; }
    const Frame &pair_frame() const { return *m_pair; }
    State state() const { return m_state; }
    bool valid;
    uint32_t match_inlier_count;
    uint32_t match_inlier_ssd;
    ScalarType error;
    Transformation T_pair_to_base;
    TransformationUncertainty T_pair_to_base_covar;
    std::vector<MatchedPoint> matched_points;
    std::vector<Point3Uncertainty> matched_points_covar;

private:
    State m_state;
    const Frame *m_base, *m_pair;
    CameraIntrinsics m_K;
    Params m_params;
};

}  // namespace mvSLAM
