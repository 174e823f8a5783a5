// C++ tests of the drop-in call surface (mvslam_amd/compat/mvslam_compat.hpp), written the way the reference's own
// tests read (test/test-sfm.cpp, test/test-lie-group.cpp, test/unit-test-helper.cpp) -- same rigs, same tolerances.
// Build + run: tests/test_compat_cpp.py (needs a GPU to run; compiles anywhere).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../mvslam_amd/compat/mvslam_compat.hpp"

static int g_fail = 0;
#define ASSERT_TRUE(c) do { if (!(c)) { std::printf("  FAILED %s:%d  %s\n", __FILE__, __LINE__, #c); ++g_fail; return; } } while (0)
#define ASSERT_EQUAL(a, b, tol) ASSERT_TRUE(std::fabs((a) - (b)) <= (tol))
#define RUN(t) do { int before = g_fail; std::printf("[ RUN  ] %s\n", #t); t(); std::printf(g_fail == before ? "[  OK  ] %s\n" : "[ FAIL ] %s\n", #t); } while (0)

using namespace mvSLAM;

enum class RIG_TYPE { CUBE, L_SHAPE };
static std::vector<Vector3Type> get_rig_points(RIG_TYPE type, const SO3 &rotation, const Vector3Type &translation, ScalarType scale)
{   // test/unit-test-helper.cpp:42-79
    std::vector<Vector3Type> p;
    if (type == RIG_TYPE::CUBE) {
        for (int x = -1; x <= 1; x += 2) for (int y = -1; y <= 1; y += 2) for (int z = -1; z <= 1; z += 2) p.emplace_back(x, y, z);
    } else {
        p = {{1, 0, 0}, {0, 0, 0}, {0, 2, 0}, {1, 0, 3}, {0, 0, 3}, {0, 2, 3}, {0.5, 0.0, 1.5}, {0.0, 1.0, 1.5}};
    }
    for (auto &q : p)
        q = rotation * Vector3Type(scale * q[0], scale * q[1], scale * q[2]) + translation;
    return p;
}

struct Rig
{
    CameraIntrinsics K = Matrix3Type::Identity();
    CameraExtrinsics P1, P2;
    std::vector<Point3> X;
    std::vector<ImagePoint> ip1, ip2;
};
static Rig make_rig(RIG_TYPE type, const SO3 &rot, ScalarType scale)
{   // test/test-sfm.cpp:19-42
    Rig r;
    Vector6Type se3_2to1{1, 0, 0, 0, 0, 0};
    r.P2 = SE3::exp(se3_2to1).inverse();
    r.X = get_rig_points(type, rot, Vector3Type(0.6, 0.0, 3.0), scale);
    r.ip1 = PinholeCamera(r.K, r.P1).project_points(r.X);
    r.ip2 = PinholeCamera(r.K, r.P2).project_points(r.X);
    return r;
}

static void sfm_solve_L_shape()
{   // test-sfm.cpp:17-90 on the non-degenerate rig (the cube is a critical configuration for the 8-point solver)
    const ScalarType tol = 0.001;
    Rig r = make_rig(RIG_TYPE::L_SHAPE, SO3(1.5, 0.7, 0.0), 0.5);
    Transformation pose2in1;
    std::vector<Point3> points;
    std::vector<size_t> idx;
    ASSERT_TRUE(sfm_solve(r.ip1, r.ip2, r.K, pose2in1, points, idx));
    const Vector6Type expect{1, 0, 0, 0, 0, 0}, got = pose2in1.ln();
    for (int i = 0; i < 6; ++i) ASSERT_EQUAL(expect[i], got[i], tol);
    ASSERT_TRUE(points.size() == r.X.size());
    for (size_t i = 0; i < points.size(); ++i) {
        ASSERT_TRUE(idx[i] == i);
        for (int j = 0; j < 3; ++j) ASSERT_EQUAL(r.X[i][j], points[i][j], tol);
    }
}

static void sfm_triangulate_cube()
{   // test-sfm.cpp:92-155
    const ScalarType tol = 0.001;
    Rig r = make_rig(RIG_TYPE::CUBE, SO3(0.0, 0.0, 0.0), 1.0);
    std::vector<Point3> points;
    std::vector<size_t> idx;
    sfm_triangulate(r.ip1, r.ip2, r.K, r.P1.inverse(), r.P2.inverse(), points, idx);
    ASSERT_TRUE(points.size() == r.X.size());
    for (size_t i = 0; i < points.size(); ++i)
        for (int j = 0; j < 3; ++j) ASSERT_EQUAL(r.X[i][j], points[i][j], tol);
}

static void sfm_solve_too_few_points_returns_false()
{
    Rig r = make_rig(RIG_TYPE::L_SHAPE, SO3(1.5, 0.7, 0.0), 0.5);
    r.ip1.resize(7);
    r.ip2.resize(7);
    Transformation T;
    std::vector<Point3> points;
    std::vector<size_t> idx;
    ASSERT_TRUE(!sfm_solve(r.ip1, r.ip2, r.K, T, points, idx));
}

static void lie_group_round_trips()
{   // test/test-lie-group.cpp:22-132, tolerance 0.01
    const ScalarType tol = 0.01;
    SO3 b(0.1, -0.2, 0.3);
    ASSERT_EQUAL(0.1, b.get_roll(), tol);
    ASSERT_EQUAL(-0.2, b.get_pitch(), tol);
    ASSERT_EQUAL(0.3, b.get_yaw(), tol);
    Matrix3Type I = b.inverse().get_matrix() * b.get_matrix();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) ASSERT_EQUAL(I(i, j), i == j ? 1.0 : 0.0, tol);
    SE3 T(b, Vector3Type(1.0, 2.0, -3.0));
    SE3 T2 = SE3::exp(T.ln());
    for (int i = 0; i < 3; ++i) {
        ASSERT_EQUAL(T2.translation()[i], T.translation()[i], tol);
        for (int j = 0; j < 3; ++j) ASSERT_EQUAL(T2.rotation().get_matrix()(i, j), b.get_matrix()(i, j), tol);
    }
}

static VisualFeature random_feature(std::mt19937 &g, int n)
{
    std::vector<KeyPoint> kp(n);
    Mat8u d;
    d.cols = 32;
    for (int i = 0; i < n; ++i) {
        kp[i] = KeyPoint{{float(g() % 640), float(g() % 480)}, 31.f, 0.f, 1.f, 0, -1};
        uint8_t row[32];
        for (auto &b : row) b = uint8_t(g());
        d.push_back_row(row);
    }
    return VisualFeature(kp, d, 640, 480);
}

static void match_visual_features_planted()
{   // no matcher test exists in the reference (test-frame-manager only checks size() > 0): planted matches
    std::mt19937 g(7);
    VisualFeature vf1 = random_feature(g, 300);
    std::vector<KeyPoint> kp2 = vf1.get_keypoints();
    Mat8u d2 = vf1.get_descriptors();
    for (int i = 0; i < 300; ++i) d2.data[(size_t)i * 32 + (i % 32)] ^= 1;   // one flipped bit per descriptor
    std::reverse(kp2.begin(), kp2.end());
    Mat8u d2r; d2r.cols = 32;
    for (int i = 299; i >= 0; --i) d2r.push_back_row(d2.row(i));
    VisualFeature vf2(kp2, d2r, 640, 480);
    auto m = VisualFeature::match_visual_features(vf1, vf2, 10);
    ASSERT_TRUE(m.size() == 300);
    for (size_t i = 0; i < m.size(); ++i) {
        ASSERT_TRUE(m[i].trainIdx == 299 - m[i].queryIdx && m[i].distance == 1.0f && m[i].imgIdx == 0);
        ASSERT_TRUE(i == 0 || m[i - 1].queryIdx < m[i].queryIdx);       // ties ordered by queryIdx
    }
    auto f = VisualFeature::match_and_filter_visual_features(vf1, vf2, 10);
    ASSERT_TRUE(f.first.size() == 300 && f.second.size() == 300 && f.second.get_descriptors().rows == 300);
    ASSERT_TRUE(f.first.get_keypoints()[5].pt.x == f.second.get_keypoints()[5].pt.x);
}

static void ransac_estimator_and_image_pair()
{
    // an L-rig seen by 2 cameras gives only 8 matches; use a random cloud for the estimator + ImagePair path
    std::mt19937 g(11);
    std::uniform_real_distribution<double> U(-1, 1);
    const int n = 200;
    CameraIntrinsics K = Matrix3Type::Identity();
    K(0, 0) = K(1, 1) = 525; K(0, 2) = 320; K(1, 2) = 240;
    SE3 P2 = SE3::exp(Vector6Type{0.3, 0.02, 0.01, 0.01, 0.03, -0.02}).inverse();
    std::vector<Point3> X;
    for (int i = 0; i < n; ++i) X.emplace_back(2 * U(g), 1.5 * U(g), 6 + 3 * U(g));
    auto ip1 = PinholeCamera(K, SE3()).project_points(X), ip2 = PinholeCamera(K, P2).project_points(X);
    hip::ransac_config().num_hypotheses = 256;
    hip::ransac_config().sampler = MVS_SAMPLER_PHILOX;
    hip::ransac_config().seed = 99;
    hip::ransac_config().max_error_sq = 1e-6;
    Transformation T;
    std::vector<Point3> pts;
    std::vector<size_t> idx;
    ASSERT_TRUE(sfm_solve(ip1, ip2, K, T, pts, idx));
    ASSERT_TRUE(pts.size() == (size_t)n);
    const Vector6Type got = T.ln();
    const ScalarType base = std::sqrt(0.3 * 0.3 + 0.02 * 0.02 + 0.01 * 0.01);
    const Vector6Type want = P2.inverse().ln();
    for (int i = 0; i < 3; ++i) ASSERT_EQUAL(got[i] * base, want[i], 2e-3);   // up to scale
    for (int i = 3; i < 6; ++i) ASSERT_EQUAL(got[i], want[i], 1e-3);
    // FundamentalMatrixEstimatorRANSAC on ideal-camera points
    std::vector<Vector3Type> q1, q2;
    for (int i = 0; i < n; ++i) {
        q1.emplace_back((ip1[i].x - 320) / 525, (ip1[i].y - 240) / 525, 1.0);
        q2.emplace_back((ip2[i].x - 320) / 525, (ip2[i].y - 240) / 525, 1.0);
    }
    FundamentalMatrixEstimatorRANSAC est(1e-6, 64);
    Matrix3Type F;
    std::vector<uint8_t> mask;
    ASSERT_TRUE(est.compute(q1, q2, F, mask));
    size_t inl = 0;
    for (auto b : mask) inl += b;
    ASSERT_TRUE(mask.size() == (size_t)n && inl == (size_t)n);
    for (int i = 0; i < n; i += 17) {
        const Vector3Type l = F * q1[i];
        ASSERT_EQUAL((q2[i][0] * l[0] + q2[i][1] * l[1]) + l[2], 0.0, 1e-6);
    }
    hip::ransac_config() = hip::RansacConfig();
}

static void pnp_solve_cube()
{   // test/test-pnp.cpp:14-60
    const ScalarType tol = 0.001;
    CameraIntrinsics K = Matrix3Type::Identity();
    Vector6Type se3{1, 0, 0, 0, 0, 0};
    CameraExtrinsics P(SE3::exp(se3).inverse());
    PinholeCamera c(K, P);
    std::vector<Point3> world = get_rig_points(RIG_TYPE::CUBE, SO3(0.0, 0.0, 0.0), Vector3Type(0.6, 0.0, 3.0), 1.0);
    std::vector<ImagePoint> image = c.project_points(world);
    Transformation pose;
    std::vector<size_t> inliers;
    ASSERT_TRUE(pnp_solve(world, image, K, pose, inliers));
    ASSERT_TRUE(inliers.size() == world.size());
    const Vector6Type got = pose.ln();
    for (int i = 0; i < 6; ++i) ASSERT_EQUAL(se3[i], got[i], tol);
}

static double get_gaussian(std::mt19937 &g, double mean, double stddev)
{   // test/unit-test-helper.cpp:6-13 (std::normal_distribution there as well; the seed is ours)
    return std::normal_distribution<double>(mean, stddev)(g);
}

static void sfm_refine_L_shape()
{   // test/test-sfm.cpp:157-286
    const ScalarType tol = 0.025;
    std::mt19937 g(2024);
    Rig r = make_rig(RIG_TYPE::L_SHAPE, SO3(1.5, 0.7, 0.0), 0.5);
    const ScalarType c_noise = 5e-3, p_noise = 5e-3;
    std::vector<Point2Estimate> p1e, p2e;
    for (const auto &p : r.ip1) {
        Point2Uncertainty C = sqr(c_noise) * Matrix2Type::Identity();
        p1e.emplace_back(Point2(p.x + get_gaussian(g, 0, c_noise), p.y + get_gaussian(g, 0, c_noise)), C);
    }
    for (const auto &p : r.ip2) {
        Point2Uncertainty C = sqr(c_noise) * Matrix2Type::Identity();
        p2e.emplace_back(Point2(p.x + get_gaussian(g, 0, c_noise), p.y + get_gaussian(g, 0, c_noise)), C);
    }
    Vector6Type delta{get_gaussian(g, 0, 5e-3), get_gaussian(g, 0, 5e-3), get_gaussian(g, 0, 5e-3),
                      get_gaussian(g, 0, 1e-2), get_gaussian(g, 0, 1e-2), get_gaussian(g, 0, 1e-2)};
    Transformation guess = SE3::exp(delta) * r.P1 * r.P2.inverse();
    std::vector<Point3> pg;
    for (const auto &p : r.X)
        pg.emplace_back(p[0] + get_gaussian(g, 0, p_noise), p[1] + get_gaussian(g, 0, p_noise), p[2] + get_gaussian(g, 0, p_noise));
    TransformationEstimate est;
    std::vector<Point3Estimate> pe;
    ScalarType error = -1;
    ASSERT_TRUE(sfm_refine(p1e, p2e, r.K, guess, pg, est, pe, error));
    const Vector6Type expect{1, 0, 0, 0, 0, 0}, got = est.mean().ln();
    for (int i = 0; i < 6; ++i) ASSERT_EQUAL(expect[i], got[i], tol);
    ASSERT_TRUE(pe.size() == r.X.size());
    for (size_t i = 0; i < pe.size(); ++i)
        for (int j = 0; j < 3; ++j) ASSERT_EQUAL(r.X[i][j], pe[i].mean()[j], tol);
    ASSERT_TRUE(error >= 0 && error < 100);
    for (int i = 0; i < 6; ++i) ASSERT_TRUE(est.covar()(i, i) > 0 && est.covar()(i, i) < 1e-3);   // a proper covariance
    for (const auto &q : pe) ASSERT_TRUE(q.covar()(0, 0) > 0 && q.covar()(2, 2) > 0);
}

static void pnp_refine_L_shape()
{   // test/test-pnp.cpp:62-160
    const ScalarType tol = 0.025;
    std::mt19937 g(7);
    CameraIntrinsics K = Matrix3Type::Identity();
    Vector6Type se3{1, 0, 0, 0, 0, 0};
    CameraExtrinsics P(SE3::exp(se3).inverse());
    PinholeCamera c(K, P);
    std::vector<Point3> world = get_rig_points(RIG_TYPE::L_SHAPE, SO3(1.5, 0.7, 0.0), Vector3Type(0.6, 0.0, 3.0), 0.5);
    std::vector<ImagePoint> image = c.project_points(world);
    std::vector<Point2Estimate> ie;
    std::vector<Point3Estimate> we;
    for (const auto &p : image) {
        Point2Uncertainty C = sqr(5e-3) * Matrix2Type::Identity();
        ie.emplace_back(Point2(p.x + get_gaussian(g, 0, 5e-3), p.y + get_gaussian(g, 0, 5e-3)), C);
    }
    for (const auto &p : world) {
        Point3Uncertainty C = Matrix3Type::Identity();
        for (int k = 0; k < 3; ++k) C(k, k) = sqr(5e-3);
        we.emplace_back(Point3(p[0] + get_gaussian(g, 0, 5e-3), p[1] + get_gaussian(g, 0, 5e-3), p[2] + get_gaussian(g, 0, 5e-3)), C);
    }
    Vector6Type delta{get_gaussian(g, 0, 5e-3), get_gaussian(g, 0, 5e-3), get_gaussian(g, 0, 5e-3),
                      get_gaussian(g, 0, 5e-3), get_gaussian(g, 0, 5e-3), get_gaussian(g, 0, 5e-3)};
    Transformation guess = SE3::exp(delta) * P.inverse();
    TransformationEstimate est;
    ScalarType error = -1;
    ASSERT_TRUE(pnp_refine(we, ie, K, guess, est, error));
    const Vector6Type got = est.mean().ln();
    for (int i = 0; i < 6; ++i) ASSERT_EQUAL(se3[i], got[i], tol);
    ASSERT_TRUE(error >= 0);
}

static void image_pair_refine()
{   // ImagePair with refine_structure_in_constructor (front-end/image-pair.cpp:67-70,176-238) on planted features
    std::mt19937 g(5);
    std::uniform_real_distribution<double> U(-1, 1);
    std::normal_distribution<double> Npx(0.0, 0.3);
    const int n = 300;
    CameraIntrinsics K = Matrix3Type::Identity();
    K(0, 0) = K(1, 1) = 525; K(0, 2) = 320; K(1, 2) = 240;
    SE3 P2 = SE3::exp(Vector6Type{0.3, 0.02, 0.01, 0.01, 0.03, -0.02}).inverse();
    std::vector<Point3> X;
    for (int i = 0; i < n; ++i) X.emplace_back(2 * U(g), 1.5 * U(g), 6 + 3 * U(g));
    auto ip1 = PinholeCamera(K, SE3()).project_points(X), ip2 = PinholeCamera(K, P2).project_points(X);
    std::vector<KeyPoint> kp1(n), kp2(n);
    Mat8u d1, d2;
    d1.cols = d2.cols = 32;
    for (int i = 0; i < n; ++i) {
        kp1[i] = KeyPoint{{(float)(ip1[i].x + Npx(g)), (float)(ip1[i].y + Npx(g))}, 31, 0, 0, 0, -1};
        kp2[i] = KeyPoint{{(float)(ip2[i].x + Npx(g)), (float)(ip2[i].y + Npx(g))}, 31, 0, 0, 0, -1};
        uint8_t row[32];
        for (auto &b : row) b = (uint8_t)(g() & 0xff);
        d1.push_back_row(row);
        row[i % 32] ^= 1;   // distance 1 to its partner, ~128 to everything else
        d2.push_back_row(row);
    }
    Frame f1{1, VisualFeature(kp1, d1, 640, 480)}, f2{2, VisualFeature(kp2, d2, 640, 480)};
    hip::ransac_config().num_hypotheses = 512;
    hip::ransac_config().sampler = MVS_SAMPLER_PHILOX;
    hip::ransac_config().seed = 3;
    hip::ransac_config().max_error_sq = 1e-2;
    ImagePair::Params prm = ImagePair::get_default_params();
    ImagePair plain(f1, f2, K, prm);
    prm.refine_structure_in_constructor = true;
    ImagePair refined(f1, f2, K, prm);
    hip::ransac_config() = hip::RansacConfig();
    ASSERT_TRUE(plain.valid && plain.state() == ImagePair::State::RECONSTRUCTED);
    ASSERT_TRUE(refined.valid && refined.state() == ImagePair::State::REFINED);
    ASSERT_TRUE(refined.matched_points.size() == plain.matched_points.size() && refined.matched_points.size() > 200);
    ASSERT_TRUE(refined.matched_points_covar.size() == refined.matched_points.size());
    // the refined rotation stays at the truth (0.3 px noise), and the pose moved off the linear estimate
    const Vector6Type want = P2.inverse().ln(), a = plain.T_pair_to_base.ln(), b = refined.T_pair_to_base.ln();
    double eb = 0, moved = 0;
    for (int i = 3; i < 6; ++i) eb += sqr(b[i] - want[i]);
    for (int i = 0; i < 6; ++i) moved += sqr(b[i] - a[i]);
    ASSERT_TRUE(eb < 1e-5 && moved > 0);
    ASSERT_TRUE(refined.error > 0 && refined.T_pair_to_base_covar(0, 0) > 0);
}

static void image_pair_update()
{   // ImagePair::update (front-end/image-pair.cpp:77-114; caller: VisualOdometer::add_frame): same frame ids -> false; a
    // candidate with fewer inliers -> false; a candidate at least as good whose refined error beats this pair's -> swapped in
    std::mt19937 g(11);
    std::uniform_real_distribution<double> U(-1, 1);
    std::normal_distribution<double> Npx(0.0, 0.3);
    const int n = 300;
    CameraIntrinsics K = Matrix3Type::Identity();
    K(0, 0) = K(1, 1) = 525; K(0, 2) = 320; K(1, 2) = 240;
    std::vector<Point3> X;
    for (int i = 0; i < n; ++i) X.emplace_back(2 * U(g), 1.5 * U(g), 6 + 3 * U(g));
    std::vector<std::vector<uint8_t>> rows(n, std::vector<uint8_t>(32));
    for (auto &r : rows) for (auto &b : r) b = (uint8_t)(g() & 0xff);
    // frame k sees the cloud from pose k; `keep` of its features carry their point's descriptor (one bit flipped per frame),
    // the others a fresh random one (no partner anywhere)
    auto make_frame = [&](uint32_t id, const SE3 &pose, int keep) {
        auto ip = PinholeCamera(K, pose).project_points(X);
        std::vector<KeyPoint> kp(n);
        Mat8u d;
        d.cols = 32;
        for (int i = 0; i < n; ++i) {
            kp[i] = KeyPoint{{(float)(ip[i].x + Npx(g)), (float)(ip[i].y + Npx(g))}, 31, 0, 0, 0, -1};
            std::vector<uint8_t> row = rows[i];
            if (i < keep)
                row[(i + id) % 32] ^= (uint8_t)(1u << (id % 8));
            else
                for (auto &b : row) b = (uint8_t)(g() & 0xff);
            d.push_back_row(row.data());
        }
        return Frame{id, VisualFeature(kp, d, 640, 480)};
    };
    const SE3 P2 = SE3::exp(Vector6Type{0.3, 0.02, 0.01, 0.01, 0.03, -0.02}).inverse();
    const SE3 P3 = SE3::exp(Vector6Type{0.45, -0.03, 0.02, -0.01, 0.02, 0.015}).inverse();
    const Frame f1 = make_frame(1, SE3(), n), f2 = make_frame(2, P2, 220), f3 = make_frame(3, P3, n), f4 = make_frame(4, P3, 120);
    hip::ransac_config().num_hypotheses = 512;
    hip::ransac_config().sampler = MVS_SAMPLER_PHILOX;
    hip::ransac_config().seed = 3;
    hip::ransac_config().max_error_sq = 1e-2;
    ImagePair pair(f1, f2, K, ImagePair::get_default_params());
    ASSERT_TRUE(pair.valid && pair.state() == ImagePair::State::RECONSTRUCTED && pair.error == infinity);
    const uint32_t count12 = pair.match_inlier_count;
    ASSERT_TRUE(count12 > 150 && count12 <= 220);
    ASSERT_TRUE(!pair.update(f1) && !pair.update(f2));                 // the pair's own frames (:80-84)
    ASSERT_TRUE(!pair.update(f4));                                     // fewer inliers than (1, 2) (:96-100)
    ASSERT_TRUE(pair.pair_frame().id == 2 && pair.match_inlier_count == count12 && pair.state() == ImagePair::State::RECONSTRUCTED);
    ASSERT_TRUE(pair.update(f3));                                      // more inliers, larger SSD, refined error < infinity
    ASSERT_TRUE(pair.pair_frame().id == 3 && pair.base_frame().id == 1 && pair.state() == ImagePair::State::REFINED);
    ASSERT_TRUE(pair.match_inlier_count > count12 && pair.error < infinity && pair.error > 0);
    ASSERT_TRUE(pair.matched_points_covar.size() == pair.matched_points.size());
    const Vector6Type want = P3.inverse().ln(), got = pair.T_pair_to_base.ln();
    double er = 0;
    for (int i = 3; i < 6; ++i) er += sqr(got[i] - want[i]);
    if (!(er < 1e-4))
        std::printf("    update: rotation error^2 %.3g; got (%.5f %.5f %.5f | %.5f %.5f %.5f) want (%.5f %.5f %.5f | %.5f %.5f %.5f), error %.4g, inliers %u\n",
                    er, got[0], got[1], got[2], got[3], got[4], got[5], want[0], want[1], want[2], want[3], want[4], want[5],
                    pair.error, pair.match_inlier_count);
    ASSERT_TRUE(er < 1e-4);   // (0.3 px noise, 300 points, baseline 0.45: the refined rotation is good to a few mrad)
    // the same candidate again: (1, 3) against itself has equal counts and SSD, its refined error is not below its own
    const Frame f5 = make_frame(5, P3, n);
    const ScalarType err13 = pair.error;
    const bool again = pair.update(f5);
    ASSERT_TRUE(again ? pair.error < err13 : pair.error == err13);
    hip::ransac_config() = hip::RansacConfig();
}

static void ba_frame_pose_and_point_two_frames()
{   // the reference's own BA entry point (vision/ba.hpp:25-36) shaped like VisualOdometer::track_refine: frame 7 anchored
    // at its own pose, frame 9 regularised, even point ids with priors, odd ones without, one observation missing
    std::mt19937 g(17);
    CameraIntrinsics K = Matrix3Type::Identity();
    K(0, 0) = K(1, 1) = 525; K(0, 2) = 320; K(1, 2) = 240;
    const SE3 Ta = SE3::exp(Vector6Type{0.4, -0.1, 0.2, 0.02, -0.1, 0.03});
    const SE3 Tb = SE3::exp(Vector6Type{0.7, -0.08, 0.25, 0.03, -0.12, 0.02});
    std::unordered_set<Id::Type> fid{7, 9}, pid;
    std::unordered_map<Id::Type, Transformation> guess{{7, Ta}, {9, SE3::exp(Vector6Type{0.003, -0.002, 0.004, 0.002, 0.001, -0.003}) * Tb}};
    std::unordered_map<Id::Type, TransformationUncertainty> fprior;
    fprior[7] = 1e-5 * Matrix6Type::Identity();
    fprior[9] = 1e-2 * Matrix6Type::Identity();
    std::unordered_map<Id::Type, Point3> pguess;
    std::unordered_map<Id::Type, Point3Uncertainty> pprior;
    std::unordered_map<Id::Type, PointIdToPoint2Estimate> fobs;
    std::unordered_map<Id::Type, Point3> truth;
    PinholeCamera ca(K, Ta.inverse()), cb(K, Tb.inverse());
    for (Id::Type i = 100; i < 140; ++i) {
        pid.insert(i);
        Point3 X(get_gaussian(g, 0, 1.0), get_gaussian(g, 0, 0.8), 6 + get_gaussian(g, 0, 1.0));
        truth[i] = X;
        pguess[i] = Point3(X[0] + get_gaussian(g, 0, 5e-3), X[1] + get_gaussian(g, 0, 5e-3), X[2] + get_gaussian(g, 0, 5e-3));
        if (i % 2 == 0) {
            Point3Uncertainty C = Matrix3Type::Identity();
            for (int k = 0; k < 3; ++k) C(k, k) = 1e-4;
            pprior[i] = C;
        }
        const auto ua = ca.project_points({X})[0], ub = cb.project_points({X})[0];
        const Point2Uncertainty C2 = 0.25 * Matrix2Type::Identity();
        fobs[7][i] = Point2Estimate(Point2(ua.x + get_gaussian(g, 0, 0.5), ua.y + get_gaussian(g, 0, 0.5)), C2);
        if (i != 104)
            fobs[9][i] = Point2Estimate(Point2(ub.x + get_gaussian(g, 0, 0.5), ub.y + get_gaussian(g, 0, 0.5)), C2);
    }
    std::unordered_map<Id::Type, TransformationEstimate> fest;
    std::unordered_map<Id::Type, Point3Estimate> pest;
    ScalarType err = -1;
    ba_frame_pose_and_point(K, fid, pid, guess, fprior, pguess, pprior, fobs, fest, pest, err);
    ASSERT_TRUE(fest.size() == 2 && pest.size() == 40 && err > 0);
    const Vector6Type a = fest[7].mean().ln(), a0 = Ta.ln(), b = fest[9].mean().ln(), b0 = Tb.ln();
    for (int i = 0; i < 6; ++i) ASSERT_EQUAL(a[i], a0[i], 1e-3);     // anchored
    for (int i = 0; i < 6; ++i) ASSERT_EQUAL(b[i], b0[i], 2e-2);     // regularised, pulled to the data
    for (const auto &kv : pest) {
        const bool has_prior = kv.first % 2 == 0;   // without a prior the depth of a point is known to ~0.16 (0.5 px, b 0.3)
        for (int j = 0; j < 3; ++j) ASSERT_EQUAL(kv.second.mean()[j], truth[kv.first][j], has_prior ? 0.05 : (j == 2 ? 1.0 : 0.2));
        ASSERT_TRUE(kv.second.covar()(2, 2) > 0 && (has_prior ? kv.second.covar()(2, 2) < 2e-4 : kv.second.covar()(2, 2) > 1e-3));
    }
    ASSERT_TRUE(fest[9].covar()(0, 0) > fest[7].covar()(0, 0));   // the anchored frame is the better known one
}

static void visual_feature_extract_and_match()
{   // VisualFeature::extract on a synthetic textured frame and its copy shifted by 7 px: the matches carry the shift
    std::mt19937 g(3);
    const int W = 320, H = 240;
    Mat8u a, b;
    a.rows = b.rows = H; a.cols = b.cols = W;
    a.data.resize((size_t)W * H); b.data.resize((size_t)W * H);
    std::vector<uint8_t> big((size_t)(W + 64) * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W + 64; ++x) {
            uint32_t hsh = (uint32_t)(x / 6) * 2654435761u ^ (uint32_t)(y / 6) * 40503u;
            hsh ^= hsh >> 13; hsh *= 0x5bd1e995u; hsh ^= hsh >> 15;
            big[(size_t)y * (W + 64) + x] = (uint8_t)(hsh & 0xff);
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            a.data[(size_t)y * W + x] = big[(size_t)y * (W + 64) + x + 7];
            b.data[(size_t)y * W + x] = big[(size_t)y * (W + 64) + x];
        }
    VisualFeature fa = VisualFeature::extract(a), fb = VisualFeature::extract(b);
    ASSERT_TRUE(fa.valid() && fb.valid() && fa.size() > 300 && fa.size() <= 500);
    ASSERT_TRUE(fa.get_descriptors().rows == (int)fa.size() && fa.get_descriptors().cols == 32);
    auto m = VisualFeature::match_visual_features(fa, fb, 30);
    ASSERT_TRUE(m.size() > 100);
    size_t good = 0;
    for (const auto &mm : m) {
        const auto &pa = fa.get_keypoints()[mm.trainIdx].pt, &pb = fb.get_keypoints()[mm.queryIdx].pt;
        good += std::fabs((pb.x - pa.x) - 7.0f) < 1.5f && std::fabs(pb.y - pa.y) < 1.5f;
    }
    ASSERT_TRUE(good * 10 >= m.size() * 9);
    auto pe = fa.get_point_estimates();
    const double sigma = (double)(1 << fa.get_keypoints()[0].octave) * 0.5;   // visual-feature.cpp:202
    ASSERT_TRUE(pe.size() == fa.size() && pe[0].covar()(0, 0) == sigma * sigma && pe[0].covar()(0, 1) == 0.0);
}

int main()
{
    try {
        RUN(lie_group_round_trips);
        RUN(sfm_solve_L_shape);
        RUN(sfm_triangulate_cube);
        RUN(sfm_solve_too_few_points_returns_false);
        RUN(match_visual_features_planted);
        RUN(ransac_estimator_and_image_pair);
        RUN(pnp_solve_cube);
        RUN(sfm_refine_L_shape);
        RUN(pnp_refine_L_shape);
        RUN(image_pair_refine);
        RUN(image_pair_update);
        RUN(ba_frame_pose_and_point_two_frames);
        RUN(visual_feature_extract_and_match);
    } catch (const std::exception &e) {
        std::printf("EXCEPTION: %s\n", e.what());
        return 2;
    }
    std::printf(g_fail ? "%d FAILED\n" : "ALL PASSED\n", g_fail);
    return g_fail ? 1 : 0;
}
