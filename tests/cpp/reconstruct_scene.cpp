// Drives the drop-in shim through the call sequence of the reference's two-image utility (utility/reconstruct-scene.cpp:
// extract both images, match_and_filter_visual_features, camera from a config file, sfm_solve) and prints what
// tests/test_compat_cpp.py checks.  Written for the test-suite: raw 8-bit frames instead of image files (the build image
// has no decoder for C++), no viewer, machine-readable output.
//   argv: frame_a.raw frame_b.raw width height camera.config max_descriptor_distance
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../mvslam_amd/compat/mvslam_compat.hpp"

namespace
{
bool read_raw_frame(const char *path, int w, int h, mvSLAM::Mat8u &frame)
{
    frame.rows = h;
    frame.cols = w;
    frame.data.assign((size_t)w * h, 0);
    std::FILE *fp = std::fopen(path, "rb");
    if (!fp)
        return false;
    const size_t got = std::fread(frame.data.data(), 1, frame.data.size(), fp);
    std::fclose(fp);
    return got == frame.data.size();
}
}  // namespace

int main(int argc, char **argv)
{
    using namespace mvSLAM;
    if (argc != 7) {
        std::fprintf(stderr, "expected: frame_a.raw frame_b.raw width height camera.config max_distance\n");
        return 64;
    }
    const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
    Mat8u frames[2];
    if (w <= 0 || h <= 0 || !read_raw_frame(argv[1], w, h, frames[0]) || !read_raw_frame(argv[2], w, h, frames[1])) {
        std::fprintf(stderr, "unreadable frames\n");
        return 65;
    }
    // a real RANSAC instead of the single hypothesis the reference ships with (sfm-solve.cpp:67): backend setting only
    hip::RansacConfig &rc = hip::ransac_config();
    rc.num_hypotheses = 2000;
    rc.sampler = MVS_SAMPLER_PHILOX;
    rc.seed = 1;
    rc.max_error_sq = 1e-3;

    const VisualFeature feat_a = VisualFeature::extract(frames[0]), feat_b = VisualFeature::extract(frames[1]);
    const auto filtered = VisualFeature::match_and_filter_visual_features(feat_a, feat_b, (ScalarType)std::atoi(argv[6]));
    const PinholeCamera cam{std::string(argv[5])};
    const CameraIntrinsics &K = cam.get_intrinsics();

    Transformation T_b_in_a;
    std::vector<Point3> cloud;
    std::vector<size_t> cloud_match_index;
    const bool solved = sfm_solve(filtered.first.get_image_points(), filtered.second.get_image_points(), K, T_b_in_a, cloud,
                                  cloud_match_index);
    std::printf("features: %zu %zu, matches: %zu\n", feat_a.size(), feat_b.size(), filtered.first.size());
    std::printf("camera intrinsics: fx %g fy %g shear %g px %g py %g\n", K(0, 0), K(1, 1), K(0, 1), K(0, 2), K(1, 2));
    if (!solved) {
        std::printf("no reconstruction\n");
        return 2;
    }
    const Vector6Type xi = T_b_in_a.ln();
    std::printf("scaled transformation (se3) = %.9f %.9f %.9f %.9f %.9f %.9f\n", xi[0], xi[1], xi[2], xi[3], xi[4], xi[5]);
    std::printf("pointsin1_scaled = %zu points\n", cloud.size());
    for (size_t k = 0; k < cloud.size() && k < 5; ++k)
        std::printf("%g, %g, %g\n", cloud[k].x(), cloud[k].y(), cloud[k].z());
    return 0;
}
