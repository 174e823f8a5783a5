// The reference's two-image driver (utility/reconstruct-scene.cpp:22-66) on top of the drop-in shim, call for call:
// load two grayscale images -> VisualFeature::extract x2 -> match_and_filter_visual_features -> PinholeCamera(file) ->
// sfm_solve -> print.  Differences: no visualiser window, and the images come as raw 8-bit files (width height on the
// command line) because the build image has no JPEG decoder for C++ (the reference uses cv::imread).
// usage: reconstruct_scene <image_1.raw> <image_2.raw> <width> <height> <intrinsics> <max_dist>
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "../../mvslam_amd/compat/mvslam_compat.hpp"

static mvSLAM::Mat8u load_image_grayscale(const std::string &filename, int width, int height)
{   // base/image.cpp:10-15
    mvSLAM::Mat8u image;
    image.rows = height;
    image.cols = width;
    image.data.resize((size_t)width * height);
    std::FILE *f = std::fopen(filename.c_str(), "rb");
    const bool ok = f && std::fread(image.data.data(), 1, image.data.size(), f) == image.data.size();
    if (f) std::fclose(f);
    if (!ok) image.rows = image.cols = 0;
    return image;
}

int main(int argc, char **argv)
{
    if (argc != 7) {
        std::printf("Usage: %s <image_1.raw> <image_2.raw> <width> <height> <intrinsics> <max_dist>\n", argv[0]);
        return 1;
    }
    const int width = std::stoi(argv[3]), height = std::stoi(argv[4]);
    mvSLAM::ScalarType max_dist = std::stoi(std::string(argv[6]));

    // input
    mvSLAM::Mat8u image1 = load_image_grayscale(argv[1], width, height);
    mvSLAM::Mat8u image2 = load_image_grayscale(argv[2], width, height);
    if (image1.rows <= 0 || image2.rows <= 0) {
        std::printf("cannot read the images.\n");
        return 1;
    }
    auto image1_vf = mvSLAM::VisualFeature::extract(image1);
    auto image2_vf = mvSLAM::VisualFeature::extract(image2);
    auto matched_vf_pair = mvSLAM::VisualFeature::match_and_filter_visual_features(image1_vf, image2_vf, max_dist);
    mvSLAM::PinholeCamera camera{std::string(argv[5])};

    // the reference as shipped scores ONE hypothesis on the first 8 matches (sfm-solve.cpp:67); a real RANSAC is a
    // setting of the backend, not of the call surface
    mvSLAM::hip::ransac_config().num_hypotheses = 2000;
    mvSLAM::hip::ransac_config().sampler = MVS_SAMPLER_PHILOX;
    mvSLAM::hip::ransac_config().seed = 1;
    mvSLAM::hip::ransac_config().max_error_sq = 1e-3;

    // output
    mvSLAM::Transformation pose2in1_scaled;
    std::vector<mvSLAM::Point3> pointsin1_scaled;
    std::vector<size_t> point_indexes;
    if (!sfm_solve(matched_vf_pair.first.get_image_points(), matched_vf_pair.second.get_image_points(),
                   camera.get_intrinsics(), pose2in1_scaled, pointsin1_scaled, point_indexes)) {
        std::printf("Reconstruction failed.\n");
        return 2;
    }
    const auto &K = camera.get_intrinsics();
    std::printf("features: %zu %zu, matches: %zu\n", image1_vf.size(), image2_vf.size(), matched_vf_pair.first.size());
    std::printf("camera intrinsics: fx %g fy %g shear %g px %g py %g\n", K(0, 0), K(1, 1), K(0, 1), K(0, 2), K(1, 2));
    const mvSLAM::Vector6Type se3 = pose2in1_scaled.ln();
    std::printf("scaled transformation (se3) = %.9f %.9f %.9f %.9f %.9f %.9f\n", se3[0], se3[1], se3[2], se3[3], se3[4], se3[5]);
    std::printf("pointsin1_scaled = %zu points\n", pointsin1_scaled.size());
    for (size_t i = 0; i < pointsin1_scaled.size() && i < 5; ++i)
        std::cout << pointsin1_scaled[i].x() << ", " << pointsin1_scaled[i].y() << ", " << pointsin1_scaled[i].z() << std::endl;
    return 0;
}
